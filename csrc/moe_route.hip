// Mixture-of-experts routing in three kernels instead of ~30 framework ops:
//
//   moe_topk      router logits -> softmax -> top-k (weights renormalised on request)
//   moe_align     counting sort of the T*k (token, expert) slots by expert and the
//                 row-tile table the grouped GEMMs consume (csrc/skinny_gemm.hip,
//                 csrc/quant_gemm.hip: tiles[n][3] = group, first row, rows)
//   moe_combine   out[t] = sum_j w[t][j] * y[position of slot (t, j)]
//
// A decode step of a sparse model spends most of its time between GEMMs at small
// batch (each framework op is a 5-8 us launch); with these the expert block of a
// layer is eight launches: router GEMM, topk, align, row gather, two grouped
// GEMMs, SiLU, combine.  Nothing here synchronises with the host, so the step
// stays hipGraph-capturable.
//
//   moe_route_small   the first three steps (top-k, slot sort + tile table, gather of the
//                 token rows into expert order) in ONE launch for decode batches of at most
//                 16 tokens: every workgroup repeats the (tiny) routing computation in LDS, so
//                 no workgroup waits for another, and then copies its share of the rows.
//                 Positions within an expert follow slot order, so all workgroups agree.
#include "common.h"

namespace {

constexpr int TOPK_MAX_VPL = 8;       // experts per lane: up to 512 experts
constexpr int TOPK_MAX_K = 16;
constexpr int ALIGN_MAX_E = 1024;
constexpr int ALIGN_THREADS = 1024;

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

// softmax + top-k of one token's logits by one wave: lane j (< k) returns the j-th pick
template <typename T>
__device__ __forceinline__ void wave_topk(const T* __restrict__ row, int E, int k, int renorm, int lane,
                                          float& out_w, int& out_i) {
  float v[TOPK_MAX_VPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < TOPK_MAX_VPL; ++i) {
    const int e = lane + 64 * i;
    v[i] = e < E ? to_f(row[e]) : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < TOPK_MAX_VPL; ++i) {
    v[i] = (lane + 64 * i) < E ? __expf(v[i] - mx) : -1.f;      // probabilities are >= 0: -1 marks "taken / absent"
    sum += fmaxf(v[i], 0.f);
  }
  sum = wave_sum(sum);
  const float inv_sum = 1.f / sum;
  float picked_w = 0.f;       // lane j keeps the j-th pick
  int picked_i = 0;
  float wsum = 0.f;
  for (int j = 0; j < k; ++j) {
    float best = -1.f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < TOPK_MAX_VPL; ++i)
      if (v[i] > best) { best = v[i]; best_e = lane + 64 * i; }      // ascending e within a lane: first maximum wins
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oe = __shfl_xor(best_e, o, 64);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
    // every lane now agrees on (best, best_e); ties went to the lowest expert id
    if ((best_e & 63) == lane) {
#pragma unroll
      for (int i = 0; i < TOPK_MAX_VPL; ++i)
        if (i == (best_e >> 6)) v[i] = -1.f;
    }
    if (lane == j) { picked_w = best * inv_sum; picked_i = best_e; }
    wsum += best * inv_sum;
  }
  out_w = renorm ? picked_w / wsum : picked_w;
  out_i = picked_i;
}

// one wave per token
template <typename T>
__global__ __launch_bounds__(256) void moe_topk_kernel(const T* __restrict__ logits, long stride,
                                                       float* __restrict__ topw, int* __restrict__ topi, int tokens,
                                                       int E, int k, int renorm) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wid;
  if (t >= tokens) return;
  float w;
  int e;
  wave_topk(logits + (long)t * stride, E, k, renorm, lane, w, e);
  if (lane < k) {
    topw[(long)t * k + lane] = w;
    topi[(long)t * k + lane] = e;
  }
}

// DeepSeek routing, one wave per token.  Scores: sigmoid of the logits (V3) or their softmax (V2); the CHOICE is
// made on `score + bias[e]` (bias may be null), optionally inside the `topk_group` best of `n_group` equal groups
// of experts - a group scored by the sum of its two best choice values (`group_top2` != 0, V3) or by its best one
// (V2) -; the WEIGHTS are the unbiased scores of the chosen experts, renormalised on request, times `scale`.
// Ties go to the lowest expert (and group) id.
template <typename T>
__global__ __launch_bounds__(256) void moe_topk_grouped_kernel(
    const T* __restrict__ logits, long stride, const float* __restrict__ bias, float* __restrict__ topw,
    int* __restrict__ topi, int tokens, int E, int k, int n_group, int topk_group, int sigmoid, int group_top2,
    int renorm, float scale) {
  __shared__ float s_choice[4][64 * TOPK_MAX_VPL];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wid;
  if (t >= tokens) return;                       // (waves of a workgroup share nothing: no barrier below)
  const T* row = logits + (long)t * stride;
  float sc[TOPK_MAX_VPL], ch[TOPK_MAX_VPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < TOPK_MAX_VPL; ++i) {
    const int e = lane + 64 * i;
    sc[i] = e < E ? to_f(row[e]) : -INFINITY;
    mx = fmaxf(mx, sc[i]);
  }
  if (sigmoid) {
#pragma unroll
    for (int i = 0; i < TOPK_MAX_VPL; ++i) sc[i] = fast_sigmoid(sc[i]);
  } else {
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TOPK_MAX_VPL; ++i) {
      sc[i] = (lane + 64 * i) < E ? __expf(sc[i] - mx) : 0.f;
      sum += sc[i];
    }
    const float inv_sum = 1.f / wave_sum(sum);
#pragma unroll
    for (int i = 0; i < TOPK_MAX_VPL; ++i) sc[i] *= inv_sum;
  }
#pragma unroll
  for (int i = 0; i < TOPK_MAX_VPL; ++i) {
    const int e = lane + 64 * i;
    ch[i] = e < E ? sc[i] + (bias != nullptr ? bias[e] : 0.f) : -INFINITY;
  }
  if (n_group > 1 && topk_group < n_group) {
    float* mine = s_choice[wid];
#pragma unroll
    for (int i = 0; i < TOPK_MAX_VPL; ++i)
      if (lane + 64 * i < E) mine[lane + 64 * i] = ch[i];
    __builtin_amdgcn_wave_barrier();               // (one wave's LDS instructions execute in order: its reads below
                                                   // see the writes above; the barrier only pins the compiler's order)
    const int gsize = E / n_group;
    float gscore = -INFINITY;                      // lane g scores group g
    if (lane < n_group) {
      float b1 = -INFINITY, b2 = -INFINITY;
      for (int j = 0; j < gsize; ++j) {
        const float v = mine[lane * gsize + j];
        if (v > b1) { b2 = b1; b1 = v; } else if (v > b2) { b2 = v; }
      }
      gscore = group_top2 ? (gsize > 1 ? b1 + b2 : b1) : b1;
    }
    unsigned long long keep = 0ull;
    for (int j = 0; j < topk_group; ++j) {
      float best = gscore;
      int best_g = lane < n_group ? lane : 0x7fffffff;
      if (lane >= n_group || ((keep >> lane) & 1ull)) { best = -INFINITY; best_g = 0x7fffffff; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int og = __shfl_xor(best_g, o, 64);
        if (ob > best || (ob == best && og < best_g)) { best = ob; best_g = og; }
      }
      if (best_g < 64) keep |= 1ull << best_g;
    }
#pragma unroll
    for (int i = 0; i < TOPK_MAX_VPL; ++i) {
      const int e = lane + 64 * i;
      if (e < E && !((keep >> (e / gsize)) & 1ull)) ch[i] = -INFINITY;
    }
  }
  float picked_w = 0.f, wsum = 0.f;
  int picked_i = 0;
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY, best_s = 0.f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < TOPK_MAX_VPL; ++i)
      if (ch[i] > best) { best = ch[i]; best_s = sc[i]; best_e = lane + 64 * i; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64), os = __shfl_xor(best_s, o, 64);
      const int oe = __shfl_xor(best_e, o, 64);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_s = os; best_e = oe; }
    }
    if (best_e == 0x7fffffff) { best_e = 0; best_s = 0.f; }     // fewer than k eligible experts: weight 0
    else if ((best_e & 63) == lane) {
#pragma unroll
      for (int i = 0; i < TOPK_MAX_VPL; ++i)
        if (i == (best_e >> 6)) ch[i] = -INFINITY;
    }
    if (lane == j) { picked_w = best_s; picked_i = best_e; }
    wsum += best_s;
  }
  if (lane < k) {
    topw[(long)t * k + lane] = (renorm ? picked_w / (wsum + 1e-20f) : picked_w) * scale;
    topi[(long)t * k + lane] = picked_i;
  }
}

// single workgroup; slots = tokens * k
__global__ __launch_bounds__(ALIGN_THREADS) void moe_align_kernel(
    const int* __restrict__ topi, int slots, int k, int E, int tile_rows, int expert_start, int local_experts,
    int* __restrict__ src_row, int* __restrict__ inv, int* __restrict__ tiles, int max_tiles) {
  __shared__ int counts[ALIGN_MAX_E], cursor[ALIGN_MAX_E], row0[ALIGN_MAX_E], tile0[ALIGN_MAX_E];
  __shared__ int total_tiles;
  const int tid = threadIdx.x;
  for (int e = tid; e < E; e += ALIGN_THREADS) { counts[e] = 0; cursor[e] = 0; }
  __syncthreads();
  for (int s = tid; s < slots; s += ALIGN_THREADS) atomicAdd(&counts[topi[s]], 1);
  __syncthreads();
  if (tid < 64) {
    // exclusive scans of rows and of tiles per expert, one wave, 64 experts a round
    int rbase = 0, tbase = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + tid;
      const int c = e < E ? counts[e] : 0;
      const int tl = (c + tile_rows - 1) / tile_rows;
      int rs = c, ts = tl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int r2 = __shfl_up(rs, o, 64), t2 = __shfl_up(ts, o, 64);
        if (tid >= o) { rs += r2; ts += t2; }
      }
      if (e < E) { row0[e] = rbase + rs - c; tile0[e] = tbase + ts - tl; }
      rbase += __shfl(rs, 63, 64);
      tbase += __shfl(ts, 63, 64);
    }
    if (tid == 0) total_tiles = tbase;
  }
  __syncthreads();
  for (int s = tid; s < slots; s += ALIGN_THREADS) {
    const int e = topi[s];
    const int pos = row0[e] + atomicAdd(&cursor[e], 1);
    inv[s] = pos;
    src_row[pos] = s / k;
  }
  for (int e = tid; e < E; e += ALIGN_THREADS) {
    const int c = counts[e];
    const bool local = local_experts < 0 || (e >= expert_start && e < expert_start + local_experts);
    const int le = local_experts < 0 ? e : min(max(e - expert_start, 0), max(local_experts - 1, 0));
    int j = tile0[e];
    for (int r = 0; r < c; r += tile_rows, ++j) {
      if (j >= max_tiles) break;
      tiles[3 * j + 0] = le;
      tiles[3 * j + 1] = row0[e] + r;
      tiles[3 * j + 2] = local ? min(tile_rows, c - r) : 0;
    }
  }
  for (int j = total_tiles + tid; j < max_tiles; j += ALIGN_THREADS) {
    tiles[3 * j + 0] = 0;
    tiles[3 * j + 1] = 0;
    tiles[3 * j + 2] = 0;
  }
}

constexpr int SMALL_MAX_TOKENS = 16;                              // one wave per token in a 1024-thread workgroup
constexpr int SMALL_MAX_SLOTS = SMALL_MAX_TOKENS * TOPK_MAX_K;
constexpr int SMALL_MAX_E = 64 * TOPK_MAX_VPL;

// grid = any number of workgroups >= 1; workgroup 0 writes the routing tables, every workgroup
// copies rows pos = blockIdx.x, blockIdx.x + gridDim.x, ... of the expert-sorted activations
template <typename T, typename TX>
__global__ __launch_bounds__(ALIGN_THREADS) void moe_route_small_kernel(
    const T* __restrict__ logits, long lstride, int tokens, int E, int k, int renorm, int tile_rows, int expert_start,
    int local_experts, float* __restrict__ topw, int* __restrict__ topi, int* __restrict__ src_row,
    int* __restrict__ inv, int* __restrict__ tiles, int max_tiles, const TX* __restrict__ x, long xstride,
    TX* __restrict__ xs, int H, int nsplit, long sstride) {
  __shared__ __attribute__((aligned(16))) int s_topi[SMALL_MAX_SLOTS];
  __shared__ int s_src[SMALL_MAX_SLOTS];
  __shared__ float s_logit[SMALL_MAX_TOKENS * SMALL_MAX_E], s_part[ALIGN_THREADS];
  __shared__ int counts[SMALL_MAX_E], row0[SMALL_MAX_E], tile0[SMALL_MAX_E];
  __shared__ int total_tiles;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool writer = blockIdx.x == 0;
  const int slots = tokens * k;
  for (int e = tid; e < E; e += ALIGN_THREADS) counts[e] = 0;
  if (nsplit > 1) {
    // Add up the router GEMM's partial sums into LDS first, several threads per logit when there are few
    // logits, each with independent loads in flight.  Fixed summation order: every workgroup gets the same
    // bits, hence the same picks.
    if constexpr (sizeof(T) == 4) {
      const int total = tokens * E;
      const int parts = max(1, min(nsplit, ALIGN_THREADS / total));
      for (int wi = tid; wi < total * parts; wi += ALIGN_THREADS) {
        const int el = wi % total, part = wi / total;
        const float* p = (const float*)logits + (long)(el / E) * lstride + (el % E);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int sp = part;
        for (; sp + 3 * parts < nsplit; sp += 4 * parts) {
          a0 += p[(long)sp * sstride];
          a1 += p[(long)(sp + parts) * sstride];
          a2 += p[(long)(sp + 2 * parts) * sstride];
          a3 += p[(long)(sp + 3 * parts) * sstride];
        }
        for (; sp < nsplit; sp += parts) a0 += p[(long)sp * sstride];
        const float a = (a0 + a1) + (a2 + a3);
        if (parts == 1) s_logit[el] = a; else s_part[part * total + el] = a;
      }
      __syncthreads();
      if (parts > 1 && tid < total) {
        float a = 0.f;
        for (int q = 0; q < parts; ++q) a += s_part[q * total + tid];
        s_logit[tid] = a;
      }
      __syncthreads();
    }
  }
  if (wid < tokens) {
    float w;
    int e;
    if (nsplit > 1) wave_topk(s_logit + wid * E, E, k, renorm, lane, w, e);
    else wave_topk(logits + (long)wid * lstride, E, k, renorm, lane, w, e);
    if (lane < k) {
      s_topi[wid * k + lane] = e;
      if (writer) {
        topw[(long)wid * k + lane] = w;
        topi[(long)wid * k + lane] = e;
      }
    }
  }
  __syncthreads();
  if (tid < slots) atomicAdd(&counts[s_topi[tid]], 1);
  __syncthreads();
  if (tid < 64) {
    int rbase = 0, tbase = 0;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + tid;
      const int c = e < E ? counts[e] : 0;
      const int tl = (c + tile_rows - 1) / tile_rows;
      int rs = c, ts = tl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int r2 = __shfl_up(rs, o, 64), t2 = __shfl_up(ts, o, 64);
        if (tid >= o) { rs += r2; ts += t2; }
      }
      if (e < E) { row0[e] = rbase + rs - c; tile0[e] = tbase + ts - tl; }
      rbase += __shfl(rs, 63, 64);
      tbase += __shfl(ts, 63, 64);
    }
    if (tid == 0) total_tiles = tbase;
  }
  __syncthreads();
  {
    // position within the expert = number of earlier slots routed to it: the same in every workgroup.
    // Four threads per slot, each counting a quarter of the earlier slots, four at a time.
    const int slot = tid >> 2, part = tid & 3;
    int rank = 0, e = -1;
    if (slot < slots) {
      e = s_topi[slot];
      const int quads = (slot + 3) >> 2;                          // int4 groups that hold slots < slot
      for (int q = part; q < quads; q += 4) {
        const int4 v = *reinterpret_cast<const int4*>(&s_topi[4 * q]);
        const int base = 4 * q;
        rank += (v.x == e && base < slot) + (v.y == e && base + 1 < slot) + (v.z == e && base + 2 < slot) +
                (v.w == e && base + 3 < slot);
      }
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    if (slot < slots && part == 0) {
      const int pos = row0[e] + rank;
      s_src[pos] = slot / k;
      if (writer) {
        inv[slot] = pos;
        src_row[pos] = slot / k;
      }
    }
  }
  if (writer) {
    for (int e = tid; e < E; e += ALIGN_THREADS) {
      const int c = counts[e];
      const bool local = local_experts < 0 || (e >= expert_start && e < expert_start + local_experts);
      const int le = local_experts < 0 ? e : min(max(e - expert_start, 0), max(local_experts - 1, 0));
      int j = tile0[e];
      for (int r = 0; r < c; r += tile_rows, ++j) {
        if (j >= max_tiles) break;
        tiles[3 * j + 0] = le;
        tiles[3 * j + 1] = row0[e] + r;
        tiles[3 * j + 2] = local ? min(tile_rows, c - r) : 0;
      }
    }
    for (int j = total_tiles + tid; j < max_tiles; j += ALIGN_THREADS) {
      tiles[3 * j + 0] = 0;
      tiles[3 * j + 1] = 0;
      tiles[3 * j + 2] = 0;
    }
  }
  __syncthreads();
  using V = typename Vec8<TX>::type;
  const int vecs = H >> 3;
  for (int pos = blockIdx.x; pos < slots; pos += gridDim.x) {
    const TX* src = x + (long)s_src[pos] * xstride;
    TX* dst = xs + (long)pos * H;
    for (int v = tid; v < vecs; v += ALIGN_THREADS) st8<TX>(dst + 8 * v, ld8<TX>(src + 8 * v));
  }
}

// grid (token, 2048-column slab); y rows are in expert-sorted order
template <typename T>
__global__ __launch_bounds__(256) void moe_combine_kernel(T* __restrict__ out, long out_stride,
                                                          const T* __restrict__ y, long y_stride,
                                                          const float* __restrict__ topw, const int* __restrict__ topi,
                                                          const int* __restrict__ inv, int k, int H, int expert_start,
                                                          int local_experts, const T* __restrict__ bias) {
  // bias (nullable): [local experts][H], added to an expert's row (and rounded to T, as a biased linear
  // layer would) before the row is weighed
  using V = typename Vec8<T>::type;
  const int t = blockIdx.x;
  __shared__ float w[TOPK_MAX_K];
  __shared__ int pos[TOPK_MAX_K];
  __shared__ int loc[TOPK_MAX_K];
  if (threadIdx.x < k) {
    const int e = topi[(long)t * k + threadIdx.x];
    const bool local = local_experts < 0 || (e >= expert_start && e < expert_start + local_experts);
    w[threadIdx.x] = local ? topw[(long)t * k + threadIdx.x] : 0.f;
    pos[threadIdx.x] = local ? inv[(long)t * k + threadIdx.x] : -1;
    loc[threadIdx.x] = e - expert_start;
  }
  __syncthreads();
  const int h = blockIdx.y * 2048 + threadIdx.x * 8;
  if (h < H) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < k; ++j) {
      if (pos[j] < 0) continue;          // an expert another rank holds: its row was never computed here
      V v = ld8<T>(y + (long)pos[j] * y_stride + h);
      if (bias != nullptr) {
        const V b = ld8<T>(bias + (long)loc[j] * H + h);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (T)((float)v[i] + (float)b[i]);
      }
      const float wj = w[j];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += wj * (float)v[i];
    }
    V o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (T)acc[i];
    st8<T>(out + (long)t * out_stride + h, o);
  }
}

// gpt-oss experts: out[row][c] = (up + 1) * gate * sigmoid(alpha * gate) with gate = min(h[row][c] + b[e][c], limit)
// and up = clamp(h[row][ip + c] + b[e][ip + c], -limit, limit), for the sorted row of every (token, choice) slot
// whose expert e is held here.  grid (slots, ceil(ip / 2048)).
template <typename T>
__global__ __launch_bounds__(256) void bias_clamped_glu_kernel(T* __restrict__ out, long out_stride,
                                                               const T* __restrict__ h, long h_stride,
                                                               const T* __restrict__ bias, const int* __restrict__ topi,
                                                               const int* __restrict__ inv, int ip, int expert_start,
                                                               int local_experts, float limit, float alpha) {
  using V = typename Vec8<T>::type;
  const int slot = blockIdx.x;
  const int le = topi[slot] - expert_start;
  if (local_experts >= 0 && (le < 0 || le >= local_experts)) return;
  const long row = inv[slot];
  const int c = blockIdx.y * 2048 + threadIdx.x * 8;
  if (c >= ip) return;
  const V g = ld8<T>(h + row * h_stride + c), u = ld8<T>(h + row * h_stride + ip + c);
  const V bg = ld8<T>(bias + (long)le * 2 * ip + c), bu = ld8<T>(bias + (long)le * 2 * ip + ip + c);
  V o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float gate = fminf((float)(T)((float)g[i] + (float)bg[i]), limit);
    const float up = fminf(fmaxf((float)(T)((float)u[i] + (float)bu[i]), -limit), limit);
    o[i] = (T)((up + 1.f) * gate * fast_sigmoid(alpha * gate));
  }
  st8<T>(out + row * out_stride + c, o);
}

// One workgroup per token: the weighted sum of its k expert rows (rounded to T, as moe_combine writes it),
// plus the residual (updated in place), then RMSNorm of the sum.  COMBINE_NORM_ITERS * 8 * blockDim >= H.
constexpr int COMBINE_NORM_ITERS = 4;
template <typename T>
__global__ void moe_combine_add_rms_norm_kernel(T* __restrict__ out, long out_stride, const T* __restrict__ y,
                                                long y_stride, const float* __restrict__ topw,
                                                const int* __restrict__ topi, const int* __restrict__ inv, int k, int H,
                                                int expert_start, int local_experts, T* __restrict__ residual,
                                                const T* __restrict__ gamma, float eps) {
  using V = typename Vec8<T>::type;
  const int t = blockIdx.x;
  __shared__ float w[TOPK_MAX_K];
  __shared__ int pos[TOPK_MAX_K];
  __shared__ float red[17];
  if (threadIdx.x < k) {
    const int e = topi[(long)t * k + threadIdx.x];
    const bool local = local_experts < 0 || (e >= expert_start && e < expert_start + local_experts);
    w[threadIdx.x] = local ? topw[(long)t * k + threadIdx.x] : 0.f;
    pos[threadIdx.x] = local ? inv[(long)t * k + threadIdx.x] : -1;
  }
  __syncthreads();
  T* rr = residual + (long)t * H;
  V vals[COMBINE_NORM_ITERS];
  float ss = 0.f;
  const int nvec = H >> 3;
#pragma unroll
  for (int it = 0; it < COMBINE_NORM_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      const V r = ld8<T>(rr + 8 * i);
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < k; ++j) {
        if (pos[j] < 0) continue;
        const V v = ld8<T>(y + (long)pos[j] * y_stride + 8 * i);
        const float wj = w[j];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += wj * (float)v[c];
      }
      V v;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v[c] = (T)((float)(T)acc[c] + (float)r[c]);
        const float f = (float)v[c];
        ss += f * f;
      }
      st8<T>(rr + 8 * i, v);
      vals[it] = v;
    }
  }
  ss = block_sum(ss, red);
  const float rinv = rsqrtf(ss / (float)H + eps);
#pragma unroll
  for (int it = 0; it < COMBINE_NORM_ITERS; ++it) {
    const int i = threadIdx.x + it * blockDim.x;
    if (i < nvec) {
      const V g = ld8<T>(gamma + 8 * i);
      V o;
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = (T)((float)vals[it][c] * rinv * (float)g[c]);
      st8<T>(out + (long)t * out_stride + 8 * i, o);
    }
  }
}

}  // namespace

VDAMD_API int vdamd_moe_topk_grouped(const void* logits, long stride, const float* bias, float* topw, int* topi,
                                     int tokens, int E, int k, int n_group, int topk_group, int sigmoid,
                                     int group_top2, int renorm, float scale, int dtype, hipStream_t s) {
  if (tokens <= 0) return 0;
  if (E < k || E > 64 * TOPK_MAX_VPL || k < 1 || k > TOPK_MAX_K) return -1;
  if (n_group < 1 || n_group > 64 || E % n_group || topk_group < 1) return -1;
  const unsigned blocks = (unsigned)((tokens + 3) / 4);
#define TOPKG(T_)                                                                                              \
  hipLaunchKernelGGL((moe_topk_grouped_kernel<T_>), dim3(blocks), dim3(256), 0, s, (const T_*)logits, stride, bias, \
                     topw, topi, tokens, E, k, n_group, topk_group, sigmoid, group_top2, renorm, scale)
  if (dtype == DT_BF16) TOPKG(bf16_t);
  else if (dtype == DT_F16) TOPKG(f16_t);
  else if (dtype == DT_F32) TOPKG(float);
  else return -2;
#undef TOPKG
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_moe_topk(const void* logits, long stride, float* topw, int* topi, int tokens, int E, int k,
                             int renorm, int dtype, hipStream_t s) {
  if (tokens <= 0) return 0;
  if (E < k || E > 64 * TOPK_MAX_VPL || k < 1 || k > TOPK_MAX_K) return -1;
  const unsigned blocks = (unsigned)((tokens + 3) / 4);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((moe_topk_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)logits, stride, topw,
                       topi, tokens, E, k, renorm);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL((moe_topk_kernel<f16_t>), dim3(blocks), dim3(256), 0, s, (const f16_t*)logits, stride, topw,
                       topi, tokens, E, k, renorm);
  else if (dtype == DT_F32)
    hipLaunchKernelGGL((moe_topk_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)logits, stride, topw,
                       topi, tokens, E, k, renorm);
  else
    return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}

// tiles must hold max_tiles >= slots / tile_rows + E entries of 3 ints
VDAMD_API int vdamd_moe_align(const int* topi, int slots, int k, int E, int tile_rows, int expert_start,
                              int local_experts, int* src_row, int* inv, int* tiles, int max_tiles, hipStream_t s) {
  if (slots <= 0) return 0;
  if (E < 1 || E > ALIGN_MAX_E || tile_rows < 1 || k < 1 || max_tiles < slots / tile_rows + E) return -1;
  hipLaunchKernelGGL(moe_align_kernel, dim3(1), dim3(ALIGN_THREADS), 0, s, topi, slots, k, E, tile_rows, expert_start,
                     local_experts, src_row, inv, tiles, max_tiles);
  HIP_LAUNCH_CHECK();
  return 0;
}

VDAMD_API int vdamd_moe_combine(void* out, long out_stride, const void* y, long y_stride, const float* topw,
                                const int* topi, const int* inv, int tokens, int k, int H, int expert_start,
                                int local_experts, const void* bias, int dtype, hipStream_t s) {
  if (tokens <= 0) return 0;
  if (H % 8 || k < 1 || k > TOPK_MAX_K) return -1;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((moe_combine_kernel<bf16_t>), dim3(tokens, (H + 2047) / 2048), dim3(256), 0, s, (bf16_t*)out, out_stride,
                       (const bf16_t*)y, y_stride, topw, topi, inv, k, H, expert_start, local_experts,
                       (const bf16_t*)bias);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL((moe_combine_kernel<f16_t>), dim3(tokens, (H + 2047) / 2048), dim3(256), 0, s, (f16_t*)out, out_stride,
                       (const f16_t*)y, y_stride, topw, topi, inv, k, H, expert_start, local_experts,
                       (const f16_t*)bias);
  else
    return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}

// top-k + slot sort + tile table + row gather for tokens <= 16 (see moe_route_small_kernel).  xs holds
// tokens * k rows of H; tiles holds max_tiles >= tokens * k / tile_rows + E entries; x and xs are 16-bit.
// nsplit > 1: logits are fp32 partial sums [nsplit][tokens][E] (sstride elements apart), added up here.
VDAMD_API int vdamd_moe_route_small(const void* logits, long lstride, int tokens, int E, int k, int renorm,
                                    int tile_rows, int expert_start, int local_experts, float* topw, int* topi,
                                    int* src_row, int* inv, int* tiles, int max_tiles, const void* x, long xstride,
                                    void* xs, int H, int logits_dtype, int x_dtype, int nsplit, long sstride,
                                    hipStream_t s) {
  if (tokens <= 0) return 0;
  if (nsplit < 1 || (nsplit > 1 && logits_dtype != DT_F32)) return -3;
  if (tokens > SMALL_MAX_TOKENS || E < k || E > SMALL_MAX_E || k < 1 || k > TOPK_MAX_K || tile_rows < 1 || H % 8 ||
      max_tiles < tokens * k / tile_rows + E)
    return -1;
  const dim3 grid((unsigned)(tokens * k)), block(ALIGN_THREADS);
#define VDAMD_ROUTE_SMALL(TL, TXX)                                                                                  \
  hipLaunchKernelGGL((moe_route_small_kernel<TL, TXX>), grid, block, 0, s, (const TL*)logits, lstride, tokens, E, k, \
                     renorm, tile_rows, expert_start, local_experts, topw, topi, src_row, inv, tiles, max_tiles,    \
                     (const TXX*)x, xstride, (TXX*)xs, H, nsplit, sstride)
  if (x_dtype == DT_BF16) {
    if (logits_dtype == DT_BF16) VDAMD_ROUTE_SMALL(bf16_t, bf16_t);
    else if (logits_dtype == DT_F32) VDAMD_ROUTE_SMALL(float, bf16_t);
    else return -2;
  } else if (x_dtype == DT_F16) {
    if (logits_dtype == DT_F16) VDAMD_ROUTE_SMALL(f16_t, f16_t);
    else if (logits_dtype == DT_F32) VDAMD_ROUTE_SMALL(float, f16_t);
    else return -2;
  } else {
    return -2;
  }
#undef VDAMD_ROUTE_SMALL
  HIP_LAUNCH_CHECK();
  return 0;
}

// residual[t] += sum_j w[t][j] * y[slot (t, j)] (in place), out[t] = RMSNorm(residual[t]) * gamma
VDAMD_API int vdamd_moe_combine_add_rms_norm(void* out, long out_stride, const void* y, long y_stride,
                                             const float* topw, const int* topi, const int* inv, int tokens, int k,
                                             int H, int expert_start, int local_experts, void* residual,
                                             const void* gamma, float eps, int dtype, hipStream_t s) {
  if (tokens <= 0) return 0;
  if (H % 8 || H > 8 * 1024 * COMBINE_NORM_ITERS || k < 1 || k > TOPK_MAX_K || residual == nullptr) return -1;
  int threads = ((H >> 3) + 63) / 64 * 64;            // one pass where the row fits: these launches are latency-bound
  if (tokens > 64) threads = (threads / COMBINE_NORM_ITERS + 63) / 64 * 64;
  if (threads > 1024) threads = 1024;
  if (threads < 64) threads = 64;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((moe_combine_add_rms_norm_kernel<bf16_t>), dim3(tokens), dim3(threads), 0, s, (bf16_t*)out,
                       out_stride, (const bf16_t*)y, y_stride, topw, topi, inv, k, H, expert_start, local_experts,
                       (bf16_t*)residual, (const bf16_t*)gamma, eps);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL((moe_combine_add_rms_norm_kernel<f16_t>), dim3(tokens), dim3(threads), 0, s, (f16_t*)out,
                       out_stride, (const f16_t*)y, y_stride, topw, topi, inv, k, H, expert_start, local_experts,
                       (f16_t*)residual, (const f16_t*)gamma, eps);
  else
    return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}

// out[row] = clamped GLU of h[row] + bias[expert of row] for the sorted rows of all `slots` (token, choice) pairs;
// h is [rows][2 ip] = [gate | up], bias [local experts][2 ip], out [rows][out_stride >= ip].
VDAMD_API int vdamd_bias_clamped_glu(void* out, long out_stride, const void* h, long h_stride, const void* bias,
                                     const int* topi, const int* inv, int slots, int ip, int expert_start,
                                     int local_experts, float limit, float alpha, int dtype, hipStream_t s) {
  if (slots <= 0) return 0;
  if (ip % 8 || h_stride % 8 || out_stride % 8) return -1;
  const dim3 grid(slots, (ip + 2047) / 2048);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL((bias_clamped_glu_kernel<bf16_t>), grid, dim3(256), 0, s, (bf16_t*)out, out_stride,
                       (const bf16_t*)h, h_stride, (const bf16_t*)bias, topi, inv, ip, expert_start, local_experts,
                       limit, alpha);
  else if (dtype == DT_F16)
    hipLaunchKernelGGL((bias_clamped_glu_kernel<f16_t>), grid, dim3(256), 0, s, (f16_t*)out, out_stride,
                       (const f16_t*)h, h_stride, (const f16_t*)bias, topi, inv, ip, expert_start, local_experts,
                       limit, alpha);
  else
    return -2;
  HIP_LAUNCH_CHECK();
  return 0;
}
