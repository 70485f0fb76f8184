// HBM read-bandwidth lab: how the per-instruction address pattern, the number of loads in
// flight and the grid shape change the achieved bandwidth of a pure streaming read.
//   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

// pattern 0: every wave instruction reads 1 KiB contiguous; a workgroup walks a contiguous chunk
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_linear(const f4* __restrict__ p, size_t n16, float* sink, int chunk16) {
  // chunk16: 16-byte elements per workgroup chunk
  f4 acc = {0, 0, 0, 0};
  const size_t nchunks = n16 / chunk16;
  for (size_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const f4* b = p + ch * chunk16;
    for (int i = threadIdx.x; i < chunk16; i += 256 * U) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (NT) v[u] = __builtin_nontemporal_load(b + i + 256 * u); else v[u] = b[i + 256 * u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
  }
  if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}

// pattern 1: MFMA A-operand shape.  A wave owns 16 rows of a [N][K] bf16 matrix; one instruction reads
// 64 contiguous bytes of each row; 4 waves of a workgroup interleave over 256-byte column blocks.
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_rows16(const f4* __restrict__ p, int N, int K, float* sink, int pad16 = 0) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const size_t row16 = (size_t)K / 8 + pad16;       // 16-byte elements per row (pad16: row padding)
  f4 acc = {0, 0, 0, 0};
  for (int tile = blockIdx.x; tile < N / 16; tile += gridDim.x) {
    const f4* r = p + ((size_t)tile * 16 + c) * row16 + g;
    const int nb = K / 128;                          // 256-byte blocks per row
    for (int kb = wid; kb < nb; kb += 4 * U) {
      f4 v[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f4* a = r + (size_t)(kb + 4 * u) * 16 + 4 * s;
          if (kb + 4 * u < nb) { if (NT) v[u][s] = __builtin_nontemporal_load(a); else v[u][s] = *a; }
          else v[u][s] = f4{0, 0, 0, 0};
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc += v[u][s];
    }
  }
  if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}

// pattern 3: what a split-K GEMM with a 64-deep K tile issues: every wave owns 16 rows and walks K alone,
// SEG bytes of each row per visit (SEG = 128: two instructions), DEPTH visits in flight.
template <int SEG, int DEPTH>
__global__ __launch_bounds__(512) void k_rows16_solo(const f4* __restrict__ p, int N, int K, float* sink, int pad16,
                                                     int ksplit) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const size_t row16 = (size_t)K / 8 + pad16;
  constexpr int PER = SEG / 64;                        // instructions per visit
  f4 acc = {0, 0, 0, 0};
  const int tiles = N / 128;                           // 8 waves x 16 rows
  const int id = blockIdx.x, ks = id % ksplit, tn = id / ksplit;
  if (tn >= tiles) return;
  const int visits = K * 2 / SEG;                      // along the whole row
  const int v_lo = visits * ks / ksplit, v_hi = visits * (ks + 1) / ksplit;
  const f4* r = p + ((size_t)tn * 128 + wid * 16 + c) * row16 + g;
  for (int v = v_lo; v < v_hi; v += DEPTH) {
    f4 t[DEPTH][PER];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int s = 0; s < PER; ++s)
        t[d][s] = (v + d < v_hi) ? r[(size_t)(v + d) * (SEG / 16) + 4 * s] : f4{0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int s = 0; s < PER; ++s) acc += t[d][s];
  }
  if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}

// pattern 2: one row per wave instruction (1 KiB contiguous of ONE row), wave walks its row; 16 rows per wave
// are visited one after the other (what a direct-to-LDS staging of W would issue).
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_row1k(const f4* __restrict__ p, int N, int K, float* sink) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t row16 = (size_t)K / 8;
  f4 acc = {0, 0, 0, 0};
  const int nseg = K / 512;                          // 1 KiB segments per row
  for (int tile = blockIdx.x; tile < N / 16; tile += gridDim.x) {
    // workgroup: 16 rows x nseg segments; wave takes segments wid, wid+4, ...; for each, the 16 rows
    for (int sg = wid; sg < nseg; sg += 4) {
      for (int r0 = 0; r0 < 16; r0 += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const f4* a = p + ((size_t)tile * 16 + r0 + u) * row16 + (size_t)sg * 64 + lane;
          if (NT) v[u] = __builtin_nontemporal_load(a); else v[u] = *a;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
      }
    }
  }
  if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}

template <typename F>
float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int N = 57344, K = 8192;                     // 70B gate_up, 940 MB bf16
  const size_t bytes = (size_t)N * K * 2;
  // two copies so that consecutive launches never hit the memory-side cache
  f4 *p[3]; float* sink;
  for (int i = 0; i < 3; ++i) { CK(hipMalloc(&p[i], bytes + (size_t)N * 8192)); CK(hipMemset(p[i], 1, bytes)); }
  CK(hipMalloc(&sink, 64));
  int which = 0;
  auto rot = [&]() { which = (which + 1) % 3; return p[which]; };
  printf("matrix %d x %d bf16 = %.0f MB\n", N, K, bytes / 1e6);
#define RUN(name, expr) { float ms = timeit([&]() { const f4* q = rot(); expr; }); \
    printf("%-44s %8.1f us  %5.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9); fflush(stdout); }
  for (int grid : {256, 512, 1024, 2048, 4096}) {
    char nm[128];
    snprintf(nm, 128, "linear U=4 chunk=64K grid=%d", grid);
    RUN(nm, (k_linear<4, false><<<grid, 256>>>(q, bytes / 16, sink, 4096)));
    snprintf(nm, 128, "linear U=8 chunk=256K grid=%d", grid);
    RUN(nm, (k_linear<8, false><<<grid, 256>>>(q, bytes / 16, sink, 16384)));
    snprintf(nm, 128, "linear U=8 nt chunk=256K grid=%d", grid);
    RUN(nm, (k_linear<8, true><<<grid, 256>>>(q, bytes / 16, sink, 16384)));
  }
  for (int pad16 : {0, 4, 8, 16, 24, 64, 72, 264}) {
    char nm[128];
    for (int grid : {1024, 3584}) {
      snprintf(nm, 128, "rows16x64B U=2 pad=%dB grid=%d", pad16 * 16, grid);
      RUN(nm, (k_rows16<2, false><<<grid, 256>>>(q, N, K, sink, pad16)));
      snprintf(nm, 128, "rows16x64B U=2 nt pad=%dB grid=%d", pad16 * 16, grid);
      RUN(nm, (k_rows16<2, true><<<grid, 256>>>(q, N, K, sink, pad16)));
      snprintf(nm, 128, "rows16x64B U=4 nt pad=%dB grid=%d", pad16 * 16, grid);
      RUN(nm, (k_rows16<4, true><<<grid, 256>>>(q, N, K, sink, pad16)));
    }
  }
  {
    char nm[128];
    const int tiles = N / 128;
    for (int ksplit : {1, 2, 4}) {
      snprintf(nm, 128, "solo 16 rows/wave, 128 B/visit, depth 4, ksplit=%d pad=256B", ksplit);
      RUN(nm, (k_rows16_solo<128, 4><<<tiles * ksplit, 512>>>(q, N, K, sink, 16, ksplit)));
      snprintf(nm, 128, "solo 16 rows/wave, 256 B/visit, depth 4, ksplit=%d pad=256B", ksplit);
      RUN(nm, (k_rows16_solo<256, 4><<<tiles * ksplit, 512>>>(q, N, K, sink, 16, ksplit)));
      snprintf(nm, 128, "solo 16 rows/wave, 512 B/visit, depth 2, ksplit=%d pad=256B", ksplit);
      RUN(nm, (k_rows16_solo<512, 2><<<tiles * ksplit, 512>>>(q, N, K, sink, 16, ksplit)));
      snprintf(nm, 128, "solo 16 rows/wave, 1024 B/visit, depth 2, ksplit=%d pad=256B", ksplit);
      RUN(nm, (k_rows16_solo<1024, 2><<<tiles * ksplit, 512>>>(q, N, K, sink, 16, ksplit)));
    }
  }
  for (int grid : {3584}) {
    char nm[128];
    snprintf(nm, 128, "rows16x64B U=1 grid=%d", grid);
    RUN(nm, (k_rows16<1, false><<<grid, 256>>>(q, N, K, sink)));
    snprintf(nm, 128, "rows16x64B U=2 grid=%d", grid);
    RUN(nm, (k_rows16<2, false><<<grid, 256>>>(q, N, K, sink)));
    snprintf(nm, 128, "rows16x64B U=4 grid=%d", grid);
    RUN(nm, (k_rows16<4, false><<<grid, 256>>>(q, N, K, sink)));
    snprintf(nm, 128, "rows16x64B U=2 nt grid=%d", grid);
    RUN(nm, (k_rows16<2, true><<<grid, 256>>>(q, N, K, sink)));
    snprintf(nm, 128, "row1KiB U=4 grid=%d", grid);
    RUN(nm, (k_row1k<4, false><<<grid, 256>>>(q, N, K, sink)));
    snprintf(nm, 128, "row1KiB U=8 grid=%d", grid);
    RUN(nm, (k_row1k<8, false><<<grid, 256>>>(q, N, K, sink)));
    snprintf(nm, 128, "row1KiB U=16 grid=%d", grid);
    RUN(nm, (k_row1k<16, false><<<grid, 256>>>(q, N, K, sink)));
  }
  return 0;
}
