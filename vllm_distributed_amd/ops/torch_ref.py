"""Plain-PyTorch implementations of every op in ``ops``.

They define the semantics (the HIP kernels are tested against these in fp32)
and are the execution path on CPU.  KV-cache layouts are the ones the CDNA4
kernels consume as MFMA operands straight from global memory:

* ``k_cache[num_blocks, num_kv_heads, block_size, head_dim]``  - a key row is
  contiguous in ``head_dim`` (the QK^T reduction axis);
* ``v_cache[num_blocks, num_kv_heads, head_dim // 16, block_size // 8, 16, 8]``  -
  V^T in MFMA-fragment order: the 8 keys of one ``head_dim`` row are a 16-byte
  chunk (the PV reduction axis), 16 rows of a key group are adjacent, and the
  key groups of a block follow each other inside a 16-row slab.  A wave's
  16-row x 32-key fragment is one contiguous 1 KiB run, and a decode step's new
  token lands in ``head_dim / 16`` runs of 256 bytes per head instead of one
  element in each of ``head_dim`` separate rows.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------- norms
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (y * weight.float()).to(x.dtype)


def fused_add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                       eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``r = x + residual``; returns ``(rms_norm(r), r)`` (r rounded to x.dtype
    before the norm, as the fused kernel stores it)."""
    r = (x.float() + residual.float()).to(x.dtype)
    return rms_norm(r, weight, eps), r


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
               eps: float) -> torch.Tensor:
    return F.layer_norm(x.float(), (x.shape[-1],), weight.float(),
                        None if bias is None else bias.float(), eps).to(x.dtype)


# -------------------------------------------------------------- activation
def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    return (F.silu(x[..., :d].float()) * x[..., d:].float()).to(x.dtype)


# ------------------------------------------------------------------ rotary
def build_cos_sin_cache(head_dim: int, max_pos: int, theta: float,
                        rope_scaling: Optional[dict] = None,
                        rotary_dim: Optional[int] = None) -> torch.Tensor:
    """``[max_pos, rotary_dim]`` fp32: cos in the first half, sin in the second."""
    rd = rotary_dim or head_dim
    inv_freq = 1.0 / (theta ** (torch.arange(0, rd, 2, dtype=torch.float64) / rd))
    if rope_scaling:
        rtype = rope_scaling.get("rope_type", rope_scaling.get("type"))
        if rtype == "llama3":
            factor = rope_scaling.get("factor", 8.0)
            lo = rope_scaling.get("low_freq_factor", 1.0)
            hi = rope_scaling.get("high_freq_factor", 4.0)
            old = rope_scaling.get("original_max_position_embeddings", 8192)
            wavelen = 2 * math.pi / inv_freq
            smooth = ((old / wavelen) - lo) / (hi - lo)
            scaled = torch.where(wavelen > old / lo, inv_freq / factor,
                                 torch.where(wavelen < old / hi, inv_freq,
                                             (1 - smooth) * inv_freq / factor + smooth * inv_freq))
            inv_freq = scaled
        elif rtype == "linear":
            inv_freq = inv_freq / rope_scaling.get("factor", 1.0)
        elif rtype == "dynamic":
            # NTK-aware base stretch, fixed for the configured context (the table is built once)
            factor = rope_scaling.get("factor", 1.0)
            old = rope_scaling.get("original_max_position_embeddings") or max_pos
            if max_pos > old:
                base = theta * ((factor * max_pos / old) - (factor - 1)) ** (rd / (rd - 2))
                inv_freq = 1.0 / (base ** (torch.arange(0, rd, 2, dtype=torch.float64) / rd))
        elif rtype == "yarn":
            # interpolate the low-frequency dims by `factor`, keep the high-frequency
            # ones, blend linearly in between; cos/sin carry the attention temperature
            factor = float(rope_scaling.get("factor", 1.0))
            old = rope_scaling.get("original_max_position_embeddings") or max_pos
            beta_fast = float(rope_scaling.get("beta_fast") or 32)
            beta_slow = float(rope_scaling.get("beta_slow") or 1)

            def corr_dim(rot: float) -> float:
                return rd * math.log(old / (rot * 2 * math.pi)) / (2 * math.log(theta))
            low = max(math.floor(corr_dim(beta_fast)), 0)
            high = min(math.ceil(corr_dim(beta_slow)), rd - 1)
            if rope_scaling.get("truncate") is False:
                low, high = max(corr_dim(beta_fast), 0), min(corr_dim(beta_slow), rd - 1)
            ramp = ((torch.arange(rd // 2, dtype=torch.float64) - low) / max(high - low, 1e-3)).clamp(0, 1)
            keep = 1.0 - ramp
            inv_freq = inv_freq / factor * (1 - keep) + inv_freq * keep
            mscale = rope_scaling.get("attention_factor")
            if mscale is None:
                ms, ms_all = rope_scaling.get("mscale"), rope_scaling.get("mscale_all_dim")

                def get_mscale(scale: float, m: float = 1.0) -> float:
                    return 1.0 if scale <= 1 else 0.1 * m * math.log(scale) + 1.0
                mscale = get_mscale(factor, ms) / get_mscale(factor, ms_all) if ms and ms_all else get_mscale(factor)
            t = torch.arange(max_pos, dtype=torch.float64)
            freqs = torch.outer(t, inv_freq)
            return (torch.cat([freqs.cos(), freqs.sin()], dim=-1) * float(mscale)).float()
        elif rtype in ("longrope", "su"):
            # LongRoPE: one divisor per frequency, a "short" set for sequences inside the original context and a
            # "long" one beyond it, and a temperature on cos / sin.  transformers switches sets when a sequence
            # grows past the original context (its cached keys keep the other rotation); here the table is built
            # once for the served context: the long set throughout when ``max_pos`` exceeds the original context,
            # the short one otherwise - what a sequence longer / shorter than that gets there.
            old = int(rope_scaling.get("original_max_position_embeddings") or max_pos)
            full = int(rope_scaling.get("max_position_embeddings") or max_pos)
            factor = float(rope_scaling.get("factor") or full / old)
            which = "long_factor" if max_pos > old else "short_factor"
            if not rope_scaling.get(which):
                raise ValueError(f"LongRoPE scaling needs short_factor and long_factor lists ({which} is missing)")
            ext = rope_scaling[which]
            if len(ext) != rd // 2:
                raise ValueError(f"LongRoPE lists {len(ext)} factors for {rd // 2} frequencies")
            inv_freq = inv_freq / torch.tensor(ext, dtype=torch.float64)
            mscale = rope_scaling.get("attention_factor")
            if mscale is None:
                mscale = 1.0 if factor <= 1.0 else math.sqrt(1 + math.log(factor) / math.log(old))
            t = torch.arange(max_pos, dtype=torch.float64)
            freqs = torch.outer(t, inv_freq)
            return (torch.cat([freqs.cos(), freqs.sin()], dim=-1) * float(mscale)).float()
        elif rtype not in (None, "default"):
            raise ValueError(f"rope scaling type {rtype!r} is not supported")
    t = torch.arange(max_pos, dtype=torch.float64)
    freqs = torch.outer(t, inv_freq)
    return torch.cat([freqs.cos(), freqs.sin()], dim=-1).float()


def rotary_embedding(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                     cos_sin_cache: torch.Tensor, head_dim: int) -> None:
    """NeoX-style rotation in place on ``q[T, Hq*D]`` and ``k[T, Hkv*D]``."""
    rd = cos_sin_cache.shape[-1]
    cs = cos_sin_cache.index_select(0, positions.long())
    cos, sin = cs[:, :rd // 2].unsqueeze(1), cs[:, rd // 2:].unsqueeze(1)
    for t in (q, k):
        v = t.view(t.shape[0], -1, head_dim)
        x1 = v[..., :rd // 2].float()
        x2 = v[..., rd // 2:rd].float()
        o1 = x1 * cos - x2 * sin
        o2 = x2 * cos + x1 * sin
        v[..., :rd // 2] = o1.to(t.dtype)
        v[..., rd // 2:rd] = o2.to(t.dtype)


# ---------------------------------------------------------------- KV cache
FP8_KV = torch.float8_e4m3fn
FP8_MAX = 448.0
V_GROUP = 8          # keys per contiguous chunk of the V cache
V_ROWS = 16          # head_dim rows per slab (the M of one MFMA fragment)


def v_cache_shape(num_blocks: int, num_kv_heads: int, head_dim: int, block_size: int) -> Tuple[int, ...]:
    if block_size % V_GROUP != 0:
        raise ValueError(f"block_size must be a multiple of {V_GROUP}")
    if head_dim % V_ROWS != 0:
        raise ValueError(f"head_dim must be a multiple of {V_ROWS}")
    return (num_blocks, num_kv_heads, head_dim // V_ROWS, block_size // V_GROUP, V_ROWS, V_GROUP)


def k_cache_shape(num_blocks: int, num_kv_heads: int, head_dim: int, block_size: int) -> Tuple[int, ...]:
    return (num_blocks, num_kv_heads, block_size, head_dim)


def _to_cache(x: torch.Tensor, cache: torch.Tensor, scale: Optional[torch.Tensor]) -> torch.Tensor:
    """``x[T, Hkv, D]`` in the cache's element type; an e4m3 cache stores ``x / scale[head]``, saturated."""
    if cache.dtype != FP8_KV:
        return x.to(cache.dtype)
    xf = x.float()
    if scale is not None:
        xf = xf / scale.float().view(1, -1, 1)
    return xf.clamp_(-FP8_MAX, FP8_MAX).to(FP8_KV)


def reshape_and_cache(k: torch.Tensor, v: torch.Tensor, k_cache: torch.Tensor,
                      v_cache: torch.Tensor, slot_mapping: torch.Tensor, kv_scales=None) -> None:
    """Scatter ``k,v[T, Hkv, D]`` into the paged caches; slot < 0 is skipped."""
    bs = k_cache.shape[2]
    valid = slot_mapping >= 0
    slots = slot_mapping[valid].long()
    blk, off = slots // bs, slots % bs
    ks, vs = kv_scales if kv_scales is not None else (None, None)
    if k_cache.dtype == FP8_KV:      # index assignment is not implemented for 8-bit floats: go through the bytes
        k_cache.view(torch.uint8)[blk, :, off, :] = _to_cache(k[valid], k_cache, ks).view(torch.uint8)
        vv = _to_cache(v[valid], v_cache, vs).view(torch.uint8)
        v_cache.view(torch.uint8)[blk, :, :, off // V_GROUP, :, off % V_GROUP] = vv.unflatten(-1, (-1, V_ROWS))
        return
    k_cache[blk, :, off, :] = k[valid].to(k_cache.dtype)
    v_cache[blk, :, :, off // V_GROUP, :, off % V_GROUP] = v[valid].to(v_cache.dtype).unflatten(-1, (-1, V_ROWS))


def rope_and_cache(positions: torch.Tensor, qkv: torch.Tensor, cos_sin_cache: torch.Tensor,
                   k_cache: torch.Tensor, v_cache: torch.Tensor, slot_mapping: torch.Tensor,
                   num_heads: int, num_kv_heads: int, head_dim: int, q_norm_weight=None, k_norm_weight=None,
                   norm_eps: float = 0.0, kv_scales=None) -> torch.Tensor:
    """Rotate q and k inside the packed ``qkv[T, (Hq+2Hkv)*D]``, write k/v to
    the cache, return the rotated ``q[T, Hq*D]`` (a view of qkv).  With norm
    weights each q and k head is RMS-normalised first."""
    qs, ks = num_heads * head_dim, num_kv_heads * head_dim
    q, k, v = qkv[:, :qs], qkv[:, qs:qs + ks], qkv[:, qs + ks:]
    if q_norm_weight is not None:
        for part, w in ((q, q_norm_weight), (k, k_norm_weight)):
            h = part.reshape(part.shape[0], -1, head_dim).float()
            h = h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + norm_eps) * w.float()
            part.copy_(h.reshape(part.shape).to(part.dtype))
    rotary_embedding(positions, q, k, cos_sin_cache, head_dim)
    reshape_and_cache(k.reshape(-1, num_kv_heads, head_dim), v.reshape(-1, num_kv_heads, head_dim),
                      k_cache, v_cache, slot_mapping, kv_scales)
    return q


def alibi_slopes(num_heads: int, kind: str = "bloom", bias_max: float = 8.0) -> torch.Tensor:
    """Per-head slopes of the linear position bias (fp32 ``[num_heads]``).  ``bloom`` (also Falcon-RW):
    a geometric sequence over the largest power of two not above the head count, the remaining heads
    taking every other step of the next finer sequence.  ``mpt``: the sequence over the next power of
    two at or above the head count with exponent step ``bias_max / n``, odd entries first when the
    count is not a power of two."""
    if kind == "mpt":
        n = 2 ** math.ceil(math.log2(num_heads))
        s = 1.0 / torch.pow(2.0, torch.arange(1, n + 1, dtype=torch.float32) * (bias_max / n))
        if n != num_heads:
            s = torch.cat([s[1::2], s[::2]])[:num_heads]
        return s.contiguous()
    n = 2 ** math.floor(math.log2(num_heads))
    base = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
    s = torch.pow(torch.tensor(base, dtype=torch.float32), torch.arange(1, n + 1, dtype=torch.float32))
    if n != num_heads:
        extra = 2.0 ** (-(2.0 ** -(math.log2(2 * n) - 3)))
        rest = min(n, num_heads - n)
        s = torch.cat([s, torch.pow(torch.tensor(extra, dtype=torch.float32),
                                    torch.arange(1, 1 + 2 * rest, 2, dtype=torch.float32))])
    return s.contiguous()


def _gather_kv(k_cache, v_cache, block_table_row, seq_len, kv_scales=None):
    bs = k_cache.shape[2]
    nb = (seq_len + bs - 1) // bs
    blocks = block_table_row[:nb].long()
    if k_cache.dtype == FP8_KV:
        k = k_cache.view(torch.uint8)[blocks].view(FP8_KV).float()
        v = v_cache.view(torch.uint8)[blocks].view(FP8_KV).float()
        if kv_scales is not None:
            k = k * kv_scales[0].float().view(1, -1, 1, 1)
            v = v * kv_scales[1].float().view(1, -1, 1, 1, 1, 1)
    else:
        k = k_cache[blocks]                   # [nb, H, bs, D]
        v = v_cache[blocks]                   # [nb, H, D/16, bs/8, 16, 8]
    k = k.permute(1, 0, 2, 3).reshape(k.shape[1], nb * bs, -1)[:, :seq_len]
    v = v.permute(1, 0, 3, 5, 2, 4).reshape(v.shape[1], nb * bs, -1)[:, :seq_len]
    return k, v                               # [H, S, D]


def paged_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                    block_table: torch.Tensor, query_start_loc: torch.Tensor,
                    seq_lens: torch.Tensor, scale: float, kv_scales=None, window: int = 0,
                    softcap: float = 0.0, sinks=None, alibi_slopes=None) -> torch.Tensor:
    """Causal attention of each sequence's query chunk against its paged KV
    (which already contains the chunk).  ``q[T, Hq, D]``; sequence ``i`` owns
    query rows ``query_start_loc[i]:query_start_loc[i+1]`` which are the last
    ``q_len`` positions of its ``seq_lens[i]`` tokens.  ``window > 0``: a query at
    position ``p`` sees keys ``p - window + 1 .. p`` only (sliding-window attention).
    ``softcap > 0``: scores are squashed to ``softcap * tanh(score / softcap)`` (Gemma-2).
    ``sinks [Hq]``: one extra score per head that joins the softmax and is then dropped (attention sinks).
    ``alibi_slopes [Hq]``: ``slope * (key position - query position)`` is added to the scores."""
    out = torch.empty_like(q)
    hq = q.shape[1]
    hkv = k_cache.shape[1]
    g = hq // hkv
    qsl = query_start_loc.tolist()
    sls = seq_lens.tolist()
    for i, sl in enumerate(sls):
        s, e = qsl[i], qsl[i + 1]
        ql = e - s
        if ql == 0:
            continue
        k, v = _gather_kv(k_cache, v_cache, block_table[i], sl, kv_scales)
        k = k.float().repeat_interleave(g, dim=0)
        v = v.float().repeat_interleave(g, dim=0)
        qi = q[s:e].float().transpose(0, 1)                     # [Hq, ql, D]
        scores = torch.matmul(qi, k.transpose(1, 2)) * scale    # [Hq, ql, sl]
        if softcap > 0:
            scores = torch.tanh(scores / softcap) * softcap
        qpos = torch.arange(sl - ql, sl, device=q.device).unsqueeze(1)
        kpos = torch.arange(sl, device=q.device).unsqueeze(0)
        if alibi_slopes is not None:
            scores = scores + alibi_slopes.float().view(-1, 1, 1) * (kpos - qpos).float()
        scores = scores.masked_fill(kpos > qpos, float("-inf"))
        if window > 0:
            scores = scores.masked_fill(kpos <= qpos - window, float("-inf"))
        if sinks is not None:
            col = sinks.float().view(-1, 1, 1).expand(hq, ql, 1)
            probs = torch.softmax(torch.cat([scores, col], dim=-1), -1)[..., :-1]
        else:
            probs = torch.softmax(scores, -1)
        out[s:e] = torch.matmul(probs, v).transpose(0, 1).to(q.dtype)
    return out


def paged_attention_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                           block_table: torch.Tensor, seq_lens: torch.Tensor,
                           scale: float, kv_scales=None, window: int = 0, softcap: float = 0.0,
                           sinks=None, alibi_slopes=None) -> torch.Tensor:
    b = q.shape[0]
    qsl = torch.arange(b + 1, device=q.device, dtype=torch.int32)
    return paged_attention(q, k_cache, v_cache, block_table, qsl, seq_lens, scale, kv_scales, window, softcap,
                           sinks, alibi_slopes)


# ---------------------------------------------------------------- sampling
def apply_penalties(logits: torch.Tensor, token_counts: torch.Tensor, prompt_mask: torch.Tensor,
                    presence: torch.Tensor, frequency: torch.Tensor,
                    repetition: torch.Tensor) -> torch.Tensor:
    """``token_counts[B,V]`` = occurrences in the output so far;
    ``prompt_mask[B,V]`` = token appears in the prompt."""
    seen = (token_counts > 0) | prompt_mask
    rep = repetition.unsqueeze(1)
    logits = torch.where(seen, torch.where(logits > 0, logits / rep, logits * rep), logits)
    logits = logits - frequency.unsqueeze(1) * token_counts
    logits = logits - presence.unsqueeze(1) * (token_counts > 0)
    return logits


def sample(logits: torch.Tensor, temperature: torch.Tensor, top_k: torch.Tensor,
           top_p: torch.Tensor, min_p: torch.Tensor, uniform: torch.Tensor) -> torch.Tensor:
    """One token per row.  ``temperature == 0`` rows are argmax.  ``uniform[B]``
    in (0,1) drives inverse-CDF sampling over the *descending-sorted* filtered
    distribution, so an identical ``uniform`` gives identical tokens on every
    implementation."""
    logits = logits.float()
    greedy = temperature == 0
    t = torch.where(greedy, torch.ones_like(temperature), temperature)
    probs = torch.softmax(logits / t.unsqueeze(1), dim=-1)
    sp, si = probs.sort(dim=-1, descending=True, stable=True)
    v = logits.shape[-1]
    ranks = torch.arange(v, device=logits.device).unsqueeze(0)
    k = torch.where(top_k <= 0, torch.full_like(top_k, v), top_k).unsqueeze(1)
    keep = ranks < k
    cum = sp.cumsum(-1)
    keep &= (cum - sp) < top_p.unsqueeze(1)          # keep until mass reaches top_p
    keep &= sp >= (min_p.unsqueeze(1) * sp[:, :1])
    keep[:, 0] = True
    sp = torch.where(keep, sp, torch.zeros_like(sp))
    total = sp.sum(-1, keepdim=True)
    cdf = sp.cumsum(-1)
    target = uniform.unsqueeze(1).to(cdf.dtype) * total
    idx = (cdf < target).sum(-1).clamp(max=v - 1)
    # guard: never pick a filtered-out entry because of rounding at the tail
    n_keep = keep.sum(-1)
    idx = torch.minimum(idx, n_keep - 1)
    sampled = si.gather(1, idx.unsqueeze(1)).squeeze(1)
    return torch.where(greedy, logits.argmax(-1), sampled)


def compute_logprobs(logits: torch.Tensor, token_ids: torch.Tensor, num_top: int):
    lp = torch.log_softmax(logits.float(), dim=-1)
    chosen = lp.gather(1, token_ids.unsqueeze(1).long()).squeeze(1)
    if num_top > 0:
        tv, ti = lp.topk(num_top, dim=-1)
    else:
        tv = ti = None
    return chosen, tv, ti


def embedding(ids: torch.Tensor, weight: torch.Tensor, vocab_start: int = 0,
              masked: bool = False) -> torch.Tensor:
    """Vocab-parallel gather: ids outside ``[vocab_start, vocab_start+rows)``
    produce zero rows (summed across ranks afterwards)."""
    local = ids.long() - vocab_start
    mask = (local >= 0) & (local < weight.shape[0])
    out = F.embedding(torch.where(mask, local, torch.zeros_like(local)), weight)
    if masked or vocab_start != 0:
        out = out * mask.unsqueeze(-1).to(out.dtype)
    return out


# ------------------------------------------------ multi-head latent attention over a latent cache (DeepSeek)
def mla_cache_write(latent: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                    slot_mapping: torch.Tensor) -> None:
    """``latent[T, DK]`` into ``k_cache[blocks, 1, bs, DK]``; its first DV values into the V^T fragment cache."""
    bs = k_cache.shape[2]
    dv = v_cache.shape[2] * V_ROWS
    valid = slot_mapping[:latent.shape[0]] >= 0
    slots = slot_mapping[:latent.shape[0]][valid].long()
    blk, off = slots // bs, slots % bs
    if k_cache.dtype == FP8_KV:      # e4m3 cache, scale 1, saturating; index assignment goes through the bytes
        x = latent[valid].float().clamp_(-FP8_MAX, FP8_MAX).to(FP8_KV).view(torch.uint8)
        k_cache.view(torch.uint8)[blk, 0, off, :] = x
        v_cache.view(torch.uint8)[blk, 0, :, off // V_GROUP, :, off % V_GROUP] = x[:, :dv].unflatten(-1, (-1, V_ROWS))
        return
    x = latent[valid].to(k_cache.dtype)
    k_cache[blk, 0, off, :] = x
    v_cache[blk, 0, :, off // V_GROUP, :, off % V_GROUP] = x[:, :dv].unflatten(-1, (-1, V_ROWS))


def mla_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_table: torch.Tensor,
                  token_seq: Optional[torch.Tensor], seq_lens: torch.Tensor, scale: float, max_seq_len: int = 0,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``q[T, Hq, DK]`` against the latent cache -> ``[T, Hq, DV]`` in fp32 arithmetic.  Keys come from the
    key-row cache, values from the V^T fragment cache (so a test of a kernel against this checks both)."""
    t, hq, dk = q.shape
    bs = k_cache.shape[2]
    dv = v_cache.shape[2] * V_ROWS
    res = torch.zeros((t, hq, dv), dtype=torch.float32, device=q.device)
    rows = token_seq.long() if token_seq is not None else torch.arange(t, device=q.device)
    lens = seq_lens[:t].long()
    for b in torch.unique(rows).tolist():
        mine = torch.nonzero(rows == b).flatten()
        n = int(lens[mine].max())
        blocks = block_table[b, :(n + bs - 1) // bs].long()
        kb, vb = k_cache, v_cache
        if k_cache.dtype == FP8_KV:
            kb, vb = k_cache.view(torch.uint8), v_cache.view(torch.uint8)
        keys = kb[blocks, 0].reshape(-1, dk)[:n]
        # [blocks, DV/16, bs/8, 16, 8] -> [blocks, bs/8, 8, DV/16, 16] -> [n, DV]
        vals = vb[blocks, 0].permute(0, 2, 4, 1, 3).reshape(-1, dv)[:n]
        if k_cache.dtype == FP8_KV:
            keys, vals = keys.contiguous().view(FP8_KV), vals.contiguous().view(FP8_KV)
        keys, vals = keys.float(), vals.float()
        s = torch.einsum("thd,nd->thn", q[mine].float(), keys) * scale
        mask = torch.arange(n, device=q.device)[None, None, :] >= lens[mine][:, None, None]
        p = torch.softmax(s.masked_fill(mask, float("-inf")), dim=-1)
        res[mine] = torch.einsum("thn,nd->thd", p, vals)
    res = res.to(q.dtype)
    if out is not None:
        out.copy_(res)
        return out
    return res
