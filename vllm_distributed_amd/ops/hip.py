"""ctypes bindings for ``libvdamd_ops.so`` (the gfx950 kernel library).

Every entry point takes raw device pointers plus the current HIP stream, so
the launches are captured when the caller is inside a hipGraph capture.
Shapes are validated here, on the host, before a kernel is launched: an
out-of-bounds access in a hand-written kernel can take the GPU down.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p
from typing import Optional, Tuple

import torch

_LIB: Optional[ctypes.CDLL] = None
# VDAMD_OPS_LIB selects another build of the library (the host-sanitizer preset of build.py)
_LIB_PATH = os.environ.get("VDAMD_OPS_LIB") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_C", "libvdamd_ops.so")
_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}
MAX_SPLITS = 16


def load() -> ctypes.CDLL:
    global _LIB, MAX_SPLITS
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise FileNotFoundError(f"{_LIB_PATH} is missing: run `python -m vllm_distributed_amd.build`")
    lib = ctypes.CDLL(_LIB_PATH)
    P, I, L, F = c_void_p, c_int, c_long, c_float
    sigs = {
        "vdamd_rms_norm": [P, P, P, P, F, I, I, L, L, I, P],
        "vdamd_layer_norm": [P, P, P, P, F, I, I, I, P],
        "vdamd_reduce_add_rms_norm": [P, P, I, P, P, F, I, I, L, I, P],
        "vdamd_silu_and_mul": [P, P, L, I, I, P],
        "vdamd_embedding": [P, P, P, I, I, I, I, I, P],
        "vdamd_mid_gemm": [P, L, P, L, P, L, P, I, I, I, I, I, I, I, I, P],
        "vdamd_stream_gemm": [P, L, P, L, P, L, P, P, L, I, I, I, I, I, I, I, I, I, P],
        "vdamd_big_gemm": [P, L, P, L, P, L, P, L, I, I, I, I, I, I, P],
        "vdamd_rope_cache": [P, P, P, P, L, L, L, P, P, P, P, I, I, I, I, I, I, P, P, F, I, I, P, P, P],
        "vdamd_rope_cache_slabs": [P, P, I, P, L, P, P, P, P, I, I, I, I, I, P, P, F, I, I, P, P, P],
        "vdamd_paged_attention_decode": [P, P, P, L, P, P, P, I, P, F, I, I, I, I, I, I, L, I, I, P, P, I, F, P, P, P],
        "vdamd_paged_attention_prefill": [P, P, L, P, P, P, I, P, P, F, I, I, I, I, I, I, L, I, I, P, P, I, F, P, P, P],
        "vdamd_argmax": [P, P, I, I, L, I, P],
        "vdamd_sample": [P, P, I, I, L, P, P, P, P, P, I, P],
        "vdamd_w4a16_gemm": [P, L, P, L, P, P, P, I, I, I, I, I, I, I, P],
        "vdamd_w4_dequant": [P, P, P, I, I, I, I, P],
        "vdamd_skinny_gemm": [P, L, P, L, P, L, P, I, I, I, I, I, I, L, P],
        "vdamd_dynamic_fp8_quant": [P, P, P, I, I, L, I, P],
        "vdamd_skinny_gemm_grouped": [P, L, P, L, P, L, L, P, I, I, I, I, I, I, P],
        "vdamd_w4a16_gemm_grouped": [P, L, P, L, P, P, L, L, P, I, I, I, I, I, I, I, P],
        "vdamd_moe_topk": [P, L, P, P, I, I, I, I, I, P],
        "vdamd_moe_align": [P, I, I, I, I, I, I, P, P, P, I, P],
        "vdamd_moe_route_small": [P, L, I, I, I, I, I, I, I, P, P, P, P, P, I, P, L, P, I, I, I, I, L, P],
        "vdamd_moe_combine": [P, L, P, L, P, P, P, I, I, I, I, I, P, I, P],
        "vdamd_bias_clamped_glu": [P, L, P, L, P, P, P, I, I, I, I, F, F, I, P],
        "vdamd_moe_topk_grouped": [P, L, P, P, P, I, I, I, I, I, I, I, I, F, I, P],
        "vdamd_mla_attention": [P, L, P, P, P, L, P, P, P, I, P, P, F, I, I, I, I, I, I, I, I, I, P],
        "vdamd_mla_cache_write": [P, L, P, P, P, I, I, I, I, I, I, P],
        "vdamd_mla_prep": [P, P, L, I, I, P, L, L, P, L, P, F, P, P, P, P, P, I, I, I, I, I, I, I, P],
        "vdamd_moe_combine_add_rms_norm": [P, L, P, L, P, P, P, I, I, I, I, I, P, P, F, I, P],
        "vdamd_fp4a16_gemm": [P, L, P, L, P, P, P, P, I, I, I, I, I, I, P],
        "vdamd_fp4_dequant": [P, P, P, P, I, I, I, P],
        "vdamd_fp4a16_gemm_grouped": [P, L, P, L, P, P, L, L, P, P, I, I, I, I, I, P],
    }
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.vdamd_decode_workspace_floats.argtypes = [I, I, I]
    lib.vdamd_decode_workspace_floats.restype = c_long
    lib.vdamd_max_splits.restype = c_int
    MAX_SPLITS = int(lib.vdamd_max_splits())
    _LIB = lib
    return lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _check(rc: int, name: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}") from None


HEAD_DIMS = (64, 96, 128, 256)


def supports_head_dim(d: int) -> bool:
    """Head sizes the attention kernels are instantiated for (96: Phi-3-mini, 256: Gemma)."""
    return d in HEAD_DIMS


def _half(t: torch.Tensor) -> bool:
    return t.dtype in (torch.bfloat16, torch.float16)


# ------------------------------------------------------------------- norms
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    if not _half(x):
        from . import torch_ref
        return torch_ref.rms_norm(x, weight, eps)
    h = x.shape[-1]
    x2 = x.reshape(-1, h)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    out = torch.empty((x2.shape[0], h), dtype=x.dtype, device=x.device)
    assert weight.numel() == h and weight.dtype == x.dtype and weight.is_contiguous()
    _check(_LIB.vdamd_rms_norm(out.data_ptr(), x2.data_ptr(), None, weight.data_ptr(), eps, x2.shape[0], h,
                               x2.stride(0), h, _dt(x), _stream()), "rms_norm")
    return out.view(x.shape)


def fused_add_rms_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """In place: ``residual <- x + residual``; ``x <- rms_norm(residual) * w``."""
    if not _half(x):
        from . import torch_ref
        return torch_ref.fused_add_rms_norm(x, residual, weight, eps)
    h = x.shape[-1]
    assert x.dim() == 2 and x.is_contiguous() and residual.is_contiguous() and residual.shape == x.shape
    assert weight.numel() == h and weight.dtype == x.dtype and residual.dtype == x.dtype
    _check(_LIB.vdamd_rms_norm(x.data_ptr(), x.data_ptr(), residual.data_ptr(), weight.data_ptr(), eps,
                               x.shape[0], h, h, h, _dt(x), _stream()), "fused_add_rms_norm")
    return x, residual


def reduce_add_rms_norm(partial: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                        eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``normed = RMSNorm(sum_k partial[k] + residual) * weight`` with ``residual`` updated in place to the
    sum: the K-split slabs ``[ks, T, H]`` (fp32) of the GEMM that fed the norm are reduced by the norm."""
    ks, t, h = partial.shape
    assert partial.dtype == torch.float32 and partial.is_contiguous() and residual.shape == (t, h)
    assert residual.is_contiguous() and _half(residual) and weight.dtype == residual.dtype
    out = torch.empty_like(residual)
    _check(_LIB.vdamd_reduce_add_rms_norm(out.data_ptr(), partial.data_ptr(), ks, residual.data_ptr(),
                                          weight.data_ptr(), float(eps), t, h, h, _dt(residual), _stream()),
           "reduce_add_rms_norm")
    return out, residual


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    if not _half(x):
        from . import torch_ref
        return torch_ref.layer_norm(x, weight, bias, eps)
    h = x.shape[-1]
    x2 = x.reshape(-1, h).contiguous()
    out = torch.empty_like(x2)
    _check(_LIB.vdamd_layer_norm(out.data_ptr(), x2.data_ptr(), weight.data_ptr(),
                                 None if bias is None else bias.data_ptr(), eps, x2.shape[0], h, _dt(x),
                                 _stream()), "layer_norm")
    return out.view(x.shape)


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    if not _half(x):
        from . import torch_ref
        return torch_ref.silu_and_mul(x)
    f = x.shape[-1] // 2
    x2 = x.reshape(-1, 2 * f)
    assert x2.is_contiguous()
    out = torch.empty((x2.shape[0], f), dtype=x.dtype, device=x.device)
    _check(_LIB.vdamd_silu_and_mul(out.data_ptr(), x2.data_ptr(), x2.shape[0], f, _dt(x), _stream()),
           "silu_and_mul")
    return out.view(x.shape[:-1] + (f,))


def embedding(ids: torch.Tensor, weight: torch.Tensor, vocab_start: int = 0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if not _half(weight):
        from . import torch_ref
        return torch_ref.embedding(ids, weight, vocab_start, True)
    assert ids.dim() == 1 and weight.is_contiguous()
    if ids.dtype != torch.int32:
        ids = ids.to(torch.int32)
    t, h = ids.shape[0], weight.shape[1]
    if out is None:
        out = torch.empty((t, h), dtype=weight.dtype, device=weight.device)
    assert out.shape == (t, h) and out.dtype == weight.dtype and out.is_contiguous()
    _check(_LIB.vdamd_embedding(out.data_ptr(), ids.data_ptr(), weight.data_ptr(), t, h, vocab_start,
                                weight.shape[0], _dt(weight), _stream()), "embedding")
    return out


# ---------------------------------------------------------- rotary + cache
FP8_KV = torch.float8_e4m3fn


def _kv_args(k_cache: torch.Tensor, kv_scales, hkv: int):
    """(kv_fp8 flag, k_scale pointer, v_scale pointer) for a cache tensor pair."""
    if k_cache.dtype != FP8_KV:
        return 0, None, None
    if kv_scales is None:
        return 1, None, None
    ks, vs = kv_scales
    for t in (ks, vs):
        assert t.dtype == torch.float32 and t.numel() == hkv and t.is_contiguous() and t.device == k_cache.device
    return 1, ks.data_ptr(), vs.data_ptr()


def _check_cache(k_cache: torch.Tensor, v_cache: torch.Tensor, hkv: int, d: int) -> int:
    assert k_cache.dtype == v_cache.dtype
    assert k_cache.dim() == 4 and v_cache.dim() == 6 and k_cache.is_contiguous() and v_cache.is_contiguous()
    nb, h, bs, dd = k_cache.shape
    assert (h, dd) == (hkv, d) and bs % 8 == 0 and tuple(v_cache.shape) == (nb, hkv, d // 16, bs // 8, 16, 8), \
        (k_cache.shape, v_cache.shape)
    assert bs % 8 == 0
    return bs


def rope_and_cache(positions, qkv, cos_sin_cache, k_cache, v_cache, slot_mapping, num_heads, num_kv_heads,
                   head_dim, q_norm_weight=None, k_norm_weight=None, norm_eps: float = 0.0,
                   kv_scales=None) -> torch.Tensor:
    """``q_norm_weight`` / ``k_norm_weight``: RMSNorm each q and k head before rotating (Qwen3)."""
    lanes = head_dim // 16          # the fused q/k norm reduces over an aligned power-of-two lane group
    if not _half(qkv) or head_dim % 16 or (q_norm_weight is not None and (lanes & (lanes - 1) or lanes > 64)):
        from . import torch_ref
        return torch_ref.rope_and_cache(positions, qkv, cos_sin_cache, k_cache, v_cache, slot_mapping,
                                        num_heads, num_kv_heads, head_dim, q_norm_weight, k_norm_weight, norm_eps,
                                        kv_scales)
    if q_norm_weight is not None:
        for w in (q_norm_weight, k_norm_weight):
            assert w.dtype == qkv.dtype and w.numel() == head_dim and w.is_contiguous()
    t = qkv.shape[0]
    qs, ks = num_heads * head_dim, num_kv_heads * head_dim
    assert qkv.dim() == 2 and qkv.shape[1] == qs + 2 * ks and qkv.stride(1) == 1
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.shape[1] == head_dim
    assert positions.dtype == torch.int32 and slot_mapping.dtype == torch.int32
    assert positions.numel() >= t and slot_mapping.numel() >= t
    bs = _check_cache(k_cache, v_cache, num_kv_heads, head_dim)
    st = qkv.stride(0)
    esz = qkv.element_size()
    base = qkv.data_ptr()
    _check(_LIB.vdamd_rope_cache(positions.data_ptr(), base, base + qs * esz, base + (qs + ks) * esz, st, st, st,
                                 cos_sin_cache.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                 slot_mapping.data_ptr(), t, num_heads, num_kv_heads, head_dim, bs, 0,
                                 None if q_norm_weight is None else q_norm_weight.data_ptr(),
                                 None if k_norm_weight is None else k_norm_weight.data_ptr(), norm_eps,
                                 _dt(qkv), *_kv_args(k_cache, kv_scales, num_kv_heads), _stream()), "rope_and_cache")
    return qkv[:, :qs]


def rope_and_cache_partials(positions, slabs, dtype, cos_sin_cache, k_cache, v_cache, slot_mapping, num_heads,
                            num_kv_heads, head_dim, q_norm_weight=None, k_norm_weight=None, norm_eps: float = 0.0,
                            kv_scales=None) -> torch.Tensor:
    """``rope_and_cache`` whose packed ``qkv`` arrives as the fp32 K-split slabs ``[ks, T, (Hq + 2 Hkv) D]`` of
    the projection (summed here, rounded to ``dtype`` first as the GEMM's own reduction would); returns the
    rotated ``q [T, Hq D]`` in a buffer of its own."""
    ks, t, n = slabs.shape
    qs = num_heads * head_dim
    lanes = head_dim // 16
    assert slabs.dtype == torch.float32 and slabs.is_contiguous() and n == qs + 2 * num_kv_heads * head_dim
    assert head_dim % 16 == 0 and dtype in (torch.bfloat16, torch.float16)
    assert q_norm_weight is None or (lanes & (lanes - 1) == 0 and lanes <= 64)
    if q_norm_weight is not None:
        for w in (q_norm_weight, k_norm_weight):
            assert w.dtype == dtype and w.numel() == head_dim and w.is_contiguous()
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.shape[1] == head_dim
    assert positions.dtype == torch.int32 and slot_mapping.dtype == torch.int32
    assert positions.numel() >= t and slot_mapping.numel() >= t
    bs = _check_cache(k_cache, v_cache, num_kv_heads, head_dim)
    q = torch.empty((t, qs), dtype=dtype, device=slabs.device)
    _check(_LIB.vdamd_rope_cache_slabs(positions.data_ptr(), slabs.data_ptr(), ks, q.data_ptr(), qs,
                                       cos_sin_cache.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                       slot_mapping.data_ptr(), t, num_heads, num_kv_heads, head_dim, bs,
                                       None if q_norm_weight is None else q_norm_weight.data_ptr(),
                                       None if k_norm_weight is None else k_norm_weight.data_ptr(), norm_eps,
                                       _DT[dtype],
                                       *_kv_args(k_cache, kv_scales, num_kv_heads), _stream()),
           "rope_and_cache_partials")
    return q


def rotary_embedding(positions, q, k, cos_sin_cache, head_dim) -> None:
    if not _half(q):
        from . import torch_ref
        return torch_ref.rotary_embedding(positions, q, k, cos_sin_cache, head_dim)
    t = q.shape[0]
    assert q.stride(-1) == 1 and k.stride(-1) == 1 and q.dim() == 2 and k.dim() == 2
    assert positions.dtype == torch.int32 and cos_sin_cache.dtype == torch.float32
    _check(_LIB.vdamd_rope_cache(positions.data_ptr(), q.data_ptr(), k.data_ptr(), None, q.stride(0),
                                 k.stride(0), 0, cos_sin_cache.data_ptr(), None, None, None, t,
                                 q.shape[1] // head_dim, k.shape[1] // head_dim, head_dim, 16, 1, None, None, 0.0,
                                 _dt(q), 0, None, None, _stream()), "rotary_embedding")


def reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, kv_scales=None) -> None:
    if not _half(k):
        from . import torch_ref
        return torch_ref.reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, kv_scales)
    t, hkv, d = k.shape
    assert k.stride(2) == 1 and v.stride(2) == 1 and k.stride(1) == d and v.stride(1) == d
    assert slot_mapping.dtype == torch.int32 and slot_mapping.numel() >= t
    bs = _check_cache(k_cache, v_cache, hkv, d)
    _check(_LIB.vdamd_rope_cache(None, None, k.data_ptr(), v.data_ptr(), 0, k.stride(0), v.stride(0), None,
                                 k_cache.data_ptr(), v_cache.data_ptr(), slot_mapping.data_ptr(), t, 0, hkv, d,
                                 bs, 2, None, None, 0.0, _dt(k), *_kv_args(k_cache, kv_scales, hkv), _stream()),
           "reshape_and_cache")


# ---------------------------------------------------------------- attention
def decode_workspace_floats(max_bs: int, num_heads: int, head_dim: int, num_kv_heads: int = 1,
                            max_seq_len: int = 1 << 30) -> int:
    """Floats of split-KV scratch that cover every batch size up to ``max_bs``
    (splits shrink as the batch grows, so this is far below max_bs * MAX_SPLITS)."""
    worst = max(b * choose_num_splits(b, num_kv_heads, max_seq_len) for b in range(1, max(1, max_bs) + 1))
    return worst * num_heads * (head_dim + 2)


def choose_num_splits(batch: int, num_kv_heads: int, max_seq_len: int) -> int:
    """Enough workgroups to fill 256 CUs twice over, each split >= 512 keys: up to 512 keys one
    workgroup per (sequence, KV head) finishes in the time the merge kernel of a split takes to launch."""
    wgs = max(1, batch * num_kv_heads)
    want = (512 + wgs - 1) // wgs
    by_len = max(1, max_seq_len // 512)
    return max(1, min(MAX_SPLITS, want, by_len))


def _per_head(t: Optional[torch.Tensor], hq: int, q: torch.Tensor):
    """Pointer of an optional fp32 ``[Hq]`` vector (attention sinks, ALiBi slopes)."""
    if t is None:
        return None
    assert t.dtype == torch.float32 and t.numel() == hq and t.is_contiguous() and t.device == q.device
    return t.data_ptr()


def paged_attention_decode(q, k_cache, v_cache, block_table, seq_lens, scale, max_seq_len=0, out=None,
                           workspace=None, num_splits: Optional[int] = None, kv_scales=None,
                           window: int = 0, softcap: float = 0.0, sinks=None, alibi_slopes=None) -> torch.Tensor:
    """``window > 0``: sliding-window attention, the query sees its last ``window`` keys only.
    ``softcap > 0``: scores become ``softcap * tanh(score / softcap)`` before the softmax (Gemma-2).
    ``sinks [Hq]`` (fp32): one extra score per query head that only enters the softmax's denominator
    (attention sinks).  ``alibi_slopes [Hq]`` (fp32): key ``k`` gets ``slope * (k - query position)`` added."""
    b, hq, d = q.shape
    hkv = k_cache.shape[1]
    bs = _check_cache(k_cache, v_cache, hkv, d)
    assert q.stride(2) == 1 and q.stride(1) == d and hq % hkv == 0
    assert block_table.dtype == torch.int32 and seq_lens.dtype == torch.int32
    assert block_table.shape[0] >= b and seq_lens.numel() >= b and block_table.stride(1) == 1
    if max_seq_len <= 0:
        max_seq_len = block_table.shape[1] * bs
    assert max_seq_len <= block_table.shape[1] * bs, "block table narrower than max_seq_len"
    if out is None:
        out = torch.empty((b, hq, d), dtype=q.dtype, device=q.device)
    assert out.is_contiguous() and out.shape == (b, hq, d)
    if num_splits is None:
        num_splits = choose_num_splits(b, hkv, min(max_seq_len, window) if window > 0 else max_seq_len)
    ws_ptr = None
    assert 0 <= num_splits <= MAX_SPLITS       # 0 = the one-wave-per-sequence kernel
    if num_splits > 1:
        need = b * hq * num_splits * (d + 2)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, dtype=torch.float32, device=q.device)
        ws_ptr = workspace.data_ptr()
    _check(_LIB.vdamd_paged_attention_decode(
        out.data_ptr(), ws_ptr, q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(),
        block_table.data_ptr(), block_table.stride(0), seq_lens.data_ptr(), float(scale), b, hq, hkv, d, bs,
        num_splits, hq * d, _dt(q), *_kv_args(k_cache, kv_scales, hkv), int(window), float(softcap),
        _per_head(sinks, hq, q), _per_head(alibi_slopes, hq, q), _stream()), "paged_attention_decode")
    return out


def paged_attention_prefill(q, k_cache, v_cache, block_table, query_start_loc, seq_lens, scale,
                            max_query_len=0, max_seq_len=0, out=None, kv_scales=None, window: int = 0,
                            softcap: float = 0.0, sinks=None, alibi_slopes=None) -> torch.Tensor:
    t, hq, d = q.shape
    hkv = k_cache.shape[1]
    bs = _check_cache(k_cache, v_cache, hkv, d)
    nseq = seq_lens.numel()
    assert q.stride(2) == 1 and q.stride(1) == d and hq % hkv == 0
    assert block_table.dtype == torch.int32 and seq_lens.dtype == torch.int32
    assert query_start_loc.dtype == torch.int32 and query_start_loc.numel() == nseq + 1
    assert block_table.shape[0] >= nseq and block_table.stride(1) == 1
    if max_query_len <= 0:
        max_query_len = t
    if max_seq_len > 0:
        assert max_seq_len <= block_table.shape[1] * bs, "block table narrower than max_seq_len"
    if out is None:
        out = torch.empty((t, hq, d), dtype=q.dtype, device=q.device)
    assert out.is_contiguous() and out.shape == (t, hq, d)
    _check(_LIB.vdamd_paged_attention_prefill(
        out.data_ptr(), q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(),
        block_table.data_ptr(), block_table.stride(0), query_start_loc.data_ptr(), seq_lens.data_ptr(),
        float(scale), nseq, max_query_len, hq, hkv, d, bs, hq * d, _dt(q), *_kv_args(k_cache, kv_scales, hkv),
        int(window), float(softcap), _per_head(sinks, hq, q), _per_head(alibi_slopes, hq, q), _stream()),
        "paged_attention_prefill")
    return out


# ----------------------------------------------------------------- sampling
def argmax(logits: torch.Tensor) -> torch.Tensor:
    assert logits.dim() == 2 and logits.stride(1) == 1
    b, v = logits.shape
    out = torch.empty(b, dtype=torch.int32, device=logits.device)
    _check(_LIB.vdamd_argmax(out.data_ptr(), logits.data_ptr(), b, v, logits.stride(0), _dt(logits),
                             _stream()), "argmax")
    return out


def sample(logits, temperature, top_k, top_p, min_p, uniform) -> torch.Tensor:
    assert logits.dim() == 2 and logits.stride(1) == 1
    b, v = logits.shape
    for t, dt in ((temperature, torch.float32), (top_k, torch.int32), (top_p, torch.float32),
                  (min_p, torch.float32), (uniform, torch.float32)):
        assert t.dtype == dt and t.numel() == b and t.is_contiguous() and t.is_cuda
    out = torch.empty(b, dtype=torch.int32, device=logits.device)
    _check(_LIB.vdamd_sample(out.data_ptr(), logits.data_ptr(), b, v, logits.stride(0), temperature.data_ptr(),
                             top_k.data_ptr(), top_p.data_ptr(), min_p.data_ptr(), uniform.data_ptr(),
                             _dt(logits), _stream()), "sample")
    return out


# ------------------------------------------------------------ W4A16 linears
W4_MAX_FUSED_M = int(os.environ.get("VDAMD_W4_MAX_FUSED_M", "256"))


def _w4_dims(qw: torch.Tensor, sz: torch.Tensor) -> Tuple[int, int, int]:
    assert qw.dim() == 4 and qw.shape[2:] == (64, 4) and qw.dtype == torch.int32 and qw.is_contiguous()
    n, k = qw.shape[0] * 16, qw.shape[1] * 128
    assert sz.dim() == 4 and sz.shape[0] == qw.shape[0] and sz.shape[2:] == (16, 2) and sz.is_contiguous()
    g = k // sz.shape[1]
    assert g * sz.shape[1] == k and g % 128 == 0, (k, sz.shape)
    return n, k, g


def w4_choose_variant(m: int, n: int, k: int) -> int:
    """0: waves split K, x from L2 (finest tiling; best for one row tile);
    1: waves split N and share x through LDS (batches above 16 rows, when
    there are enough 256-row workgroups to go round)."""
    override = os.environ.get("VDAMD_W4_VARIANT")
    if override is not None:
        return int(override)
    if m <= 16 or (m <= 32 and n < 32768):      # measured inside a captured graph on 70B / 480B projections
        return 0
    wgs = ((n // 16 + 15) // 16) * max(1, (k // 128) // 4)
    return 1 if wgs >= 128 else 0


def w4_choose_ksplit(m: int, n: int, k: int, variant: int = 0) -> int:
    """Split K until ~2-3 workgroups per CU exist (variant 0: 64 rows per
    workgroup, at least two 128-column blocks per wave; variant 1: 256 output
    rows per workgroup up to 64 batch rows, 128 above (row chunks of <= 128),
    at least four blocks per workgroup)."""
    if variant == 1:
        if m > 64:
            base = ((n // 16 + 7) // 8) * ((m + 127) // 128)
        else:
            base = (n // 16 + 15) // 16
        return max(1, min((512 + base - 1) // base, (k // 128) // 4))
    mt = min(4, (m + 15) // 16)
    mz = (m + 16 * mt - 1) // (16 * mt)
    base = ((n // 16 + 3) // 4) * mz
    ks = max(1, min((512 + base - 1) // base, (k // 128) // 8))
    return 1 << (ks.bit_length() - 1)       # power-of-two splits measured at or next to the optimum; 3 and 6 are not


def w4_dequant(qw: torch.Tensor, sz: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, k, g = _w4_dims(qw, sz)
    if out is None:
        out = torch.empty((n, k), dtype=sz.dtype, device=qw.device)
    assert out.is_contiguous() and out.numel() >= n * k and out.dtype == sz.dtype
    _check(_LIB.vdamd_w4_dequant(out.data_ptr(), qw.data_ptr(), sz.data_ptr(), n, k, g, _dt(sz), _stream()),
           "w4_dequant")
    return out[:n * k].view(n, k) if out.dim() == 1 else out


def w4a16_gemm(x: torch.Tensor, qw: torch.Tensor, sz: torch.Tensor, variant: Optional[int] = None,
               ksplit: Optional[int] = None, partials: bool = False) -> Optional[torch.Tensor]:
    """``x [M, K]`` (bf16/fp16, rows 16-byte aligned) times the packed weights.  ``partials``: the fp32
    K-split slabs ``[ksplit, M, N]`` instead of their sum, for a consumer that adds them up while it
    reads them; None when the shape runs unsplit."""
    n, k, g = _w4_dims(qw, sz)
    assert x.dim() == 2 and x.shape[1] == k and x.dtype == sz.dtype and x.stride(1) == 1
    assert x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
    m = x.shape[0]
    if variant is None:
        variant = w4_choose_variant(m, n, k)
    if ksplit is None:
        ksplit = w4_choose_ksplit(m, n, k, variant)
    if partials and (ksplit <= 1 or m == 0):
        return None
    out = None if partials else torch.empty((m, n), dtype=x.dtype, device=x.device)
    if m == 0:
        return out
    assert 1 <= ksplit <= k // 128
    partial = torch.empty((ksplit, m, n), dtype=torch.float32, device=x.device) if ksplit > 1 else None
    _check(_LIB.vdamd_w4a16_gemm(None if out is None else out.data_ptr(), n, x.data_ptr(), x.stride(0), qw.data_ptr(),
                                 sz.data_ptr(), None if partial is None else partial.data_ptr(), m, n, k, g, ksplit,
                                 variant, _dt(x), _stream()), "w4a16_gemm")
    return partial if partials else out


# ------------------------------------------------------------ NVFP4 linears
def _fp4_dims(qw: torch.Tensor, fs: torch.Tensor, alpha: Optional[torch.Tensor]) -> Tuple[int, int]:
    assert qw.dim() == 4 and qw.shape[2:] == (64, 4) and qw.dtype == torch.int32 and qw.is_contiguous()
    assert fs.shape == qw.shape[:2] + (64,) and fs.dtype == torch.int32 and fs.is_contiguous()
    n, k = qw.shape[0] * 16, qw.shape[1] * 128
    if alpha is not None:
        assert alpha.dtype == torch.float32 and alpha.numel() == n and alpha.is_contiguous()
    return n, k


def fp4_dequant(qw: torch.Tensor, fs: torch.Tensor, alpha: Optional[torch.Tensor], dtype: torch.dtype,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n, k = _fp4_dims(qw, fs, alpha)
    if out is None:
        out = torch.empty((n, k), dtype=dtype, device=qw.device)
    assert out.is_contiguous() and out.numel() >= n * k and out.dtype == dtype
    _check(_LIB.vdamd_fp4_dequant(out.data_ptr(), qw.data_ptr(), fs.data_ptr(),
                                  None if alpha is None else alpha.data_ptr(), n, k, _DT[dtype], _stream()),
           "fp4_dequant")
    return out[:n * k].view(n, k) if out.dim() == 1 else out


def fp4a16_gemm(x: torch.Tensor, qw: torch.Tensor, fs: torch.Tensor, alpha: Optional[torch.Tensor] = None,
                variant: Optional[int] = None, ksplit: Optional[int] = None) -> torch.Tensor:
    """``x [M, K]`` (bf16/fp16) times NVFP4 weights (``ops.quant.pack_fp4``), ``alpha [N]`` fp32 per-row scale."""
    n, k = _fp4_dims(qw, fs, alpha)
    assert x.dim() == 2 and x.shape[1] == k and _half(x) and x.stride(1) == 1
    assert x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
    m = x.shape[0]
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    if m == 0:
        return out
    if variant is None:
        variant = w4_choose_variant(m, n, k)
    if ksplit is None:
        ksplit = w4_choose_ksplit(m, n, k, variant)
    assert 1 <= ksplit <= k // 128
    partial = torch.empty(ksplit * m * n, dtype=torch.float32, device=x.device) if ksplit > 1 else None
    _check(_LIB.vdamd_fp4a16_gemm(out.data_ptr(), n, x.data_ptr(), x.stride(0), qw.data_ptr(), fs.data_ptr(),
                                  None if alpha is None else alpha.data_ptr(),
                                  None if partial is None else partial.data_ptr(), m, n, k, ksplit, variant,
                                  _dt(x), _stream()), "fp4a16_gemm")
    return out


# ---------------------------------------------------------- skinny bf16 GEMM
SKINNY_MAX_M = 128


def skinny_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.dim() == 2 and w.dim() == 2 and 0 < x.shape[0] <= SKINNY_MAX_M and x.dtype == w.dtype
            and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] == w.shape[1]
            and w.shape[1] % 128 == 0 and w.shape[0] % 16 == 0 and x.stride(1) == 1 and w.stride(1) == 1
            and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0)


def skinny_choose(m: int, n: int, k: int) -> Tuple[int, int]:
    """``(ntw, ksplit)``: output tiles per wave and K split.  Fewer tiles per
    wave when N alone cannot fill the chip; then K is split until there are
    about two workgroups per CU, keeping >= 4 column blocks per workgroup.
    Long-K shapes (the down projection) and wide short-K ones (an 8B-class
    gate_up) use the measured optimum."""
    tiles = n // 16
    if k >= 2 * n and k >= 8192:
        if tiles <= 256:                      # e.g. 4096 x 14336: one tile per wave, 64 workgroups x 8
            return 1, max(1, min(8, (k // 128) // 4))
        ntw = 1 if m <= 32 else (4 if m <= 64 else 2)
        wgs = (tiles + 4 * ntw - 1) // (4 * ntw)
        return ntw, max(1, min(4 if m <= 32 else 8, (1024 + wgs - 1) // wgs, (k // 128) // 4))
    if m <= 16 and n >= 16384 and k <= 4096:
        return 2, max(1, min(4, (k // 128) // 4))
    ntw = 4 if m <= 64 else 2
    while ntw > 1 and (tiles + 4 * ntw - 1) // (4 * ntw) < 512:
        ntw //= 2
    wgs = (tiles + 4 * ntw - 1) // (4 * ntw)
    ksplit = max(1, min((512 + wgs - 1) // wgs, (k // 128) // 4))
    return ntw, ksplit


def skinny_profitable(m: int, n: int, k: int) -> bool:
    """Where this kernel beats the library GEMM on MI355X (measured in captured
    graphs, benchmarks/skinny_gemm_bench.py): long reductions into a narrow
    output, for which the library's short-batch kernels reach only 1.6-4.8 TB/s
    (from 24 rows up when the output is 4096 wide or less - below that the
    two are level), very wide outputs over a short reduction at <= 16 rows, and
    4096-square projections at <= 16 rows."""
    if m > SKINNY_MAX_M:
        return False
    if k >= 2 * n and k >= 8192:
        return m >= 24 or n > 4096
    if m <= 16 and 2048 <= n <= 4096 and 4096 <= k < 8192:      # a square 4096 projection: 11.2 vs 12.5 us at one row
        return True
    return m <= 16 and n >= 16384 and k <= 4096


def skinny_gemm(x: torch.Tensor, w: torch.Tensor, ntw: Optional[int] = None, ksplit: Optional[int] = None,
                out: Optional[torch.Tensor] = None, silu_mul: bool = False) -> torch.Tensor:
    """``x [M<=128, K] @ w [N, K]^T`` streaming ``w`` once at HBM speed.  ``silu_mul``: ``x`` is
    ``[M, 2K] = [gate | up]`` and the left operand is ``silu(gate) * up``, formed inside the kernel."""
    up_off = 0
    if silu_mul:
        assert x.dim() == 2 and x.shape[1] == 2 * w.shape[1] and x.stride(1) == 1
        up_off = w.shape[1]
        x = x[:, :up_off]                      # the gate half; the kernel finds the up half up_off further on
    assert skinny_supported(x, w), (x.shape, x.stride(), w.shape, w.stride(), x.dtype)
    m, k = x.shape
    n = w.shape[0]
    a, b = skinny_choose(m, n, k)
    ntw = a if ntw is None else ntw
    ksplit = b if ksplit is None else ksplit
    assert 1 <= ksplit <= k // 128
    if out is None:
        out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    assert out.shape == (m, n) and out.dtype == x.dtype and out.is_contiguous() and out.data_ptr() % 16 == 0
    partial = torch.empty(ksplit * m * n, dtype=torch.float32, device=x.device) if ksplit > 1 else None
    _check(_LIB.vdamd_skinny_gemm(out.data_ptr(), n, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                                  None if partial is None else partial.data_ptr(), m, n, k, ksplit, ntw,
                                  _dt(x), up_off, _stream()), "skinny_gemm")
    return out


def skinny_gemm_partials(x: torch.Tensor, w: torch.Tensor) -> Optional[torch.Tensor]:
    """``skinny_gemm`` without its reduction: the fp32 partial sums ``[ksplit, M, N]`` of ``x @ w^T``, for a
    consumer that adds them up while it loads them (the decode routing kernel).  None when the shape is
    not split over K (there is then nothing to save) or not one the kernel takes."""
    if not skinny_supported(x, w):
        return None
    m, k = x.shape
    n = w.shape[0]
    ntw, ksplit = skinny_choose(m, n, k)
    if ksplit <= 1:
        return None
    partial = torch.empty((ksplit, m, n), dtype=torch.float32, device=x.device)
    _check(_LIB.vdamd_skinny_gemm(None, n, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), partial.data_ptr(),
                                  m, n, k, ksplit, ntw, _dt(x), 0, _stream()), "skinny_gemm (partial sums)")
    return partial


# ------------------------------------------- weight-streaming GEMM, M <= 16
STREAM_MAX_M = 16
EPI_NONE, EPI_RESIDUAL, EPI_SILU_MUL = 0, 1, 2


def stream_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.dim() == 2 and w.dim() == 2 and 0 < x.shape[0] <= STREAM_MAX_M and x.dtype == w.dtype
            and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] == w.shape[1]
            and w.shape[1] % 128 == 0 and w.shape[0] % 16 == 0 and x.stride(1) == 1 and w.stride(1) == 1
            and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0)


STREAM_WIRED_MAX_M = 8


def stream_profitable(m: int, n: int, k: int) -> bool:
    """Where this kernel beats the library GEMM on MI355X (benchmarks/stream_silu_probe.py and
    stream_gemm_sweep.py, padded weight rows, inside graphs): up to 8 rows into a very wide output - a gate_up
    projection (8B-class: 48 us with the activation in the epilogue against 67 us for library + activation
    kernel; 70B: 164 against 166 us) or an LM head (356 against 369 us).  Narrow outputs (a TP=8 shard) lose."""
    return 0 < m <= STREAM_WIRED_MAX_M and n >= 24576 and k >= 2048


def stream_choose(n: int, k: int, silu: bool = False) -> Tuple[int, int, int]:
    """``(ntw, ksplit, depth)``.  One output tile per workgroup until N alone gives several
    workgroups per CU; K is split across workgroups only when even that leaves CUs idle."""
    tiles = n // 16
    ntw = 2 if (silu or tiles >= 4096) else 1
    wgs = tiles // ntw
    ksplit = 1
    while wgs * ksplit < 1024 and (k // 128) // (ksplit * 2) >= 8:
        ksplit *= 2
    return ntw, ksplit, 3


def stream_gemm(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, silu_mul: bool = False, ntw: Optional[int] = None,
                ksplit: Optional[int] = None, depth: Optional[int] = None, nt: bool = False) -> torch.Tensor:
    """``x [M<=16, K] @ w [N, K]^T`` reading ``w`` once at HBM speed.  ``residual [M, N]`` is added in
    the epilogue; ``silu_mul`` treats ``w`` as ``[gate ; up]`` and returns ``silu(gate) * up`` (``N/2`` wide)."""
    assert stream_supported(x, w), (x.shape, x.stride(), w.shape, w.stride(), x.dtype)
    assert not (silu_mul and residual is not None)
    m, k = x.shape
    n = w.shape[0]
    a, b, c = stream_choose(n, k, silu_mul)
    ntw = a if ntw is None else ntw
    ksplit = b if ksplit is None else ksplit
    depth = c if depth is None else depth
    n_out = n // 2 if silu_mul else n
    if out is None:
        out = torch.empty((m, n_out), dtype=x.dtype, device=x.device)
    assert out.shape == (m, n_out) and out.dtype == x.dtype and out.stride(1) == 1 and out.data_ptr() % 8 == 0
    epi = EPI_SILU_MUL if silu_mul else (EPI_RESIDUAL if residual is not None else EPI_NONE)
    if residual is not None:
        assert residual.shape == (m, n) and residual.dtype == x.dtype and residual.stride(1) == 1
    partial = torch.empty(ksplit * m * n, dtype=torch.float32, device=x.device) if ksplit > 1 else None
    _check(_LIB.vdamd_stream_gemm(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), w.data_ptr(),
                                  w.stride(0), None if partial is None else partial.data_ptr(),
                                  None if residual is None else residual.data_ptr(),
                                  0 if residual is None else residual.stride(0), m, n, k, ksplit, ntw, depth,
                                  int(nt), epi, _dt(x), _stream()), "stream_gemm")
    return out


# ---------------------------------------- split-K LDS-DMA GEMM, 16 < M <= 512
MID_MAX_M = 512
# -1: the LDS-DMA / four-tile-prefetch kernel for 256-row tiles (2-7 % faster there), the register-staged
# one for smaller tiles (where the deep variant is 2-5 % slower); 0 / 1 force one of them
MID_VARIANT = int(os.environ.get("VDAMD_MID_GEMM_VARIANT", "-1"))


def mid_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.dim() == 2 and w.dim() == 2 and 0 < x.shape[0] <= MID_MAX_M and x.dtype == w.dtype
            and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] == w.shape[1]
            and w.shape[1] % 64 == 0 and w.shape[0] % 16 == 0 and x.stride(1) == 1 and w.stride(1) == 1
            and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
            and w.data_ptr() % 16 == 0)


# measured time of one 64-deep K tile of one workgroup, us (70B o-projection, unsplit): the part of it
# that does not shrink with the tile is the HBM latency the two-tile prefetch cannot cover
_MID_TILE_US = {(256, 256): 1.6, (256, 128): 1.05, (128, 256): 1.2, (128, 128): 0.75, (64, 256): 1.0}


# K-split counts the choice is made from (VDAMD_MID_KSPLITS=1,2,4,8,16 restricts it, for A/B measurements)
_MID_KSPLITS = tuple(int(v) for v in os.environ.get("VDAMD_MID_KSPLITS", "1,2,3,4,5,6,7,8,12,16").split(","))


def _mid_estimate(m: int, n: int, k: int, bm: int, bn: int, ksplit: int) -> float:
    """Modelled run time in us: rounds of workgroups x K tiles x time per tile (at least the HBM time
    of W), plus the fp32 slab round trip and the reduce launch when K is split."""
    blocks = -(-n // bn) * ksplit * -(-m // bm)
    rounds = -(-blocks // 256)
    main = max(rounds * (k / 64 / ksplit) * _MID_TILE_US[(bm, bn)], n * k * 2 / 5.0e6)
    slabs = (3.0 + 2 * ksplit * m * n * 4 / 3.5e6) if ksplit > 1 else 0.0
    return 6.0 + main + slabs


def mid_choose(m: int, n: int, k: int) -> Tuple[int, int, int]:
    """``(bm, bn, ksplit)`` with the smallest modelled time.  The row tile covers all of M up to 256
    (W is then read from HBM once); splitting K buys parallelism and costs slab traffic."""
    best = None
    for (bm, bn) in _MID_TILE_US:
        if bm < min(m, 256) or (bm == 64 and m > 48 and k < 2 * n):
            continue
        # (counts that are not powers of two as well: a 70B qkv projection's 40 column tiles fill 240 of the
        # 256 CUs at 6, 160 at 4)
        for ksplit in _MID_KSPLITS:
            if k // 64 // ksplit < 4:
                break
            e = _mid_estimate(m, n, k, bm, bn, ksplit)
            if best is None or e < best[0]:
                best = (e, bm, bn, ksplit)
    return best[1], best[2], best[3]


def mid_profitable(m: int, n: int, k: int) -> bool:
    """Where this kernel beats the library GEMM on MI355X (benchmarks/stream_gemm_sweep.py, padded weight
    rows): outputs too narrow to fill the chip with the library's tiles.  Down-projection shapes
    (K >= 2 N) from 48 rows, square-ish ones from 128 rows (from 48 when K >= 8192 and N <= 8192);
    very wide outputs (a 70B gate_up) stay with the library, which already has enough tiles there."""
    if m <= 16 or m > 256 or n > 12288 or k < 2048:
        return False
    if k >= 2 * n:
        return m >= 48
    if m >= 128:
        return True
    return m >= 48 and n <= 8192 and k >= 8192


def mid_gemm(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, ksplit: Optional[int] = None,
             bm: Optional[int] = None, bn: Optional[int] = None, variant: Optional[int] = None,
             partials: bool = False) -> Optional[torch.Tensor]:
    """``x [M<=512, K] @ w [N, K]^T`` with W streamed once and K split over the chip.  ``partials``: the
    fp32 K-split slabs ``[ksplit, M, N]`` instead of their sum (None when the shape runs unsplit)."""
    assert mid_supported(x, w), (x.shape, x.stride(), w.shape, w.stride(), x.dtype)
    m, k = x.shape
    n = w.shape[0]
    a, b, c = mid_choose(m, n, k)
    bm = a if bm is None else bm
    bn = b if bn is None else bn
    ksplit = c if ksplit is None else ksplit
    assert 1 <= ksplit <= k // 64
    if variant is None:
        variant = MID_VARIANT
    if variant < 0:
        variant = 1 if bm == 256 and bn <= 256 else 0
    if partials:
        if ksplit <= 1:
            return None
        partial = torch.empty((ksplit, m, n), dtype=torch.float32, device=x.device)
        _check(_LIB.vdamd_mid_gemm(None, n, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), partial.data_ptr(),
                                   m, n, k, ksplit, bm, bn, variant, _dt(x), _stream()), "mid_gemm (partial sums)")
        return partial
    if out is None:
        out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    assert out.shape == (m, n) and out.dtype == x.dtype and out.stride(1) == 1 and out.data_ptr() % 8 == 0
    partial = torch.empty(ksplit * m * n, dtype=torch.float32, device=x.device) if ksplit > 1 else None
    _check(_LIB.vdamd_mid_gemm(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                               None if partial is None else partial.data_ptr(), m, n, k, ksplit, bm, bn,
                               variant, _dt(x), _stream()), "mid_gemm")
    return out


# ------------------------------------- 256 x 256 tile LDS-DMA GEMM, M >= 512
BIG_MIN_M = 512
# with the SiLU*mul epilogue a single full row of tiles already beats library GEMM + activation kernel
# (70B gate_up at 256 rows: 216 us against 229 + 9; profiles/r2_big_gemm_vs_library.txt)
BIG_MIN_M_SILU = 240
BIG_VARIANT = int(os.environ.get("VDAMD_BIG_GEMM_VARIANT", "0"))


def big_supported(x: torch.Tensor, w: torch.Tensor, silu_mul: bool = False) -> bool:
    return (x.dim() == 2 and w.dim() == 2 and x.shape[0] > 0 and x.dtype == w.dtype
            and x.dtype in (torch.bfloat16, torch.float16) and x.shape[1] == w.shape[1]
            and w.shape[1] % 128 == 0 and w.shape[0] % (32 if silu_mul else 16) == 0
            and x.stride(1) == 1 and w.stride(1) == 1 and x.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0
            and x.shape[0] * x.stride(0) * 2 < (1 << 32) - 64 and w.shape[0] * w.stride(0) * 2 < (1 << 32) - 64)


def big_profitable(m: int, n: int, k: int, silu_mul: bool = False) -> bool:
    """Where the 256 x 256-tile kernel is level with or ahead of the library GEMM on MI355X
    (benchmarks/big_gemm_bench.py, profiles/r2_big_gemm_vs_library.txt): enough tiles to fill the
    chip's 256 CUs in whole rounds.  With the SiLU*mul epilogue it also saves the activation kernel
    and wins at every batch size measured (512 .. 16384 rows); without it the library is 1-4 % ahead
    from 4096 rows up and stays in charge there."""
    if m < (BIG_MIN_M_SILU if silu_mul else BIG_MIN_M) or k < 1024:
        return False
    if m < BIG_MIN_M and m > 256:
        return False                  # between one and two full rows of tiles the second row is mostly padding
    tiles = -(-m // 256) * (-(-(n // 2) // 128) if silu_mul else -(-n // 256))
    rounds = -(-tiles // 256)
    eff = tiles / (rounds * 256.0)
    if silu_mul:
        return tiles >= 200 and eff >= 0.85
    return m <= 3072 and tiles >= 224 and eff >= 0.9


def big_gemm(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None,
             residual: Optional[torch.Tensor] = None, silu_mul: bool = False,
             variant: Optional[int] = None) -> torch.Tensor:
    """``x [M, K] @ w [N, K]^T`` on 256 x 256 tiles with both operands fed by LDS-DMA.  ``residual
    [M, N]`` is added in the epilogue; ``silu_mul`` treats ``w`` as ``[gate ; up]`` and returns
    ``silu(gate) * up`` (``N/2`` wide)."""
    assert big_supported(x, w, silu_mul), (x.shape, x.stride(), w.shape, w.stride(), x.dtype)
    assert not (silu_mul and residual is not None)
    m, k = x.shape
    n = w.shape[0]
    n_out = n // 2 if silu_mul else n
    if out is None:
        out = torch.empty((m, n_out), dtype=x.dtype, device=x.device)
    assert out.shape == (m, n_out) and out.dtype == x.dtype and out.stride(1) == 1 and out.data_ptr() % 16 == 0
    assert out.stride(0) % 8 == 0 and m * out.stride(0) * 2 < (1 << 32) - 64
    epi = EPI_SILU_MUL if silu_mul else (EPI_RESIDUAL if residual is not None else EPI_NONE)
    if residual is not None:
        assert (residual.shape == (m, n) and residual.dtype == x.dtype and residual.stride(1) == 1
                and residual.stride(0) % 4 == 0 and residual.data_ptr() % 8 == 0)
    _check(_LIB.vdamd_big_gemm(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                               None if residual is None else residual.data_ptr(),
                               0 if residual is None else residual.stride(0), m, n, k, epi,
                               BIG_VARIANT if variant is None else variant, _dt(x), _stream()), "big_gemm")
    return out


# ------------------------------------------------- grouped GEMMs (MoE experts)
def _check_tiles(tiles: torch.Tensor, x: torch.Tensor, tile_rows: int) -> int:
    assert tiles.dtype == torch.int32 and tiles.dim() == 2 and tiles.shape[1] == 3 and tiles.is_contiguous()
    assert tiles.device == x.device and tile_rows % 16 == 0
    assert x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
    return tiles.shape[0]


def grouped_gemm(x: torch.Tensor, w: torch.Tensor, tiles: torch.Tensor, tile_rows: int,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[r] = x[r] @ w[g(r)]^T`` for group-sorted rows: ``w [G, N, K]``,
    ``tiles [n, 3] int32`` = (group, first row, rows <= tile_rows; 0 = unused).
    Every row covered by a tile must lie inside ``x``; rows outside all tiles
    are left untouched in ``out``."""
    nt = _check_tiles(tiles, x, tile_rows)
    g, n, k = w.shape
    assert w.is_contiguous() and w.dtype == x.dtype and x.shape[1] == k and 16 <= tile_rows <= 128
    assert k % 128 == 0 and n % 16 == 0
    if out is None:
        out = torch.empty((x.shape[0], n), dtype=x.dtype, device=x.device)
    ntw = 1 if tile_rows > 64 or n <= 4096 else 2
    _check(_LIB.vdamd_skinny_gemm_grouped(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), w.data_ptr(), k,
                                          n * k, tiles.data_ptr(), nt, tile_rows, n, k, ntw, _dt(x), _stream()),
           "skinny_gemm_grouped")
    return out


W4_GROUPED_SILU_MAX_ROWS = 32      # tile sizes whose kernel forms silu(gate) * up while loading its operand


def grouped_w4a16_gemm(x: torch.Tensor, qw: torch.Tensor, sz: torch.Tensor, tiles: torch.Tensor,
                       tile_rows: int, out: Optional[torch.Tensor] = None, silu_mul: bool = False) -> torch.Tensor:
    """Grouped form of ``w4a16_gemm``: ``qw [G, N/16, K/128, 64, 4]``, ``sz [G, N/16, K/g, 16, 2]``.
    ``silu_mul`` (``tile_rows <= 32``): ``x`` is ``[rows, 2K] = [gate | up]`` and the left operand is
    ``silu(gate) * up``, formed inside the kernel."""
    nt = _check_tiles(tiles, x, tile_rows)
    assert qw.dim() == 5 and sz.dim() == 5 and qw.shape[0] == sz.shape[0] and tile_rows <= 64
    n, k, grp = _w4_dims(qw[0], sz[0])
    up_off = 0
    if silu_mul:
        assert tile_rows <= W4_GROUPED_SILU_MAX_ROWS and x.shape[1] == 2 * k
        up_off = k
    else:
        assert x.shape[1] == k
    assert qw.is_contiguous() and sz.is_contiguous() and x.dtype == sz.dtype
    if out is None:
        out = torch.empty((x.shape[0], n), dtype=x.dtype, device=x.device)
    _check(_LIB.vdamd_w4a16_gemm_grouped(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), qw.data_ptr(),
                                         sz.data_ptr(), qw[0].numel(), sz[0].numel(), tiles.data_ptr(), nt,
                                         tile_rows, n, k, grp, _dt(x), up_off, _stream()), "w4a16_gemm_grouped")
    return out


def grouped_fp4a16_gemm(x: torch.Tensor, qw: torch.Tensor, fs: torch.Tensor, alpha: Optional[torch.Tensor],
                        tiles: torch.Tensor, tile_rows: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Grouped form of ``fp4a16_gemm``: ``qw [G, N/16, K/128, 64, 4]``, ``fs [G, N/16, K/128, 64]``,
    ``alpha [G, N]`` or None."""
    nt = _check_tiles(tiles, x, tile_rows)
    assert qw.dim() == 5 and fs.dim() == 4 and qw.shape[0] == fs.shape[0] and tile_rows <= 64
    n, k = _fp4_dims(qw[0], fs[0], None)
    assert qw.is_contiguous() and fs.is_contiguous() and x.shape[1] == k and _half(x)
    if alpha is not None:
        assert alpha.shape == (qw.shape[0], n) and alpha.dtype == torch.float32 and alpha.is_contiguous()
    if out is None:
        out = torch.empty((x.shape[0], n), dtype=x.dtype, device=x.device)
    _check(_LIB.vdamd_fp4a16_gemm_grouped(out.data_ptr(), out.stride(0), x.data_ptr(), x.stride(0), qw.data_ptr(),
                                          fs.data_ptr(), qw[0].numel(), fs[0].numel() * 2,
                                          None if alpha is None else alpha.data_ptr(), tiles.data_ptr(), nt,
                                          tile_rows, n, k, _dt(x), _stream()), "fp4a16_gemm_grouped")
    return out


# ------------------------------------------------------------- MoE routing
MOE_MAX_EXPERTS = 512
MOE_MAX_TOPK = 16


def moe_topk(logits: torch.Tensor, k: int, renormalize: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """softmax + top-k over router logits ``[T, E]``: ``(weights [T, k] fp32, experts [T, k] int32)``;
    ties go to the lowest expert id."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype in _DT
    t, e = logits.shape
    assert k <= e <= MOE_MAX_EXPERTS and 1 <= k <= MOE_MAX_TOPK
    topw = torch.empty((t, k), dtype=torch.float32, device=logits.device)
    topi = torch.empty((t, k), dtype=torch.int32, device=logits.device)
    _check(_LIB.vdamd_moe_topk(logits.data_ptr(), logits.stride(0), topw.data_ptr(), topi.data_ptr(), t, e, k,
                               int(renormalize), _dt(logits), _stream()), "moe_topk")
    return topw, topi


def moe_topk_grouped(logits: torch.Tensor, bias: Optional[torch.Tensor], k: int, n_group: int, topk_group: int,
                     sigmoid: bool, group_top2: bool, renormalize: bool, scale: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """DeepSeek routing in one launch: scores = sigmoid (V3) or softmax (V2) of ``logits [T, E]``; experts chosen
    by ``score + bias`` inside the ``topk_group`` best of ``n_group`` groups (a group scored by its two best
    choices' sum, or by its best); weights = the chosen experts' unbiased scores, renormalised on request, times
    ``scale``.  Returns ``(weights [T, k] fp32, experts [T, k] int32)``."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype in _DT
    t, e = logits.shape
    assert k <= e <= MOE_MAX_EXPERTS and 1 <= k <= MOE_MAX_TOPK and 1 <= n_group <= 64 and e % n_group == 0
    assert bias is None or (bias.dtype == torch.float32 and bias.numel() == e and bias.is_contiguous())
    topw = torch.empty((t, k), dtype=torch.float32, device=logits.device)
    topi = torch.empty((t, k), dtype=torch.int32, device=logits.device)
    _check(_LIB.vdamd_moe_topk_grouped(logits.data_ptr(), logits.stride(0), None if bias is None else bias.data_ptr(),
                                       topw.data_ptr(), topi.data_ptr(), t, e, k, n_group, max(1, topk_group),
                                       int(sigmoid), int(group_top2), int(renormalize), float(scale), _dt(logits),
                                       _stream()), "moe_topk_grouped")
    return topw, topi


def moe_align(topi: torch.Tensor, num_experts: int, tile_rows: int, expert_start: int = 0, local_experts: int = -1,
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Sort the ``T*k`` slots by expert on the device: ``(src_row [T*k] = token of each sorted row,
    inv [T*k] = sorted position of each slot, tiles [T*k // tile_rows + E, 3] int32)``.  Tiles of
    experts outside ``[expert_start, expert_start + local_experts)`` have 0 rows."""
    assert topi.dtype == torch.int32 and topi.dim() == 2 and topi.is_contiguous() and num_experts <= 1024
    t, k = topi.shape
    slots = t * k
    n = slots // tile_rows + num_experts
    src_row = torch.empty(slots, dtype=torch.int32, device=topi.device)
    inv = torch.empty(slots, dtype=torch.int32, device=topi.device)
    tiles = torch.empty((n, 3), dtype=torch.int32, device=topi.device)
    _check(_LIB.vdamd_moe_align(topi.data_ptr(), slots, k, num_experts, tile_rows, expert_start, local_experts,
                                src_row.data_ptr(), inv.data_ptr(), tiles.data_ptr(), n, _stream()), "moe_align")
    return src_row, inv, tiles


MOE_SMALL_MAX_TOKENS = 16


def moe_route_small(logits: torch.Tensor, x: torch.Tensor, k: int, renormalize: bool, tile_rows: int,
                    expert_start: int = 0, local_experts: int = -1):
    """``moe_topk`` + ``moe_align`` + the gather ``x[src_row]`` in one launch, for at most 16 tokens:
    ``(topw, topi, inv, tiles, xs [T*k, H])``.  Rows of one expert keep slot order."""
    nsplit, sstride = 1, 0
    if logits.dim() == 3:                       # fp32 partial sums of a K-split router GEMM, [splits, T, E]
        assert logits.dtype == torch.float32 and logits.is_contiguous()
        nsplit, sstride = logits.shape[0], logits.stride(0)
        logits = logits[0]
    t, e = logits.shape
    h = x.shape[1]
    assert logits.stride(1) == 1 and x.shape[0] == t and _half(x) and x.stride(1) == 1 and h % 8 == 0
    assert x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
    assert 1 <= t <= MOE_SMALL_MAX_TOKENS and k <= e <= MOE_MAX_EXPERTS and 1 <= k <= MOE_MAX_TOPK
    assert logits.dtype in (x.dtype, torch.float32)
    slots = t * k
    n = slots // tile_rows + e
    dev = x.device
    topw = torch.empty((t, k), dtype=torch.float32, device=dev)
    ints = torch.empty(3 * slots + 3 * n, dtype=torch.int32, device=dev)          # one allocation for the tables
    topi, src_row, inv = ints[:slots].view(t, k), ints[slots:2 * slots], ints[2 * slots:3 * slots]
    tiles = ints[3 * slots:].view(n, 3)
    xs = torch.empty((slots, h), dtype=x.dtype, device=dev)
    _check(_LIB.vdamd_moe_route_small(logits.data_ptr(), logits.stride(0), t, e, k, int(renormalize), tile_rows,
                                      expert_start, local_experts, topw.data_ptr(), topi.data_ptr(),
                                      src_row.data_ptr(), inv.data_ptr(), tiles.data_ptr(), n, x.data_ptr(),
                                      x.stride(0), xs.data_ptr(), h, _dt(logits), _dt(x), nsplit, sstride, _stream()),
           "moe_route_small")
    return topw, topi, inv, tiles, xs


def moe_combine(y: torch.Tensor, topw: torch.Tensor, topi: torch.Tensor, inv: torch.Tensor, expert_start: int = 0,
                local_experts: int = -1, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[t] = sum_j topw[t, j] * y[inv[t*k + j]]`` in fp32 (slots of experts this rank does not hold are skipped).
    ``bias [local experts, H]``: each row first gains its expert's bias (rounded to the row's dtype)."""
    t, k = topi.shape
    h = y.shape[1]
    if bias is not None:
        assert bias.dtype == y.dtype and bias.dim() == 2 and bias.shape[1] == h and bias.is_contiguous()
        assert bias.shape[0] >= (local_experts if local_experts >= 0 else 1)
    assert _half(y) and y.stride(1) == 1 and y.stride(0) % 8 == 0 and h % 8 == 0 and y.data_ptr() % 16 == 0
    assert topw.dtype == torch.float32 and topw.is_contiguous() and topi.is_contiguous() and inv.numel() == t * k
    out = torch.empty((t, h), dtype=y.dtype, device=y.device)
    _check(_LIB.vdamd_moe_combine(out.data_ptr(), h, y.data_ptr(), y.stride(0), topw.data_ptr(), topi.data_ptr(),
                                  inv.data_ptr(), t, k, h, expert_start, local_experts,
                                  None if bias is None else bias.data_ptr(), _dt(y), _stream()),
           "moe_combine")
    return out


def bias_clamped_glu(h: torch.Tensor, bias: torch.Tensor, topi: torch.Tensor, inv: torch.Tensor, expert_start: int,
                     local_experts: int, limit: float, alpha: float) -> torch.Tensor:
    """gpt-oss expert activation on expert-sorted rows ``h [rows, 2 ip] = [gate | up]``: every row gains the
    ``[2 ip]`` bias of its expert, then ``(up + 1) * gate * sigmoid(alpha * gate)`` with ``gate`` clamped from
    above and ``up`` on both sides; returns ``[rows, ip]``.  Rows of slots whose expert lives elsewhere are
    left unwritten (their weights are zero in the combine)."""
    rows, two_ip = h.shape
    ip = two_ip // 2
    slots = topi.numel()
    assert _half(h) and h.stride(1) == 1 and ip % 8 == 0 and inv.numel() == slots and rows >= slots
    assert bias.dtype == h.dtype and bias.shape[1] == two_ip and bias.is_contiguous()
    assert topi.dtype == torch.int32 and inv.dtype == torch.int32 and topi.is_contiguous() and inv.is_contiguous()
    out = torch.empty((rows, ip), dtype=h.dtype, device=h.device)
    _check(_LIB.vdamd_bias_clamped_glu(out.data_ptr(), ip, h.data_ptr(), h.stride(0), bias.data_ptr(), topi.data_ptr(),
                                       inv.data_ptr(), slots, ip, expert_start, local_experts, float(limit),
                                       float(alpha), _dt(h), _stream()), "bias_clamped_glu")
    return out


def moe_combine_add_rms_norm(y: torch.Tensor, topw: torch.Tensor, topi: torch.Tensor, inv: torch.Tensor,
                             expert_start: int, local_experts: int, residual: torch.Tensor, weight: torch.Tensor,
                             eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """``moe_combine`` followed by ``fused_add_rms_norm`` in one launch: ``residual`` (in place) gains the
    weighted sum of each token's expert rows, and its RMSNorm is returned next to it."""
    t, k = topi.shape
    h = y.shape[1]
    assert y.stride(1) == 1 and h % 8 == 0 and h <= 32768 and k <= 16 and _half(y)
    assert residual.shape == (t, h) and residual.is_contiguous() and residual.dtype == y.dtype == weight.dtype
    assert topw.dtype == torch.float32 and topw.is_contiguous() and topi.is_contiguous() and inv.is_contiguous()
    out = torch.empty_like(residual)
    _check(_LIB.vdamd_moe_combine_add_rms_norm(out.data_ptr(), h, y.data_ptr(), y.stride(0), topw.data_ptr(),
                                               topi.data_ptr(), inv.data_ptr(), t, k, h, expert_start, local_experts,
                                               residual.data_ptr(), weight.data_ptr(), float(eps), _dt(y), _stream()),
           "moe_combine_add_rms_norm")
    return out, residual


# ------------------------------------------------------------------ FP8 W8A8
def dynamic_fp8_quant(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-token dynamic quantisation to OCP e4m3: ``(xq [M, K] float8_e4m3fn, scale [M, 1] fp32)``
    with ``x ~ xq * scale``."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
    m, k = x.shape
    out = torch.empty((m, k), dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty((m, 1), dtype=torch.float32, device=x.device)
    _check(_LIB.vdamd_dynamic_fp8_quant(out.data_ptr(), scale.data_ptr(), x.data_ptr(), m, k, x.stride(0),
                                        _dt(x), _stream()), "dynamic_fp8_quant")
    return out, scale


# ------------------------------------------------ multi-head latent attention (DeepSeek) over a latent cache
MLA_KERNEL_DIMS = ((576, 512), (320, 256), (96, 64))


def mla_supported(dk: int, dv: int, dtype: torch.dtype) -> bool:
    return (dk, dv) in MLA_KERNEL_DIMS and dtype in (torch.bfloat16, torch.float16)


def _mla_fp8(k_cache: torch.Tensor, v_cache: torch.Tensor, dtype: torch.dtype) -> int:
    """1 when the latent caches hold e4m3 bytes, 0 when elements of the model type."""
    assert k_cache.dtype == v_cache.dtype and k_cache.dtype in (dtype, torch.float8_e4m3fn), (k_cache.dtype, dtype)
    return int(k_cache.dtype == torch.float8_e4m3fn)


def mla_cache_write(latent: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                    slot_mapping: torch.Tensor) -> None:
    """``latent[T, DK]`` (normalised c_kv | rotated k_rope) into the key-row cache and, its first DV values,
    into the V^T fragment cache."""
    t, dk = latent.shape
    dv = v_cache.shape[2] * v_cache.shape[4]
    bs = k_cache.shape[2]
    assert latent.stride(1) == 1 and k_cache.shape[1] == 1 and k_cache.shape[3] == dk and k_cache.is_contiguous()
    fp8 = _mla_fp8(k_cache, v_cache, latent.dtype)
    assert v_cache.is_contiguous() and v_cache.shape[3] * v_cache.shape[5] == bs
    assert slot_mapping.dtype == torch.int32 and slot_mapping.numel() >= t
    _check(_LIB.vdamd_mla_cache_write(latent.data_ptr(), latent.stride(0), k_cache.data_ptr(), v_cache.data_ptr(),
                                      slot_mapping.data_ptr(), t, dk, dv, bs, _dt(latent), fp8, _stream()),
           "mla_cache_write")


def mla_num_splits(tokens: int, q_tiles: int, max_seq_len: int) -> int:
    """KV splits: about one workgroup per CU in total and no split shorter than ~192 keys.  A workgroup
    streams ~36 GB/s by itself (one group of 32 keys in flight ahead of the one being multiplied), so the
    chip's bandwidth needs a few hundred of them; beyond that, more splits only add partial sums to write and
    merge (scripts/dev/mla_split_sweep.py: the fastest count is within ~5 % of this rule from batch 1 to 256)."""
    want = -(-256 // max(1, tokens * q_tiles))
    return int(max(1, min(want, max_seq_len // 192, 64)))


def mla_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, block_table: torch.Tensor,
                  token_seq: Optional[torch.Tensor], seq_lens: torch.Tensor, scale: float, max_seq_len: int,
                  out: Optional[torch.Tensor] = None, splits: Optional[int] = None,
                  mode: Optional[int] = None) -> torch.Tensor:
    """``q[T, Hq, DK]`` against the latent cache -> ``[T, Hq, DV]``; token ``t`` reads block-table row
    ``token_seq[t]`` (``t`` itself when None) and sees keys ``0 .. seq_lens[t] - 1``.  ``mode`` (default: by
    shape): 0 = a workgroup per (token, 16 heads); 1 = four consecutive tokens of a sequence per workgroup
    (steps with ``token_seq``: prefill chunks, verification - the tokens of a sequence must be adjacent rows at
    consecutive positions); 2 = four head tiles of a token per workgroup (64 heads per rank and more)."""
    t, hq, dk = q.shape
    dv = v_cache.shape[2] * v_cache.shape[4]
    bs = k_cache.shape[2]
    assert mla_supported(dk, dv, q.dtype), (dk, dv, q.dtype)
    assert q.stride(2) == 1 and q.stride(1) == dk and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert block_table.dtype == torch.int32 and block_table.stride(1) == 1 and seq_lens.dtype == torch.int32
    assert seq_lens.numel() >= t and (token_seq is None or (token_seq.dtype == torch.int32 and token_seq.numel() >= t))
    assert token_seq is not None or block_table.shape[0] >= t
    if out is None:
        out = torch.empty((t, hq, dv), dtype=q.dtype, device=q.device)
    assert out.stride(2) == 1 and out.stride(1) == dv
    q_tiles = -(-hq // 16)
    if mode is None:
        # (four head tiles per workgroup need enough workgroups left to fill the chip)
        mode = 1 if token_seq is not None else (2 if q_tiles >= 4 and t * q_tiles >= 128 else 0)
    if splits is None:
        splits = mla_num_splits(-(-t // 4) if mode == 1 else t, -(-q_tiles // 4) if mode == 2 else q_tiles, max_seq_len)
    part_o = part_ml = None
    if splits > 1:
        part_o = torch.empty((t * hq * splits, dv), dtype=torch.float32, device=q.device)
        part_ml = torch.empty((t * hq * splits, 2), dtype=torch.float32, device=q.device)
    _check(_LIB.vdamd_mla_attention(
        out.data_ptr(), out.stride(0), None if part_o is None else part_o.data_ptr(),
        None if part_ml is None else part_ml.data_ptr(), q.data_ptr(), q.stride(0), k_cache.data_ptr(),
        v_cache.data_ptr(), block_table.data_ptr(), block_table.stride(0),
        None if token_seq is None else token_seq.data_ptr(), seq_lens.data_ptr(), float(scale), t, hq, dk, dv, bs,
        splits, int(mode), _dt(q), _mla_fp8(k_cache, v_cache, q.dtype), _stream()), "mla_attention")
    return out


def mla_prep(q: torch.Tensor, nope: int, q_lat: torch.Tensor, latent: torch.Tensor, norm_weight: torch.Tensor,
             eps: float, cos_sin: torch.Tensor, positions: torch.Tensor, k_cache: Optional[torch.Tensor],
             v_cache: Optional[torch.Tensor], slot_mapping: Optional[torch.Tensor]) -> torch.Tensor:
    """One launch between the projections and the attention of a latent-cache layer: the latent row of every
    token normalised (first ``rank`` values) and rotated (last ``rope`` values) into both cache forms, and the
    absorbed query ``[T, H, rank + rope] = [q_lat | rotated q[..., nope:]]`` returned.  ``q [T, H, nope + rope]``
    and ``q_lat [H, T, rank]`` may be strided views; without a cache (``slot_mapping`` None) only the query is made."""
    t, h, qd = q.shape
    rank = q_lat.shape[2]
    rope = qd - nope
    assert q.stride(2) == 1 and q_lat.stride(2) == 1 and latent.stride(1) == 1 and latent.shape == (t, rank + rope)
    assert q_lat.shape == (h, t, rank) and cos_sin.dtype == torch.float32 and cos_sin.shape[1] == rope
    assert positions.dtype == torch.int32 and norm_weight.dtype == q.dtype == q_lat.dtype == latent.dtype
    out = torch.empty((t, h, rank + rope), dtype=q.dtype, device=q.device)
    bs, fp8 = 8, 0
    if slot_mapping is not None:
        bs = k_cache.shape[2]
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape[3] == rank + rope
        assert slot_mapping.dtype == torch.int32 and slot_mapping.numel() >= t
        fp8 = _mla_fp8(k_cache, v_cache, q.dtype)
    _check(_LIB.vdamd_mla_prep(out.data_ptr(), q.data_ptr(), q.stride(0), q.stride(1), nope, q_lat.data_ptr(),
                               q_lat.stride(0), q_lat.stride(1), latent.data_ptr(), latent.stride(0),
                               norm_weight.data_ptr(), float(eps), cos_sin.data_ptr(), positions.data_ptr(),
                               None if slot_mapping is None else k_cache.data_ptr(),
                               None if slot_mapping is None else v_cache.data_ptr(),
                               None if slot_mapping is None else slot_mapping.data_ptr(), t, h, rank, rope, bs,
                               _dt(q), fp8, _stream()), "mla_prep")
    return out
