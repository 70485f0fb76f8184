"""Op dispatch: hand-written gfx950 HIP kernels on GPU, PyTorch reference on CPU.

``VDAMD_OPS_BACKEND`` = ``auto`` (default: HIP for CUDA tensors, torch for CPU
tensors), ``hip`` (fail if the extension is missing) or ``torch`` (reference
everywhere).  On a GPU box a missing extension is an error, not a silent
fallback: numbers measured on the reference path would be meaningless.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import envs
from . import torch_ref
from .torch_ref import k_cache_shape, v_cache_shape  # noqa: F401  (the paged-cache layouts every backend shares)

_hip = None
_hip_error: Optional[BaseException] = None


def _load_hip():
    global _hip, _hip_error
    if _hip is not None or _hip_error is not None:
        return _hip
    try:
        from . import hip as hip_mod
        hip_mod.load()
        _hip = hip_mod
    except BaseException as e:  # noqa: BLE001
        _hip_error = e
    return _hip


def hip_available() -> bool:
    return _load_hip() is not None


def hip_supports_head_dim(d: int) -> bool:
    return hip_available() and _hip.supports_head_dim(d)


# How often attention ran on the HIP kernels and how often a GPU tensor took the PyTorch reference instead
# (a head size the kernels are not built for).  Workers report it in ``get_stats``; the reference path is
# announced once per head size, loudly: numbers measured on it say nothing about the kernels.
attention_calls = {"hip": 0, "reference_on_gpu": 0}
_announced_fallbacks = set()


def _note_attention(q, on_hip: bool) -> None:
    if on_hip:
        attention_calls["hip"] += 1
        return
    if not q.is_cuda:
        return
    attention_calls["reference_on_gpu"] += 1
    d = int(q.shape[-1])
    if d not in _announced_fallbacks:
        _announced_fallbacks.add(d)
        from ..logger import init_logger
        init_logger(__name__).warning(
            "attention with head size %d runs on the PyTorch REFERENCE path on the GPU (the HIP kernels are built for "
            "%s): correct, but slow and not hipGraph-friendly", d, sorted(_hip.HEAD_DIMS) if _hip is not None else "?")


def _use_hip(t: torch.Tensor) -> bool:
    if not t.is_cuda:
        return False
    backend = envs.VDAMD_OPS_BACKEND
    if backend == "torch":
        return False
    if _load_hip() is None:
        raise RuntimeError(
            "HIP kernel library is not available on a GPU tensor path "
            f"({_hip_error!r}); build it with `python -m vllm_distributed_amd.build` "
            "or set VDAMD_OPS_BACKEND=torch to use the slow reference ops")
    return True


def rms_norm(x, weight, eps):
    if _use_hip(x):
        return _hip.rms_norm(x, weight, eps)
    return torch_ref.rms_norm(x, weight, eps)


def fused_add_rms_norm(x, residual, weight, eps) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns ``(normed, new_residual)``; the HIP path updates both in place."""
    if _use_hip(x):
        return _hip.fused_add_rms_norm(x, residual, weight, eps)
    return torch_ref.fused_add_rms_norm(x, residual, weight, eps)


def reduce_add_rms_norm(partial, residual, weight, eps) -> Tuple[torch.Tensor, torch.Tensor]:
    """``fused_add_rms_norm`` whose first operand arrives as fp32 K-split slabs ``[ks, T, H]``."""
    if _use_hip(residual):
        return _hip.reduce_add_rms_norm(partial, residual, weight, eps)
    return torch_ref.fused_add_rms_norm(partial.sum(0).to(residual.dtype), residual, weight, eps)


def linear_partials(x, weight):
    """``x @ weight^T`` left as the fp32 K-split slabs ``[ks, M, N]`` of the short-batch kernels, for a
    consumer that reduces them itself (the add + RMSNorm that follows a row-parallel projection); None
    where the shape runs unsplit or on another kernel."""
    if not (x.is_cuda and x.dim() == 2 and 0 < x.shape[0] <= 256 and envs.VDAMD_SKINNY_GEMM
            and envs.VDAMD_FUSED_REDUCE_NORM and _use_hip(x)):
        return None
    m, n, k = x.shape[0], weight.shape[0], weight.shape[1]
    if _hip.mid_profitable(m, n, k) and _hip.mid_supported(x, weight):
        return _hip.mid_gemm(x, weight, partials=True)
    if m <= 128 and _hip.skinny_profitable(m, n, k) and _hip.skinny_supported(x, weight):
        return _hip.skinny_gemm_partials(x, weight)
    return None


def w4a16_partials(x, qw, sz):
    """The fused 4-bit GEMM left as its fp32 K-split slabs (None: unsplit shape or a batch it does not take)."""
    if not (x.dim() == 2 and _use_hip(x) and x.dtype in (torch.bfloat16, torch.float16)
            and 0 < x.shape[0] <= _hip.W4_MAX_FUSED_M and envs.VDAMD_FUSED_REDUCE_NORM):
        return None
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    return _hip.w4a16_gemm(x, qw, sz, partials=True)


def layer_norm(x, weight, bias, eps):
    if _use_hip(x):
        return _hip.layer_norm(x, weight, bias, eps)
    return torch_ref.layer_norm(x, weight, bias, eps)


def silu_and_mul(x):
    if _use_hip(x):
        return _hip.silu_and_mul(x)
    return torch_ref.silu_and_mul(x)


def rotary_embedding(positions, q, k, cos_sin_cache, head_dim):
    # the HIP kernel rotates whole heads; a partial rotation (GPT-NeoX's rotary_pct) takes the reference ops
    if _use_hip(q) and cos_sin_cache.shape[-1] == head_dim:
        return _hip.rotary_embedding(positions, q, k, cos_sin_cache, head_dim)
    return torch_ref.rotary_embedding(positions, q, k, cos_sin_cache, head_dim)


def reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, kv_scales=None):
    """``kv_scales = (k_scale[Hkv], v_scale[Hkv])`` fp32: only read for an e4m3 cache."""
    if _use_hip(k):
        return _hip.reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, kv_scales)
    return torch_ref.reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, kv_scales)


def rope_and_cache(positions, qkv, cos_sin_cache, k_cache, v_cache, slot_mapping,
                   num_heads, num_kv_heads, head_dim, q_norm_weight=None, k_norm_weight=None, norm_eps=0.0,
                   kv_scales=None):
    if _use_hip(qkv):
        return _hip.rope_and_cache(positions, qkv, cos_sin_cache, k_cache, v_cache, slot_mapping,
                                   num_heads, num_kv_heads, head_dim, q_norm_weight, k_norm_weight, norm_eps,
                                   kv_scales)
    return torch_ref.rope_and_cache(positions, qkv, cos_sin_cache, k_cache, v_cache, slot_mapping,
                                    num_heads, num_kv_heads, head_dim, q_norm_weight, k_norm_weight, norm_eps,
                                    kv_scales)


def rope_cache_fused_ok(head_dim: int) -> bool:
    """Whether the HIP rotary + cache kernel takes this head size (per-head norm included)."""
    lanes = head_dim // 16
    return head_dim % 16 == 0 and lanes & (lanes - 1) == 0 and lanes <= 64 and hip_available()


def rope_and_cache_partials(positions, slabs, dtype, cos_sin_cache, k_cache, v_cache, slot_mapping,
                            num_heads, num_kv_heads, head_dim, q_norm_weight=None, k_norm_weight=None, norm_eps=0.0,
                            kv_scales=None):
    """``rope_and_cache`` on a projection left as K-split slabs (``linear_partials`` / ``w4a16_partials``;
    HIP only): returns the rotated ``q``."""
    return _hip.rope_and_cache_partials(positions, slabs, dtype, cos_sin_cache, k_cache, v_cache, slot_mapping,
                                        num_heads, num_kv_heads, head_dim, q_norm_weight, k_norm_weight, norm_eps,
                                        kv_scales)


def paged_attention(q, k_cache, v_cache, block_table, query_start_loc, seq_lens, scale,
                    max_query_len: int = 0, max_seq_len: int = 0, out=None, kv_scales=None, window: int = 0,
                    softcap: float = 0.0, sinks=None, alibi_slopes=None):
    on_hip = _use_hip(q) and _hip.supports_head_dim(q.shape[-1])
    _note_attention(q, on_hip)
    if on_hip:
        return _hip.paged_attention_prefill(q, k_cache, v_cache, block_table, query_start_loc,
                                            seq_lens, scale, max_query_len, max_seq_len, out, kv_scales, window,
                                            softcap, sinks, alibi_slopes)
    res = torch_ref.paged_attention(q, k_cache, v_cache, block_table, query_start_loc,
                                    seq_lens, scale, kv_scales, window, softcap, sinks, alibi_slopes)
    if out is not None:
        out.copy_(res)
        return out
    return res


def paged_attention_decode(q, k_cache, v_cache, block_table, seq_lens, scale,
                           max_seq_len: int = 0, out=None, workspace=None, kv_scales=None, window: int = 0,
                           softcap: float = 0.0, sinks=None, alibi_slopes=None):
    on_hip = _use_hip(q) and _hip.supports_head_dim(q.shape[-1])
    _note_attention(q, on_hip)
    if on_hip:
        return _hip.paged_attention_decode(q, k_cache, v_cache, block_table, seq_lens, scale,
                                           max_seq_len, out, workspace, kv_scales=kv_scales, window=window,
                                           softcap=softcap, sinks=sinks, alibi_slopes=alibi_slopes)
    res = torch_ref.paged_attention_decode(q, k_cache, v_cache, block_table, seq_lens, scale, kv_scales, window,
                                           softcap, sinks, alibi_slopes)
    if out is not None:
        out.copy_(res)
        return out
    return res


def sample(logits, temperature, top_k, top_p, min_p, uniform):
    if _use_hip(logits):
        return _hip.sample(logits, temperature, top_k, top_p, min_p, uniform)
    return torch_ref.sample(logits, temperature, top_k, top_p, min_p, uniform)


def argmax(logits):
    if _use_hip(logits):
        return _hip.argmax(logits)
    return logits.argmax(-1)


def embedding(ids, weight, vocab_start=0, masked=False, out=None):
    if _use_hip(weight):
        return _hip.embedding(ids, weight, vocab_start, out)
    res = torch_ref.embedding(ids, weight, vocab_start, masked)
    if out is not None:
        out.copy_(res)
        return out
    return res


def linear(x, weight, bias=None, out=None):
    """``x @ weight^T (+ bias)``: the library GEMM, except for short batches on
    long-reduction shapes where the weight-streaming HIP kernel is faster.
    ``out`` (2-D ``x`` only): write the result there (e.g. straight into the
    buffer the following all-reduce reads)."""
    if x.is_cuda and x.dim() == 2 and x.shape[0] <= 256 and envs.VDAMD_SKINNY_GEMM and _use_hip(x):
        m, n, k = x.shape[0], weight.shape[0], weight.shape[1]
        if (envs.VDAMD_STREAM_GEMM and _hip.stream_profitable(m, n, k) and _hip.stream_supported(x, weight)
                and (out is None or out.stride(0) == n)):
            y = _hip.stream_gemm(x, weight, out=out)
            return y if bias is None else y.add_(bias)
        if _hip.mid_profitable(m, n, k) and _hip.mid_supported(x, weight):
            y = _hip.mid_gemm(x, weight, out=out)
            return y if bias is None else y.add_(bias)
        if m <= 128 and _hip.skinny_profitable(m, n, k) and _hip.skinny_supported(x, weight):
            y = _hip.skinny_gemm(x, weight, out=out)
            return y if bias is None else y.add_(bias)
    if (x.is_cuda and x.dim() == 2 and envs.VDAMD_BIG_GEMM and _use_hip(x) and x.shape[0] >= _hip.BIG_MIN_M
            and _hip.big_profitable(x.shape[0], weight.shape[0], weight.shape[1])
            and _hip.big_supported(x, weight) and _big_out_ok(out, x.shape[0])):
        y = _hip.big_gemm(x, weight, out=out)
        return y if bias is None else y.add_(bias)
    if out is not None and x.dim() == 2:
        if bias is None:
            return torch.mm(x, weight.t(), out=out)
        return torch.addmm(bias, x, weight.t(), out=out)
    return torch.nn.functional.linear(x, weight, bias)


_GELU_KINDS = {"gelu": "none", "gelu_new": "tanh", "gelu_pytorch_tanh": "tanh", "gelu_fast": "tanh",
               "gelu_tanh": "tanh"}


def activation(x, kind: str):
    """Element-wise activation by its Hugging Face name."""
    if kind in ("silu", "swish"):
        return torch.nn.functional.silu(x)
    if kind == "relu":
        return torch.relu(x)
    if kind == "relu2":
        return torch.relu(x).square_()
    if kind in _GELU_KINDS:
        return torch.nn.functional.gelu(x, approximate=_GELU_KINDS[kind])
    raise ValueError(f"unsupported activation {kind!r}")


def act_and_mul(x, kind: str):
    """``act(x[..., :F]) * x[..., F:]`` for a gated MLP with an activation other than SiLU (GeGLU)."""
    if kind in ("silu", "swish"):
        return silu_and_mul(x)
    f = x.shape[-1] // 2
    return activation(x[..., :f], kind) * x[..., f:]


def _big_out_ok(out, m):
    return out is None or (out.stride(1) == 1 and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0
                           and m * out.stride(0) * 2 < (1 << 32) - 64)


def linear_split(x, weight, split: int):
    """``x @ weight^T`` as two GEMMs over ``weight[:split]`` and ``weight[split:]`` writing one buffer,
    when the first part alone fills the chip in whole rounds of the large-batch kernel's tiles while the
    whole matrix would not (a 70B qkv projection at 2048 rows: 8192 q columns = 256 tiles = one round,
    10240 = 1.25 rounds; 249 us against 276 us for the library on the whole matrix); None otherwise."""
    if not (x.is_cuda and x.dim() == 2 and envs.VDAMD_BIG_GEMM and _use_hip(x) and x.shape[0] >= _hip.BIG_MIN_M):
        return None
    m, n, k = x.shape[0], weight.shape[0], weight.shape[1]
    if (not 0 < split < n or split % 8 or _hip.big_profitable(m, n, k) or not _hip.big_profitable(m, split, k)
            or not _hip.big_supported(x, weight[:split])):
        return None
    out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    _hip.big_gemm(x, weight[:split], out=out[:, :split])
    torch.mm(x, weight[split:].t(), out=out[:, split:])
    return out


def linear_add(x, weight, residual) -> bool:
    """``residual += x @ weight^T`` inside the large-batch GEMM (its epilogue reads and rewrites the
    residual tile it has just computed); False when that kernel does not take the shape."""
    if (x.is_cuda and x.dim() == 2 and envs.VDAMD_BIG_GEMM and _use_hip(x) and x.shape[0] >= _hip.BIG_MIN_M
            and _hip.big_profitable(x.shape[0], weight.shape[0], weight.shape[1])
            and _hip.big_supported(x, weight) and residual.shape == (x.shape[0], weight.shape[0])
            and residual.dtype == x.dtype and _big_out_ok(residual, x.shape[0])):
        _hip.big_gemm(x, weight, out=residual, residual=residual)
        return True
    return False


def gate_up_silu(x, weight):
    """``silu(x @ Wg^T) * (x @ Wu^T)`` for ``weight = [Wg ; Wu]`` in one kernel (the large-batch GEMM
    forms the activation in its epilogue: the ``[M, 2F]`` intermediate is never written), or ``None``
    when that kernel does not take the shape - the caller then runs the two-step path.  A handful of rows
    take the barrier-free weight-streaming kernel, which has the same epilogue."""
    if (x.is_cuda and x.dim() == 2 and envs.VDAMD_STREAM_GEMM and envs.VDAMD_SKINNY_GEMM and _use_hip(x)
            and _hip.stream_profitable(x.shape[0], weight.shape[0], weight.shape[1])
            and _hip.stream_supported(x, weight) and weight.shape[0] % 64 == 0):
        return _hip.stream_gemm(x, weight, silu_mul=True)
    if (x.is_cuda and x.dim() == 2 and envs.VDAMD_BIG_GEMM and _use_hip(x) and x.shape[0] >= _hip.BIG_MIN_M_SILU
            and _hip.big_profitable(x.shape[0], weight.shape[0], weight.shape[1], True)
            and _hip.big_supported(x, weight, True)):
        return _hip.big_gemm(x, weight, silu_mul=True)
    return None


def silu_mul_linear(gate_up, weight, out=None):
    """``(silu(gate) * up) @ weight^T`` for ``gate_up = [gate | up]``: one kernel when the
    weight-streaming GEMM takes the shape (it forms the activation while staging its operand),
    otherwise the activation kernel followed by ``linear``."""
    if (gate_up.is_cuda and gate_up.dim() == 2 and gate_up.shape[0] <= 128 and envs.VDAMD_SKINNY_GEMM
            and envs.VDAMD_FUSE_SILU_GEMM and _use_hip(gate_up)):
        m, n, k = gate_up.shape[0], weight.shape[0], weight.shape[1]
        if (gate_up.shape[1] == 2 * k and not _hip.mid_profitable(m, n, k) and _hip.skinny_profitable(m, n, k)
                and _hip.skinny_supported(gate_up[:, :k], weight)):
            return _hip.skinny_gemm(gate_up, weight, out=out, silu_mul=True)
    return linear(silu_and_mul(gate_up), weight, None, out)


FP8_MAX = 448.0        # largest finite OCP e4m3 value


def fp8_quantize_weight(w):
    """Per-output-channel symmetric quantisation of ``w [N, K]`` to e4m3:
    ``(wq float8_e4m3fn, scale [N] fp32)`` with ``w ~ wq * scale[:, None]``."""
    wf = w.to(torch.float32)
    scale = wf.abs().amax(dim=1).clamp_min(1e-12) / FP8_MAX
    wq = (wf / scale.unsqueeze(1)).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return wq, scale


def fp8_linear(x, wq, w_scale, bias=None):
    """W8A8 linear: activations quantised per token on the fly (HIP kernel),
    weights per output channel, product on the fp8 matrix cores through
    hipBLASLt (``torch._scaled_mm``, fp32 accumulation, row-wise scaling).
    Off the GPU the same arithmetic is done in fp32 from the dequantised
    operands."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    if _use_hip(x2) and x2.dtype in (torch.bfloat16, torch.float16) and x2.shape[0] > 0:
        if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        xq, xs = _hip.dynamic_fp8_quant(x2)
        y = torch._scaled_mm(xq, wq.t(), scale_a=xs, scale_b=w_scale.view(1, -1), bias=bias, out_dtype=x2.dtype)
    else:
        xf = x2.to(torch.float32)
        xs = xf.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / FP8_MAX
        xq = (xf / xs).clamp_(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
        y = (xq.to(torch.float32) * xs) @ (wq.to(torch.float32) * w_scale.view(-1, 1)).t()
        if bias is not None:
            y = y + bias.to(torch.float32)
        y = y.to(x.dtype)
    return y.view(shape[:-1] + (y.shape[-1],))


_w4_scratch = {}


def _scratch_key(device, dtype):
    d = torch.device(device)
    return (d.index if d.index is not None else torch.cuda.current_device(), dtype)


def w4_reserve_scratch(numel: int, device, dtype) -> None:
    """Size the dequantisation scratch for the largest quantised matrix up
    front: captured hipGraphs keep its address, so it must never be
    reallocated while serving."""
    if torch.device(device).type != "cuda":
        return
    key = _scratch_key(device, dtype)
    buf = _w4_scratch.get(key)
    if buf is None or buf.numel() < numel:
        _w4_scratch[key] = torch.empty(numel, dtype=dtype, device=device)


def w4a16_linear(x, qw, sz, bias=None, dense=None):
    """``y = x . dequant(qw, sz)^T (+ bias)`` for weights packed by
    ``ops.quant.pack_w4``.  Small batches run the fused 4-bit MFMA kernel (the
    weights cross HBM once, as 4-bit); large batches dequantise into a shared
    scratch matrix and use the library GEMM, which is compute-bound there
    (``dense``: a resident dequantised copy to use instead, see
    ``VDAMD_W4_DENSE_SHADOW``)."""
    from . import quant
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    if _use_hip(x2) and x2.dtype in (torch.bfloat16, torch.float16):
        if x2.shape[0] <= _hip.W4_MAX_FUSED_M:
            if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
                x2 = x2.contiguous()
            y = _hip.w4a16_gemm(x2, qw, sz)
        elif dense is not None:
            y = torch.nn.functional.linear(x2, dense)
        else:
            n, k = qw.shape[0] * 16, qw.shape[1] * 128
            key = _scratch_key(x2.device, sz.dtype)
            buf = _w4_scratch.get(key)
            if buf is None or buf.numel() < n * k:
                buf = _w4_scratch[key] = torch.empty(n * k, dtype=sz.dtype, device=x2.device)
            y = torch.nn.functional.linear(x2, _hip.w4_dequant(qw, sz, buf))
    else:
        y = torch.nn.functional.linear(x2, quant.w4_dequant(qw, sz, x2.dtype))
    if bias is not None:
        y = y + bias
    return y.view(shape[:-1] + (y.shape[-1],))


def fp4_linear(x, qw, fs, alpha=None, bias=None, dense=None):
    """``y = x . dequant(qw, fs, alpha)^T (+ bias)`` for NVFP4 weights packed by
    ``ops.quant.pack_fp4`` (weight-only: activations stay bf16/fp16).  Same
    split as ``w4a16_linear``: fused 4-bit kernel for small batches, dequantise
    + library GEMM for large ones."""
    from . import quant
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    if _use_hip(x2) and x2.dtype in (torch.bfloat16, torch.float16):
        if x2.shape[0] <= _hip.W4_MAX_FUSED_M:
            if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
                x2 = x2.contiguous()
            y = _hip.fp4a16_gemm(x2, qw, fs, alpha)
        elif dense is not None:
            y = torch.nn.functional.linear(x2, dense)
        else:
            n, k = qw.shape[0] * 16, qw.shape[1] * 128
            key = _scratch_key(x2.device, x2.dtype)
            buf = _w4_scratch.get(key)
            if buf is None or buf.numel() < n * k:
                buf = _w4_scratch[key] = torch.empty(n * k, dtype=x2.dtype, device=x2.device)
            y = torch.nn.functional.linear(x2, _hip.fp4_dequant(qw, fs, alpha, x2.dtype, buf))
    else:
        y = torch.nn.functional.linear(x2, quant.fp4_dequant(qw, fs, alpha, x2.dtype))
    if bias is not None:
        y = y + bias
    return y.view(shape[:-1] + (y.shape[-1],))


def moe_align(topi, num_experts: int, tile_rows: int, expert_start: int = 0, local_experts: int = -1):
    """Sort the ``T*k`` (token, expert) slots by expert and cut each expert's
    run into row tiles, entirely on the device (no host sync, so a decode
    step with experts stays hipGraph-capturable).

    Returns ``(order, tiles)``: ``order [T*k]`` maps sorted position -> flat
    slot ``t*k + j``; ``tiles [n, 3] int32`` = (local expert, first sorted row,
    rows), ``n = T*k // tile_rows + E`` (an upper bound), rows = 0 for unused
    entries and for experts this rank does not hold."""
    flat = topi.reshape(-1)
    tk = flat.numel()
    order = flat.argsort(stable=True)
    counts = torch.zeros(num_experts, dtype=torch.int64, device=flat.device)
    counts.scatter_add_(0, flat, torch.ones_like(flat))
    ends = counts.cumsum(0)
    tpe = (counts + tile_rows - 1) // tile_rows
    tends = tpe.cumsum(0)
    n = tk // tile_rows + num_experts
    j = torch.arange(n, device=flat.device)
    e = torch.searchsorted(tends, j, right=True).clamp_(max=num_experts - 1)
    within = j - (tends - tpe)[e]
    row0 = (ends - counts)[e] + within * tile_rows
    rows = (counts[e] - within * tile_rows).clamp_(min=0, max=tile_rows)
    rows = torch.where(j < tends[-1], rows, torch.zeros_like(rows))
    if local_experts >= 0:
        local = (e >= expert_start) & (e < expert_start + local_experts)
        rows = torch.where(local, rows, torch.zeros_like(rows))
        e = (e - expert_start).clamp_(min=0, max=max(local_experts - 1, 0))
    tiles = torch.stack([e, row0, rows], dim=1).to(torch.int32).contiguous()
    return order, tiles


def moe_route_fused_ok(logits, num_experts: int, top_k: int) -> bool:
    """Whether the three-kernel HIP routing path (``csrc/moe_route.hip``) covers this router."""
    return (_use_hip(logits) and logits.dim() == 2 and num_experts <= _hip.MOE_MAX_EXPERTS
            and top_k <= _hip.MOE_MAX_TOPK and logits.dtype in (torch.bfloat16, torch.float16, torch.float32))


def moe_topk(logits, k: int, renormalize: bool):
    return _hip.moe_topk(logits, k, renormalize)


def moe_topk_grouped(logits, bias, k: int, n_group: int, topk_group: int, sigmoid: bool, group_top2: bool,
                     renormalize: bool, scale: float):
    return _hip.moe_topk_grouped(logits, bias, k, n_group, topk_group, sigmoid, group_top2, renormalize, scale)


def moe_align_fused(topi, num_experts: int, tile_rows: int, expert_start: int = 0, local_experts: int = -1):
    return _hip.moe_align(topi, num_experts, tile_rows, expert_start, local_experts)


def moe_route_small_ok(logits, x) -> bool:
    """Whether the one-launch routing path takes this (decode-sized) batch (``logits`` None: not computed yet)."""
    return (x.dim() == 2 and 0 < x.shape[0] <= _hip.MOE_SMALL_MAX_TOKENS and x.dtype in (torch.bfloat16, torch.float16)
            and (logits is None or logits.dtype in (x.dtype, torch.float32)) and x.stride(1) == 1
            and x.stride(0) % 8 == 0 and x.shape[1] % 8 == 0 and x.data_ptr() % 16 == 0)


def router_logit_partials(x, weight):
    """Router GEMM for a decode-sized batch with its K-split reduction left to the routing kernel: fp32
    ``[splits, T, E]``, or None when the plain GEMM should run."""
    if not (_use_hip(x) and weight.dtype == x.dtype and moe_route_small_ok(None, x)):
        return None
    return _hip.skinny_gemm_partials(x, weight)


def moe_route_small(logits, x, k: int, renormalize: bool, tile_rows: int, expert_start: int = 0,
                    local_experts: int = -1):
    return _hip.moe_route_small(logits, x, k, renormalize, tile_rows, expert_start, local_experts)


def moe_combine(y, topw, topi, inv, expert_start: int = 0, local_experts: int = -1, bias=None):
    return _hip.moe_combine(y, topw, topi, inv, expert_start, local_experts, bias)


def bias_clamped_glu(h, bias, topi, inv, expert_start: int, local_experts: int, limit: float, alpha: float):
    """gpt-oss expert activation with the per-expert bias, on expert-sorted rows (HIP only)."""
    return _hip.bias_clamped_glu(h, bias, topi, inv, expert_start, local_experts, limit, alpha)


def moe_combine_add_rms_norm(y, topw, topi, inv, expert_start, local_experts, residual, weight, eps):
    """``fused_add_rms_norm(moe_combine(...), residual, weight, eps)`` in one launch (HIP only)."""
    return _hip.moe_combine_add_rms_norm(y, topw, topi, inv, expert_start, local_experts, residual, weight, eps)


def grouped_linear(x, tiles, tile_rows: int, weight=None, qw=None, sz=None, alpha=None, silu_mul: bool = False):
    """Row-grouped ``x @ W[g]^T`` (dense ``weight [G, N, K]``, packed 4-bit
    integer ``qw/sz``, or NVFP4 ``qw`` + int32 ``sz`` + ``alpha [G, N]``) for rows
    sorted by group - see ``moe_align``.  HIP only.  ``silu_mul``: ``x`` is ``[rows, 2K] = [gate | up]``
    and the product is taken with ``silu(gate) * up`` (inside the kernel where ``grouped_silu_fused`` says
    so, through the activation kernel otherwise)."""
    assert _use_hip(x)
    if silu_mul and not grouped_silu_fused(tile_rows, qw, sz):
        x, silu_mul = silu_and_mul(x), False
    if qw is not None:
        if sz.dtype == torch.int32:
            return _hip.grouped_fp4a16_gemm(x, qw, sz, alpha, tiles, tile_rows)
        return _hip.grouped_w4a16_gemm(x, qw, sz, tiles, tile_rows, silu_mul=silu_mul)
    return _hip.grouped_gemm(x, weight, tiles, tile_rows)


def grouped_silu_fused(tile_rows: int, qw, sz) -> bool:
    """Whether the grouped GEMM of these weights forms ``silu(gate) * up`` itself at this tile size
    (integer 4-bit experts, few rows per expert: decode)."""
    return (qw is not None and sz.dtype != torch.int32 and tile_rows <= _hip.W4_GROUPED_SILU_MAX_ROWS
            and envs.VDAMD_MOE_FUSED_SILU)


apply_penalties = torch_ref.apply_penalties
compute_logprobs = torch_ref.compute_logprobs
build_cos_sin_cache = torch_ref.build_cos_sin_cache


# ------------------------------------------------ multi-head latent attention over a latent cache (DeepSeek)
def mla_kernel_ok(dk: int, dv: int, dtype: torch.dtype) -> bool:
    """Whether the HIP latent-attention kernel is built for this key / value width."""
    return hip_available() and _hip.mla_supported(dk, dv, dtype)


def mla_cache_write(latent, k_cache, v_cache, slot_mapping):
    if _use_hip(latent) and latent.dtype in (torch.bfloat16, torch.float16):
        return _hip.mla_cache_write(latent, k_cache, v_cache, slot_mapping)      # (the cache may be e4m3)
    return torch_ref.mla_cache_write(latent, k_cache, v_cache, slot_mapping)


def mla_attention(q, k_cache, v_cache, block_table, token_seq, seq_lens, scale, max_seq_len, out=None):
    on_hip = _use_hip(q) and _hip.mla_supported(q.shape[2], v_cache.shape[2] * v_cache.shape[4], q.dtype)
    _note_attention(q, on_hip)
    if on_hip:
        return _hip.mla_attention(q, k_cache, v_cache, block_table, token_seq, seq_lens, scale, max_seq_len, out)
    return torch_ref.mla_attention(q, k_cache, v_cache, block_table, token_seq, seq_lens, scale, max_seq_len, out)


def mla_prep_ok(x: torch.Tensor) -> bool:
    """Whether the one-launch query assembly + latent cache write (HIP) takes this tensor."""
    return _use_hip(x) and x.dtype in (torch.bfloat16, torch.float16)


def mla_prep(q, nope, q_lat, latent, norm_weight, eps, cos_sin, positions, k_cache, v_cache, slot_mapping):
    return _hip.mla_prep(q, nope, q_lat, latent, norm_weight, eps, cos_sin, positions, k_cache, v_cache, slot_mapping)
