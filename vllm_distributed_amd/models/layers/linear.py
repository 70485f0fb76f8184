"""Tensor-parallel linear layers and embeddings.

Column-parallel layers shard the output dimension (no communication); the
row-parallel layer that follows shards the input dimension and all-reduces its
partial sums - two all-reduces per transformer layer.  Weights are stored
``[out, in]`` and each rank loads only its slice (``weight_loader``).
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Iterator, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import envs, ops
from ..utils import materialize
from ...parallel.state import (get_tp_group, get_tensor_model_parallel_rank,
                               get_tensor_model_parallel_world_size,
                               tensor_model_parallel_all_gather,
                               tensor_model_parallel_all_reduce)


def _param(*shape: int, dtype: torch.dtype) -> nn.Parameter:
    return nn.Parameter(torch.empty(*shape, dtype=dtype), requires_grad=False)


# ------------------------------------------------------------ quantisation
@dataclass(frozen=True)
class QuantSpec:
    """Quantisation of the decoder's linear layers.  ``awq``: weight-only 4-bit
    asymmetric, ``group_size`` input columns per (scale, zero) pair, checkpoint
    tensors ``qweight [in, out/8]``, ``qzeros [in/g, out/8]``, ``scales [in/g, out]``.
    ``fp8``: W8A8 on the fp8 matrix cores - weights quantised per output channel
    to OCP e4m3 once loaded, activations per token at run time.
    ``nvfp4``: weight-only E2M1 with one e4m3 scale per 16 input columns and one
    fp32 scale per checkpoint matrix: ``weight_packed [out, in/2] uint8``,
    ``weight_scale [out, in/16] float8_e4m3fn``, ``weight_global_scale [1]``
    (the reference image installs its fp4 kernels "for nvfp4", ``Dockerfile:5-6``)."""
    method: str = "awq"
    group_size: int = 128
    pack: int = 8


_ACTIVE_QUANT: Optional[QuantSpec] = None


@contextlib.contextmanager
def quantized_linears(spec: Optional[QuantSpec]) -> Iterator[None]:
    """Linear layers constructed inside this context are built quantised."""
    global _ACTIVE_QUANT
    prev, _ACTIVE_QUANT = _ACTIVE_QUANT, spec
    try:
        yield
    finally:
        _ACTIVE_QUANT = prev


def active_quant() -> Optional[QuantSpec]:
    return _ACTIVE_QUANT


def _tag(p: nn.Parameter, out_dim: int, pack: int = 1, in_div: int = 1) -> nn.Parameter:
    """Sharding metadata read by the weight loaders: which dim is the output
    dim, how many logical output columns one stored column packs, and by how
    much the input dim is divided (quantisation groups)."""
    p.out_dim, p.pack, p.in_div = out_dim, pack, in_div      # type: ignore[attr-defined]
    return p


def _meta(p: nn.Parameter):
    return getattr(p, "out_dim", 0), getattr(p, "pack", 1), getattr(p, "in_div", 1)


def _scalar_param(n: int) -> nn.Parameter:
    """One fp32 value per output row, loaded from per-matrix scalars: each
    fused sub-matrix (q, k, v / gate, up) fills its own rows."""
    p = nn.Parameter(torch.ones(n, dtype=torch.float32), requires_grad=False)
    p.per_shard_scalar = True                                  # type: ignore[attr-defined]
    return _tag(p, 0)


def _fill_scalar(dst: torch.Tensor, w: torch.Tensor) -> None:
    dst.copy_(materialize(w).reshape(-1)[:1].to(torch.float32).expand(dst.shape))


def nvfp4_rename(name: str, w: torch.Tensor) -> Optional[Tuple[str, torch.Tensor]]:
    """Checkpoint tensor of an NVFP4 linear -> (parameter name, tensor), or None
    for tensors this weight-only path does not use (activation scales).
    compressed-tensors stores ``weight_global_scale`` as a divisor, modelopt
    stores the multiplier ``weight_scale_2`` and keeps the bytes in ``weight``."""
    base, _, kind = name.rpartition(".")
    if kind in ("input_global_scale", "input_scale", "k_scale", "v_scale"):
        return None
    if kind == "weight_scale":
        return base + ".weight_block_scale", w
    if kind == "weight_global_scale":
        return base + ".weight_alpha", w.to(torch.float32).reciprocal()
    if kind == "weight_scale_2":
        return base + ".weight_alpha", w.to(torch.float32)
    if kind == "weight" and w.dtype == torch.uint8:
        return base + ".weight_packed", w
    return name, w


def nvfp4_dequantize(packed: torch.Tensor, block_scale: torch.Tensor, alpha: torch.Tensor,
                     dtype: torch.dtype) -> torch.Tensor:
    """NVFP4 shard tensors -> dense ``[out, in]`` (shapes the 4-bit kernel does not tile)."""
    from ...ops import quant
    codes = torch.stack([packed & 0xF, packed >> 4], dim=-1).reshape(packed.shape[0], -1)
    w = quant.fp4_lut(packed.device)[codes.long()]
    w = w * block_scale.to(torch.float32).repeat_interleave(16, dim=1) * alpha.reshape(-1, 1)
    return w.to(dtype)


def fp4_layout_ok(in_local: int, out_local: int) -> bool:
    return in_local % 128 == 0 and out_local % 16 == 0


def w4_layout_ok(in_local: int, out_local: int, group: int) -> bool:
    return in_local % 128 == 0 and out_local % 16 == 0 and group % 128 == 0 and in_local % group == 0


def awq_to_native(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, dtype: torch.dtype):
    """AWQ shard tensors -> ``(qw, sz)`` in the kernel layout (``ops.quant.pack_w4``)."""
    from ...ops import quant
    q = quant.awq_unpack(qweight).t().contiguous()
    z = quant.awq_unpack(qzeros).t().contiguous()
    return quant.pack_w4(q, z, scales.t().contiguous(), dtype)


class _LinearBase(nn.Module):
    """Dense ``weight [out, in]`` or, under ``quantized_linears``, the AWQ
    checkpoint triple which ``finalize_weights`` repacks once loading is done."""

    def _alloc(self, in_local: int, out_local: int, bias: bool, dtype: torch.dtype,
               quant: Optional[bool] = None) -> None:
        spec = _ACTIVE_QUANT if quant is not False else None
        self.quant = spec
        self.in_local, self.out_local, self.dtype = in_local, out_local, dtype
        self.register_buffer("qw", None)
        self.register_buffer("sz", None)
        self.register_buffer("weight_scale", None)
        self.register_buffer("alpha", None)
        self.register_buffer("dense_shadow", None)
        self.lora = None          # adapter slots, attached by lora.LoRAManager
        if spec is not None and spec.method == "nvfp4":
            assert in_local % 16 == 0, f"NVFP4 shard [{out_local} x {in_local}] splits a 16-column block"
            self.weight_packed = _tag(_param(out_local, in_local // 2, dtype=torch.uint8), 0, 1, 2)
            self.weight_block_scale = _tag(_param(out_local, in_local // 16, dtype=torch.float8_e4m3fn), 0, 1, 16)
            self.weight_alpha = _scalar_param(out_local)
        elif spec is None or spec.method == "fp8":
            # fp8: loaded in the checkpoint dtype, quantised per output channel in finalize_weights
            self.weight = _tag(_param(out_local, in_local, dtype=dtype), 0)
        else:
            g = spec.group_size
            assert in_local % g == 0 and out_local % spec.pack == 0, \
                f"quantised shard [{out_local} x {in_local}] does not align with groups of {g}"
            self.qweight = _tag(_param(in_local, out_local // spec.pack, dtype=torch.int32), 1, spec.pack, 1)
            self.qzeros = _tag(_param(in_local // g, out_local // spec.pack, dtype=torch.int32), 1, spec.pack, g)
            self.scales = _tag(_param(in_local // g, out_local, dtype=dtype), 1, 1, g)
        self.bias = _tag(_param(out_local, dtype=dtype), 0) if bias else None

    def _set_loaders(self, loader, bias_loader=None) -> None:
        for name in ("weight", "qweight", "qzeros", "scales", "weight_packed", "weight_block_scale", "weight_alpha"):
            p = getattr(self, name, None)
            if isinstance(p, nn.Parameter):
                p.weight_loader = loader                         # type: ignore[attr-defined]
        if self.bias is not None:
            self.bias.weight_loader = bias_loader or loader      # type: ignore[attr-defined]

    def finalize_weights(self) -> None:
        if self.quant is None or self.qw is not None or self.weight_scale is not None:
            return
        if self.quant.method == "fp8":
            if self.in_local % 16 or self.out_local % 16:       # hipBLASLt fp8 alignment
                self.quant = None
                return
            wq, scale = ops.fp8_quantize_weight(self.weight.data)
            self.weight = nn.Parameter(wq, requires_grad=False)
            self.weight_scale = scale
            return
        if self.quant.method == "nvfp4":
            packed, bs, alpha = self.weight_packed.data, self.weight_block_scale.data, self.weight_alpha.data
            del self.weight_packed, self.weight_block_scale, self.weight_alpha
            if fp4_layout_ok(self.in_local, self.out_local):
                from ...ops import quant
                self.qw, self.sz = quant.pack_fp4(packed, bs)
                self.alpha = alpha.contiguous()
                ops.w4_reserve_scratch(self.in_local * self.out_local, self.qw.device, self.dtype)
                self._make_shadow()
            else:
                self.quant = None
                self.weight = nn.Parameter(nvfp4_dequantize(packed, bs, alpha, self.dtype), requires_grad=False)
            return
        qweight, qzeros, scales = self.qweight.data, self.qzeros.data, self.scales.data
        del self.qweight, self.qzeros, self.scales
        if w4_layout_ok(self.in_local, self.out_local, self.quant.group_size):
            qw, sz = awq_to_native(qweight, qzeros, scales, self.dtype)
            self.qw, self.sz = qw, sz
            ops.w4_reserve_scratch(qw.shape[0] * 16 * qw.shape[1] * 128, qw.device, self.dtype)
            self._make_shadow()
        else:      # shapes the 4-bit kernel does not tile: dense fallback
            from ..loader import awq_dequantize
            self.quant = None
            self.weight = nn.Parameter(awq_dequantize(qweight, qzeros, scales, self.dtype), requires_grad=False)

    def pad_weight_rows(self, pad: int) -> bool:
        """Re-home a dense 16-bit GPU weight in a buffer whose rows are ``pad`` elements longer; the
        parameter becomes the ``[:, :in]`` view.  Rows of a power-of-two length all start on the same
        HBM channels: a GEMM that streams 16 rows at a time then reads at 5.3-5.5 TB/s, with 256 bytes
        of padding at 6.0-6.2 (measured with ``benchmarks/lab/read_bw.hip``; the library GEMMs at
        M <= 64 gain 5-10 % from the same change)."""
        w = getattr(self, "weight", None)
        if (pad <= 0 or self.qw is not None or self.weight_scale is not None or not isinstance(w, nn.Parameter)
                or w.dim() != 2 or not w.is_cuda or w.dtype not in (torch.bfloat16, torch.float16)
                or not w.is_contiguous() or (w.shape[1] * 2) % 1024 or w.shape[0] < 64):
            return False
        n, k = w.shape
        buf = torch.empty((n, k + pad), dtype=w.dtype, device=w.device)
        buf[:, :k].copy_(w.data)
        buf[:, k:].zero_()
        w.data = buf[:, :k]
        return True

    def _matmul(self, x: torch.Tensor, bias: Optional[torch.Tensor],
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        y = self._base_matmul(x, bias, out)
        if self.lora is not None:
            y = self.lora.apply(x, y)
        return y

    def _make_shadow(self) -> None:
        """With ``VDAMD_W4_DENSE_SHADOW=1`` a 4-bit layer also keeps its weights
        dequantised: batches above the fused kernels' range then run the
        library GEMM on the resident copy instead of dequantising on every
        call.  4.5x the weight memory, which 288 GB of HBM can afford for
        models up to ~70B parameters."""
        from ... import envs
        if not envs.VDAMD_W4_DENSE_SHADOW or self.qw is None or not self.qw.is_cuda or not ops.hip_available():
            return
        from ...ops import hip
        if self.alpha is not None:
            self.dense_shadow = hip.fp4_dequant(self.qw, self.sz, self.alpha, self.dtype)
        else:
            self.dense_shadow = hip.w4_dequant(self.qw, self.sz)

    def _base_matmul(self, x: torch.Tensor, bias: Optional[torch.Tensor],
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``out`` is a hint (dense 16-bit weights honour it): callers use what is returned."""
        if self.qw is not None:
            if self.alpha is not None:
                return ops.fp4_linear(x, self.qw, self.sz, self.alpha, bias, self.dense_shadow)
            return ops.w4a16_linear(x, self.qw, self.sz, bias, self.dense_shadow)
        if self.weight_scale is not None:
            return ops.fp8_linear(x, self.weight, self.weight_scale, bias)
        return ops.linear(x, self.weight, bias, out if x.dim() == 2 else None)


def pad_dense_weight_rows(model: nn.Module, pad: int) -> int:
    """``pad_weight_rows`` on every dense linear of ``model``; returns how many were re-homed."""
    n, freed = 0, 0
    for m in model.modules():
        if isinstance(m, _LinearBase) and m.pad_weight_rows(pad):
            n += 1
            freed += m.weight.numel() * m.weight.element_size()
            if freed > (4 << 30):        # the old, shorter buffers cannot host the new ones: hand them back
                torch.cuda.empty_cache()
                freed = 0
    if n:
        torch.cuda.empty_cache()
    return n


def divide(a: int, b: int) -> int:
    assert a % b == 0, f"{a} is not divisible by {b}"
    return a // b


class ReplicatedLinear(_LinearBase):
    def __init__(self, in_features: int, out_features: int, bias: bool, dtype: torch.dtype,
                 quant: Optional[bool] = None):
        super().__init__()
        self._alloc(in_features, out_features, bias, dtype, quant)
        self._set_loaders(self._load)

    @staticmethod
    def _load(param: nn.Parameter, w: torch.Tensor, shard_id=None) -> None:
        if getattr(param, "per_shard_scalar", False):
            return _fill_scalar(param.data, w)
        assert param.shape == w.shape, (param.shape, w.shape)
        param.data.copy_(materialize(w))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._matmul(x, self.bias)


class ColumnParallelLinear(_LinearBase):
    """``y = x W^T`` with ``W`` split along ``out``.  ``output_sizes`` lists the
    logical sub-matrices fused into this layer (e.g. gate and up), each sharded
    independently so a rank's slice stays ``[gate_i | up_i]``."""

    def __init__(self, in_features: int, output_sizes: Sequence[int], bias: bool,
                 dtype: torch.dtype, gather_output: bool = False):
        super().__init__()
        self.tp = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.output_sizes = list(output_sizes)
        self.shard_sizes = [divide(s, self.tp) for s in self.output_sizes]
        self.gather_output = gather_output
        self._alloc(in_features, sum(self.shard_sizes), bias, dtype)
        self._set_loaders(self.weight_loader)

    def weight_loader(self, param: nn.Parameter, w: torch.Tensor, shard_id: Optional[int] = None) -> None:
        """``shard_id`` = index into ``output_sizes``; None = the full fused matrix."""
        od, pack, _ = _meta(param)
        if getattr(param, "per_shard_scalar", False):
            if shard_id is None:
                return _fill_scalar(param.data, w)
            return _fill_scalar(param.data.narrow(0, sum(self.shard_sizes[:shard_id]), self.shard_sizes[shard_id]), w)
        if shard_id is None:
            assert w.shape[od] * pack == sum(self.output_sizes), (w.shape, self.output_sizes)
            off = 0
            for i, s in enumerate(self.output_sizes):
                self.weight_loader(param, w.narrow(od, off // pack, s // pack), i)
                off += s
            return
        size = self.shard_sizes[shard_id] // pack
        dst_off = sum(self.shard_sizes[:shard_id]) // pack
        src = w.narrow(od, self.tp_rank * size, size)
        param.data.narrow(od, dst_off, size).copy_(materialize(src))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self._matmul(x, self.bias)
        if self.gather_output and self.tp > 1:
            y = tensor_model_parallel_all_gather(y, dim=-1)
        return y

    def forward_silu(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """``silu(gate) * up`` straight from ``x`` for a ``[gate ; up]`` layer when one kernel can do
        it (dense 16-bit weights, large batch), else None: the caller falls back to ``forward`` plus
        the activation."""
        if (self.qw is not None or self.weight_scale is not None or self.lora is not None or self.bias is not None
                or self.gather_output or len(self.shard_sizes) != 2 or self.shard_sizes[0] != self.shard_sizes[1]):
            return None
        return ops.gate_up_silu(x, self.weight)


class QKVParallelLinear(_LinearBase):
    """Fused q/k/v projection.  Query heads are split across ranks; KV heads
    are split when there are at least ``tp`` of them, else replicated."""

    def __init__(self, hidden_size: int, head_dim: int, num_heads: int, num_kv_heads: int,
                 bias: bool, dtype: torch.dtype):
        super().__init__()
        self.tp = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.head_dim = head_dim
        self.total_heads, self.total_kv_heads = num_heads, num_kv_heads
        self.num_heads = divide(num_heads, self.tp)
        if num_kv_heads >= self.tp:
            self.num_kv_heads = divide(num_kv_heads, self.tp)
            self.kv_replicas = 1
        else:
            self.num_kv_heads = 1
            self.kv_replicas = divide(self.tp, num_kv_heads)
        self.q_size = self.num_heads * head_dim
        self.kv_size = self.num_kv_heads * head_dim
        self._alloc(hidden_size, self.q_size + 2 * self.kv_size, bias, dtype)
        self._set_loaders(self.weight_loader)

    def weight_loader(self, param: nn.Parameter, w: torch.Tensor, shard_id: Optional[str] = None) -> None:
        od, pack, _ = _meta(param)
        if getattr(param, "per_shard_scalar", False):
            if shard_id is None:
                return _fill_scalar(param.data, w)
            size = self.q_size if shard_id == "q" else self.kv_size
            dst = 0 if shard_id == "q" else self.q_size + (0 if shard_id == "k" else self.kv_size)
            return _fill_scalar(param.data.narrow(0, dst, size), w)
        if shard_id is None:      # full fused [q | k | v]
            qs, ks = self.total_heads * self.head_dim // pack, self.total_kv_heads * self.head_dim // pack
            self.weight_loader(param, w.narrow(od, 0, qs), "q")
            self.weight_loader(param, w.narrow(od, qs, ks), "k")
            self.weight_loader(param, w.narrow(od, qs + ks, ks), "v")
            return
        if shard_id == "q":
            size, dst, src_idx = self.q_size, 0, self.tp_rank
        else:
            size = self.kv_size
            dst = self.q_size + (0 if shard_id == "k" else self.kv_size)
            src_idx = self.tp_rank // self.kv_replicas
        size, dst = size // pack, dst // pack
        param.data.narrow(od, dst, size).copy_(materialize(w.narrow(od, src_idx * size, size)))

    def forward_partials(self, x: torch.Tensor) -> Optional["PartialSums"]:
        """The packed ``[q | k | v]`` left as the fp32 K-split slabs of the short-batch kernels (dense
        16-bit or AWQ weights, no bias, no adapter) for the rotary + cache kernel to sum; None when the
        shape does not run split or the layer is not of that kind."""
        if (self.weight_scale is not None or self.lora is not None or self.bias is not None
                or self.alpha is not None or x.dim() != 2 or not x.is_cuda or not envs.VDAMD_FUSED_REDUCE_ROPE):
            return None
        slabs = ops.w4a16_partials(x, self.qw, self.sz) if self.qw is not None else ops.linear_partials(x, self.weight)
        return None if slabs is None else PartialSums(slabs, x.dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.qw is None and self.weight_scale is None and self.lora is None and x.dim() == 2:
            # large batches: the q columns alone tile the chip exactly, the k/v columns follow separately
            y = ops.linear_split(x, self.weight, self.q_size)
            if y is not None:
                return y if self.bias is None else y.add_(self.bias)
        return self._matmul(x, self.bias)


class RowParallelLinear(_LinearBase):
    def __init__(self, in_features: int, out_features: int, bias: bool, dtype: torch.dtype,
                 reduce_results: bool = True):
        super().__init__()
        self.tp = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.reduce_results = reduce_results
        self._alloc(divide(in_features, self.tp), out_features, bias, dtype)
        self._set_loaders(self.weight_loader, ReplicatedLinear._load)

    def weight_loader(self, param: nn.Parameter, w: torch.Tensor, shard_id=None) -> None:
        if getattr(param, "per_shard_scalar", False):
            return _fill_scalar(param.data, w)
        od, _, in_div = _meta(param)
        n = self.in_local // in_div
        param.data.copy_(materialize(w.narrow(1 - od, self.tp_rank * n, n)))

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False) -> torch.Tensor:
        """``reduce=False`` returns this rank's partial sums; the caller owns the
        all-reduce (used to overlap it with another micro-batch's compute, or to
        fuse it with the following add + norm).  ``staged``: the partial sums are
        written where the TP group's next direct all-reduce reads its input."""
        out = None
        if staged and self.tp > 1 and x.dim() == 2:
            out = get_tp_group().stage_view((x.shape[0], self.out_local), x.dtype)
        # bias is added once (rank 0) so the all-reduce does not multiply it
        y = self._matmul(x, self.bias if self.tp_rank == 0 else None, out)
        if reduce and self.reduce_results and self.tp > 1:
            y = tensor_model_parallel_all_reduce(y)
        return y

    def forward_partials(self, x: torch.Tensor) -> Optional["PartialSums"]:
        """The product left as the fp32 K-split slabs of the short-batch kernels (an unsharded dense
        16-bit or AWQ layer, no bias, no adapter) for the add + RMSNorm that follows to sum; None when
        the shape does not run split or the layer is not of that kind."""
        if (self.tp > 1 or self.weight_scale is not None or self.lora is not None or self.bias is not None
                or self.alpha is not None or x.dim() != 2 or not x.is_cuda):
            return None
        slabs = ops.w4a16_partials(x, self.qw, self.sz) if self.qw is not None else ops.linear_partials(x, self.weight)
        return None if slabs is None else PartialSums(slabs, x.dtype)

    def forward_add(self, x: torch.Tensor, residual: torch.Tensor) -> bool:
        """``residual += x W^T`` in the GEMM's epilogue, for an unsharded dense 16-bit layer and a
        batch the large-batch kernel takes; False (nothing done) otherwise.  The add + RMSNorm that
        follows then only has to normalise ``residual``."""
        if (self.tp > 1 or self.qw is not None or self.weight_scale is not None or self.lora is not None
                or self.bias is not None or x.dim() != 2):
            return False
        return ops.linear_add(x, self.weight, residual)

    def forward_gated(self, gate_up: torch.Tensor, reduce: bool = True, staged: bool = False) -> torch.Tensor:
        """``forward(silu(gate) * up)`` for ``gate_up = [gate | up]``.  A dense 16-bit layer hands the
        pair to the GEMM, which forms the activation while it stages its operand (short batches on
        the weight-streaming kernel); every other layer runs the activation kernel first."""
        if (self.qw is not None or self.weight_scale is not None or self.lora is not None or self.bias is not None
                or gate_up.dim() != 2):
            return self.forward(ops.silu_and_mul(gate_up), reduce, staged)
        out = None
        if staged and self.tp > 1:
            out = get_tp_group().stage_view((gate_up.shape[0], self.out_local), gate_up.dtype)
        y = ops.silu_mul_linear(gate_up, self.weight, out)
        if reduce and self.reduce_results and self.tp > 1:
            y = tensor_model_parallel_all_reduce(y)
        return y


class PartialSums:
    """A projection's output as fp32 slabs ``[ks, T, H]`` whose sum over ``ks`` is the value.  Only an
    ``RMSNorm`` called with a residual consumes it as is; everybody else asks for ``materialize()``."""
    __slots__ = ("slabs", "dtype")

    def __init__(self, slabs: torch.Tensor, dtype: torch.dtype):
        self.slabs, self.dtype = slabs, dtype

    @property
    def shape(self):
        return self.slabs.shape[1:]

    def materialize(self) -> torch.Tensor:
        return self.slabs.sum(0).to(self.dtype)

    def add_rms_norm(self, residual: torch.Tensor, weight: torch.Tensor, eps: float):
        return ops.reduce_add_rms_norm(self.slabs, residual, weight, eps)


def pad_vocab(vocab_size: int, tp: int, multiple: int = 64) -> int:
    m = multiple * tp
    return (vocab_size + m - 1) // m * m


class VocabParallelEmbedding(nn.Module):
    def __init__(self, vocab_size: int, hidden_size: int, dtype: torch.dtype, out_scale: Optional[float] = None):
        super().__init__()
        # factor on the looked-up rows (not on the weight: a tied LM head reads it unscaled); rounded
        # to the activation dtype first, as the defining implementations do
        self.out_scale = out_scale
        self.tp = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.vocab_size = vocab_size
        self.padded_vocab = pad_vocab(vocab_size, self.tp)
        self.rows = self.padded_vocab // self.tp
        self.vocab_start = self.tp_rank * self.rows
        self.weight = _param(self.rows, hidden_size, dtype=dtype)
        self.weight.weight_loader = self.weight_loader      # type: ignore[attr-defined]

    def weight_loader(self, param: nn.Parameter, w: torch.Tensor, shard_id=None) -> None:
        start = self.vocab_start
        end = min(start + self.rows, w.shape[0])
        n = max(0, end - start)
        if n > 0:
            param.data[:n].copy_(materialize(w[start:end]))
        if n < self.rows:
            param.data[n:].zero_()

    def forward(self, ids: torch.Tensor, reduce: bool = True, staged: bool = False) -> torch.Tensor:
        """``reduce=False`` returns this rank's partial rows (zero for ids of
        other shards); the caller owns the reduction (``staged``: see ``RowParallelLinear``)."""
        dst = None
        if staged and self.tp > 1:
            dst = get_tp_group().stage_view((ids.shape[0], self.weight.shape[1]), self.weight.dtype)
        out = ops.embedding(ids, self.weight, self.vocab_start, masked=self.tp > 1, out=dst)
        if self.out_scale is not None:
            # (a host scalar rounded to the activation dtype: a device tensor made here would be a host-to-device
            # copy in every step, which a hipGraph capture refuses)
            out.mul_(float(torch.tensor(self.out_scale, dtype=out.dtype)))
        if reduce and self.tp > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class ParallelLMHead(VocabParallelEmbedding):
    """Vocab-sharded output projection; ``logits`` all-gathers so every rank of
    the TP group samples from the same full-vocab row."""

    def __init__(self, vocab_size: int, hidden_size: int, dtype: torch.dtype, bias: bool = False):
        super().__init__(vocab_size, hidden_size, dtype)
        self.bias = _param(self.rows, dtype=dtype) if bias else None
        if bias:
            self.bias.weight_loader = self.weight_loader      # type: ignore[attr-defined]  (same vocab shard)

    def logits(self, hidden: torch.Tensor) -> torch.Tensor:
        y = ops.linear(hidden, self.weight, self.bias)
        if self.tp > 1:
            y = tensor_model_parallel_all_gather(y, dim=-1)
        return y[..., :self.vocab_size]

    def greedy_tokens(self, hidden: torch.Tensor) -> torch.Tensor:
        """Vocab-parallel argmax: each rank reduces its own shard to one
        ``(max, index)`` pair per row and only those pairs are all-gathered -
        ``8 * B`` bytes per rank instead of a ``[B, vocab]`` logits gather.
        Ties resolve to the lowest token id, like a full-row argmax."""
        y = ops.linear(hidden, self.weight, self.bias)
        valid = max(0, min(self.rows, self.vocab_size - self.vocab_start))
        if valid == 0:
            val = torch.full((y.shape[0],), float("-inf"), dtype=torch.float32, device=y.device)
            idx = torch.zeros(y.shape[0], dtype=torch.int64, device=y.device)
        else:
            idx = ops.argmax(y[:, :valid]).long()
            val = y[:, :valid].gather(1, idx.unsqueeze(1)).squeeze(1).float()
        if self.tp == 1:
            return idx.to(torch.int32)
        pack = torch.stack([val, (idx + self.vocab_start).float()], dim=1)      # ids < 2^24: exact in fp32
        g = tensor_model_parallel_all_gather(pack, dim=0).view(self.tp, -1, 2)
        vals, ids = g[..., 0], g[..., 1]
        best = vals.max(dim=0, keepdim=True).values
        ids = torch.where(vals == best, ids, torch.full_like(ids, float(1 << 30)))
        return ids.min(dim=0).values.to(torch.int32)
