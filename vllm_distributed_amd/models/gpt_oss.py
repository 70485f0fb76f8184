"""gpt-oss (``GptOssForCausalLM``): Llama-shaped attention with q/k/v/o biases, alternating sliding-window and
global layers, YaRN rotary scaling and one learned *attention sink* per head (an extra score that only enters
the softmax's denominator: ``csrc/attention.hip`` ``sinks``), over a sparse MLP whose experts carry biases and
a clamped SwiGLU ``(up + 1) * gate * sigmoid(1.702 * gate)``; the router has a bias and weighs the chosen
experts by a softmax over their logits.

Checkpoints hold the experts stacked: ``gate_up_proj [E, H, 2I]`` with gate and up columns interleaved,
``down_proj [E, I, H]``, and one bias row per expert for each; the original release stores the two matrices as
MXFP4 (``*_blocks`` 16 bytes = 32 E2M1 values, ``*_scales`` one power-of-two exponent per block), which is
widened to the model dtype at load (parity unpinned: no such checkpoint is at hand, the decoder is tested
against an encoder written from the format's definition).

The real shapes (hidden = intermediate = 2880) are not multiples of the 128-deep K tiles of the grouped expert
GEMMs, so the expert matrices are zero-padded along their reduction dimension (2880 -> 2944) when they are
allocated; zero rows / columns change nothing, and decode steps then run sync-free (hipGraph-capturable).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import VllmConfig
from ..parallel.state import tensor_model_parallel_all_reduce
from .layers.linear import ReplicatedLinear
from .layers.moe import FusedMoE
from .llama import LlamaAttention, LlamaDecoderLayer, LlamaForCausalLM
from .utils import materialize

_EP = False
FP4_VALUES = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)


def _ceil128(n: int) -> int:
    return (n + 127) // 128 * 128


def mxfp4_dequantize(blocks: torch.Tensor, scales: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``blocks [..., G, 16] uint8`` (two E2M1 values per byte, low nibble first) and ``scales [..., G] uint8``
    (biased power-of-two exponents, one per 32 values) -> ``[..., G * 32]``."""
    lut = torch.tensor(FP4_VALUES + tuple(-v for v in FP4_VALUES), dtype=torch.float32, device=blocks.device)
    b = blocks.to(torch.int64)
    vals = torch.stack([lut[b & 0xF], lut[b >> 4]], dim=-1).reshape(*blocks.shape[:-1], 32)
    vals = torch.ldexp(vals, (scales.to(torch.int32) - 127).unsqueeze(-1))
    return vals.reshape(*blocks.shape[:-2], blocks.shape[-2] * 32).to(dtype)


def mxfp4_quantize(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Round-to-nearest MXFP4 of ``w [..., K]`` (``K % 32 == 0``): the inverse of ``mxfp4_dequantize`` up to
    rounding, used by the tests to build checkpoints of the original release's shape."""
    g = w.float().reshape(*w.shape[:-1], w.shape[-1] // 32, 32)
    amax = g.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -126)
    exp = torch.ceil(torch.log2(amax / 6.0)).clamp(-127, 127)
    lut = torch.tensor(FP4_VALUES, dtype=torch.float32, device=w.device)
    scaled = g / torch.exp2(exp)
    idx = (scaled.abs().unsqueeze(-1) - lut).abs().argmin(-1)
    idx = idx + 8 * (scaled < 0).to(idx.dtype)
    blocks = (idx[..., 0::2] | (idx[..., 1::2] << 4)).to(torch.uint8)
    return blocks, (exp.squeeze(-1) + 127).to(torch.uint8)


class GptOssMoE(FusedMoE):
    ALPHA = 1.702

    def __init__(self, cfg: dict, dtype: torch.dtype, expert_parallel: bool):
        super().__init__(cfg["hidden_size"], cfg["intermediate_size"], cfg["num_local_experts"],
                         cfg["num_experts_per_tok"], True, dtype, expert_parallel=expert_parallel,
                         allocate_experts=False)
        if self.quant is not None:
            raise NotImplementedError("packed 4-bit gpt-oss experts (AWQ / NVFP4); MXFP4 checkpoints are widened at load")
        self.gate = ReplicatedLinear(self.hidden, self.num_experts, True, dtype, quant=False)
        self.limit = float(cfg.get("swiglu_limit", 7.0))
        E, I, H = self.local_experts, self.inter_local, self.hidden
        self.h_pad, self.i_pad = _ceil128(H), _ceil128(I)

        def zeros(*shape):
            return nn.Parameter(torch.zeros(*shape, dtype=dtype), requires_grad=False)
        # [gate rows ; up rows], each padded to i_pad rows, reduction padded to h_pad; and [H, i_pad]
        self.w13 = zeros(E, 2 * self.i_pad, self.h_pad)
        self.w2 = zeros(E, H, self.i_pad)
        self.b13 = zeros(E, 2 * self.i_pad)
        self.b2 = zeros(E, H)

    # ------------------------------------------------------------------ loading (stacked checkpoint tensors)
    def _rows(self) -> slice:
        lo = 0 if self.expert_parallel else self.tp_rank * self.inter_local
        return slice(lo, lo + self.inter_local)

    def load_stacked(self, kind: str, w: torch.Tensor, scales: Optional[torch.Tensor] = None) -> None:
        """``kind``: ``gate_up_proj`` ``[E, H, 2I]`` / ``gate_up_proj_bias`` ``[E, 2I]`` / ``down_proj``
        ``[E, I, H]`` / ``down_proj_bias`` ``[E, H]``; with ``scales`` the matrix arrives as MXFP4 blocks
        in the original release's orientation (``[E, 2I, H/32, 16]`` and ``[E, H, I/32, 16]``)."""
        t = materialize(w)
        s = None if scales is None else materialize(scales)
        I, ip, H, rows = self.inter_local, self.i_pad, self.hidden, self._rows()
        for eid in range(self.num_experts):
            e = self._local_index(eid)
            if e is None:
                continue
            if kind == "gate_up_proj":
                m = mxfp4_dequantize(t[eid], s[eid], self.dtype) if s is not None else t[eid].t()   # [2I, H]
                self.w13.data[e].zero_()                 # (the padding included: the loader poisons fresh parameters)
                self.w13.data[e, :I, :H].copy_(m[0::2][rows])
                self.w13.data[e, ip:ip + I, :H].copy_(m[1::2][rows])
            elif kind == "gate_up_proj_bias":
                self.b13.data[e].zero_()
                self.b13.data[e, :I].copy_(t[eid][0::2][rows])
                self.b13.data[e, ip:ip + I].copy_(t[eid][1::2][rows])
            elif kind == "down_proj":
                m = mxfp4_dequantize(t[eid], s[eid], self.dtype) if s is not None else t[eid].t()   # [H, I]
                self.w2.data[e].zero_()
                self.w2.data[e, :, :I].copy_(m[:, rows])
            elif kind == "down_proj_bias":
                # the partial sums of sharded experts are added up over the ranks: the bias goes in once
                self.b2.data[e].zero_()
                if self.expert_parallel or self.tp_rank == 0:
                    self.b2.data[e].copy_(t[eid])
            else:
                raise KeyError(kind)

    # ------------------------------------------------------------------ compute
    def _act(self, h: torch.Tensor) -> torch.Tensor:
        gate, up = h[..., :self.i_pad], h[..., self.i_pad:]
        gate = gate.clamp(max=self.limit)
        up = up.clamp(min=-self.limit, max=self.limit)
        return (up + 1) * (gate * torch.sigmoid(gate * self.ALPHA))

    def _pad(self, x: torch.Tensor) -> torch.Tensor:
        return x if self.h_pad == self.hidden else F.pad(x, (0, self.h_pad - self.hidden))

    def _expert_mlp(self, e: int, x: torch.Tensor) -> torch.Tensor:
        h = F.linear(self._pad(x), self.w13[e], self.b13[e])
        return F.linear(self._act(h), self.w2[e], self.b2[e])

    def _grouped_tile_rows(self, x: torch.Tensor) -> int:
        if self.num_experts > 512 or self.top_k > 16:
            return 0
        return super()._grouped_tile_rows(x)

    def _forward_fused_routing(self, x: torch.Tensor, logits: torch.Tensor, tile_rows: int, deferred: bool = False):
        """Top-k + softmax over the chosen logits, slot sort and tile table on the device, two grouped
        GEMMs with each row's expert bias added in between and after, weighted sum: no host sync."""
        local = self.local_experts if self.expert_parallel else -1
        topw, topi = ops.moe_topk(logits, self.top_k, True)
        src_row, inv, tiles = ops.moe_align_fused(topi, self.num_experts, tile_rows, self.expert_start, local)
        xs = self._pad(x).index_select(0, src_row)
        h = ops.grouped_linear(xs, tiles, tile_rows, weight=self.w13)
        a = ops.bias_clamped_glu(h, self.b13, topi, inv, self.expert_start, local, self.limit, self.ALPHA)
        y = ops.grouped_linear(a, tiles, tile_rows, weight=self.w2)
        return ops.moe_combine(y, topw, topi, inv, self.expert_start, local, bias=self.b2)

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False, deferred: bool = False):
        logits = self.gate(x)
        tile_rows = self._grouped_tile_rows(x)
        if tile_rows and x.dim() == 2 and ops.moe_route_fused_ok(logits, self.num_experts, self.top_k):
            out = self._forward_fused_routing(x, logits, tile_rows)
        else:
            topw, topi = self.route(logits.float())       # softmax, top-k, renormalise == softmax over the top-k logits
            topw = topw.to(x.dtype)
            out = torch.zeros_like(x)
            flat_e = topi.reshape(-1)
            order = flat_e.argsort(stable=True)
            counts = torch.bincount(flat_e, minlength=self.num_experts).tolist()
            tok = order // self.top_k
            wts = topw.reshape(-1)[order]
            start = 0
            for eid, n in enumerate(counts):
                if n == 0:
                    continue
                seg = slice(start, start + n)
                start += n
                e = self._local_index(eid)
                if e is None:
                    continue
                rows = tok[seg]
                out.index_add_(0, rows, self._expert_mlp(e, x.index_select(0, rows)) * wts[seg].unsqueeze(1))
        if reduce and self.tp > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class GptOssAttention(LlamaAttention):
    def __init__(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor):
        super().__init__(cfg, layer_idx, dtype, cos_sin)
        # one sink per local query head, in fp32 as the kernels read it
        self.sinks = nn.Parameter(torch.zeros(self.num_heads, dtype=torch.float32), requires_grad=False)
        self.sinks.weight_loader = self._load_sinks

    def _load_sinks(self, param: nn.Parameter, w: torch.Tensor, shard_id=None) -> None:
        qp = self.qkv_proj
        param.data.copy_(materialize(w).float()[qp.tp_rank * self.num_heads:(qp.tp_rank + 1) * self.num_heads])

    def forward(self, *args, **kwargs):
        self.attn.sinks = self.sinks
        return super().forward(*args, **kwargs)


class GptOssDecoderLayer(LlamaDecoderLayer):
    def _make_attn(self, cfg: dict, layer_idx: int, dtype: torch.dtype, cos_sin: torch.Tensor) -> nn.Module:
        return GptOssAttention(cfg, layer_idx, dtype, cos_sin)

    def _make_mlp(self, cfg: dict, dtype: torch.dtype) -> nn.Module:
        return GptOssMoE(cfg, dtype, _EP)


class GptOssForCausalLM(LlamaForCausalLM):
    layer_cls = GptOssDecoderLayer
    # the experts are widened to the model dtype at load whatever the checkpoint says about them
    packed_quantization = False

    def __init__(self, vllm_config: VllmConfig):
        global _EP
        _EP = vllm_config.parallel_config.enable_expert_parallel
        cfg = vllm_config.model_config.hf_config
        cfg.setdefault("attention_bias", True)
        super().__init__(vllm_config)

    def map_weight_name(self, name: str) -> Optional[str]:
        name = super().map_weight_name(name)
        if name is None:
            return None
        return name.replace(".mlp.router.", ".mlp.gate.")

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> List[str]:
        loaded: List[str] = []
        pending: Dict[str, torch.Tensor] = {}        # MXFP4: a matrix's blocks and scales arrive separately

        def others():
            for name, w in weights:
                mapped = self.map_weight_name(name)
                if mapped is None or ".mlp.experts." not in mapped:
                    yield name, w
                    continue
                if not self._owns_layer(mapped):
                    continue
                prefix, kind = mapped.split(".mlp.experts.")
                moe = self.layers[prefix.split("layers.")[1].split(".")[0]].mlp
                if kind.endswith(("_blocks", "_scales")):
                    base = kind[:-7]
                    other = pending.pop(f"{prefix}.{base}", None)
                    if other is None:
                        pending[f"{prefix}.{base}"] = materialize(w)      # (valid only while its file is open)
                        continue
                    blocks, scales = (materialize(w), other) if kind.endswith("_blocks") else (other, materialize(w))
                    moe.load_stacked(base, blocks, scales)
                else:
                    moe.load_stacked(kind, w)
                loaded.append(mapped)
        dense = super().load_weights(others())
        if pending:
            raise ValueError(f"MXFP4 expert tensors without their counterpart: {sorted(pending)}")
        return loaded + dense
