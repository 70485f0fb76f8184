"""Sparse mixture-of-experts MLP (Qwen3-MoE routing).

Two sharding modes over the TP group:

* tensor-sharded experts (default, what the reference's TP=2 deployment of
  Qwen3-Coder-480B-A35B uses - ``.env.server:11`` has no expert-parallel flag):
  every rank holds all experts with ``intermediate/tp`` columns;
* expert parallel (``enable_expert_parallel``): every rank holds ``E/tp`` whole
  experts and only computes the tokens routed to them.

Either way partial outputs are summed with one all-reduce.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ..utils import materialize
import torch.nn.functional as F

from ... import envs, ops
from ...parallel.state import (get_tensor_model_parallel_rank,
                               get_tensor_model_parallel_world_size,
                               tensor_model_parallel_all_reduce)
from .linear import (ReplicatedLinear, _fill_scalar, active_quant, awq_to_native, divide, fp4_layout_ok,
                     nvfp4_dequantize, w4_layout_ok)


class ExpertRows:
    """A sparse block's output before the weighted sum over each token's experts: rows ``y`` in
    expert-sorted order plus the routing tables.  An ``RMSNorm`` called with a residual sums, adds and
    normalises in one launch; everybody else asks for ``materialize()``."""
    __slots__ = ("y", "topw", "topi", "inv", "expert_start", "local")

    def __init__(self, y, topw, topi, inv, expert_start, local):
        self.y, self.topw, self.topi, self.inv, self.expert_start, self.local = y, topw, topi, inv, expert_start, local

    @property
    def shape(self):
        return (self.topi.shape[0], self.y.shape[1])

    def materialize(self) -> torch.Tensor:
        return ops.moe_combine(self.y, self.topw, self.topi, self.inv, self.expert_start, self.local)

    def add_rms_norm(self, residual: torch.Tensor, weight: torch.Tensor, eps: float):
        return ops.moe_combine_add_rms_norm(self.y, self.topw, self.topi, self.inv, self.expert_start, self.local,
                                            residual, weight, eps)


class FusedMoE(nn.Module):
    def __init__(self, hidden: int, inter: int, num_experts: int, top_k: int,
                 renormalize: bool, dtype: torch.dtype, expert_parallel: bool = False,
                 allocate_experts: bool = True):
        # allocate_experts = False: a subclass lays out its own (dense) expert tensors
        super().__init__()
        self.tp = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.num_experts, self.top_k, self.renormalize = num_experts, top_k, renormalize
        self.expert_parallel = expert_parallel and self.tp > 1
        if self.expert_parallel:
            self.local_experts = divide(num_experts, self.tp)
            self.expert_start = self.tp_rank * self.local_experts
            self.inter_local = inter
        else:
            self.local_experts, self.expert_start = num_experts, 0
            self.inter_local = divide(inter, self.tp)
        self.inter = inter
        self.hidden, self.dtype = hidden, dtype
        self.gate = ReplicatedLinear(hidden, num_experts, False, dtype, quant=False)   # router stays dense
        self.quant = active_quant()
        if self.quant is not None and self.quant.method not in ("awq", "nvfp4"):
            self.quant = None           # fp8 covers the dense linears; experts stay in the model dtype
        E, I = self.local_experts, self.inter_local
        self.register_buffer("qw13", None)
        self.register_buffer("sz13", None)
        self.register_buffer("qw2", None)
        self.register_buffer("sz2", None)
        self.register_buffer("alpha13", None)
        self.register_buffer("alpha2", None)

        def param(*shape, dt=dtype, out_dim=0, pack=1, in_div=1, loader=None):
            p = nn.Parameter(torch.empty(*shape, dtype=dt), requires_grad=False)
            p.out_dim, p.pack, p.in_div, p.weight_loader = out_dim, pack, in_div, loader
            return p
        if not allocate_experts:
            return
        if self.quant is None:
            self.w13 = param(E, 2 * I, hidden, loader=self._load_w13)
            self.w2 = param(E, hidden, I, loader=self._load_w2)
        elif self.quant.method == "nvfp4":
            assert hidden % 16 == 0 and I % 16 == 0, f"NVFP4 expert shard [{I} x {hidden}] splits a 16-column block"
            u8, f8, f32 = torch.uint8, torch.float8_e4m3fn, torch.float32
            self.w13_weight_packed = param(E, 2 * I, hidden // 2, dt=u8, in_div=2, loader=self._load_w13)
            self.w13_weight_block_scale = param(E, 2 * I, hidden // 16, dt=f8, in_div=16, loader=self._load_w13)
            self.w13_weight_alpha = param(E, 2 * I, dt=f32, loader=self._load_w13)
            self.w2_weight_packed = param(E, hidden, I // 2, dt=u8, in_div=2, loader=self._load_w2)
            self.w2_weight_block_scale = param(E, hidden, I // 16, dt=f8, in_div=16, loader=self._load_w2)
            self.w2_weight_alpha = param(E, hidden, dt=f32, loader=self._load_w2)
            self.w13_weight_alpha.per_shard_scalar = self.w2_weight_alpha.per_shard_scalar = True
        else:
            g, pk = self.quant.group_size, self.quant.pack
            assert hidden % g == 0 and I % g == 0, \
                f"expert shard [{I} x {hidden}] does not align with quantisation groups of {g}"
            i32 = torch.int32
            self.w13_qweight = param(E, hidden, 2 * I // pk, dt=i32, out_dim=1, pack=pk, loader=self._load_w13)
            self.w13_qzeros = param(E, hidden // g, 2 * I // pk, dt=i32, out_dim=1, pack=pk, in_div=g,
                                    loader=self._load_w13)
            self.w13_scales = param(E, hidden // g, 2 * I, out_dim=1, in_div=g, loader=self._load_w13)
            self.w2_qweight = param(E, I, hidden // pk, dt=i32, out_dim=1, pack=pk, loader=self._load_w2)
            self.w2_qzeros = param(E, I // g, hidden // pk, dt=i32, out_dim=1, pack=pk, in_div=g,
                                   loader=self._load_w2)
            self.w2_scales = param(E, I // g, hidden, out_dim=1, in_div=g, loader=self._load_w2)

    def _local_index(self, expert_id: int) -> Optional[int]:
        e = expert_id - self.expert_start
        return e if 0 <= e < self.local_experts else None

    def _load_w13(self, param: nn.Parameter, w: torch.Tensor, shard_id) -> None:
        expert_id, which = shard_id          # which: 0 = gate, 1 = up
        e = self._local_index(expert_id)
        if e is None:
            return
        od, pack = param.out_dim, param.pack
        n = self.inter_local // pack
        if getattr(param, "per_shard_scalar", False):
            return _fill_scalar(param.data[e].narrow(0, which * n, n), w)
        if not self.expert_parallel:
            w = w.narrow(od, self.tp_rank * n, n)
        param.data[e].narrow(od, which * n, n).copy_(materialize(w))

    def _load_w2(self, param: nn.Parameter, w: torch.Tensor, shard_id) -> None:
        expert_id, _ = shard_id
        e = self._local_index(expert_id)
        if e is None:
            return
        if getattr(param, "per_shard_scalar", False):
            return _fill_scalar(param.data[e], w)
        if not self.expert_parallel:
            n = self.inter_local // param.in_div
            w = w.narrow(1 - param.out_dim, self.tp_rank * n, n)
        param.data[e].copy_(materialize(w))

    def finalize_weights(self) -> None:
        """Repack the experts' AWQ shards into the 4-bit kernel layout (or
        dequantise them when a shard shape does not tile)."""
        if self.quant is None or self.qw13 is not None:
            return
        E, I, H, g = self.local_experts, self.inter_local, self.hidden, self.quant.group_size
        if self.quant.method == "nvfp4":
            return self._finalize_nvfp4()
        native = w4_layout_ok(H, 2 * I, g) and w4_layout_ok(I, H, g)
        parts = {"13": [], "2": []}
        for key in ("13", "2"):
            qweight = getattr(self, f"w{key}_qweight").data
            qzeros = getattr(self, f"w{key}_qzeros").data
            scales = getattr(self, f"w{key}_scales").data
            for e in range(E):
                if native:
                    parts[key].append(awq_to_native(qweight[e], qzeros[e], scales[e], self.dtype))
                else:
                    from ..loader import awq_dequantize
                    parts[key].append(awq_dequantize(qweight[e], qzeros[e], scales[e], self.dtype))
            delattr(self, f"w{key}_qweight")
            delattr(self, f"w{key}_qzeros")
            delattr(self, f"w{key}_scales")
        if native:
            self.qw13 = torch.stack([p[0] for p in parts["13"]])
            self.sz13 = torch.stack([p[1] for p in parts["13"]])
            self.qw2 = torch.stack([p[0] for p in parts["2"]])
            self.sz2 = torch.stack([p[1] for p in parts["2"]])
            ops.w4_reserve_scratch(2 * I * H, self.qw13.device, self.dtype)
        else:
            self.quant = None
            self.w13 = nn.Parameter(torch.stack(parts["13"]), requires_grad=False)
            self.w2 = nn.Parameter(torch.stack(parts["2"]), requires_grad=False)

    def _finalize_nvfp4(self) -> None:
        from ...ops import quant
        E, I, H = self.local_experts, self.inter_local, self.hidden
        native = fp4_layout_ok(H, 2 * I) and fp4_layout_ok(I, H)
        out = {}
        for key in ("13", "2"):
            packed = getattr(self, f"w{key}_weight_packed").data
            bs = getattr(self, f"w{key}_weight_block_scale").data
            alpha = getattr(self, f"w{key}_weight_alpha").data
            for name in ("weight_packed", "weight_block_scale", "weight_alpha"):
                delattr(self, f"w{key}_{name}")
            if native:
                parts = [quant.pack_fp4(packed[e], bs[e]) for e in range(E)]
                out[key] = (torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), alpha.contiguous())
            else:
                out[key] = torch.stack([nvfp4_dequantize(packed[e], bs[e], alpha[e], self.dtype) for e in range(E)])
        if native:
            self.qw13, self.sz13, self.alpha13 = out["13"]
            self.qw2, self.sz2, self.alpha2 = out["2"]
            ops.w4_reserve_scratch(2 * I * H, self.qw13.device, self.dtype)
        else:
            self.quant = None
            self.w13 = nn.Parameter(out["13"], requires_grad=False)
            self.w2 = nn.Parameter(out["2"], requires_grad=False)

    def _expert_mlp(self, e: int, x: torch.Tensor) -> torch.Tensor:
        if self.alpha13 is not None:
            h = ops.fp4_linear(x, self.qw13[e], self.sz13[e], self.alpha13[e])
            return ops.fp4_linear(ops.silu_and_mul(h), self.qw2[e], self.sz2[e], self.alpha2[e])
        if self.qw13 is not None:
            h = ops.w4a16_linear(x, self.qw13[e], self.sz13[e])
            return ops.w4a16_linear(ops.silu_and_mul(h), self.qw2[e], self.sz2[e])
        return F.linear(ops.silu_and_mul(F.linear(x, self.w13[e])), self.w2[e])

    def _grouped_tile_rows(self, x: torch.Tensor) -> int:
        """Row-tile size for the grouped HIP GEMMs, or 0 to use the per-expert
        library GEMMs (large prefill batches, where each expert has hundreds of
        rows and hipBLASLt is compute-efficient; that path syncs with the host,
        so it is never taken while a hipGraph is being captured)."""
        if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and ops.hip_available()):
            return 0
        if self.quant is None and (self.w13.shape[2] % 128 or self.w2.shape[2] % 128 or self.w13.shape[1] % 16):
            return 0
        cap = 64 if self.qw13 is not None else 128
        avg = x.shape[0] * self.top_k / max(1, self.num_experts)
        if avg > envs.VDAMD_MOE_GROUPED_MAX_ROWS * cap / 64 and not torch.cuda.is_current_stream_capturing():
            return 0
        for rows in (16, 32, 64, 128):
            if avg <= rows or rows == cap:
                return rows
        return cap

    def _forward_grouped(self, x: torch.Tensor, topw: torch.Tensor, topi: torch.Tensor, tile_rows: int,
                         ) -> torch.Tensor:
        """Experts as two grouped GEMMs over expert-sorted rows: no host sync,
        every expert's weights stream from HBM once per step."""
        t, k = topi.shape
        order, tiles = ops.moe_align(topi, self.num_experts, tile_rows, self.expert_start,
                                     self.local_experts if self.expert_parallel else -1)
        xs = x.index_select(0, order // k)
        if self.qw13 is not None:
            h = ops.grouped_linear(xs, tiles, tile_rows, qw=self.qw13, sz=self.sz13, alpha=self.alpha13)
            a = ops.silu_and_mul(h)
            y = ops.grouped_linear(a, tiles, tile_rows, qw=self.qw2, sz=self.sz2, alpha=self.alpha2)
        else:
            h = ops.grouped_linear(xs, tiles, tile_rows, weight=self.w13)
            a = ops.silu_and_mul(h)
            y = ops.grouped_linear(a, tiles, tile_rows, weight=self.w2)
        w_sorted = topw.reshape(-1)[order]
        if self.expert_parallel:        # rows of experts held elsewhere were never written
            local = (topi.reshape(-1)[order] >= self.expert_start) & \
                    (topi.reshape(-1)[order] < self.expert_start + self.local_experts)
            y = torch.where(local.unsqueeze(1), y, torch.zeros_like(y))
        y = y * w_sorted.unsqueeze(1)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=order.device)
        return y.index_select(0, inv).view(t, k, -1).sum(dim=1)      # fixed order: deterministic

    def _experts_grouped(self, xs: torch.Tensor, tiles: torch.Tensor, tile_rows: int) -> torch.Tensor:
        if self.qw13 is not None:
            h = ops.grouped_linear(xs, tiles, tile_rows, qw=self.qw13, sz=self.sz13, alpha=self.alpha13)
            return ops.grouped_linear(h, tiles, tile_rows, qw=self.qw2, sz=self.sz2, alpha=self.alpha2,
                                      silu_mul=True)
        h = ops.grouped_linear(xs, tiles, tile_rows, weight=self.w13)
        return ops.grouped_linear(ops.silu_and_mul(h), tiles, tile_rows, weight=self.w2)

    def _forward_fused_routing(self, x: torch.Tensor, logits: torch.Tensor, tile_rows: int, deferred: bool = False):
        """Routing, expert GEMMs and the weighted sum in eight launches (router
        GEMM included): top-k, slot sort + tile table and the combine are one HIP
        kernel each.  ``deferred`` (decode-sized batches): the weighted sum is left to the add + RMSNorm
        that follows (``ExpertRows``)."""
        local = self.local_experts if self.expert_parallel else -1
        if ops.moe_route_small_ok(logits, x):
            # decode-sized batch: top-k, slot sort, tile table and the row gather are one launch
            topw, topi, inv, tiles, xs = ops.moe_route_small(logits, x, self.top_k, self.renormalize, tile_rows,
                                                             self.expert_start, local)
            y = self._experts_grouped(xs, tiles, tile_rows)
            if deferred:
                return ExpertRows(y, topw, topi, inv, self.expert_start, local)
            return ops.moe_combine(y, topw, topi, inv, self.expert_start, local)
        topw, topi = ops.moe_topk(logits, self.top_k, self.renormalize)
        src_row, inv, tiles = ops.moe_align_fused(topi, self.num_experts, tile_rows, self.expert_start, local)
        y = self._experts_grouped(x.index_select(0, src_row), tiles, tile_rows)
        return ops.moe_combine(y, topw, topi, inv, self.expert_start, local)

    # True in subclasses that override ``route``: the fused top-k kernel (softmax routing) is skipped
    custom_routing = False

    def router_logits(self, x: torch.Tensor) -> torch.Tensor:
        """``[T, E]`` scores of the router (families that keep the router in fp32 override this)."""
        return self.gate(x)

    def route(self, logits: torch.Tensor):
        """Router logits ``[T, E]`` (fp32) -> (weights ``[T, k]`` fp32, expert ids ``[T, k]``)."""
        probs = torch.softmax(logits, dim=-1)
        topw, topi = probs.topk(self.top_k, dim=-1)
        if self.renormalize:
            topw = topw / topw.sum(-1, keepdim=True)
        return topw, topi

    def forward(self, x: torch.Tensor, reduce: bool = True, staged: bool = False, deferred: bool = False):
        # `deferred`: the caller feeds the result to an add + RMSNorm and accepts ``ExpertRows``
        # `staged` (write the partial sums where the next direct all-reduce reads) is a hint only
        tile_rows = self._grouped_tile_rows(x)
        logits = None
        if (tile_rows and not self.custom_routing and self.gate.bias is None
                and getattr(self.gate, "lora", None) is None and self.num_experts <= 512 and self.top_k <= 16):
            # decode: the routing kernel adds up the router GEMM's K-split partial sums itself
            logits = ops.router_logit_partials(x, self.gate.weight)
            if logits is not None:
                # unsharded experts only: a sharded block's sum still has to cross the TP group
                defer = deferred and self.tp == 1 and envs.VDAMD_FUSED_REDUCE_NORM
                out = self._forward_fused_routing(x, logits, tile_rows, defer)
                if reduce and self.tp > 1:
                    out = tensor_model_parallel_all_reduce(out)
                return out
        logits = self.router_logits(x)
        if (tile_rows and x.dim() == 2 and not self.custom_routing
                and ops.moe_route_fused_ok(logits, self.num_experts, self.top_k)):
            out = self._forward_fused_routing(x, logits, tile_rows)
            if reduce and self.tp > 1:
                out = tensor_model_parallel_all_reduce(out)
            return out
        topw, topi = self.route(logits.float())
        topw = topw.to(x.dtype)
        if tile_rows:
            out = self._forward_grouped(x, topw, topi, tile_rows)
            if reduce and self.tp > 1:
                out = tensor_model_parallel_all_reduce(out)
            return out
        out = torch.zeros_like(x)
        # group token-slots by expert: one sort, then contiguous segments
        flat_e = topi.reshape(-1)
        order = flat_e.argsort(stable=True)
        counts = torch.bincount(flat_e, minlength=self.num_experts).tolist()
        tok = (order // self.top_k)
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
            h = self._expert_mlp(e, x.index_select(0, rows))
            out.index_add_(0, rows, h * wts[seg].unsqueeze(1))
        if reduce and self.tp > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out
