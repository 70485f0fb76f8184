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
import torch.nn.functional as F

from ... import ops
from ...parallel.state import (get_tensor_model_parallel_rank,
                               get_tensor_model_parallel_world_size,
                               tensor_model_parallel_all_reduce)
from .linear import ReplicatedLinear, divide


class FusedMoE(nn.Module):
    def __init__(self, hidden: int, inter: int, num_experts: int, top_k: int,
                 renormalize: bool, dtype: torch.dtype, expert_parallel: bool = False):
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
        self.gate = ReplicatedLinear(hidden, num_experts, False, dtype)
        self.w13 = nn.Parameter(torch.empty(self.local_experts, 2 * self.inter_local, hidden,
                                            dtype=dtype), requires_grad=False)
        self.w2 = nn.Parameter(torch.empty(self.local_experts, hidden, self.inter_local,
                                           dtype=dtype), requires_grad=False)
        self.w13.weight_loader = self._load_w13     # type: ignore[attr-defined]
        self.w2.weight_loader = self._load_w2       # type: ignore[attr-defined]

    def _local_index(self, expert_id: int) -> Optional[int]:
        e = expert_id - self.expert_start
        return e if 0 <= e < self.local_experts else None

    def _load_w13(self, param: nn.Parameter, w: torch.Tensor, shard_id) -> None:
        expert_id, which = shard_id          # which: 0 = gate, 1 = up
        e = self._local_index(expert_id)
        if e is None:
            return
        if not self.expert_parallel:
            w = w.narrow(0, self.tp_rank * self.inter_local, self.inter_local)
        param.data[e].narrow(0, which * self.inter_local, self.inter_local).copy_(w)

    def _load_w2(self, param: nn.Parameter, w: torch.Tensor, shard_id) -> None:
        expert_id, _ = shard_id
        e = self._local_index(expert_id)
        if e is None:
            return
        if not self.expert_parallel:
            w = w.narrow(1, self.tp_rank * self.inter_local, self.inter_local)
        param.data[e].copy_(w)

    def forward(self, x: torch.Tensor, reduce: bool = True) -> torch.Tensor:
        logits = self.gate(x).float()
        probs = torch.softmax(logits, dim=-1)
        topw, topi = probs.topk(self.top_k, dim=-1)
        if self.renormalize:
            topw = topw / topw.sum(-1, keepdim=True)
        topw = topw.to(x.dtype)
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
            h = F.linear(x.index_select(0, rows), self.w13[e])
            h = F.linear(ops.silu_and_mul(h), self.w2[e])
            out.index_add_(0, rows, h * wts[seg].unsqueeze(1))
        if reduce and self.tp > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out
