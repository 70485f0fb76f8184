"""LoRA adapters served next to the base model.

The reference's executor forwards ``add_lora`` and friends to every worker
through ``collective_rpc`` (reference ``src/launch.py:351-384``; they belong to
the ``Executor`` base class it subclasses at ``:60``).  Here an adapter is a
PEFT directory (``adapter_config.json`` + ``adapter_model.safetensors``); each
worker keeps up to ``max_loras`` adapters resident in per-layer slot tensors,
already cut to its tensor-parallel shard:

* column-parallel layers (qkv, gate_up): ``A`` whole, ``B`` rows of the rank;
* row-parallel layers (o_proj, down_proj): ``A`` columns of the rank, ``B``
  whole - the low-rank product is a partial sum like the base product and rides
  the same all-reduce.

A step names, per token, which adapter applies; the base GEMM runs once for the
whole batch and each active adapter adds ``(x_sel A^T) B^T`` on its own rows.
Steps with any adapter active run eagerly (captured graphs hold the base model
only).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ..config import LoRAConfig
from ..logger import init_logger

logger = init_logger(__name__)

TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
# checkpoint projection -> (fused layer attribute, sub-matrix index)
_FUSED = {"q_proj": ("qkv_proj", 0), "k_proj": ("qkv_proj", 1), "v_proj": ("qkv_proj", 2),
          "o_proj": ("o_proj", 0), "gate_proj": ("gate_up_proj", 0), "up_proj": ("gate_up_proj", 1),
          "down_proj": ("down_proj", 0)}


@dataclass(frozen=True)
class LoRARequest:
    lora_name: str
    lora_int_id: int
    lora_path: str

    def __post_init__(self) -> None:
        if self.lora_int_id < 1:
            raise ValueError("lora_int_id must be >= 1 (0 means the base model)")


@dataclass
class LoRAAdapter:
    rank: int
    scaling: float
    # "layers.3.self_attn.q_proj" -> (A [r, in], B [out, r])
    weights: Dict[str, Tuple[torch.Tensor, torch.Tensor]]


def load_peft_adapter(path: str) -> LoRAAdapter:
    with open(os.path.join(path, "adapter_config.json")) as f:
        cfg = json.load(f)
    rank = int(cfg["r"])
    scaling = float(cfg.get("lora_alpha", rank)) / rank
    st = os.path.join(path, "adapter_model.safetensors")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        tensors = load_file(st)
    else:
        tensors = torch.load(os.path.join(path, "adapter_model.bin"), map_location="cpu", weights_only=True)
    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for name, w in tensors.items():
        for tag in ("lora_A", "lora_B"):
            marker = f".{tag}."
            if marker in name:
                mod = name.split(marker)[0]
                for prefix in ("base_model.model.", "model."):
                    if mod.startswith(prefix):
                        mod = mod[len(prefix):]
                pairs.setdefault(mod, {})[tag] = w
    weights = {}
    for mod, ab in pairs.items():
        if mod.rsplit(".", 1)[-1] not in TARGETS:
            raise ValueError(f"LoRA on {mod!r} is not supported (targets: {', '.join(TARGETS)})")
        if "lora_A" not in ab or "lora_B" not in ab:
            raise ValueError(f"adapter {path}: {mod} lacks lora_A or lora_B")
        weights[mod] = (ab["lora_A"], ab["lora_B"])
    return LoRAAdapter(rank, scaling, weights)


# ------------------------------------------------------------ per-step context
class LoRAStep:
    """Which adapter slot applies to which rows of this step's token batch.
    ``groups``: (slot, row indices on the device, or None when the slot covers
    every row)."""

    def __init__(self, groups: List[Tuple[int, Optional[torch.Tensor]]]):
        self.groups = groups


_STEP: Optional[LoRAStep] = None


def set_lora_step(step: Optional[LoRAStep]) -> None:
    global _STEP
    _STEP = step


def lora_active() -> bool:
    return _STEP is not None


class LoRALayerSlots(nn.Module):
    """Adapter slots of one (possibly fused) linear layer."""

    def __init__(self, in_local: int, sub_sizes: Sequence[int], slots: int, max_rank: int,
                 dtype: torch.dtype, device):
        super().__init__()
        self.sub_sizes = list(sub_sizes)
        self.offsets = [sum(self.sub_sizes[:i]) for i in range(len(self.sub_sizes))]
        self.max_rank = max_rank
        n = len(self.sub_sizes)
        self.register_buffer("a", torch.zeros(slots, n * max_rank, in_local, dtype=dtype, device=device),
                             persistent=False)
        self.register_buffer("b", torch.zeros(slots, sum(self.sub_sizes), max_rank, dtype=dtype, device=device),
                             persistent=False)
        # which sub-matrices of a slot hold an adapter (host-side, decides the work done)
        self.used = [[False] * n for _ in range(slots)]

    def clear(self, slot: int) -> None:
        self.a[slot].zero_()
        self.b[slot].zero_()
        self.used[slot] = [False] * len(self.sub_sizes)

    def write(self, slot: int, sub: int, a: torch.Tensor, b: torch.Tensor) -> None:
        r = a.shape[0]
        assert r <= self.max_rank and b.shape == (self.sub_sizes[sub], r), (a.shape, b.shape, self.sub_sizes)
        self.a[slot, sub * self.max_rank: sub * self.max_rank + r].copy_(a)
        self.b[slot, self.offsets[sub]: self.offsets[sub] + self.sub_sizes[sub], :r].copy_(b)
        self.used[slot][sub] = True

    def apply(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        step = _STEP
        if step is None:
            return y
        x2, y2 = x.reshape(-1, x.shape[-1]), y.view(-1, y.shape[-1])
        r = self.max_rank
        for slot, rows in step.groups:
            used = self.used[slot]
            if not any(used):
                continue
            xs = x2 if rows is None else x2.index_select(0, rows)
            h = xs @ self.a[slot].t()                               # [t, n_sub * r]
            if all(used):
                d = torch.cat([h[:, i * r:(i + 1) * r] @ self.b[slot, o:o + s].t()
                               for i, (o, s) in enumerate(zip(self.offsets, self.sub_sizes))], dim=1)
                if rows is None:
                    y2 += d
                else:
                    y2.index_add_(0, rows, d)
                continue
            for i, (o, s) in enumerate(zip(self.offsets, self.sub_sizes)):
                if not used[i]:
                    continue
                d = h[:, i * r:(i + 1) * r] @ self.b[slot, o:o + s].t()
                if rows is None:
                    y2[:, o:o + s] += d
                else:
                    y2[:, o:o + s].index_add_(0, rows, d)
        return y


# ------------------------------------------------------------- worker manager
class LoRAManager:
    """Adapter slots of one worker's model shard."""

    def __init__(self, model: nn.Module, config: LoRAConfig, dtype: torch.dtype, device):
        from ..models.layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear
        self.config = config
        self.dtype, self.device = dtype, device
        self.layers: Dict[str, nn.Module] = {}
        for name, m in model.named_modules():
            leaf = name.rsplit(".", 1)[-1]
            if leaf not in ("qkv_proj", "o_proj", "gate_up_proj", "down_proj"):
                continue
            if isinstance(m, QKVParallelLinear):
                subs = [m.q_size, m.kv_size, m.kv_size]
            elif isinstance(m, ColumnParallelLinear):
                subs = list(m.shard_sizes)
            elif isinstance(m, RowParallelLinear):
                subs = [m.out_local]
            else:
                continue
            m.lora = LoRALayerSlots(m.in_local, subs, config.max_loras, config.max_lora_rank, dtype, device)
            self.layers[name] = m
        if not self.layers:
            raise ValueError("the model has no layers LoRA adapters can attach to")
        self.slot_of: Dict[int, int] = {}          # lora_int_id -> slot
        self.names: Dict[int, str] = {}
        self.pinned: set = set()
        self._clock = 0
        self._last_used: Dict[int, int] = {}

    # -- sharding ---------------------------------------------------------
    def _shard(self, layer: nn.Module, sub: int, a: torch.Tensor, b: torch.Tensor):
        from ..models.layers.linear import QKVParallelLinear, RowParallelLinear
        if isinstance(layer, RowParallelLinear):
            n = layer.in_local
            return a[:, layer.tp_rank * n:(layer.tp_rank + 1) * n], b
        size = layer.lora.sub_sizes[sub]
        idx = layer.tp_rank
        if isinstance(layer, QKVParallelLinear) and sub > 0:
            idx = layer.tp_rank // layer.kv_replicas
        return a, b[idx * size:(idx + 1) * size]

    # -- adapter lifecycle -------------------------------------------------
    def list_loras(self) -> List[int]:
        return sorted(self.slot_of)

    def _free_slot(self, protect: Sequence[int]) -> int:
        used = set(self.slot_of.values())
        for s in range(self.config.max_loras):
            if s not in used:
                return s
        victims = [i for i in self.slot_of if i not in self.pinned and i not in protect]
        if not victims:
            raise RuntimeError(f"all {self.config.max_loras} LoRA slots are pinned or in use")
        victim = min(victims, key=lambda i: self._last_used.get(i, 0))
        slot = self.slot_of[victim]
        self.remove_lora(victim)
        return slot

    def add_lora(self, req: LoRARequest, protect: Sequence[int] = ()) -> bool:
        self._clock += 1
        self._last_used[req.lora_int_id] = self._clock
        if req.lora_int_id in self.slot_of:
            return False
        adapter = load_peft_adapter(req.lora_path)
        if adapter.rank > self.config.max_lora_rank:
            raise ValueError(f"adapter {req.lora_name!r} has rank {adapter.rank} > max_lora_rank "
                             f"{self.config.max_lora_rank}")
        slot = self._free_slot(protect)
        for m in self.layers.values():
            m.lora.clear(slot)
        loaded = 0
        for mod, (a, b) in adapter.weights.items():
            parent, proj = mod.rsplit(".", 1)
            fused, sub = _FUSED[proj]
            layer = self.layers.get(f"{parent}.{fused}")
            if layer is None:
                continue                    # a layer of another pipeline stage
            a_s, b_s = self._shard(layer, sub, a, b)
            layer.lora.write(slot, sub, a_s.to(self.device, self.dtype),
                             (b_s.to(torch.float32) * adapter.scaling).to(self.device, self.dtype))
            loaded += 1
        self.slot_of[req.lora_int_id] = slot
        self.names[req.lora_int_id] = req.lora_name
        logger.info("LoRA adapter %r (id %d, rank %d) -> slot %d, %d matrices", req.lora_name, req.lora_int_id,
                    adapter.rank, slot, loaded)
        return True

    def remove_lora(self, lora_id: int) -> bool:
        slot = self.slot_of.pop(lora_id, None)
        if slot is None:
            return False
        self.names.pop(lora_id, None)
        self.pinned.discard(lora_id)
        self._last_used.pop(lora_id, None)
        for m in self.layers.values():
            m.lora.clear(slot)
        return True

    def pin_lora(self, lora_id: int) -> bool:
        if lora_id not in self.slot_of:
            return False
        self.pinned.add(lora_id)
        return True

    # -- per step -----------------------------------------------------------
    def make_step(self, token_lora_ids, to_dev) -> Optional[LoRAStep]:
        """``token_lora_ids``: numpy int array, one adapter id (0 = none) per token row."""
        import numpy as np
        ids = np.unique(token_lora_ids)
        ids = ids[ids != 0]
        if ids.size == 0:
            return None
        groups = []
        for i in ids.tolist():
            slot = self.slot_of[i]
            self._clock += 1
            self._last_used[i] = self._clock
            where = np.nonzero(token_lora_ids == i)[0]
            groups.append((slot, None if where.size == token_lora_ids.size else to_dev(where, torch.int64)))
        return LoRAStep(groups)
