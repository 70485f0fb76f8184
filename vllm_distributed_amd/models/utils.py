"""Pipeline-parallel helpers shared by the model definitions."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass
class IntermediateTensors:
    """What crosses a pipeline-stage boundary."""
    hidden: torch.Tensor
    residual: Optional[torch.Tensor]


def get_pp_layer_range(num_layers: int, pp_rank: int, pp_size: int,
                       partition: Optional[str] = None) -> Tuple[int, int]:
    """Layers ``[start, end)`` owned by stage ``pp_rank``.

    ``partition`` is the ``VLLM_PP_LAYER_PARTITION`` string, e.g. ``"30,50"``
    (reference ``docker-compose.yml:38``); default is an even split with the
    remainder given to the *middle* stages first, so the first stage
    (embedding) and the last (LM head + sampling) stay lightest.
    """
    if partition:
        parts = [int(x) for x in str(partition).split(",")]
        if len(parts) != pp_size or sum(parts) != num_layers or any(p <= 0 for p in parts):
            raise ValueError(f"invalid PP layer partition {partition!r} for "
                             f"{num_layers} layers / {pp_size} stages")
    else:
        base, rem = divmod(num_layers, pp_size)
        parts = [base] * pp_size
        order = sorted(range(pp_size), key=lambda i: (i in (0, pp_size - 1), abs(i - (pp_size - 1) / 2)))
        for i in order[:rem]:
            parts[i] += 1
    start = sum(parts[:pp_rank])
    return start, start + parts[pp_rank]


_SAFETENSORS_DTYPES = {"F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16,
                       "I64": torch.int64, "I32": torch.int32, "I16": torch.int16, "I8": torch.int8,
                       "U8": torch.uint8, "BOOL": torch.bool, "F8_E4M3": torch.float8_e4m3fn,
                       "F8_E5M2": torch.float8_e5m2}


class LazyTensor:
    """A checkpoint tensor that has not been read yet.

    ``narrow`` and leading-dimension slicing compose index ranges without touching the
    file; ``materialize`` reads exactly the remaining hyper-rectangle through the
    safetensors slice API.  A tensor-parallel rank that cuts its shard out of a
    ``[out, in]`` matrix therefore reads ``1/TP`` of the bytes instead of the whole
    tensor (with 8 ranks on a node: the checkpoint crosses the page cache once, not
    eight times).  Any other tensor method materialises first and is forwarded."""

    def __init__(self, handle, key: str, ranges=None, shape=None, dtype=None):
        self._handle, self._key = handle, key
        if ranges is None:
            sl = handle.get_slice(key)
            shape = tuple(sl.get_shape())
            dtype = _SAFETENSORS_DTYPES.get(str(sl.get_dtype()), None)
            ranges = tuple((0, n) for n in shape)
        self._ranges = tuple(ranges)
        self._full_shape = tuple(shape)
        self.dtype = dtype

    @property
    def shape(self) -> torch.Size:
        return torch.Size(b - a for a, b in self._ranges)

    def dim(self) -> int:
        return len(self._ranges)

    def numel(self) -> int:
        n = 1
        for a, b in self._ranges:
            n *= b - a
        return n

    def narrow(self, dim: int, start: int, length: int) -> "LazyTensor":
        if dim < 0:
            dim += len(self._ranges)
        a, b = self._ranges[dim]
        if not (0 <= start and start + length <= b - a):
            raise IndexError(f"narrow({dim}, {start}, {length}) outside a dimension of {b - a}")
        r = list(self._ranges)
        r[dim] = (a + start, a + start + length)
        return LazyTensor(self._handle, self._key, r, self._full_shape, self.dtype)

    def __getitem__(self, idx):
        if isinstance(idx, slice) and idx.step in (None, 1) and self._ranges:
            n = self._ranges[0][1] - self._ranges[0][0]
            start, stop, _ = idx.indices(n)
            return self.narrow(0, start, max(0, stop - start))
        return self.materialize()[idx]

    def materialize(self) -> torch.Tensor:
        sl = self._handle.get_slice(self._key)
        if not self._ranges:
            return self._handle.get_tensor(self._key)
        return sl[tuple(slice(a, b) for a, b in self._ranges)]

    def __getattr__(self, name: str):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)


def materialize(x):
    """The tensor behind a ``LazyTensor`` (read now), or ``x`` itself."""
    return x.materialize() if isinstance(x, LazyTensor) else x
