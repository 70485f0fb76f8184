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
