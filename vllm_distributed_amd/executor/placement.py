"""Rank placement as a pure function.

Ranks are laid out ``rank = pp_rank * TP + tp_rank``.  Stages are filled in
order: head-local GPUs first, then registered remote nodes
(reference ``src/launch.py:211-247``).  Differences from the reference, all
deliberate:

* a node that cannot host a stage is skipped - the *stage* is never skipped
  (the reference ``continue``s past the stage, ``src/launch.py:227-231``);
* devices on a node are handed out in ascending ``local_rank`` order, so the
  rank -> device map does not depend on TCP connection arrival order
  (the reference pops from the tail of an arrival-ordered list, ``:235``);
* a TP group may be assembled from the head's local devices plus remote
  worker processes running on the *same host* (``same_host=True``), which is
  how "head + 7 workers, TP=8" on one 8-GPU node is expressed.  TP groups
  never span hosts: their all-reduces must stay on xGMI.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple


class PlacementError(RuntimeError):
    """Not enough devices registered (yet) to place every rank."""


@dataclass(frozen=True)
class NodeInfo:
    node_id: str
    local_ranks: Tuple[int, ...]        # free device indices on that node, any order
    same_host: bool = False             # runs on the head's physical host


@dataclass(frozen=True)
class Assignment:
    rank: int
    pp_rank: int
    tp_rank: int
    node_id: Optional[str]              # None = spawned by the head itself
    local_rank: int


def plan_placement(tp: int, pp: int, head_local_devices: int,
                   nodes: Sequence[NodeInfo]) -> List[Assignment]:
    """Returns one assignment per rank, in rank order, or raises ``PlacementError``."""
    if tp < 1 or pp < 1:
        raise ValueError("tp and pp must be >= 1")
    local_free = list(range(max(0, head_local_devices)))
    free: Dict[str, List[int]] = {n.node_id: sorted(n.local_ranks) for n in nodes}
    order = [n.node_id for n in nodes]
    same_host = {n.node_id for n in nodes if n.same_host}
    out: List[Assignment] = []
    for stage in range(pp):
        base = stage * tp
        picked: Optional[List[Tuple[Optional[str], int]]] = None
        if len(local_free) >= tp:
            picked = [(None, local_free.pop(0)) for _ in range(tp)]
        else:
            for nid in order:
                if len(free[nid]) >= tp:
                    picked = [(nid, free[nid].pop(0)) for _ in range(tp)]
                    break
            if picked is None:
                # one host, several worker groups: head-local devices + same-host remotes
                pool: List[Tuple[Optional[str], int]] = [(None, d) for d in local_free]
                for nid in order:
                    if nid in same_host:
                        pool += [(nid, d) for d in free[nid]]
                if len(pool) >= tp:
                    picked = pool[:tp]
                    for nid, d in picked:
                        (local_free if nid is None else free[nid]).remove(d)
        if picked is None:
            have = len(local_free) + sum(len(v) for v in free.values())
            raise PlacementError(
                f"cannot place pipeline stage {stage}: need {tp} devices on one host, "
                f"{have} free device(s) registered across {len(order)} remote node(s)")
        for tp_rank, (nid, dev) in enumerate(picked):
            out.append(Assignment(rank=base + tp_rank, pp_rank=stage, tp_rank=tp_rank,
                                  node_id=nid, local_rank=dev))
    return out


def output_rank(world_size: int, tp: int) -> int:
    """First TP rank of the last pipeline stage (reference ``src/launch.py:304-314``)."""
    return world_size - tp


def max_concurrent_batches(pp: int, async_scheduling: bool) -> int:
    """Reference ``src/launch.py:298-302``."""
    return 2 if async_scheduling else pp
