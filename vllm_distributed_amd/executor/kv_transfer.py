"""KV-transfer (disaggregated prefill/decode) plumbing on the executor side.

With a KV connector configured every worker reports which requests finished
sending / receiving KV blocks; a request only counts as finished once *all*
workers have reported it.  ``KVOutputAggregator`` folds the per-worker
``ModelRunnerOutput``s into the output rank's (reference ``src/launch.py:28,
295-296,338-349``).
"""
from __future__ import annotations

import concurrent.futures
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Set

from ..config import VllmConfig
from ..engine.types import ModelRunnerOutput


class KVOutputAggregator:
    def __init__(self, world_size: int):
        self.world_size = world_size
        self._send_remaining: Dict[str, int] = defaultdict(lambda: world_size)
        self._recv_remaining: Dict[str, int] = defaultdict(lambda: world_size)

    def _update(self, remaining: Dict[str, int], ids: Optional[Set[str]], done: Set[str]) -> None:
        for rid in ids or ():
            remaining[rid] -= 1
            if remaining[rid] <= 0:
                del remaining[rid]
                done.add(rid)

    def aggregate(self, outputs: Sequence[Optional[ModelRunnerOutput]], output_rank: int = 0,
                  ) -> Optional[ModelRunnerOutput]:
        sent: Set[str] = set()
        recvd: Set[str] = set()
        for o in outputs:
            if o is None:
                continue
            self._update(self._send_remaining, o.finished_sending, sent)
            self._update(self._recv_remaining, o.finished_recving, recvd)
        out = outputs[output_rank]
        if out is None:
            out = ModelRunnerOutput()
        out.finished_sending = sent or None
        out.finished_recving = recvd or None
        return out

    def async_aggregate(self, futures: Sequence["concurrent.futures.Future"], output_rank: int = 0,
                        ) -> "concurrent.futures.Future":
        result: concurrent.futures.Future = concurrent.futures.Future()
        outputs: List[Optional[ModelRunnerOutput]] = [None] * len(futures)
        pending = [len(futures)]

        def make_cb(i: int):
            def cb(f: "concurrent.futures.Future") -> None:
                if result.done():
                    return
                try:
                    outputs[i] = f.result()
                except BaseException as e:  # noqa: BLE001
                    result.set_exception(e)
                    return
                pending[0] -= 1
                if pending[0] == 0:
                    result.set_result(self.aggregate(outputs, output_rank))
            return cb
        for i, f in enumerate(futures):
            f.add_done_callback(make_cb(i))
        return result


class SchedulerKVConnector:
    """Scheduler-side hook: attaches connector metadata to each step.  The
    bundled connector is a no-op placeholder that keeps the aggregation path
    exercised; real transports plug in by name."""

    def __init__(self, vllm_config: VllmConfig):
        self.config = vllm_config.kv_transfer_config

    def build_connector_meta(self, scheduler_output) -> dict:
        return {"role": self.config.kv_role, "step": scheduler_output.step_id}


def make_scheduler_connector(vllm_config: VllmConfig) -> SchedulerKVConnector:
    return SchedulerKVConnector(vllm_config)
