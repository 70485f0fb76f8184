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


def kv_prefix_key(token_ids: Sequence[int], num_tokens: int, salt: str = "") -> str:
    """Name of the stored KV of the first ``num_tokens`` tokens of a prompt."""
    import hashlib
    h = hashlib.sha256(salt.encode())
    h.update(b"|" + str(num_tokens).encode() + b"|")
    h.update(",".join(str(int(t)) for t in token_ids[:num_tokens]).encode())
    return h.hexdigest()[:40]


def transferable_tokens(num_prompt_tokens: int, block_size: int) -> int:
    """Whole blocks of a prompt that can be handed over: at least one prompt token stays
    behind for the receiver to run, because a forward pass is what produces its first logits."""
    return max(0, (num_prompt_tokens - 1) // block_size) * block_size


class SchedulerKVConnector:
    """Scheduler side of the shared-storage KV connector (disaggregated prefill / decode).

    A *producer* engine writes the KV blocks of every prompt it has prefilled under
    ``kv_connector_extra_config["shared_storage_path"]``; a *consumer* engine that meets the
    same prompt finds them there, has its workers copy them into freshly allocated blocks and
    schedules only the rest of the prompt.  The scheduler decides (storage lookups happen here,
    once, not on every rank), the workers move bytes and report ``finished_sending`` /
    ``finished_recving``, and the executor counts a transfer as done when every rank has
    reported it (``KVOutputAggregator``).  A producer keeps a finished request's blocks until
    its save has been confirmed."""

    def __init__(self, vllm_config: VllmConfig):
        import os
        self.config = vllm_config.kv_transfer_config
        extra = self.config.kv_connector_extra_config or {}
        self.path = extra.get("shared_storage_path")
        self.block_size = vllm_config.cache_config.block_size
        self.produce = self.path is not None and self.config.kv_role in ("kv_producer", "kv_both")
        self.consume = self.path is not None and self.config.kv_role in ("kv_consumer", "kv_both")
        mc, pc = vllm_config.model_config, vllm_config.parallel_config
        # KV of another model, dtype or sharding must never be mistaken for this engine's
        self.salt = "|".join(str(x) for x in (mc.model, mc.dtype, vllm_config.cache_config.block_size,
                                              pc.tensor_parallel_size, pc.pipeline_parallel_size,
                                              getattr(vllm_config.cache_config, "cache_dtype", "auto")))
        if self.path:
            os.makedirs(self.path, exist_ok=True)
        self._loads: List[dict] = []              # decided while scheduling, shipped with the step
        self._saves: List[dict] = []
        self.sending: Set[str] = set()            # saves asked for and not yet confirmed by every rank
        self.held_until_sent: Dict[str, object] = {}
        self.num_loaded_tokens = 0
        self.num_saved_requests = 0

    # -- consumer -----------------------------------------------------------------------
    def get_num_new_matched_tokens(self, request, num_local_tokens: int) -> int:
        """Prompt tokens beyond the ``num_local_tokens`` already in this engine's cache whose
        KV can be loaded from storage."""
        import os
        if not self.consume:
            return 0
        n = transferable_tokens(request.num_prompt_tokens, self.block_size)
        if n <= num_local_tokens:
            return 0
        key = kv_prefix_key(request.prompt_token_ids, n, self.salt)
        if not os.path.isfile(os.path.join(self.path, key, "meta.json")):      # written last: the commit mark
            return 0
        return n - num_local_tokens

    def note_load(self, request, first_token: int, num_tokens: int, block_ids: List[int]) -> None:
        key = kv_prefix_key(request.prompt_token_ids, first_token + num_tokens, self.salt)
        self._loads.append({"req_id": request.request_id, "key": key, "first_block": first_token // self.block_size,
                            "num_blocks": num_tokens // self.block_size, "block_ids": list(block_ids)})
        self.num_loaded_tokens += num_tokens

    # -- producer -----------------------------------------------------------------------
    def note_prefill_scheduled(self, request, block_ids: List[int]) -> None:
        """Called when the step that computes the last prompt token has been scheduled."""
        import os
        if not self.produce or request.request_id in self.sending:
            return
        n = transferable_tokens(request.num_prompt_tokens, self.block_size)
        if n == 0:
            return
        key = kv_prefix_key(request.prompt_token_ids, n, self.salt)
        if os.path.isfile(os.path.join(self.path, key, "meta.json")):
            return
        self._saves.append({"req_id": request.request_id, "key": key, "num_blocks": n // self.block_size,
                            "block_ids": list(block_ids[:n // self.block_size])})
        self.sending.add(request.request_id)
        self.num_saved_requests += 1

    def build_connector_meta(self, scheduler_output, pending: bool = False) -> dict:
        """``pending``: transfers started by earlier steps are still in flight (an otherwise empty step is
        then worth running: its output carries the workers' completion reports)."""
        loads, self._loads = self._loads, []
        saves, self._saves = self._saves, []
        return {"role": self.config.kv_role, "step": scheduler_output.step_id, "path": self.path,
                "load": loads, "save": saves, "pending": bool(pending or loads or saves)}

    def update_from_output(self, runner_output) -> List[object]:
        """Fold the aggregated worker reports in; returns the requests whose blocks were being
        held for a save and may be freed now."""
        done = []
        for rid in (getattr(runner_output, "finished_sending", None) or ()):
            self.sending.discard(rid)
            req = self.held_until_sent.pop(rid, None)
            if req is not None:
                done.append(req)
        return done

    def request_finished(self, request) -> bool:
        """True: keep the request's blocks, a save still reads them."""
        if request.request_id in self.sending:
            self.held_until_sent[request.request_id] = request
            return True
        return False


class WorkerKVConnector:
    """Worker side: copies KV blocks between this rank's cache and the shared storage, off the step.

    One file per (prompt prefix, rank): ``<path>/<key>/rank<r>.safetensors`` with ``k.<i>`` /
    ``v.<i>`` = the blocks of the rank's i-th attention layer in cache layout.  ``meta.json``
    appears once every rank's file is in place (the last writer renames it in).

    ``execute_model`` only queues the transfers a step asks for (``start``) and collects the names of
    those that have finished since it last asked (``poll``); file I/O and the copies between host and
    device run on one side thread, on a stream of its own.  That is safe without further ordering
    because the scheduler keeps everybody else off the blocks involved: blocks being loaded belong to a
    parked request that is scheduled only after every rank has reported the load, and blocks being
    saved are whole prompt blocks, never rewritten, held until every rank has reported the save (the
    side thread waits for the event that marks the end of the step that wrote them)."""

    def __init__(self, vllm_config: VllmConfig, rank: int, world_size: int):
        import queue
        import threading
        self.rank, self.world_size = rank, world_size
        self._jobs: "queue.Queue" = queue.Queue()
        self._lock = threading.Lock()
        self._sent: Set[str] = set()
        self._recvd: Set[str] = set()
        self._error: Optional[BaseException] = None
        self._inflight = 0
        self._thread: Optional[threading.Thread] = None
        self.transfers_started = 0
        self.max_step_block_s = 0.0          # longest time one start() / poll() call held the step thread

    # -- step thread ---------------------------------------------------------------------
    def start(self, meta: dict, kv_caches, kind: str, after=None) -> None:
        """Queue the ``kind`` (``"load"`` / ``"save"``) operations of a step's connector metadata.
        ``after``: a device event the side thread waits for first (the step that fills the blocks)."""
        import time
        ops = meta.get(kind) or ()
        if not ops:
            return
        t0 = time.perf_counter()
        self._ensure_thread(kv_caches[0][0].device)
        with self._lock:
            self._inflight += len(ops)
        self.transfers_started += len(ops)
        for op in ops:
            self._jobs.put((kind, op, meta["path"], kv_caches, after))
        self.max_step_block_s = max(self.max_step_block_s, time.perf_counter() - t0)

    def poll(self):
        """``(finished_sending, finished_recving)`` since the last call (None when empty).  A transfer that
        failed fails the step: the engine must not go on believing in blocks that never arrived."""
        with self._lock:
            if self._error is not None:
                err, self._error = self._error, None
                raise RuntimeError(f"KV transfer failed on rank {self.rank}: {err!r}") from err
            sent, self._sent = self._sent, set()
            recvd, self._recvd = self._recvd, set()
        return sent or None, recvd or None

    def idle(self) -> bool:
        with self._lock:
            return self._inflight == 0

    def shutdown(self) -> None:
        if self._thread is not None:
            self._jobs.put(None)
            self._thread.join(timeout=30)
            self._thread = None

    # -- side thread ---------------------------------------------------------------------
    def _ensure_thread(self, device) -> None:
        import threading
        if self._thread is None or not self._thread.is_alive():
            self._thread = threading.Thread(target=self._run, args=(device,), name=f"kv-transfer-{self.rank}",
                                            daemon=True)
            self._thread.start()

    def _run(self, device) -> None:
        import contextlib

        import torch
        stream = None
        if device.type == "cuda":
            torch.cuda.set_device(device)
            stream = torch.cuda.Stream(device)
        with torch.inference_mode():
            while True:
                job = self._jobs.get()
                if job is None:
                    return
                kind, op, path, kv_caches, after = job
                try:
                    if after is not None:
                        after.synchronize()
                    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                        (self._load if kind == "load" else self._save)(op, path, kv_caches)
                        if stream is not None:
                            stream.synchronize()
                    with self._lock:
                        (self._recvd if kind == "load" else self._sent).add(op["req_id"])
                        self._inflight -= 1
                except BaseException as e:  # noqa: BLE001 - reported by the next poll()
                    with self._lock:
                        self._error = e
                        self._inflight -= 1

    def _load(self, op: dict, path: str, kv_caches) -> None:
        import os

        import torch
        from safetensors import safe_open
        fn = os.path.join(path, op["key"], f"rank{self.rank}.safetensors")
        fb, nb = op["first_block"], op["num_blocks"]
        dev = kv_caches[0][0].device
        dst = torch.as_tensor(op["block_ids"][fb:fb + nb], dtype=torch.int64, device=dev)
        with safe_open(fn, framework="pt", device="cpu") as f:
            for i, (k, v) in enumerate(kv_caches):
                k.index_copy_(0, dst, f.get_slice(f"k.{i}")[fb:fb + nb].to(dev))
                v.index_copy_(0, dst, f.get_slice(f"v.{i}")[fb:fb + nb].to(dev))

    def _save(self, op: dict, path: str, kv_caches) -> None:
        import json
        import os

        import torch
        from safetensors.torch import save_file
        d = os.path.join(path, op["key"])
        os.makedirs(d, exist_ok=True)
        src = torch.as_tensor(op["block_ids"], dtype=torch.int64, device=kv_caches[0][0].device)
        tensors = {}
        for i, (k, v) in enumerate(kv_caches):
            tensors[f"k.{i}"] = k.index_select(0, src).cpu().contiguous()
            tensors[f"v.{i}"] = v.index_select(0, src).cpu().contiguous()
        tmp = os.path.join(d, f".rank{self.rank}.tmp")
        save_file(tensors, tmp)
        os.replace(tmp, os.path.join(d, f"rank{self.rank}.safetensors"))
        have = [f for f in os.listdir(d) if f.startswith("rank") and f.endswith(".safetensors")]
        if len(have) == self.world_size:           # every rank's part is there: commit
            tmpm = os.path.join(d, f".meta.{self.rank}.tmp")
            with open(tmpm, "w") as f:
                json.dump({"num_blocks": op["num_blocks"], "world_size": self.world_size}, f)
            os.replace(tmpm, os.path.join(d, "meta.json"))


def make_scheduler_connector(vllm_config: VllmConfig) -> SchedulerKVConnector:
    return SchedulerKVConnector(vllm_config)
