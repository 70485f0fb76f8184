"""KV-transfer (disaggregated prefill/decode) plumbing on the executor side.

With a KV connector configured every worker reports which requests finished
sending / receiving KV blocks; a request only counts as finished once *all*
workers have reported it.  ``KVOutputAggregator`` folds the per-worker
``ModelRunnerOutput``s into the output rank's (reference ``src/launch.py:28,
295-296,338-349``).

Two connectors share the scheduler / worker halves below:

* ``SharedStorageConnector`` - KV blocks travel as safetensors files under ``shared_storage_path``.
* ``DeviceIpcConnector`` - producer and consumer on one machine: the producer gathers a prompt's blocks
  into a device buffer and publishes its hipIpc handle (a small JSON file under ``shared_storage_path``,
  which only serves as the mailbox); the consumer maps the buffer and copies device to device (over
  xGMI when the engines sit on different GPUs), then acknowledges, and the producer frees the buffer.
  The bytes never touch the host.  Buffers nobody claimed are dropped after ``device_buffer_ttl_s``; a
  hand-over is single-use.  Without a GPU (or beyond ``device_buffer_bytes`` of unclaimed buffers) a
  save falls back to the file form, which every consumer also reads.
"""
from __future__ import annotations

import concurrent.futures
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Set

from ..config import VllmConfig
from ..engine.types import ModelRunnerOutput

DEVICE_CONNECTORS = ("DeviceIpcConnector",)


def _pid_alive(pid: int) -> bool:
    import os
    try:
        os.kill(int(pid), 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    return True


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
            os.makedirs(self.path, mode=0o700, exist_ok=True)      # hand-over descriptors name mappable device memory
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
        if not self._committed(os.path.join(self.path, key)):
            return 0
        return n - num_local_tokens

    @staticmethod
    def _committed(d: str) -> bool:
        """``meta.json`` is written last: the commit mark.  A device-buffer hand-over is only usable while
        the process that owns the buffer lives, and only on its machine."""
        import json
        import os
        import socket
        try:
            with open(os.path.join(d, "meta.json")) as f:
                meta = json.load(f)
        except (OSError, ValueError):
            return False
        if meta.get("mode") != "ipc":
            return True
        return meta.get("host") == socket.gethostname() and _pid_alive(meta.get("pid", -1))

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
        if self._committed(os.path.join(self.path, key)):
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
        cfg = vllm_config.kv_transfer_config
        extra = (cfg.kv_connector_extra_config or {}) if cfg is not None else {}
        self.device_mode = cfg is not None and cfg.kv_connector in DEVICE_CONNECTORS
        self.device_buffer_bytes = int(extra.get("device_buffer_bytes", 8 << 30))
        self.device_buffer_ttl_s = float(extra.get("device_buffer_ttl_s", 600.0))
        self._exports: Dict[str, dict] = {}          # side thread only: key -> device buffer handed out
        self._exported_bytes = 0
        self._last_reap = 0.0
        self._ipc = None
        self.device_bytes_sent = 0
        self.device_bytes_received = 0
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
        if self._exports and self._thread is not None:
            import time
            now = time.monotonic()
            if now - self._last_reap > 0.5:          # claimed or expired device buffers go back to the pool
                self._last_reap = now
                self._jobs.put(("reap", None, None, None, None))
        return sent or None, recvd or None

    def idle(self) -> bool:
        with self._lock:
            return self._inflight == 0

    def shutdown(self) -> None:
        if self._thread is not None:
            self._jobs.put(None)
            self._thread.join(timeout=30)
            self._thread = None
        for key in list(self._exports):                  # nobody can map these once this process is gone
            self._drop_export(key)

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
                if kind == "reap":
                    try:
                        self._reap()
                    except Exception:  # noqa: BLE001 - housekeeping only
                        pass
                    continue
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
        desc = os.path.join(path, op["key"], f"rank{self.rank}.ipc.json")
        if os.path.isfile(desc):
            return self._load_device(desc, op["key"], fb, nb, dst, kv_caches)
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
        if self.device_mode and src.is_cuda and self._save_device(op, d, src, kv_caches):
            return
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
                json.dump({"num_blocks": op["num_blocks"], "world_size": self.world_size, "mode": "file"}, f)
            os.replace(tmpm, os.path.join(d, "meta.json"))

    # -- device-to-device hand-over -----------------------------------------------------
    def _ipc_lib(self):
        if self._ipc is None:
            import ctypes

            from ..ops import hip
            lib = hip.load()
            lib.vdamd_ipc_handle_bytes.restype = ctypes.c_int
            lib.vdamd_ipc_alloc.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
            lib.vdamd_ipc_alloc.restype = ctypes.c_void_p
            lib.vdamd_ipc_free.argtypes = [ctypes.c_void_p]
            lib.vdamd_ipc_free.restype = ctypes.c_int
            lib.vdamd_ipc_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
            lib.vdamd_ipc_open.restype = ctypes.c_void_p
            lib.vdamd_ipc_close.argtypes = [ctypes.c_void_p]
            lib.vdamd_ipc_close.restype = ctypes.c_int
            self._ipc = lib
        return self._ipc

    @staticmethod
    def _view(buf, entry: dict):
        """The tensor an entry of a hand-over descriptor names inside the mapped byte buffer."""
        import torch
        dtype = getattr(torch, entry["dtype"])
        n = 1
        for x in entry["shape"]:
            n *= x
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return buf[entry["offset"]:entry["offset"] + nbytes].view(dtype).view(entry["shape"])

    @staticmethod
    def _raw(t):
        """One-byte element types as plain bytes (gather / scatter kernels exist for those everywhere)."""
        import torch
        return t.view(torch.uint8) if t.element_size() == 1 else t

    def _save_device(self, op: dict, d: str, src, kv_caches) -> bool:
        """Gather the blocks into a fresh exportable device buffer and publish its handle.  False: no room
        (or no buffer to be had): the caller writes files instead."""
        import ctypes
        import json
        import os
        import socket
        import time

        import torch

        from ..parallel.custom_all_reduce import _DevicePointer
        self._reap()
        entries, off = [], 0
        nb = src.numel()
        for i, (k, v) in enumerate(kv_caches):
            for name, t in (("k", k), ("v", v)):
                shape = [nb] + list(t.shape[1:])
                nbytes = nb * t[0].numel() * t.element_size()
                entries.append({"name": f"{name}.{i}", "offset": off, "shape": shape,
                                "dtype": str(t.dtype).split(".")[1]})
                off += (nbytes + 255) // 256 * 256
        total = off
        if total == 0 or self._exported_bytes + total > self.device_buffer_bytes:
            return False
        lib = self._ipc_lib()
        handle = ctypes.create_string_buffer(int(lib.vdamd_ipc_handle_bytes()))
        ptr = lib.vdamd_ipc_alloc(total, handle)
        if not ptr:
            return False
        buf = torch.as_tensor(_DevicePointer(ptr, total), device=src.device)
        for e in entries:
            kind, layer = e["name"].split(".")
            t = kv_caches[int(layer)][0 if kind == "k" else 1]
            torch.index_select(self._raw(t), 0, src, out=self._raw(self._view(buf, e)))
        torch.cuda.current_stream().synchronize()        # the bytes are in place before anybody can find them
        del buf
        key = os.path.basename(d)
        self._exports[key] = {"ptr": ptr, "nbytes": total, "dir": d, "born": time.monotonic()}
        self._exported_bytes += total
        self.device_bytes_sent += total
        tmp = os.path.join(d, f".rank{self.rank}.ipc.tmp")
        with open(tmp, "w") as f:
            json.dump({"handle": handle.raw.hex(), "nbytes": total, "entries": entries, "pid": os.getpid(),
                       "host": socket.gethostname(), "num_blocks": nb}, f)
        os.replace(tmp, os.path.join(d, f"rank{self.rank}.ipc.json"))
        have = [f for f in os.listdir(d) if f.startswith("rank") and f.endswith((".ipc.json", ".safetensors"))]
        if len(have) == self.world_size:
            tmpm = os.path.join(d, f".meta.{self.rank}.tmp")
            with open(tmpm, "w") as f:
                json.dump({"num_blocks": op["num_blocks"], "world_size": self.world_size, "mode": "ipc",
                           "pid": os.getpid(), "host": socket.gethostname()}, f)
            os.replace(tmpm, os.path.join(d, "meta.json"))
        return True

    def _load_device(self, desc_path: str, key: str, fb: int, nb: int, dst, kv_caches) -> None:
        import ctypes
        import json
        import os

        import torch

        from ..parallel.custom_all_reduce import _DevicePointer
        with open(desc_path) as f:
            desc = json.load(f)
        lib = self._ipc_lib()
        own = self._exports.get(key) if desc.get("pid") == os.getpid() else None
        mapped = None
        if own is not None:                  # this very process handed it out (kv_both): no mapping needed
            ptr = own["ptr"]
        else:
            err = ctypes.c_int(0)
            mapped = ptr = lib.vdamd_ipc_open(bytes.fromhex(desc["handle"]), ctypes.byref(err))
            if not ptr:
                raise RuntimeError(f"hipIpcOpenMemHandle failed with {err.value} for {desc_path}")
        try:
            buf = torch.as_tensor(_DevicePointer(ptr, desc["nbytes"]), device=dst.device)
            for e in desc["entries"]:
                kind, layer = e["name"].split(".")
                t = kv_caches[int(layer)][0 if kind == "k" else 1]
                self._raw(t).index_copy_(0, dst, self._raw(self._view(buf, e)[fb:fb + nb]))
            torch.cuda.current_stream().synchronize()
            del buf
        finally:
            if mapped is not None:
                lib.vdamd_ipc_close(mapped)
        self.device_bytes_received += desc["nbytes"]
        ack = os.path.join(os.path.dirname(desc_path), f"rank{self.rank}.ack")
        with open(ack + ".tmp", "w") as f:
            f.write("ok")
        os.replace(ack + ".tmp", ack)

    def _reap(self) -> None:
        """Free the device buffers whose hand-over has been acknowledged, or that nobody came for."""
        import os
        import time
        now = time.monotonic()
        for key, ex in list(self._exports.items()):
            if os.path.isfile(os.path.join(ex["dir"], f"rank{self.rank}.ack")) \
                    or now - ex["born"] > self.device_buffer_ttl_s:
                self._drop_export(key)

    def _drop_export(self, key: str) -> None:
        import os
        ex = self._exports.pop(key, None)
        if ex is None:
            return
        d = ex["dir"]
        # the commit mark goes first: nobody new may pick this hand-over up
        for name in ("meta.json", f"rank{self.rank}.ipc.json", f"rank{self.rank}.ack"):
            try:
                os.remove(os.path.join(d, name))
            except OSError:
                pass
        try:
            os.rmdir(d)
        except OSError:
            pass
        self._ipc_lib().vdamd_ipc_free(ex["ptr"])
        self._exported_bytes -= ex["nbytes"]


def make_scheduler_connector(vllm_config: VllmConfig) -> SchedulerKVConnector:
    return SchedulerKVConnector(vllm_config)
