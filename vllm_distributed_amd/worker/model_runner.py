"""Model runner: turns a ``SchedulerOutput`` into device inputs, runs the model
(eagerly or by replaying a captured hipGraph), samples, and reports tokens.

Batch order is decode-first (all ``q_len == 1`` sequences, then prompt chunks)
so the two attention kernels each see a contiguous slice.  Pure-decode steps
whose padded batch size has a captured graph are replayed from static buffers:
one graph launch instead of ~10 kernel launches per layer.
"""
from __future__ import annotations

import os
import threading
import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import envs, ops
from ..config import VllmConfig
from ..engine.types import KVCacheSpec, ModelRunnerOutput, SchedulerOutput
from ..logger import init_logger
from ..lora import lora_active, set_lora_step
from ..utils.tracing import trace_range
from ..models.layers.attention import AttentionMetadata, set_forward_context
from ..models.loader import load_model
from ..models.utils import IntermediateTensors
from ..parallel.state import CommTimer, get_pp_group, get_tp_group
from ..utils import cdiv
from .input_batch import InputBatch

logger = init_logger(__name__)


class StepMarks:
    """Device-side stamps of one step: start of its device work, end of the forward,
    end of sampling (= the event the token readback waits for), and the end stamp of
    the previous step on the same stream.  GPU: timing events, resolved once the
    last one has completed; CPU: ``perf_counter`` values (the forward is synchronous)."""

    __slots__ = ("start", "fwd", "end", "prev_end", "gpu")

    def __init__(self, gpu: bool):
        self.gpu = gpu
        self.start = self.fwd = self.end = self.prev_end = None

    def stamp(self, which: str) -> None:
        if self.gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            setattr(self, which, ev)
        else:
            setattr(self, which, time.perf_counter())

    def resolve(self) -> Dict[str, float]:
        """Seconds per segment; the end stamp must have completed (GPU)."""
        if self.start is None or self.end is None:
            return {}
        fwd = self.fwd if self.fwd is not None else self.end
        if self.gpu:
            try:
                d = {"forward": self.start.elapsed_time(fwd) * 1e-3, "sample": fwd.elapsed_time(self.end) * 1e-3}
                if self.prev_end is not None:
                    d["device_idle"] = max(0.0, self.prev_end.elapsed_time(self.start) * 1e-3)
                return d
            except RuntimeError:          # a measurement must never take a step down
                return {}
        d = {"forward": fwd - self.start, "sample": self.end - fwd}
        if self.prev_end is not None:
            d["device_idle"] = max(0.0, self.start - self.prev_end)
        return d


class PendingSample:
    """Tokens sampled on the device whose values have not reached the host yet.

    ``wait()`` blocks until they have and builds the step's output; it touches no
    runner state, so a reply thread may call it while the worker thread is already
    preparing the next step.  ``publish()`` folds the token values into the
    persistent batch and belongs to the thread that runs ``execute_model``."""

    lazy_step_result = True       # executors call .get() / .wait() on results carrying this mark
    __slots__ = ("runner", "step_id", "req_ids", "nreq", "sel", "rows", "gens", "pos", "tokens",
                 "event", "logprobs", "timing", "marks", "_result", "prompt_logprobs", "_lock", "_toks",
                 "_published", "kv_sent", "kv_recv", "pooled", "counts", "drafts", "_accepted", "proposals")

    def __init__(self, runner: "ModelRunner", step_id: int, req_ids: List[str], sel: Optional[np.ndarray],
                 rows: Optional[np.ndarray], pos: Optional[np.ndarray], tokens: Optional[torch.Tensor],
                 event: Optional["torch.cuda.Event"], logprobs, timing: Optional[Dict[str, float]],
                 marks: Optional[StepMarks] = None):
        self.runner, self.step_id, self.req_ids, self.nreq = runner, step_id, req_ids, len(req_ids)
        self.sel, self.rows, self.pos, self.tokens, self.event = sel, rows, pos, tokens, event
        self.gens = runner.batch.row_gen[rows].copy() if rows is not None else None
        self.logprobs, self.timing, self.marks = logprobs, timing, marks
        self.prompt_logprobs = None
        self.pooled = None                   # (req ids, [k, H] float32 device tensor)
        # speculative decoding: sampled positions per selected request, and each one's draft tokens (or None)
        self.counts: Optional[np.ndarray] = None
        self.drafts: Optional[List[Optional[List[int]]]] = None
        self._accepted: Optional[List[List[int]]] = None
        self.proposals: Optional[Dict[str, List[int]]] = None      # the draft model's tokens for the next step
        self.kv_sent = self.kv_recv = None
        self._result: Optional[ModelRunnerOutput] = None
        self._toks: Optional[np.ndarray] = None
        self._lock = threading.Lock()
        self._published = False

    @property
    def resolved(self) -> bool:
        return self._result is not None

    def wait(self) -> ModelRunnerOutput:
        """Wait for the device and build the output.  Thread-safe; no runner state is modified."""
        with self._lock:
            if self._result is not None:
                return self._result
            out_tokens: List[List[int]] = [[] for _ in range(self.nreq)]
            logprobs_out = None
            if self.event is not None:
                self.event.synchronize()
            if self.sel is not None and self.counts is None:
                self._toks = toks = self.tokens.numpy()
                tok_list = toks.tolist()
                for j, i in enumerate(self.sel.tolist()):
                    out_tokens[i] = [tok_list[j]]
                if self.logprobs is not None:
                    logprobs_out = [None] * self.nreq
                    for j, i in enumerate(self.sel.tolist()):
                        logprobs_out[i] = self.logprobs[j]
            elif self.sel is not None:
                # some requests were sampled at several positions (their last token + drafts): keep
                # the drafts the model reproduced, and the token it put after the last of them
                tok_list = self.tokens.numpy().tolist()
                self._accepted = []
                if self.logprobs is not None:
                    logprobs_out = [None] * self.nreq
                off = 0
                for j, i in enumerate(self.sel.tolist()):
                    c = int(self.counts[j])
                    mine, d = tok_list[off:off + c], self.drafts[j]
                    n_acc = 0
                    if d:
                        while n_acc < len(d) and mine[n_acc] == d[n_acc]:
                            n_acc += 1
                    out_tokens[i] = mine[:n_acc + 1]
                    self._accepted.append(out_tokens[i])
                    if self.logprobs is not None:
                        logprobs_out[i] = self.logprobs[off + c - 1]
                    off += c
            pooled = None
            if self.pooled is not None:
                ids, vecs = self.pooled
                pooled = dict(zip(ids, vecs.cpu().tolist()))
            timing = self.timing
            if timing is not None and self.marks is not None:
                timing.update(self.marks.resolve())
            self._result = ModelRunnerOutput(step_id=self.step_id, req_ids=self.req_ids,
                                             sampled_token_ids=out_tokens, logprobs=logprobs_out, timing=timing,
                                             prompt_logprobs=self.prompt_logprobs, pooled=pooled,
                                             finished_sending=self.kv_sent or None,
                                             finished_recving=self.kv_recv or None,
                                             draft_token_ids=self.proposals)
            return self._result

    def publish(self) -> None:
        """Fold the token values into the persistent batch (unless the row changed
        owner meanwhile).  Only the thread that runs ``execute_model`` calls this."""
        if self._published:
            return
        self.wait()
        self._published = True
        if self.sel is not None and self.counts is None:
            b = self.runner.batch
            ok = b.row_gen[self.rows] == self.gens
            rows, pos = self.rows[ok], self.pos[ok]
            b.token_ids[rows, pos] = self._toks[ok]
            b.num_known[rows] = np.maximum(b.num_known[rows], pos + 1)
        elif self.sel is not None:
            b = self.runner.batch
            for j, (r, p0) in enumerate(zip(self.rows.tolist(), self.pos.tolist())):
                if b.row_gen[r] != self.gens[j]:
                    continue
                acc = self._accepted[j]
                m, ran = len(acc), int(self.counts[j])
                b.token_ids[r, p0:p0 + m] = acc
                b.num_tokens[r] = p0 + m                  # execute_model had counted one token
                b.num_computed[r] -= ran - m              # positions of rejected drafts were not computed
                b.num_known[r] = p0 + m
        try:
            self.runner._pending.remove(self)
        except ValueError:
            pass

    def get(self) -> ModelRunnerOutput:
        self.publish()
        return self._result


class _Discard:
    """Resolves a pending sample for its side effects on a non-output rank."""

    lazy_step_result = True

    def __init__(self, pending: PendingSample):
        self.pending = pending

    def get(self) -> None:
        self.pending.get()
        return None

    def wait(self) -> None:
        self.pending.wait()
        return None


class DecodeGraph:
    """A captured decode step for one padded batch size."""

    def __init__(self, batch_size: int):
        self.batch_size = batch_size
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.output: Optional[Any] = None
        self.aux: Optional[torch.Tensor] = None      # the model's ``last_aux`` of the captured step (EAGLE-3)


class ModelRunner:
    def __init__(self, vllm_config: VllmConfig, device: torch.device):
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.cache_config = vllm_config.cache_config
        self.scheduler_config = vllm_config.scheduler_config
        self.parallel_config = vllm_config.parallel_config
        self.device = device
        self.is_gpu = device.type == "cuda"
        self.block_size = self.cache_config.block_size
        self.max_model_len = self.model_config.max_model_len
        self.max_num_reqs = self.scheduler_config.max_num_seqs
        self.max_num_tokens = self.scheduler_config.max_num_batched_tokens
        self.model: Optional[torch.nn.Module] = None
        self.batch = InputBatch(self.max_num_reqs, self.max_model_len, self.block_size,
                                self.model_config.vocab_size, self.model_config.seed)
        self.kv_caches: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self.graphs: Dict[int, DecodeGraph] = {}
        # batch size -> [(context limit, graph)] ascending; the last entry is also graphs[batch size]
        self.graph_ctx: Dict[int, List[Tuple[int, DecodeGraph]]] = {}
        self.graph_sizes: List[int] = []
        self.graph_pool = None
        self.graph_fallback_reason: Optional[str] = "not captured yet"
        self._static: Dict[str, torch.Tensor] = {}
        self.decode_workspace: Optional[torch.Tensor] = None
        self.draft = None                 # worker.draft_model.DraftProposer (draft-model speculation)
        # vision-language requests: their images (multimodal.MultiModalInput) and the projected features of
        # each, computed when the first prompt chunk over the image runs and kept until the request leaves
        self._mm_inputs: Dict[str, list] = {}
        self._mm_embeds: Dict[Tuple[str, int], torch.Tensor] = {}
        self.step_count = 0
        self.report_on_all_ranks = False
        self.comm_stream = None
        self.lora_manager = None
        self.sampled_dev: Optional[torch.Tensor] = None      # last sampled token per batch row (device)
        self._pending: List[PendingSample] = []
        self._last_end_mark = None        # end stamp of the previous step (device idle time between steps)
        self._token_table = None          # vocabulary bytes for constrained decoding, built on first use
        self.defer_output = False        # set by in-process executors under async scheduling
        self.timing_enabled = vllm_config.observability_config.collect_step_timing
        self.kv_connector = None
        if vllm_config.kv_transfer_config is not None and \
                (vllm_config.kv_transfer_config.kv_connector_extra_config or {}).get("shared_storage_path"):
            from ..executor.kv_transfer import WorkerKVConnector
            from ..parallel.state import get_world_group
            w = get_world_group()
            self.kv_connector = WorkerKVConnector(vllm_config, w.rank, w.world_size)
        CommTimer.configure()
        self._pool_acc: Dict[str, torch.Tensor] = {}        # running sums of "mean" pooling requests

    # ------------------------------------------------------------------ setup
    def load_model(self) -> None:
        t0 = time.time()
        self.model = load_model(self.vllm_config, self.device)
        if self.is_gpu and envs.VDAMD_WEIGHT_ROW_PAD > 0:
            from ..models.layers.linear import pad_dense_weight_rows
            n = pad_dense_weight_rows(self.model, int(envs.VDAMD_WEIGHT_ROW_PAD))
            logger.info("padded the rows of %d dense weights by %d elements", n, envs.VDAMD_WEIGHT_ROW_PAD)
        self.pp = get_pp_group()
        self.tp = get_tp_group()
        mode = envs.VDAMD_TP_OVERLAP
        if (self.is_gpu and self.pp.world_size == 1 and hasattr(self.model, "forward_overlapped")
                and getattr(self.model, "tp_fusions", True) and (mode == "force" or (mode == "auto" and self.tp.world_size > 1))):
            self.comm_stream = torch.cuda.Stream(priority=-1)
        self.pp_link = None
        if self.pp.world_size > 1:
            from ..parallel.pipeline import PipelineLink
            self.pp_link = PipelineLink(self.pp, self.device, use_rccl=envs.VDAMD_COMM_BACKEND in ("auto", "rccl"))
        sc = self.vllm_config.speculative_config
        if sc is not None and sc.worker_side:
            from .draft_model import DraftProposer, HiddenStateProposer, MedusaProposer
            if sc.reads_target_hidden:
                cls = MedusaProposer if sc.method == "medusa" else HiddenStateProposer
                self.draft = cls(self.vllm_config, self.device)
                self.draft.bind_target(self.model)
                if getattr(self.draft, "wants_aux", False):
                    self.comm_stream = None       # (the tapped layer loop is the plain one: no overlapped form)
            else:
                self.draft = DraftProposer(self.vllm_config, self.device)
        if self.vllm_config.lora_config is not None:
            from ..lora import LoRAManager
            self.lora_manager = LoRAManager(self.model, self.vllm_config.lora_config,
                                            self.model_config.torch_dtype, self.device)
        logger.info("model loaded in %.1fs", time.time() - t0)

    def get_kv_cache_spec(self) -> KVCacheSpec:
        layers = self.model.attention_layers()
        a = layers[0]
        fp8 = self.vllm_config.cache_config.kv_fp8
        windows = [int(getattr(l, "window", 0) or 0) for l in layers]
        dtype_bytes = 1 if fp8 else self.model_config.dtype_bytes
        if self.draft is not None:        # the draft reads every key it ever wrote unless it is windowed too
            windows += [int(getattr(l, "window", 0) or 0) for l in self.draft.model.attention_layers()]
        heads, width = a.num_kv_heads, a.head_dim
        if getattr(a, "latent", False):
            # one shared latent row per token, held in two layouts: (rank + rope) + rank values, which the
            # spec's "2 x heads x head_dim" states as one head of half that sum
            heads, width = 1, a.kv_values_per_token() // 2
        return KVCacheSpec(num_layers=len(layers), num_kv_heads=heads, head_dim=width,
                           block_size=self.block_size, dtype_bytes=dtype_bytes,
                           sliding_window=max(windows) if windows and min(windows) > 0 else None,
                           extra_bytes_per_block=(self.draft.kv_bytes_per_block(self.block_size, dtype_bytes)
                                                  if self.draft is not None else 0))

    @torch.inference_mode()
    def profile_run(self) -> None:
        """Forward of the largest token batch with no KV cache bound, to expose
        peak activation memory to the caller's measurement."""
        t = self.max_num_tokens         # a step packs many sequences: not bounded by max_model_len
        ids = torch.zeros(t, dtype=torch.int32, device=self.device)
        pos = torch.arange(t, dtype=torch.int32, device=self.device) % self.max_model_len
        set_forward_context(None, pos)
        inter = None
        if not self.pp.is_first_rank:
            h = self.model_config.hidden_size
            z = torch.zeros(t, h, dtype=self.model_config.torch_dtype, device=self.device)
            inter = IntermediateTensors(z, z.clone())
        out = self.model(ids, pos, inter)
        if self.pp.is_last_rank:
            n = min(self.max_num_reqs, t)
            self.model.compute_logits(out[:n])
        self.tp.end_step()
        if self.is_gpu:
            torch.cuda.synchronize()

    def initialize_kv_cache(self, num_blocks: int) -> None:
        spec = self.get_kv_cache_spec()
        # an e4m3 cache halves the bytes decode attention streams; K/V are widened to the model
        # dtype in registers before the MFMA, with one fp32 scale per KV head (1.0 unless a
        # checkpoint supplies k_scale / v_scale)
        fp8 = self.vllm_config.cache_config.kv_fp8
        dtype = torch.float8_e4m3fn if fp8 else self.model_config.torch_dtype
        self.num_blocks = num_blocks
        self.kv_caches = []
        for attn in self.model.attention_layers():
            k_shape, v_shape = attn.kv_cache_shapes(num_blocks, self.block_size)
            k = torch.zeros(k_shape, dtype=dtype, device=self.device)
            v = torch.zeros(v_shape, dtype=dtype, device=self.device)
            scales = attn.kv_scales if fp8 else None
            if scales is not None:
                scales = tuple(t.to(self.device) for t in scales)
            attn.bind_kv_cache(k, v, scales)
            self.kv_caches.append((k, v))
        if self.draft is not None:
            self.draft.allocate_kv(num_blocks, self.block_size, dtype, fp8)
        logger.info("KV cache: %d blocks x %d tokens (%.2f GiB)", num_blocks, self.block_size,
                    num_blocks * spec.bytes_per_block / 2**30)

    # ------------------------------------------------------------ state sync
    def _update_states(self, so: SchedulerOutput) -> None:
        b = self.batch
        for rid in so.finished_req_ids:
            b.remove_request(rid)
            self._drop_mm(rid)
        for rid in so.preempted_req_ids:
            b.remove_request(rid)
            self._drop_mm(rid)
        for nr in so.scheduled_new_reqs:
            if getattr(nr, "mm_inputs", None):
                self._mm_inputs[nr.req_id] = nr.mm_inputs
            lora_id = 0
            if nr.lora_request is not None:
                if self.lora_manager is None:
                    raise RuntimeError("request names a LoRA adapter but the engine runs without --enable-lora")
                self.lora_manager.add_lora(nr.lora_request, protect=self.loras_in_batch())
                lora_id = nr.lora_request.lora_int_id
            b.add_request(nr.req_id, nr.prompt_token_ids, nr.output_token_ids, nr.sampling_params,
                          nr.block_ids, nr.num_computed_tokens, lora_id)
        c = so.scheduled_cached_reqs
        samples_here = self.pp.is_last_rank
        for i, rid in enumerate(c.req_ids):
            row = b.row_of.get(rid)
            if row is None:
                continue
            b.append_blocks(row, c.new_block_ids[i])
            toks = c.new_token_ids[i] if c.new_token_ids else None
            if toks and not samples_here:
                # stages that do not sample learn new tokens from the scheduler
                b.append_tokens(row, toks)
        if so.scheduled_spec_decode_tokens:
            # draft tokens sit behind the last real token as this step's extra inputs; they are not
            # part of the request (num_tokens stays) until the step has verified them
            for rid, drafts in so.scheduled_spec_decode_tokens.items():
                row = b.row_of.get(rid)
                if row is None:
                    continue
                nt = int(b.num_tokens[row])
                b.token_ids[row, nt:nt + len(drafts)] = drafts
                b.num_known[row] = nt + len(drafts)

    def _drop_mm(self, rid: str) -> None:
        if self._mm_inputs.pop(rid, None) is not None:
            for key in [k for k in self._mm_embeds if k[0] == rid]:
                del self._mm_embeds[key]

    def _image_rows(self, p: dict) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        """Rows of this step's flat token list that sit on image placeholders, and the image features that
        belong there.  A prompt may be prefilled in several chunks, and a chunk may start or end inside an
        image: each chunk takes the slice of the image's features its positions cover."""
        if not self._mm_inputs or not hasattr(self.model, "encode_images") or not getattr(self.model, "is_first", True):
            return None
        rows, feats = [], []
        starts = p["seq_lens"] - p["n_sched"]
        for i, rid in enumerate(p["req_ids"]):
            items = self._mm_inputs.get(rid)
            if not items:
                continue
            lo, hi = int(starts[i]), int(p["seq_lens"][i])
            for k, m in enumerate(items):
                a, z = max(lo, m.offset), min(hi, m.offset + m.length)
                if a >= z:
                    continue
                emb = self._mm_embeds.get((rid, k))
                if emb is None:
                    px = torch.from_numpy(np.ascontiguousarray(m.pixel_values)).to(self.device)
                    if px.dim() == 4:       # an any-resolution image: base view + tiles, packed by the model
                        emb = self.model.encode_anyres_image(px, m.image_size)
                    else:
                        emb = self.model.encode_images(px.unsqueeze(0))[0]
                    self._mm_embeds[(rid, k)] = emb
                    if emb.shape[0] != m.length:
                        raise RuntimeError(f"image of {rid}: {emb.shape[0]} features for {m.length} placeholder tokens")
                rows.append(np.arange(a - lo, z - lo, dtype=np.int64) + int(p["qsl"][i]))
                feats.append(emb[a - m.offset:z - m.offset])
        if not rows:
            return None
        return self._to_dev(np.concatenate(rows), torch.int64), torch.cat(feats)

    def loras_in_batch(self):
        b = self.batch
        rows = list(b.row_of.values())
        return set(int(i) for i in np.unique(b.lora_id[rows]) if i) if rows else set()

    # --------------------------------------------------------- input building
    def _prepare_inputs(self, so: SchedulerOutput):
        b = self.batch
        req_ids = list(so.num_scheduled_tokens.keys())
        n_sched = np.fromiter(so.num_scheduled_tokens.values(), dtype=np.int32, count=len(req_ids))
        rows = np.fromiter((b.row_of[r] for r in req_ids), dtype=np.int64, count=len(req_ids))
        # decode-first, stable
        is_prefill = n_sched != 1
        if is_prefill.any():
            order = np.argsort(is_prefill, kind="stable")
            rows, n_sched = rows[order], n_sched[order]
            req_ids = [req_ids[i] for i in order.tolist()]
            num_decodes = int((~is_prefill).sum())
        else:
            num_decodes = len(req_ids)
        nreq = len(req_ids)
        starts = b.num_computed[rows]
        seq_lens = (starts + n_sched).astype(np.int32)
        qsl = np.zeros(nreq + 1, dtype=np.int32)
        np.cumsum(n_sched, out=qsl[1:])
        total = int(qsl[-1])
        if num_decodes == nreq:
            positions = starts.astype(np.int32)
            tok_rows = rows
        else:
            req_idx = np.repeat(np.arange(nreq), n_sched)
            positions = (starts[req_idx] + (np.arange(total, dtype=np.int32) - qsl[req_idx])).astype(np.int32)
            tok_rows = rows[req_idx]
        input_ids = b.token_ids[tok_rows, positions]
        unknown = positions >= b.num_known[tok_rows]      # values still on the device only
        blk = b.block_table[tok_rows, positions // self.block_size]
        slot_mapping = (blk * self.block_size + positions % self.block_size).astype(np.int32)
        max_blocks = int(cdiv(int(seq_lens.max()), self.block_size))
        # which requests reach the end of their known tokens -> sample
        sample_mask = seq_lens >= b.num_tokens[rows]
        spec = so.scheduled_spec_decode_tokens
        return dict(spec=spec if spec else None,
                    req_ids=req_ids, rows=rows, n_sched=n_sched, num_decodes=num_decodes,
                    seq_lens=seq_lens, qsl=qsl, total=total, positions=positions,
                    input_ids=input_ids, slot_mapping=slot_mapping, max_blocks=max_blocks,
                    sample_mask=sample_mask, unknown=unknown if unknown.any() else None,
                    tok_rows=tok_rows)

    def _to_dev(self, arr: np.ndarray, dtype=torch.int32) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if t.dtype != dtype:
            t = t.to(dtype)
        if self.is_gpu:
            return t.pin_memory().to(self.device, non_blocking=True)
        return t

    def _build_metadata(self, p: dict) -> Tuple[AttentionMetadata, torch.Tensor, torch.Tensor]:
        b = self.batch
        nd, nreq = p["num_decodes"], len(p["req_ids"])
        block_table = self._to_dev(b.block_table[p["rows"], :p["max_blocks"]])
        seq_lens = self._to_dev(p["seq_lens"])
        qsl = self._to_dev(p["qsl"])
        slot = self._to_dev(p["slot_mapping"])
        n_sched = p["n_sched"]
        md = AttentionMetadata(
            num_decodes=nd, num_decode_tokens=nd, num_prefills=nreq - nd,
            num_prefill_tokens=p["total"] - nd, slot_mapping=slot, block_table=block_table,
            seq_lens=seq_lens, query_start_loc=qsl,
            max_query_len=int(n_sched.max()), max_seq_len=int(p["seq_lens"].max()),
            max_decode_seq_len=int(p["seq_lens"][:nd].max()) if nd else 0,
            decode_workspace=self.decode_workspace)
        if nreq > nd:
            pq = p["qsl"][nd:] - p["qsl"][nd]
            md.prefill_query_start_loc = self._to_dev(pq)
            md.max_prefill_query_len = int(n_sched[nd:].max())
            md.prefill_fresh = bool((p["seq_lens"][nd:] == n_sched[nd:]).all())
        input_ids = self._to_dev(p["input_ids"])
        self._patch_unknown_inputs(input_ids, p)
        positions = self._to_dev(p["positions"])
        return md, input_ids, positions

    def _patch_unknown_inputs(self, input_ids: torch.Tensor, p: dict) -> None:
        """Input tokens sampled by a step whose result has not been read back yet
        come straight from the device-side copy of the last sampled tokens."""
        unknown = p["unknown"]
        if unknown is None:
            return
        where = np.nonzero(unknown)[0]
        rows = self._to_dev(p["tok_rows"][where], torch.int64)
        input_ids.index_copy_(0, self._to_dev(where, torch.int64), self.sampled_dev.index_select(0, rows))

    # ------------------------------------------------------------ execution
    @torch.inference_mode()
    def execute_model(self, so: SchedulerOutput) -> Optional[ModelRunnerOutput]:
        t0 = time.perf_counter()
        with trace_range("step.prepare"):
            for pend in list(self._pending):      # values a reply thread has already read back
                if pend.resolved or so.scheduled_spec_decode_tokens or pend.counts is not None:
                    pend.publish()                # (a verified step changes token counts: always fold it in first)
            self._update_states(so)
            self.step_count += 1
            kv_meta = so.kv_connector_metadata if self.kv_connector is not None else None
            if kv_meta:
                self.kv_connector.start(kv_meta, self.kv_caches, "load")     # copied in by a side thread
            if so.total_num_scheduled_tokens == 0:
                if self.kv_connector is not None:         # every rank reports its finished transfers
                    kv_sent, kv_recv = self.kv_connector.poll()
                    return ModelRunnerOutput(step_id=so.step_id, finished_sending=kv_sent, finished_recving=kv_recv)
                return ModelRunnerOutput(step_id=so.step_id) if self._is_output_rank() else None   # nothing to wait for
            p = self._prepare_inputs(so)
        nreq, nd, total = len(p["req_ids"]), p["num_decodes"], p["total"]
        t1 = time.perf_counter()
        marks = None
        if self.timing_enabled:
            marks = StepMarks(self.is_gpu)
            marks.prev_end = self._last_end_mark
            marks.stamp("start")

        lora_step = None
        if self.lora_manager is not None:
            lora_step = self.lora_manager.make_step(self.batch.lora_id[p["tok_rows"]], self._to_dev)
        # (a one-token prompt chunk counts as a decode row and may sit on an image placeholder: look at every step
        # while a request with images is resident)
        image_rows = self._image_rows(p) if self._mm_inputs else None
        if image_rows is not None:
            self.model.set_mm_step(*image_rows)
        graph = None
        if self.graphs and nd == nreq and lora_step is None and image_rows is None:   # graphs hold the text model only
            graph = self._pick_graph(nreq, p["max_blocks"])
        with trace_range("step.forward.graph" if graph is not None else "step.forward.eager"):
            if graph is not None:
                hidden = self._run_graph(graph, p)
            else:
                md, input_ids, positions = self._build_metadata(p)
                set_lora_step(lora_step)
                try:
                    hidden = self._run_eager(md, input_ids, positions, total)
                finally:
                    set_lora_step(None)
                    if image_rows is not None:
                        self.model.set_mm_step(None)
        if self.draft is not None:
            with trace_range("step.draft.catch_up"):
                if graph is not None:                    # the graph ran from its own static inputs
                    md, input_ids, positions = self._build_metadata(p)
                if not self.draft.needs_hidden:          # (a hidden-state head runs once the step has sampled)
                    self.draft.catch_up(md, input_ids, positions)

        # advance computed-token counts (the scheduler did the same when it scheduled)
        self.batch.num_computed[p["rows"]] += p["n_sched"]
        kv_sent = kv_recv = None
        if self.kv_connector is not None:
            if kv_meta and kv_meta.get("save"):
                # this step wrote the last prompt blocks: the side thread copies them out once it has run
                done = None
                if self.is_gpu:
                    done = torch.cuda.Event()
                    done.record()
                self.kv_connector.start(kv_meta, self.kv_caches, "save", after=done)
            kv_sent, kv_recv = self.kv_connector.poll()       # whatever has finished since the last step

        if not self.pp.is_last_rank:
            if marks is not None:
                marks.stamp("end")
                self._last_end_mark = marks.end
            if kv_sent or kv_recv:                 # every rank's report is counted by the executor
                return ModelRunnerOutput(step_id=so.step_id, finished_sending=kv_sent or None,
                                         finished_recving=kv_recv or None)
            return None
        if marks is not None:
            marks.stamp("fwd")
        prompt_lps = None
        if nd < nreq and (self.batch.num_prompt_logprobs[p["rows"][nd:]] >= 0).any():
            prompt_lps = self._prompt_logprobs(hidden, p)
        pooled = None
        if nd < nreq and self.batch.pooling[p["rows"][nd:]].any():
            pooled = self._pool(hidden, p)
        sample_mask = p["sample_mask"]
        sel = rows_sel = pos_sel = host_tokens = event = logprobs_sel = None
        counts = drafts_sel = None
        if sample_mask.any():
            sel = np.nonzero(sample_mask)[0]
            last_idx = p["qsl"][1:][sel] - 1
            rows_sel = p["rows"][sel]
            rows_smp = rows_sel
            if p["spec"]:
                # a request with drafts is sampled at every one of its positions
                drafts_sel = [p["spec"].get(p["req_ids"][i]) for i in sel.tolist()]
                counts = np.array([1 + len(d) if d else 1 for d in drafts_sel], dtype=np.int64)
                ends = np.cumsum(counts)
                within = np.arange(int(ends[-1])) - np.repeat(ends - counts, counts)
                idx = np.repeat(last_idx - (counts - 1), counts) + within
                rows_smp = np.repeat(rows_sel, counts)
                smp_idx = idx
                hs = hidden.index_select(0, self._to_dev(idx, torch.int64))
            elif len(sel) == total:        # every token is a sampling position (pure decode)
                smp_idx = np.arange(total)
                hs = hidden[:total]
            else:
                smp_idx = last_idx
                hs = hidden.index_select(0, self._to_dev(last_idx, torch.int64))
            with trace_range("step.sample"):
                tokens, logprobs_sel = self._sample(hs, rows_smp)
            b = self.batch
            if self.sampled_dev is None:
                self.sampled_dev = torch.zeros(self.max_num_reqs, dtype=torch.int32, device=self.device)
            # every TP rank samples the same token; keep it on the device for the next step
            self.sampled_dev.index_copy_(0, self._to_dev(rows_sel, torch.int64),
                                         tokens if counts is None else
                                         tokens.index_select(0, self._to_dev(np.cumsum(counts) - 1, torch.int64)))
            pos_sel = b.num_tokens[rows_sel].copy()
            b.num_tokens[rows_sel] = pos_sel + 1
            if self.is_gpu:
                host_tokens = torch.empty(tokens.shape[0], dtype=torch.int32, pin_memory=True)
                host_tokens.copy_(tokens, non_blocking=True)
            else:
                host_tokens = tokens
        proposals = None
        if self.draft is not None and self.draft.needs_hidden:
            with trace_range("step.draft.propose"):
                proposals = self._propose_from_hidden(p, md, input_ids, positions, hidden, sel, rows_sel, pos_sel,
                                                      tokens if sel is not None else None,
                                                      smp_idx if sel is not None else None, counts, drafts_sel)
        elif self.draft is not None and sel is not None:
            with trace_range("step.draft.propose"):
                proposals = self._propose_drafts(p, sel, rows_sel, pos_sel, tokens, counts, drafts_sel)
        self.tp.end_step()                # sampling may have gathered over the TP group
        if marks is not None:
            marks.stamp("end")
            self._last_end_mark = marks.end
            if self.is_gpu:
                event = marks.end
        elif self.is_gpu and sel is not None:
            event = torch.cuda.Event()
            event.record()
        # host time of this call: building inputs, and launching (not running) the device work
        timing = {"prepare": t1 - t0, "launch": time.perf_counter() - t1} if self.timing_enabled else None
        if timing is not None and CommTimer.enabled:
            timing["comm"] = CommTimer.drain()
        pending = PendingSample(self, so.step_id, p["req_ids"], sel, rows_sel, pos_sel, host_tokens, event,
                                logprobs_sel, timing, marks)
        pending.prompt_logprobs = prompt_lps
        pending.counts, pending.drafts = counts, drafts_sel
        pending.proposals = proposals
        pending.pooled = pooled
        pending.kv_sent, pending.kv_recv = kv_sent, kv_recv
        self._pending.append(pending)
        report_all = self.kv_connector is not None      # the executor aggregates transfer reports of all ranks
        if self.defer_output:
            return pending if (self._is_output_rank() or report_all) else _Discard(pending)
        out = pending.get()
        return out if (self._is_output_rank() or report_all) else None

    def _propose_drafts(self, p: dict, sel: np.ndarray, rows_sel: np.ndarray, pos_sel: np.ndarray,
                        tokens: torch.Tensor, counts: Optional[np.ndarray], drafts_sel) -> Dict[str, List[int]]:
        """Run the draft model's proposal loop after this step's sampled tokens.  The values are read
        back here (speculative decoding schedules one step at a time, so nothing runs ahead of
        this sync): which of the step's drafts the target accepted decides where each request's
        text now ends."""
        from .draft_model import accepted_tails
        toks = tokens.cpu().numpy()
        last_idx, n_acc = accepted_tails(toks, counts, drafts_sel)
        last = tokens.index_select(0, self._to_dev(last_idx, torch.int64))
        b = self.batch
        start = pos_sel.astype(np.int64) + n_acc          # position of the token each request ends on now
        out = self.draft.propose(last, start, b.block_table[rows_sel], b.num_blocks[rows_sel], self.block_size,
                                 self.max_model_len, self._to_dev)
        ids = p["req_ids"]
        return {ids[i]: d for i, d in zip(sel.tolist(), out) if d}

    def _propose_from_hidden(self, p: dict, md, input_ids: torch.Tensor, positions: torch.Tensor,
                             hidden: torch.Tensor, sel, rows_sel, pos_sel, tokens: Optional[torch.Tensor],
                             smp_idx, counts, drafts_sel) -> Optional[Dict[str, List[int]]]:
        """EAGLE / multi-token-prediction heads.  The head pairs the target's hidden state at every position
        of this step with the token that FOLLOWS it: the next input token inside a request, the next prompt
        token at the end of an unfinished prompt chunk, and at a sampled position the token the target has
        just produced there (for a draft under verification that is the draft itself when it was accepted;
        positions behind a rejected draft get a wrong pair, are not counted as computed, and are run again
        by the next step)."""
        from .draft_model import accepted_tails
        b, total = self.batch, p["total"]
        ids = input_ids[:total]
        shifted = torch.empty_like(ids)
        shifted[:-1] = ids[1:]
        ends = p["qsl"][1:] - 1                            # flat index of every request's last scheduled token
        nxt = b.num_computed[p["rows"]]                    # (already advanced: the position after that token)
        known = nxt < b.num_known[p["rows"]]
        follow = np.where(known, b.token_ids[p["rows"], np.minimum(nxt, b.token_ids.shape[1] - 1)], 0)
        shifted.index_copy_(0, self._to_dev(ends, torch.int64), self._to_dev(follow.astype(np.int32)).to(shifted.dtype))
        if sel is not None:
            shifted.index_copy_(0, self._to_dev(smp_idx, torch.int64), tokens.to(shifted.dtype))
        if getattr(self.draft, "wants_aux", False):
            hidden = self.model.last_aux          # the residual stream at three depths, side by side
        feats = self.draft.run_step_tokens(md, shifted, positions[:total], hidden[:total])
        if sel is None:
            return None
        toks = tokens.cpu().numpy()
        last_idx, n_acc = accepted_tails(toks, counts, drafts_sel)
        # flat position of each sampled request's last kept token: its first sampled position + accepted drafts
        first = (p["qsl"][1:][sel] - 1) - ((counts - 1) if counts is not None else 0)
        keep = first + n_acc
        start = pos_sel.astype(np.int64) + n_acc          # position of the token each request ends on now
        out = self.draft.propose(feats.index_select(0, self._to_dev(keep, torch.int64)), start,
                                 b.block_table[rows_sel], b.num_blocks[rows_sel], self.block_size,
                                 self.max_model_len, self._to_dev)
        ids_ = p["req_ids"]
        return {ids_[i]: d for i, d in zip(sel.tolist(), out) if d}

    def _overlap_bounds(self, nreq: int):
        """Micro-batch row ranges for all-reduce/compute overlap, or None."""
        if self.comm_stream is None or nreq < 2:
            return None
        if envs.VDAMD_TP_OVERLAP != "force" and (nreq < envs.VDAMD_TP_OVERLAP_MIN_BATCH or self.tp.rccl is None):
            # without the RCCL binding the in-place all-reduce is host-staged: nothing to overlap with,
            # and it could not be captured into a hipGraph
            return None
        half = (nreq // 2 + 7) // 8 * 8 if nreq >= 32 else nreq // 2
        return [(0, half), (half, nreq)]

    @staticmethod
    def _slice_decode_metadata(md: AttentionMetadata, s: int, e: int) -> AttentionMetadata:
        n = e - s
        return AttentionMetadata(
            num_decodes=n, num_decode_tokens=n, num_prefills=0, num_prefill_tokens=0,
            slot_mapping=md.slot_mapping[s:e], block_table=md.block_table[s:e], seq_lens=md.seq_lens[s:e],
            query_start_loc=md.query_start_loc[:n + 1], max_query_len=1, max_seq_len=md.max_seq_len,
            max_decode_seq_len=md.max_decode_seq_len, decode_workspace=md.decode_workspace)

    def _forward_decode(self, md: AttentionMetadata, input_ids: torch.Tensor, positions: torch.Tensor):
        """Pure-decode forward on a single pipeline stage, overlapped when enabled.  Leaves the TP
        group's peer-mapped buffers at parity 0, so a capture of this call can be replayed anywhere."""
        try:
            bounds = self._overlap_bounds(md.num_decodes)
            if bounds is None:
                set_forward_context(md, positions)
                out = self.model(input_ids, positions, None)
                set_forward_context(None)
                return out
            mds = [self._slice_decode_metadata(md, s, e) for s, e in bounds]
            sp_ok = getattr(self.model, "sequence_parallel_ok", None)
            if sp_ok is not None and sp_ok(md.num_decodes) and all(sp_ok(e - s) for s, e in bounds):
                return self.model.forward_overlapped_sp(input_ids, positions, mds, bounds, self.comm_stream)
            return self.model.forward_overlapped(input_ids, positions, mds, bounds, self.comm_stream)
        finally:
            self.tp.end_step()

    def _is_output_rank(self) -> bool:
        return self.pp.is_last_rank and (self.tp.rank_in_group == 0 or self.report_on_all_ranks)

    def _run_eager(self, md: AttentionMetadata, input_ids: torch.Tensor, positions: torch.Tensor,
                   total: int):
        # adapter row indices refer to the whole batch: no micro-batch split while one is active
        if md.num_prefills == 0 and self.pp.world_size == 1 and self.comm_stream is not None and not lora_active():
            return self._forward_decode(md, input_ids, positions)
        inter = None
        if not self.pp.is_first_rank:
            inter = self._recv_intermediate(total)
        set_forward_context(md, positions)
        out = self.model(input_ids, positions, inter)
        set_forward_context(None)
        self.tp.end_step()
        if not self.pp.is_last_rank:
            self._send_intermediate(out)
            return None
        return out

    def _recv_intermediate(self, total: int, static: bool = False) -> IntermediateTensors:
        """``(hidden, residual)`` of the previous stage for ``total`` tokens: one message, received
        on the pipeline link's own stream.  ``static``: into the buffers a captured graph reads."""
        h = self.model_config.hidden_size
        dtype = self.model_config.torch_dtype
        want_res = self._pp_sends_residual()
        if static:
            hidden = self._static["pp_hidden"][:total]
            residual = self._static["pp_residual"][:total] if want_res else None
        else:
            hidden = torch.empty((total, h), dtype=dtype, device=self.device)
            residual = torch.empty((total, h), dtype=dtype, device=self.device) if want_res else None
        self.pp_link.recv(hidden, residual)
        return IntermediateTensors(hidden, residual)

    def _send_intermediate(self, out: IntermediateTensors, total: Optional[int] = None) -> None:
        hidden, residual = out.hidden, out.residual if self._pp_sends_residual() else None
        if total is not None:                  # a padded graph batch: only the real rows travel
            hidden = hidden[:total]
            residual = residual[:total] if residual is not None else None
        self.pp_link.send(hidden.contiguous(), None if residual is None else residual.contiguous())

    def _pp_sends_residual(self) -> bool:
        return self.model_config.architecture != "OPTForCausalLM"

    # -------------------------------------------------------------- sampling
    def _sample(self, hidden: torch.Tensor, rows: np.ndarray):
        b = self.batch
        temps = b.temperature[rows]
        need_lp = int(b.num_logprobs[rows].max()) if len(rows) else -1
        adjust = bool(b.has_adjust[rows].any())
        if (self.tp.world_size > 1 and need_lp < 0 and not temps.any() and not b.has_penalty[rows].any()
                and not adjust and hasattr(self.model, "compute_greedy_tokens")):
            # all-greedy batch under TP: argmax per vocabulary shard, gather only the winners
            return self.model.compute_greedy_tokens(hidden).to(torch.int32), None
        logits = self.model.compute_logits(hidden)
        if b.has_penalty[rows].any():
            for pend in list(self._pending):      # penalties count every token generated so far
                pend.get()
            logits = self._apply_penalties(logits.float(), rows)
        if adjust:
            logits = self._adjust_logits(logits.float(), rows)
        if not temps.any():
            tokens = ops.argmax(logits)
        else:
            uniform = b.draw_uniform(rows)
            tokens = ops.sample(
                logits, self._to_dev(temps, torch.float32), self._to_dev(b.top_k[rows], torch.int32),
                self._to_dev(b.top_p[rows], torch.float32), self._to_dev(b.min_p[rows], torch.float32),
                self._to_dev(uniform, torch.float32))
        logprobs = None
        if need_lp >= 0:
            chosen, tv, ti = ops.compute_logprobs(logits, tokens, max(need_lp, 0))
            chosen = chosen.tolist()
            tv_l = tv.tolist() if tv is not None else None
            ti_l = ti.tolist() if ti is not None else None
            tk = tokens.tolist()
            logprobs = []
            for j, r in enumerate(rows.tolist()):
                k = int(b.num_logprobs[r])
                if k < 0:
                    logprobs.append(None)
                    continue
                d = {tk[j]: chosen[j]}
                for t_id, t_lp in zip((ti_l[j][:k] if ti_l else []), (tv_l[j][:k] if tv_l else [])):
                    d.setdefault(t_id, t_lp)
                logprobs.append([d])
        return tokens.to(torch.int32), logprobs

    def _pool(self, hidden: torch.Tensor, p: dict):
        """Pooled final hidden states of the prefill rows that asked for them (embedding / scoring
        requests; reference ``src/launch.py:429`` serves vLLM's pooling routes with the rest of its
        app).  "last": the prompt's last position.  "mean": the sum over this chunk joins the
        request's running sum; the mean leaves with the chunk that completes the prompt.  Returns
        ``(req ids, [k, H] float32 on the device)`` for the prompts completed by this step."""
        b = self.batch
        nd, qsl, rows = p["num_decodes"], p["qsl"], p["rows"]
        ids, vecs = [], []
        for i in range(nd, len(rows)):
            r = int(rows[i])
            mode = int(b.pooling[r])
            if mode == 0:
                continue
            lo, hi = int(qsl[i]), int(qsl[i + 1])
            done = bool(p["sample_mask"][i])
            rid = b.req_ids[r]
            if mode == 1:
                if done:
                    ids.append(rid)
                    vecs.append(hidden[hi - 1].float())
                continue
            part = hidden[lo:hi].float().sum(0)
            acc = self._pool_acc.pop(rid, None)
            part = part if acc is None else acc + part
            if done:
                ids.append(rid)
                vecs.append(part / float(b.num_prompt_tokens[r]))
            else:
                self._pool_acc[rid] = part
        if len(self._pool_acc) > 4 * self.max_num_reqs:      # sums of requests aborted mid-prompt
            live = set(b.req_ids)
            self._pool_acc = {k: v for k, v in self._pool_acc.items() if k in live}
        return (ids, torch.stack(vecs)) if ids else None

    def _prompt_logprobs(self, hidden: torch.Tensor, p: dict) -> Optional[dict]:
        """Log-probabilities of the prompt tokens this step's prefill chunks predict:
        the hidden state at position i scores prompt token i + 1.  Logits are formed
        1024 positions at a time so a long chunk never holds a full [tokens, vocab]
        matrix.  Every TP rank runs the same slices (the logits gather is collective)."""
        b = self.batch
        rows, n_sched, seq_lens, qsl = p["rows"], p["n_sched"], p["seq_lens"], p["qsl"]
        out = {}
        for i in range(p["num_decodes"], len(rows)):
            r = int(rows[i])
            k = int(b.num_prompt_logprobs[r])
            if k < 0:
                continue
            n, npt = int(n_sched[i]), int(b.num_prompt_tokens[r])
            start = int(seq_lens[i]) - n
            lo, hi = start, min(start + n, npt - 1)
            if hi <= lo:
                continue
            off = int(qsl[i]) - start
            entries: list = []
            for c0 in range(lo, hi, 1024):
                c1 = min(hi, c0 + 1024)
                logits = self.model.compute_logits(hidden[off + c0:off + c1])
                targets = b.token_ids[r, c0 + 1:c1 + 1]
                chosen, tv, ti = ops.compute_logprobs(logits, self._to_dev(targets, torch.int64), k)
                chosen = chosen.tolist()
                tv_l = tv.tolist() if tv is not None else None
                ti_l = ti.tolist() if ti is not None else None
                for j, t in enumerate(targets.tolist()):
                    d = {t: chosen[j]}
                    if ti_l:
                        for t_id, t_lp in zip(ti_l[j], tv_l[j]):
                            d.setdefault(t_id, t_lp)
                    entries.append(d)
            out[p["req_ids"][i]] = (lo + 1, entries)
        return out or None

    def _adjust_logits(self, logits: torch.Tensor, rows: np.ndarray) -> torch.Tensor:
        """Per-request logit edits, in the order bias -> allow list -> ban lists:
        ``logit_bias`` adds, ``allowed_token_ids`` masks everything else,
        ``bad_words`` bans a word's last token while the generated tail spells
        the rest of it, and ``min_tokens`` bans EOS / stop tokens until enough
        tokens exist.  The edits are gathered on the host into one sparse
        update so the device sees two index kernels whatever the batch mix."""
        b = self.batch
        v = logits.shape[-1]
        neg = float("-inf")
        add_r, add_c, add_v = [], [], []
        ban_r, ban_c = [], []
        allow = []
        synced = False
        for j, r in enumerate(rows.tolist()):
            if not b.has_adjust[r]:
                continue
            sp = b.params[r]
            if sp.logit_bias:
                for t, val in sp.logit_bias.items():
                    if t < v:
                        add_r.append(j)
                        add_c.append(t)
                        add_v.append(val)
            if sp.allowed_token_ids is not None:
                allow.append((j, [t for t in sp.allowed_token_ids if t < v]))
            n_out = int(b.num_tokens[r]) - int(b.num_prompt_tokens[r])
            if sp.min_tokens > 0 and n_out < sp.min_tokens:
                for t in sp.min_tokens_ban_ids:
                    if t < v:
                        ban_r.append(j)
                        ban_c.append(t)
            if sp.bad_words_token_ids:
                if not synced:                      # the tail must hold every token generated so far
                    for pend in list(self._pending):
                        pend.get()
                    synced = True
                nt = int(b.num_tokens[r])
                npt = int(b.num_prompt_tokens[r])
                for w in sp.bad_words_token_ids:
                    k = len(w) - 1
                    if w[-1] >= v or k > nt - npt:
                        continue                    # the prefix has to come from generated tokens
                    if k == 0 or b.token_ids[r, nt - k:nt].tolist() == list(w[:-1]):
                        ban_r.append(j)
                        ban_c.append(w[-1])
        dev = logits.device
        for j, r in enumerate(rows.tolist()):
            if b.grammar[r] is not None:
                if not synced:
                    for pend in list(self._pending):
                        pend.get()
                    synced = True
                keep = self._grammar_mask(r, v)
                if keep is not None:
                    logits[j].masked_fill_(~torch.from_numpy(keep).to(dev).bool(), neg)
        if add_r:
            logits.index_put_((torch.tensor(add_r, device=dev), torch.tensor(add_c, device=dev)),
                              torch.tensor(add_v, dtype=logits.dtype, device=dev), accumulate=True)
        for j, ids in allow:
            keep = logits[j, ids].clone()
            logits[j].fill_(neg)
            logits[j, ids] = keep
        if ban_r:
            logits[torch.tensor(ban_r, device=dev), torch.tensor(ban_c, device=dev)] = neg
        return logits

    def _grammar_mask(self, r: int, v: int) -> Optional[np.ndarray]:
        """uint8 [v] of the tokens a constrained request may produce next: the grammar is first
        advanced over the output tokens it has not seen yet.  Once the object is complete (or
        nothing else fits) only EOS remains."""
        b = self.batch
        if self._token_table is None or self._token_table.n != v:
            from ..ops.grammar import TokenTable
            from ..tokenizer import get_tokenizer
            self._token_table = TokenTable(get_tokenizer(self.model_config).token_bytes(), v)
        table, g = self._token_table, b.grammar[r]
        npt, nt = int(b.num_prompt_tokens[r]), int(b.num_tokens[r])
        for t in b.token_ids[r, npt + int(b.grammar_pos[r]):nt].tolist():
            if not g.advance(table.tokens[t] if t < table.n else b""):
                # tokens from before a constraint could apply (should not happen): stop constraining
                logger.warning("request %s left its output grammar; constraint dropped", b.req_ids[r])
                b.grammar[r] = None
                return None
        b.grammar_pos[r] = nt - npt
        keep = None if g.done else g.mask(table)
        eos = b.params[r].eos_token_id
        if keep is None or not keep.any():
            keep = np.zeros(v, dtype=np.uint8)
            if eos is not None and eos < v:
                keep[eos] = 1
            else:
                return None
        elif getattr(g, "may_stop", False) and eos is not None and eos < v:
            keep = keep.copy()                    # complete already, but a longer alternative exists
            keep[eos] = 1
        return keep

    def _apply_penalties(self, logits: torch.Tensor, rows: np.ndarray) -> torch.Tensor:
        b = self.batch
        v = logits.shape[-1]
        counts = np.zeros((len(rows), v), dtype=np.float32)
        prompt_mask = np.zeros((len(rows), v), dtype=bool)
        for j, r in enumerate(rows.tolist()):
            if not b.has_penalty[r]:
                continue
            npt, nt = int(b.num_prompt_tokens[r]), int(b.num_tokens[r])
            out_t = b.token_ids[r, npt:nt]
            out_t = out_t[out_t < v]
            np.add.at(counts[j], out_t, 1.0)
            pt = b.token_ids[r, :npt]
            prompt_mask[j, pt[pt < v]] = True
        dev = logits.device
        return ops.apply_penalties(
            logits, torch.from_numpy(counts).to(dev), torch.from_numpy(prompt_mask).to(dev),
            torch.from_numpy(b.presence[rows]).to(dev), torch.from_numpy(b.frequency[rows]).to(dev),
            torch.from_numpy(b.repetition[rows]).to(dev))

    # ---------------------------------------------------------- hipGraph path
    def _pick_graph(self, nreq: int, max_blocks: int) -> Optional[DecodeGraph]:
        if max_blocks > self._static["block_table"].shape[1]:
            return None
        for s in self.graph_sizes:
            if s >= nreq:
                # the variant captured for the shortest context that covers this batch: its split-KV
                # factor was chosen for that length (a short context is not cut into needless splits)
                ctx = max_blocks * self.block_size
                for limit, g in self.graph_ctx.get(s, ()):
                    if limit >= ctx:
                        return g
                return self.graphs[s]
        return None

    CONTEXT_BUCKETS = (512, 4096, 32768)

    def _context_variants(self, bs: int) -> List[int]:
        """Context limits worth a graph of their own at this batch size: those at which the decode
        kernel's split-KV factor differs from the next larger limit's."""
        if not ops.hip_available():
            return [self.max_model_len]
        from ..ops import hip
        hkv = self.model.attention_layers()[0].num_kv_heads
        limits = [c for c in self.CONTEXT_BUCKETS if c < self.max_model_len] + [self.max_model_len]
        keep = [limits[-1]]
        for c in reversed(limits[:-1]):
            if hip.choose_num_splits(bs, hkv, c) != hip.choose_num_splits(bs, hkv, keep[0]):
                keep.insert(0, c)
        return keep

    def _alloc_static(self, max_bs: int) -> None:
        dev = self.device
        mb = self.batch.max_blocks
        self._static = {
            "input_ids": torch.zeros(max_bs, dtype=torch.int32, device=dev),
            "positions": torch.zeros(max_bs, dtype=torch.int32, device=dev),
            "slot_mapping": torch.full((max_bs,), -1, dtype=torch.int32, device=dev),
            "seq_lens": torch.zeros(max_bs, dtype=torch.int32, device=dev),
            "block_table": torch.zeros(max_bs, mb, dtype=torch.int32, device=dev),
            "qsl": torch.arange(max_bs + 1, dtype=torch.int32, device=dev),
        }
        if self.is_gpu:
            # pinned staging so one H2D per tensor per step
            self._stage = {k: torch.zeros_like(v, device="cpu").pin_memory()
                           for k, v in self._static.items() if k != "qsl"}

    def _static_metadata(self, bs: int, max_seq_len: int) -> AttentionMetadata:
        s = self._static
        return AttentionMetadata(
            num_decodes=bs, num_decode_tokens=bs, num_prefills=0, num_prefill_tokens=0,
            slot_mapping=s["slot_mapping"][:bs], block_table=s["block_table"][:bs],
            seq_lens=s["seq_lens"][:bs], query_start_loc=s["qsl"][:bs + 1],
            max_query_len=1, max_seq_len=max_seq_len, max_decode_seq_len=max_seq_len,
            decode_workspace=self.decode_workspace)

    @torch.inference_mode()
    def capture_graphs(self) -> None:
        cc = self.vllm_config.compilation_config
        self.graph_fallback_reason = None
        if not self.is_gpu:
            return
        if not (cc.use_hip_graph and envs.VDAMD_USE_HIP_GRAPH):
            self.graph_fallback_reason = "disabled (enforce_eager / VDAMD_USE_HIP_GRAPH=0)"
            return
        first = self.model.attention_layers()[0]
        kernel_head_dim = first.head_dim      # (padded to a kernel size where needed)
        if getattr(first, "latent", False):
            kernel_ok = ops.mla_kernel_ok(first.head_dim, first.v_dim, self.model_config.torch_dtype)
        else:
            kernel_ok = ops.hip_supports_head_dim(kernel_head_dim)
        if not kernel_ok:
            self.graph_fallback_reason = f"no HIP attention kernel for head_dim {kernel_head_dim}"
            logger.info("hipGraph decode capture needs the HIP attention kernels (head_dim 64 or 128): "
                        "running eagerly")
            return
        if self.tp.world_size > 1 and self.tp.device_group is not None and not self.tp.graph_capturable():
            # torch.distributed collectives keep host-side work objects; the direct RCCL binding
            # and the peer-mapped kernels are pure stream enqueues that a capture can record
            self.graph_fallback_reason = ("neither the direct RCCL communicator nor the peer-mapped collectives "
                                          "are active (torch.distributed collectives cannot be captured)")
            logger.info("hipGraph decode capture needs capturable TP collectives: running eagerly")
            return
        sizes = [s for s in cc.sizes(self.max_num_reqs) if s <= self.max_num_reqs]
        if not sizes:
            self.graph_fallback_reason = "no capture sizes configured"
            return
        t0 = time.time()
        self.graphs.clear()
        self.graph_ctx.clear()
        self._alloc_static(max(sizes))
        if not self.pp.is_first_rank:         # what the previous stage sends lands here, in front of the graph
            h, dt = self.model_config.hidden_size, self.model_config.torch_dtype
            self._static["pp_hidden"] = torch.zeros(max(sizes), h, dtype=dt, device=self.device)
            self._static["pp_residual"] = torch.zeros(max(sizes), h, dtype=dt, device=self.device)
        self._ensure_decode_workspace(max(sizes))
        error: Optional[BaseException] = None
        try:
            for bs in sorted(sizes, reverse=True):
                variants = self._context_variants(bs)
                for ctx in reversed(variants):             # the full-length one first: it sizes the pool
                    g = self._capture_one(bs, ctx)
                    self.graph_ctx.setdefault(bs, []).insert(0, (ctx, g))
                self.graphs[bs] = self.graph_ctx[bs][-1][1]
        except Exception as e:  # capture is an optimisation: fall back to eager
            error = e
            logger.warning("hipGraph capture raised", exc_info=True)
            torch.cuda.synchronize()
        # every rank of the group takes the same decision: a rank replaying a graph and a
        # rank running eagerly would still issue the same collectives, but a communicator
        # that failed mid-capture has to be re-checked by all of its ranks together
        verdicts = self.tp.all_gather_object(None if error is None else repr(error))
        failed = [(r, v) for r, v in enumerate(verdicts) if v is not None]
        if failed:
            self.graph_fallback_reason = "capture failed on tp rank(s) " + ", ".join(f"{r}: {v}" for r, v in failed)
            logger.warning("hipGraph %s: running decode eagerly", self.graph_fallback_reason)
            self.graphs.clear()
            self.graph_ctx.clear()
            self.graph_sizes = []
            self.graph_pool = None
            if self.tp.world_size > 1 and (self.tp.rccl is not None or self.tp.custom_ar is not None):
                from ..parallel.accel import reverify_after_failed_capture
                reverify_after_failed_capture(self.tp)
            return
        self.graph_sizes = sorted(self.graphs)
        logger.info("captured %d decode graphs (batch sizes %s, %d context variants) in %.1fs",
                    sum(len(v) for v in self.graph_ctx.values()), self.graph_sizes,
                    sum(len(v) - 1 for v in self.graph_ctx.values()), time.time() - t0)

    def _ensure_decode_workspace(self, max_bs: int) -> None:
        if not self.is_gpu or not ops.hip_available():
            return
        from ..ops import hip
        attn = self.model.attention_layers()[0]
        n = hip.decode_workspace_floats(max(max_bs, self.max_num_reqs), attn.num_heads, attn.head_dim,
                                        attn.num_kv_heads, self.max_model_len)
        self.decode_workspace = torch.empty(n, dtype=torch.float32, device=self.device)

    def _forward_stage(self, md: AttentionMetadata, input_ids: torch.Tensor, positions: torch.Tensor,
                       inter: Optional[IntermediateTensors]):
        """This pipeline stage's layers on a decode batch (no transfers: those stay outside a capture)."""
        try:
            set_forward_context(md, positions)
            out = self.model(input_ids, positions, inter)
            set_forward_context(None)
            return out
        finally:
            self.tp.end_step()

    def _capture_one(self, bs: int, ctx: Optional[int] = None) -> DecodeGraph:
        s = self._static
        md = self._static_metadata(bs, ctx or self.max_model_len)
        ids, pos = s["input_ids"][:bs], s["positions"][:bs]
        inter = None
        if not self.pp.is_first_rank:
            inter = IntermediateTensors(s["pp_hidden"][:bs],
                                        s["pp_residual"][:bs] if self._pp_sends_residual() else None)
        if self.pp.world_size > 1:
            def run():
                return self._forward_stage(md, ids, pos, inter)
        else:
            def run():
                return self._forward_decode(md, ids, pos)
        # warm up on a side stream (allocator + lazy init), then capture
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        g = DecodeGraph(bs)
        g.graph = torch.cuda.CUDAGraph()
        # thread_local: API calls on other threads (RCCL watchdog, HTTP loop) must not
        # invalidate this capture
        with torch.cuda.graph(g.graph, pool=self.graph_pool, capture_error_mode="thread_local"):
            g.output = run()
            g.aux = getattr(self.model, "last_aux", None)
        set_forward_context(None)
        if self.graph_pool is None:
            self.graph_pool = g.graph.pool()
        self.graphs[bs] = g
        return g

    def _run_graph(self, g: DecodeGraph, p: dict):
        s, st = self._static, self._stage
        n, bs = len(p["req_ids"]), g.batch_size
        mb = p["max_blocks"]
        st["input_ids"][:n] = torch.from_numpy(p["input_ids"])
        st["positions"][:n] = torch.from_numpy(p["positions"])
        st["slot_mapping"][:n] = torch.from_numpy(p["slot_mapping"])
        st["seq_lens"][:n] = torch.from_numpy(p["seq_lens"])
        st["block_table"][:n, :mb] = torch.from_numpy(self.batch.block_table[p["rows"], :mb])
        if n < bs:      # padding rows: no cache write, zero-length attention
            st["slot_mapping"][n:bs] = -1
            st["seq_lens"][n:bs] = 0
            st["input_ids"][n:bs] = 0
            st["positions"][n:bs] = 0
        for k in ("input_ids", "positions", "slot_mapping", "seq_lens"):
            s[k][:bs].copy_(st[k][:bs], non_blocking=True)
        s["block_table"][:n, :mb].copy_(st["block_table"][:n, :mb], non_blocking=True)
        self._patch_unknown_inputs(s["input_ids"], p)
        if self.pp_link is not None:
            # the stage's layers are the graph; the transfers around it run on the link's streams
            if not self.pp.is_first_rank:
                self._recv_intermediate(n, static=True)
            self.pp_link.wait_send()           # the previous batch's transfer has read the output buffers
            g.graph.replay()
            if not self.pp.is_last_rank:
                self._send_intermediate(g.output, n)
                return None
            return g.output
        g.graph.replay()
        if g.aux is not None:
            self.model.last_aux = g.aux
        return g.output

    # ----------------------------------------------------------------- warmup
    @torch.inference_mode()
    def warm_up(self) -> None:
        """Run one tiny prefill + decode through the real kernels so lazy
        initialisation (hipBLASLt heuristics, RCCL channels) is off the
        request path, then capture decode graphs."""
        if self.decode_workspace is None:
            self._ensure_decode_workspace(self.max_num_reqs)
        tuned = self.is_gpu and envs.VDAMD_TUNE_GEMM and self._enable_gemm_tuning()
        self.capture_graphs()
        if tuned:
            # every captured batch size has run (and timed) its GEMM shapes: freeze the
            # choices so no tuning run ever lands on the request path
            torch.cuda.tunable.tuning_enable(False)
            if hasattr(torch.cuda.tunable, "write_file"):
                torch.cuda.tunable.write_file()      # otherwise the library writes the file at exit

    def _enable_gemm_tuning(self) -> bool:
        """Let the library time its candidate kernels for each decode GEMM shape
        during the warm-up passes that precede graph capture (measured on
        MI355X: Llama-3-8B decode at batch 256 -5% per step; Llama-3-70B at batch
        2048 between -1.7% and +3.4% from run to run, the short timing runs being
        noisy at that size - hence opt-in).  Results persist in ``VDAMD_TUNE_GEMM_FILE``."""
        try:
            tun = torch.cuda.tunable
            tun.enable(True)
            tun.tuning_enable(True)
            tun.set_max_tuning_duration(50)
            tun.set_filename(envs.VDAMD_TUNE_GEMM_FILE, insert_device_ordinal=True)
            os.makedirs(os.path.dirname(envs.VDAMD_TUNE_GEMM_FILE) or ".", exist_ok=True)
            return True
        except Exception as e:  # noqa: BLE001
            logger.warning("GEMM tuning is unavailable: %s", e)
            return False
