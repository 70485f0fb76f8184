"""Headline benchmark: decode throughput (output tokens/s, whole node) and p50
time-to-first-token for Llama-3-70B bf16, tensor-parallel over N MI355X GPUs.

    python bench.py --gpus 1 --steps 32 --warmup 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" is one engine step (scheduler -> executor -> worker forward ->
sampling -> scheduler update) in which every running sequence produces one
token.  The batch is 2048 sequences per GPU (weak scaling: global batch 2048*N,
TP = N), 128-token synthetic prompts, random-init weights of the real
architecture, greedy sampling.  Prefill of the whole batch and W decode steps
are warm-up; exactly K decode steps are timed between barrier+synchronize
pairs and the maximum over ranks is reported.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _fit_batch_per_gpu(args) -> int:
    """Largest per-GPU batch (<= --batch-per-gpu) whose KV cache fits next to
    the weights for the requested number of steps.  Sized for N=1, where one
    GPU holds all the weights, and used at every N so the per-GPU work is the
    same along the scaling curve."""
    if args.device != "cuda":
        return args.batch_per_gpu
    import math

    import torch

    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import hf_weight_shapes
    mc = ModelConfig(model=args.model, dtype=args.dtype, load_format="dummy")
    esz = 2 if args.dtype != "float32" else 4
    weights = sum(math.prod(s) for s in hf_weight_shapes(mc).values()) * esz
    index = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    total = torch.cuda.get_device_properties(index).total_memory
    kv_budget = total * args.gpu_memory_utilization - weights - (10 << 30)     # activations, graphs, reserve
    per_token = 2 * mc.num_hidden_layers * mc.num_kv_heads * mc.head_dim * esz
    block = 64
    per = args.batch_per_gpu
    while per > 64:
        prefill_steps = -(-per * args.prompt_len // args.prefill_budget_per_gpu)
        tokens = args.prompt_len + prefill_steps + args.warmup + args.steps + 6
        if per * (-(-tokens // block)) * block * per_token <= kv_budget:
            break
        per -= 64
    return per


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", type=str, default="llama-3-70b")
    ap.add_argument("--batch-per-gpu", type=int, default=2048)
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--ttft-requests", type=int, default=5)
    ap.add_argument("--dtype", type=str, default="bfloat16")
    ap.add_argument("--enforce-eager", action="store_true")
    ap.add_argument("--quantization", type=str, default=None, choices=[None, "fp8", "awq", "nvfp4"],
                    help="NOT the headline configuration (that is bf16): reduced-precision linears")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture the decode step in a hipGraph also for N>1 (default there: eager - at "
                         "thousands of sequences per GPU the launches are hidden behind ~300 us kernels)")
    ap.add_argument("--executor", type=str, default="auto", choices=["auto", "uni", "spmd"])
    ap.add_argument("--sync-scheduling", action="store_true",
                    help="schedule each step only after the previous step's tokens reached the host")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.92)
    ap.add_argument("--prefill-budget-per-gpu", type=int, default=16384,
                    help="tokens per prefill step and GPU (max_num_batched_tokens = this x N)")
    ap.add_argument("--num-gpu-blocks-override", type=int, default=None,
                    help="fixed KV-cache size in blocks (rehearsals with several ranks on one device)")
    ap.add_argument("--simulate-tp", type=int, default=1,
                    help="one GPU stands in for one rank of a TP group of this size (per-rank compute proxy; "
                         "collectives move no data and the result is NOT a benchmark number)")
    ap.add_argument("--device", type=str, default="cuda", choices=["cuda", "cpu"],
                    help="cpu = gloo rehearsal of the multi-rank path with a tiny model (not a measurement)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    n = args.gpus
    if args.simulate_tp > 1:
        assert n == 1, "--simulate-tp runs on one GPU"
        os.environ["VDAMD_SIMULATE_TP"] = str(args.simulate_tp)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n > 1 and world != n:
        raise SystemExit(f"--gpus {n} needs one rank per GPU: launch with "
                         f"`python -m torch.distributed.run --nproc-per-node {n} bench.py --gpus {n} ...`")
    os.environ.setdefault("VLLM_LOGGING_LEVEL", "INFO" if rank == 0 else "WARNING")
    if n > 1:
        # report the time inside collectives next to the step time (event pairs, no extra syncs)
        os.environ.setdefault("VDAMD_TIME_COLLECTIVES", "1")

    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.sampling_params import SamplingParams

    # prefill budget scales with N so the batch is prefilled in the same number of steps at every N
    shards = n * args.simulate_tp
    prefill_budget = args.prefill_budget_per_gpu * shards
    per_gpu = _fit_batch_per_gpu(args)
    batch = per_gpu * shards
    # continuous batching staggers the sequences by the number of prefill steps
    prefill_steps = -(-batch * args.prompt_len // max(prefill_budget, batch))
    total_new = args.warmup + args.steps + prefill_steps + 10
    max_len = args.prompt_len + total_new + 16
    backend = args.executor
    if backend == "auto":
        backend = "spmd" if n > 1 else "uni"
    engine = LLMEngine.from_engine_args(EngineArgs(
        model=args.model, dtype=args.dtype, load_format="dummy", tensor_parallel_size=n, device=args.device,
        quantization=args.quantization,
        distributed_executor_backend=backend, max_model_len=max_len, max_num_seqs=batch,
        max_num_batched_tokens=max(prefill_budget, batch), enable_prefix_caching=False,
        enforce_eager=args.enforce_eager or (n > 1 and not args.hip_graph), max_capture_size=batch, hip_graph_sizes=[batch],
        gpu_memory_utilization=args.gpu_memory_utilization,
        num_gpu_blocks_override=args.num_gpu_blocks_override, seed=0, async_scheduling=not args.sync_scheduling))
    vocab = engine.model_config.vocab_size
    rng = np.random.default_rng(1234)
    # the KV cache the workers could actually allocate has the last word on the batch
    blocks_per_seq = -(-(args.prompt_len + total_new) // engine.vllm_config.cache_config.block_size)
    fit = engine.vllm_config.cache_config.num_gpu_blocks // blocks_per_seq
    if fit < batch:
        batch = max(8 * n, fit // (8 * n) * (8 * n))

    on_gpu = args.device == "cuda"

    def sync() -> None:
        if dist.is_initialized():
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    # ---- p50 TTFT: single requests on an idle engine
    ttfts = []
    for i in range(args.ttft_requests + 1):
        ids = rng.integers(0, vocab, size=args.prompt_len).tolist()
        sync()
        t0 = time.perf_counter()
        rid = engine.add_request(f"ttft-{i}", ids, SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True,
                                                                  detokenize=False))
        got = False
        while not got:
            for out in engine.step():
                got = got or (out.request_id == rid and len(out.outputs[0].token_ids) >= 1)
        if on_gpu:
            torch.cuda.synchronize()
        if i > 0:                      # first one pays lazy initialisation
            ttfts.append(time.perf_counter() - t0)
        while engine.has_unfinished_requests():
            engine.step()
    ttft_p50_ms = 1e3 * float(np.median(ttfts)) if ttfts else None

    # ---- throughput: fill the batch, prefill it, warm up, time K decode steps
    sp = SamplingParams(temperature=0.0, max_tokens=total_new, ignore_eos=True, detokenize=False)
    for i in range(batch):
        engine.add_request(f"b-{i}", rng.integers(0, vocab, size=args.prompt_len).tolist(), sp)
    sched = engine.core.scheduler
    t_prefill0 = time.perf_counter()
    while sched.waiting or any(r.num_computed_tokens < r.num_prompt_tokens for r in sched.running):
        engine.step()
    if on_gpu:
        torch.cuda.synchronize()
    prefill_s = time.perf_counter() - t_prefill0
    if not args.sync_scheduling:
        engine.step()          # fill the one-step-ahead pipeline: from here every call returns one full step
    for _ in range(args.warmup):
        engine.step()
    produced = 0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = engine.step()
        produced += sum(len(o.new_token_ids) for o in outs)
    sync()
    elapsed = time.perf_counter() - t0
    assert produced == batch * args.steps, (produced, batch, args.steps)
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if on_gpu and dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        runner = engine.core.executor.driver.worker.model_runner
        tps = batch * args.steps / elapsed
        dt = {"bfloat16": "bf16", "float16": "fp16", "float32": "fp32"}[args.dtype]
        if args.quantization:
            dt = f"{args.quantization}-weights/{dt}"
        pretty = {"llama-3-70b": "Llama-3-70B", "llama-3-8b": "Llama-3-8B"}.get(args.model, args.model)
        print(json.dumps({
            "metric": (f"output tokens/sec (whole node), {pretty} {dt} TP over N GPUs, decode"
                       if args.simulate_tp == 1 else
                       f"NOT A BENCHMARK: per-rank compute proxy, one GPU simulating rank 0 of TP={args.simulate_tp}"),
            "value": round(tps, 2), "unit": "tokens/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt,
            "data": "synthetic prompts (random token ids), random-init weights",
            "ttft_p50_ms": None if ttft_p50_ms is None else round(ttft_p50_ms, 2),
            "prefill_tokens_per_s": round(batch * args.prompt_len / prefill_s, 1),
            "config": {"model": args.model, "global_batch": batch, "seq_len": args.prompt_len,
                       "parallelism": f"tp{n}", "executor": backend,
                       "async_scheduling": not args.sync_scheduling,
                       "hip_graph": bool(runner.graphs), "block_size": runner.block_size,
                       "step_breakdown_ms": {k: round(v, 3) for k, v in engine.core.timer.summary().items()}},
        }))
    engine.shutdown()


if __name__ == "__main__":
    main()
