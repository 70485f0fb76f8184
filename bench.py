"""Headline benchmark: decode throughput (output tokens/s, whole node) and p50
time-to-first-token for Llama-3-70B bf16, tensor-parallel over N MI355X GPUs,
driven through the serving stack's own executor.

    python bench.py --gpus 1 --steps 32 --warmup 4
    python bench.py --gpus 8                      # head + 8 spawned workers, no launcher needed
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

For N > 1 the engine runs exactly as ``launch.py serve -tp N`` runs it, minus HTTP:
a head process (scheduler, KV block manager, no GPU context) fans every step out to
one worker process per GPU over the RPC control plane, and the workers talk RCCL /
peer-mapped all-reduce over xGMI.  Started bare, the head spawns all N workers
itself.  Started under ``torch.distributed.run`` (one process per GPU), rank 0
becomes the head plus a spawned worker for GPU 0 and ranks 1..N-1 register with it
as ``launch.py remote`` workers do - "head + N-1 workers on one host".
``--executor spmd`` keeps the launcher-only lower bound (every rank runs its own
scheduler, no control-plane traffic).

One "step" is one engine step (scheduler -> executor -> worker forward ->
sampling -> scheduler update) in which every running sequence produces one
token.  Headline: 2048 sequences per GPU (weak scaling: global batch 2048*N,
TP = N), 128-token synthetic prompts, random-init weights of the real
architecture, greedy sampling.  Prefill of the whole batch and W decode steps
are warm-up; exactly K decode steps are timed between two "every worker has
drained its device" barriers.  The same run then measures a fixed global batch
of 256 sequences (strong scaling: the latency regime) and reports it next to
the headline.

The headline layout is tensor parallelism over all N GPUs (the named configuration).  ``--parallelism``
selects another split of the same N GPUs (``pp``: N pipeline stages, ``tpAxppB``), and for N > 1 the
run ends by measuring the pipeline-parallel layout in a child process and reporting it next to the
headline as ``alt_parallelism``: on a point-to-point xGMI mesh a weak-scaled decode batch moves
``2 * layers`` all-reduces of ``[2048 N, hidden]`` under TP and one ``[2048, hidden]`` pair per stage
boundary under PP.
"""
from __future__ import annotations

import argparse
import json
import os
import secrets
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MI355X_HBM_BYTES = 288 << 30


def _fit_batch_per_gpu(args, query_device: bool) -> int:
    """Largest per-GPU batch (<= --batch-per-gpu) whose KV cache fits next to
    the weights for the requested number of steps.  Sized for N=1, where one
    GPU holds all the weights, and used at every N so the per-GPU work is the
    same along the scaling curve."""
    if args.device != "cuda":
        return args.batch_per_gpu
    import math

    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import hf_weight_shapes
    mc = ModelConfig(model=args.model, dtype=args.dtype, load_format="dummy")
    esz = 2 if args.dtype != "float32" else 4
    weights = sum(math.prod(s) for s in hf_weight_shapes(mc).values()) * esz
    total = MI355X_HBM_BYTES
    if query_device:           # never in a head process: it must stay free of any GPU context
        import torch
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


def _torchrun_rendezvous(rank: int) -> tuple:
    """(rpc_port, rpc_secret) agreed between the launcher's ranks through the
    launcher's own key-value store; rank 0 picks them."""
    from datetime import timedelta

    from torch.distributed import TCPStore

    from vllm_distributed_amd.utils import get_open_port
    addr, port = os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"])
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
    store = TCPStore(addr, port, None, rank == 0 and not agent_store, timedelta(seconds=600))
    key = "vdamd_bench_rpc/" + os.environ.get("TORCHELASTIC_RUN_ID", "0") + "/" + os.environ.get(
        "TORCHELASTIC_RESTART_COUNT", "0")
    if rank == 0:
        value = f"{get_open_port()}:{secrets.token_hex(16)}"
        store.set(key, value)
    else:
        value = store.get(key).decode()
    p, s = value.split(":")
    return int(p), s


def _parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", type=str, default="llama-3-70b")
    ap.add_argument("--batch-per-gpu", type=int, default=2048)
    ap.add_argument("--strong-batch", type=int, default=256,
                    help="fixed global batch of the strong-scaling point (0 = skip it)")
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--ttft-requests", type=int, default=5)
    ap.add_argument("--dtype", type=str, default="bfloat16")
    ap.add_argument("--enforce-eager", action="store_true")
    ap.add_argument("--quantization", type=str, default=None, choices=[None, "fp8", "awq", "nvfp4"],
                    help="NOT the headline configuration (that is bf16): reduced-precision linears")
    ap.add_argument("--kv-cache-dtype", type=str, default="auto",
                    help="NOT the headline configuration (that is bf16): e.g. fp8 KV cache")
    ap.add_argument("--executor", type=str, default="auto", choices=["auto", "uni", "distributed", "spmd"],
                    help="auto: in-process worker for N=1, head + workers over the RPC control plane for N>1; "
                         "spmd: launcher-only lower bound (needs torch.distributed.run)")
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
    ap.add_argument("--parallelism", type=str, default="tp",
                    help="tp (default: TP = N), pp (PP = N), or tpAxppB with A * B = N")
    ap.add_argument("--alt-parallelism", type=str, default="pp",
                    help="second layout measured in a child process after the headline when N > 1 ('none' = skip)")
    ap.add_argument("--alt-timeout", type=float, default=900.0)
    ap.add_argument("--device", type=str, default="cuda", choices=["cuda", "cpu"],
                    help="cpu = gloo rehearsal of the multi-rank path with a tiny model (not a measurement)")
    return ap.parse_args()


def _split(spec: str, n: int) -> tuple:
    """``(tp, pp)`` of a ``--parallelism`` value for N GPUs."""
    spec = spec.strip().lower()
    if spec == "tp":
        return n, 1
    if spec == "pp":
        return 1, n
    import re
    m = re.fullmatch(r"tp(\d+)xpp(\d+)", spec)
    if not m or int(m.group(1)) * int(m.group(2)) != n:
        raise SystemExit(f"--parallelism {spec!r}: expected tp, pp or tpAxppB with A * B = {n}")
    return int(m.group(1)), int(m.group(2))


def _measure_alt_layout(args, n: int) -> dict:
    """The same weak point under ``--alt-parallelism``, in a child process started bare (its head spawns
    all N workers) once this process's workers have gone.  A failure there never costs the headline."""
    import subprocess
    tp, pp = _split(args.alt_parallelism, n)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k not in (
        "RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR",
        "MASTER_PORT", "VLLM_SERVER_PORT", "VLLM_RPC_SECRET", "VLLM_HEAD_LOCAL_DEVICES")}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--model", args.model, "--batch-per-gpu", str(args.batch_per_gpu), "--strong-batch", "0",
           "--prompt-len", str(args.prompt_len), "--ttft-requests", "1", "--dtype", args.dtype, "--parallelism",
           args.alt_parallelism, "--alt-parallelism", "none", "--device", args.device,
           "--gpu-memory-utilization", str(args.gpu_memory_utilization),
           "--prefill-budget-per-gpu", str(args.prefill_budget_per_gpu)]
    if args.num_gpu_blocks_override is not None:
        cmd += ["--num-gpu-blocks-override", str(args.num_gpu_blocks_override)]
    if args.enforce_eager:
        cmd.append("--enforce-eager")
    if args.sync_scheduling:
        cmd.append("--sync-scheduling")
    try:
        time.sleep(3.0)                    # the previous engine's workers release their devices as they exit
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.alt_timeout)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"parallelism": f"tp{tp}xpp{pp}", "error": (r.stderr or r.stdout)[-600:]}
        j = json.loads(lines[-1])
        return {"parallelism": j["config"]["parallelism"], "tokens_per_s": j["value"], "ms_per_step": j["ms_per_step"],
                "global_batch": j["config"]["global_batch"], "ttft_p50_ms": j.get("ttft_p50_ms"),
                "hip_graph": j["config"]["hip_graph"], "step_breakdown_ms": j["config"]["step_breakdown_ms"]}
    except Exception as e:  # noqa: BLE001
        return {"parallelism": f"tp{tp}xpp{pp}", "error": repr(e)[-600:]}


def main() -> None:
    args = _parse()
    n = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = world > 1                                  # torch.distributed.run gave every GPU a process
    backend = args.executor
    if backend == "auto":
        backend = "uni" if n == 1 else "distributed"
    if args.simulate_tp > 1:
        assert n == 1, "--simulate-tp runs on one GPU"
        os.environ["VDAMD_SIMULATE_TP"] = str(args.simulate_tp)
    if launched and world != n:
        raise SystemExit(f"launcher started {world} ranks for --gpus {n}")
    if backend == "spmd" and n > 1 and not launched:
        raise SystemExit("--executor spmd needs one launched rank per GPU: "
                         f"python -m torch.distributed.run --nproc-per-node {n} bench.py --gpus {n} --executor spmd")
    os.environ.setdefault("VLLM_LOGGING_LEVEL", "INFO" if rank == 0 else "WARNING")

    if backend == "distributed":
        # control plane on loopback with a one-off secret; the data plane rendezvous is made by the head
        os.environ["VLLM_HOST_IP"] = "127.0.0.1"
        os.environ["VLLM_RPC_BIND"] = "127.0.0.1"
        if launched:
            port, secret = _torchrun_rendezvous(rank)
            os.environ["VLLM_SERVER_PORT"], os.environ["VLLM_RPC_SECRET"] = str(port), secret
            os.environ.setdefault("VLLM_REMOTE_RETRY_SECONDS", "0.5")
            local_rank = int(os.environ.get("LOCAL_RANK", rank))
            # the launcher's own rendezvous variables must not leak into the workers' data-plane
            # rendezvous (the head makes that one): with the launcher's store flag set, nobody
            # would host the tcp:// store the workers are told to meet at
            for k in [k for k in os.environ if k.startswith("TORCHELASTIC_")] + [
                    "RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                    "MASTER_ADDR", "MASTER_PORT"]:
                os.environ.pop(k, None)
            if rank > 0:
                # this rank is a worker node offering one device, like `launch.py remote 127.0.0.1`
                from vllm_distributed_amd.launch import remote_worker_main
                remote_worker_main("127.0.0.1", n - 1, local_rank, "launched-workers", port)
                return                                       # not reached: the worker leaves through os._exit
            os.environ["VLLM_HEAD_LOCAL_DEVICES"] = "1"       # the head's own worker takes device 0
        else:
            from vllm_distributed_amd.utils import get_open_port
            os.environ["VLLM_SERVER_PORT"] = str(get_open_port())
            os.environ["VLLM_RPC_SECRET"] = secrets.token_hex(16)
            os.environ["VLLM_HEAD_LOCAL_DEVICES"] = str(n)
        os.environ.setdefault("VLLM_WORKER_REGISTRATION_TIMEOUT", "600")
    run_head(args, n, rank, backend, launched)


def run_head(args, n: int, rank: int, backend: str, launched: bool) -> None:
    import numpy as np

    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.sampling_params import SamplingParams

    spmd = backend == "spmd" and n > 1
    on_gpu = args.device == "cuda"
    tp, pp = _split(args.parallelism, n)
    if spmd and pp > 1:
        raise SystemExit("--executor spmd runs tensor parallelism only")
    if n > 1 and not spmd:
        os.environ.setdefault("VDAMD_TIME_COLLECTIVES", "0")

    # prefill budget scales with N so the batch is prefilled in the same number of steps at every N
    shards = n * args.simulate_tp
    # (a pipeline stage is not sharded by PP: its prefill chunk is sized by the TP degree alone)
    prefill_budget = args.prefill_budget_per_gpu * (shards if args.parallelism == "tp" else _split(args.parallelism, n)[0])
    per_gpu = _fit_batch_per_gpu(args, query_device=backend != "distributed")
    batch = per_gpu * shards
    strong = min(args.strong_batch, batch) if args.strong_batch > 0 else 0
    # continuous batching staggers the sequences by the number of prefill steps
    prefill_steps = -(-batch * args.prompt_len // max(prefill_budget, batch))
    total_new = args.warmup + args.steps + prefill_steps + 10
    max_len = args.prompt_len + total_new + 16
    # under PP the running sequences are dealt out over PP micro-batches: those are the sizes decode runs at
    sizes = sorted({-(-s // pp) for s in (strong, batch) if s > 0} | ({s for s in (strong, batch) if s > 0} if pp == 1 else set()))
    engine = LLMEngine.from_engine_args(EngineArgs(
        model=args.model, dtype=args.dtype, load_format="dummy", tensor_parallel_size=tp, pipeline_parallel_size=pp,
        device=args.device,
        quantization=args.quantization, kv_cache_dtype=args.kv_cache_dtype,
        distributed_executor_backend=backend, max_model_len=max_len, max_num_seqs=batch,
        max_num_batched_tokens=max(prefill_budget, batch), enable_prefix_caching=False,
        enforce_eager=args.enforce_eager, max_capture_size=max(sizes), hip_graph_sizes=sizes,
        gpu_memory_utilization=args.gpu_memory_utilization,
        num_gpu_blocks_override=args.num_gpu_blocks_override, seed=0, async_scheduling=not args.sync_scheduling))
    executor = engine.core.executor
    vocab = engine.model_config.vocab_size
    rng = np.random.default_rng(1234)
    # the KV cache the workers could actually allocate has the last word on the batch
    blocks_per_seq = -(-(args.prompt_len + total_new) // engine.vllm_config.cache_config.block_size)
    fit = engine.vllm_config.cache_config.num_gpu_blocks // blocks_per_seq
    if fit < batch:
        batch = max(8 * n, fit // (8 * n) * (8 * n))

    def sync() -> None:
        """Every worker has finished what was queued on its device (and, by the
        call returning, every rank has been heard from)."""
        executor.collective_rpc("synchronize")
        if spmd:
            import torch.distributed as dist
            dist.barrier()

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
        if i > 0:                      # first one pays lazy initialisation
            ttfts.append(time.perf_counter() - t0)
        while engine.has_unfinished_requests():
            engine.step()
    ttft_p50_ms = 1e3 * float(np.median(ttfts)) if ttfts else None

    sched = engine.core.scheduler

    def measure(nseq: int, tag: str) -> dict:
        """Fill a batch of ``nseq`` sequences, prefill it, warm up, time K decode steps."""
        sp = SamplingParams(temperature=0.0, max_tokens=total_new, ignore_eos=True, detokenize=False)
        rids = [engine.add_request(f"{tag}-{i}", rng.integers(0, vocab, size=args.prompt_len).tolist(), sp)
                for i in range(nseq)]
        sync()
        t_prefill0 = time.perf_counter()
        while sched.waiting or any(r.num_computed_tokens < r.num_prompt_tokens for r in sched.running):
            engine.step()
        sync()
        prefill_s = time.perf_counter() - t_prefill0
        if not args.sync_scheduling or pp > 1:
            for _ in range(max(1, pp - 1) if pp > 1 else 1):
                engine.step()      # fill the pipeline: from here every call returns one batch's tokens
        for _ in range(args.warmup):
            engine.step()
        produced = 0
        sync()
        engine.core.timer.reset()
        t0 = time.perf_counter()
        if pp == 1:
            for _ in range(args.steps):
                outs = engine.step()
                produced += sum(len(o.new_token_ids) for o in outs)
        else:
            # one "step" = every sequence gains one token = PP micro-batches through the pipeline
            while produced < nseq * args.steps:
                produced += sum(len(o.new_token_ids) for o in engine.step())
        sync()
        elapsed = time.perf_counter() - t0
        assert produced >= nseq * args.steps, (produced, nseq, args.steps)
        steps_done = produced / nseq
        if spmd:                   # every rank timed its own copy of the loop: report the slowest
            import torch
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device="cuda" if on_gpu and dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        breakdown = {k: round(v, 3) for k, v in engine.core.timer.summary().items()}
        engine.abort_request(rids)
        while engine.has_unfinished_requests():
            engine.step()
        sync()
        return {"global_batch": nseq, "tokens_per_s": round(produced / elapsed, 2),
                "ms_per_step": round(1e3 * elapsed / steps_done, 3),
                "prefill_tokens_per_s": round(nseq * args.prompt_len / prefill_s, 1),
                "step_breakdown_ms": breakdown}

    weak = measure(batch, "b")
    strong_res = measure(strong, "s") if strong and strong != batch else None

    stats = executor.collective_rpc("get_stats") if rank == 0 else []
    engine.shutdown()
    if rank == 0:
        alt = None
        if n > 1 and not spmd and args.alt_parallelism.lower() not in ("none", args.parallelism.lower()) \
                and args.simulate_tp == 1:
            alt = _measure_alt_layout(args, n)
        s0 = stats[0] if stats else {}
        dt = {"bfloat16": "bf16", "float16": "fp16", "float32": "fp32"}[args.dtype]
        if args.quantization:
            dt = f"{args.quantization}-weights/{dt}"
        if args.kv_cache_dtype not in ("auto", args.dtype):
            dt = f"{dt}/kv-{args.kv_cache_dtype}"
        pretty = {"llama-3-70b": "Llama-3-70B", "llama-3-8b": "Llama-3-8B"}.get(args.model, args.model)
        bd = weak["step_breakdown_ms"]
        config = {"model": args.model, "global_batch": batch, "seq_len": args.prompt_len,
                  "parallelism": f"tp{tp}" if pp == 1 else (f"pp{pp}" if tp == 1 else f"tp{tp}xpp{pp}"),
                  "executor": backend,
                  "launch": ("torch.distributed.run: head + %d registered workers" % (n - 1)
                             if backend == "distributed" and launched
                             else "head spawns every worker" if backend == "distributed" else backend),
                  "async_scheduling": not args.sync_scheduling,
                  "hip_graph": bool(s0.get("hip_graph_sizes")), "hip_graph_sizes": s0.get("hip_graph_sizes"),
                  "hip_graph_fallback": s0.get("hip_graph_fallback"),
                  "tp_collectives": {"backend": s0.get("tp_backend"), "direct_rccl": s0.get("tp_rccl"),
                                     "direct_all_reduce": s0.get("tp_direct_all_reduce")},
                  "block_size": s0.get("block_size"), "step_breakdown_ms": bd}
        print(json.dumps({
            "metric": (f"output tokens/sec (whole node), {pretty} {dt} "
                       f"{'TP' if pp == 1 else ('PP' if tp == 1 else 'TPxPP')} over N GPUs, decode"
                       if args.simulate_tp == 1 else
                       f"NOT A BENCHMARK: per-rank compute proxy, one GPU simulating rank 0 of TP={args.simulate_tp}"),
            "value": weak["tokens_per_s"], "unit": "tokens/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": weak["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt,
            "data": "synthetic prompts (random token ids), random-init weights",
            "ttft_p50_ms": None if ttft_p50_ms is None else round(ttft_p50_ms, 2),
            "prefill_tokens_per_s": weak["prefill_tokens_per_s"],
            "rpc_overhead_ms": bd.get("rpc_overhead_ms"),
            "strong_scaling": None if strong_res is None else dict(strong_res, scaling="strong"),
            "alt_parallelism": alt,
            "config": config,
        }), flush=True)


if __name__ == "__main__":
    main()
