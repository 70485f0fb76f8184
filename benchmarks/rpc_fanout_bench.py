"""Control-plane and head-side cost of one engine step, with no model behind it.

    python benchmarks/rpc_fanout_bench.py --world 8 --batches 256 2048 16384 --steps 40

The head spawns ``world`` stub workers (pipes) or, with ``--remote``, keeps one and takes the others as
``launch.py remote`` processes over loopback TCP - the two transports of the reference
(``src/launch.py:157,255``).  Reported per batch size: p50 / p99 of the blocking ``execute_model`` round
trip (pickle once, fan out, worker thread hop, reply from the output rank), and the head's own
``schedule`` / ``update_from_output`` / output-processing time per step.  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--batches", type=int, nargs="+", default=[256, 2048, 16384])
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--remote", action="store_true", help="one spawned worker + (world-1) TCP workers")
    ap.add_argument("--pp", type=int, default=1)
    ap.add_argument("--sync-scheduling", action="store_true")
    args = ap.parse_args()
    import numpy as np

    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.utils import get_open_port

    port = get_open_port()
    os.environ.update(VLLM_SERVER_PORT=str(port), VLLM_RPC_SECRET="bench-secret", VLLM_HOST_IP="127.0.0.1",
                      VLLM_RPC_BIND="127.0.0.1", VLLM_LOGGING_LEVEL="WARNING", VLLM_REMOTE_RETRY_SECONDS="0.3",
                      VLLM_HEAD_LOCAL_DEVICES="1" if args.remote else str(args.world),
                      PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "benchmarks"), os.environ.get("PYTHONPATH", "")]))
    remote = None
    if args.remote and args.world > 1:
        remote = subprocess.Popen([sys.executable, os.path.join(ROOT, "launch.py"), "remote", "127.0.0.1",
                                   "--num-devices", str(args.world - 1), "--device-offset", "1"], env=dict(os.environ))
    big = max(args.batches)
    eng = LLMEngine.from_engine_args(EngineArgs(
        model="llama-tiny", device="cpu", dtype="float32", max_model_len=512, block_size=64,
        num_gpu_blocks_override=big * 8 + 64, max_num_seqs=big, max_num_batched_tokens=max(big * 4, 8192),
        tensor_parallel_size=args.world // args.pp, pipeline_parallel_size=args.pp,
        distributed_executor_backend="distributed", worker_cls="fast_stub_worker.FastStubWorker",
        enable_prefix_caching=False, async_scheduling=not args.sync_scheduling,
        hf_overrides={"num_attention_heads": 8, "num_key_value_heads": 8, "head_dim": 16}))
    out = {"world": args.world, "transport": "tcp-loopback" if args.remote else "unix-socketpair", "pp": args.pp,
           "async_scheduling": not args.sync_scheduling, "results": []}
    try:
        for b in args.batches:
            sp = SamplingParams(temperature=0.0, max_tokens=args.steps + 40, ignore_eos=True, detokenize=False)
            rids = [eng.add_request(f"b{b}-{i}", [1, 2, 3, 4], sp) for i in range(b)]
            for _ in range(6):
                eng.step()
            timer = eng.core.timer
            timer.reset()
            ex = eng.core.executor
            rtts, walls, procs = [], [], []
            orig = ex.execute_model

            def timed_execute(so, non_block=False):
                if non_block:
                    return orig(so, non_block=True)
                t0 = time.perf_counter()
                r = orig(so)
                rtts.append((time.perf_counter() - t0) * 1e3)
                return r
            ex.execute_model = timed_execute
            produced = 0
            t_all = time.perf_counter()
            for _ in range(args.steps * args.pp):
                t0 = time.perf_counter()
                outs = eng.step()
                walls.append((time.perf_counter() - t0) * 1e3)
                produced += sum(len(o.new_token_ids) for o in outs)
            total = time.perf_counter() - t_all
            ex.execute_model = orig
            s = timer.summary()
            row = {"batch": b, "tokens_per_step_call": produced / max(1, len(walls)),
                   "step_wall_ms_p50": round(float(np.percentile(walls, 50)), 3),
                   "step_wall_ms_p99": round(float(np.percentile(walls, 99)), 3),
                   "head_tokens_per_s_ceiling": round(produced / total, 1),
                   "schedule_ms": round(s.get("schedule_ms", 0.0), 3), "update_ms": round(s.get("update_ms", 0.0), 3),
                   "execute_blocked_ms": round(s.get("execute_ms", 0.0), 3), "launch_ms": round(s.get("launch_ms", 0.0), 3),
                   "rpc_overhead_ms": s.get("rpc_overhead_ms")}
            if rtts:
                row.update(execute_rtt_ms_p50=round(float(np.percentile(rtts, 50)), 3),
                           execute_rtt_ms_p99=round(float(np.percentile(rtts, 99)), 3))
            out["results"].append(row)
            eng.abort_request(rids)
            while eng.has_unfinished_requests():
                eng.step()
        # bare round trip of a call every worker answers (no scheduler output in it)
        ex = eng.core.executor
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            ex.collective_rpc("check_health")
            lat.append((time.perf_counter() - t0) * 1e3)
        out["noop_all_reply_ms_p50"] = round(float(np.percentile(lat, 50)), 3)
        out["noop_all_reply_ms_p99"] = round(float(np.percentile(lat, 99)), 3)
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            ex.collective_rpc("check_health", unique_reply_rank=ex.output_rank)
            lat.append((time.perf_counter() - t0) * 1e3)
        out["noop_one_reply_ms_p50"] = round(float(np.percentile(lat, 50)), 3)
        out["noop_one_reply_ms_p99"] = round(float(np.percentile(lat, 99)), 3)
    finally:
        eng.shutdown()
        if remote is not None:
            remote.terminate()
            remote.wait(timeout=10)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
