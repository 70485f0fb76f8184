"""One engine of a disaggregated prefill / decode pair, run as a child process by the tests:

    python tests/kv_handover_proc.py <role> <mailbox dir> [device]

``role``: ``reference`` (no connector), ``producer`` (prefills, hands the KV over, then serves commands
read from stdin: ``reap`` runs another request so that finished hand-overs are collected, ``exit``),
``consumer`` (loads the KV and generates).  Prints ``RESULT {json}`` lines.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PROMPTS = [list(range(3, 3 + 70)), [9, 8, 7] * 15, [4, 5]]           # 4, 2 and 0 whole 16-token blocks


def main():
    role, path = sys.argv[1], sys.argv[2]
    device = sys.argv[3] if len(sys.argv) > 3 else "cuda"
    connector = os.environ.get("KV_CONNECTOR", "DeviceIpcConnector")
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    kv = None
    if role != "reference":
        kv = {"kv_connector": connector, "kv_role": "kv_producer" if role == "producer" else "kv_consumer",
              "kv_connector_extra_config": {"shared_storage_path": path}}
    eng = LLMEngine.from_engine_args(EngineArgs(
        model="llama-tiny", device=device, dtype="float32" if device == "cpu" else "bfloat16", max_model_len=128,
        block_size=16, num_gpu_blocks_override=64, max_num_seqs=8, kv_transfer_config=kv,
        enable_prefix_caching=False, gpu_memory_utilization=0.2))

    def run(prompts, max_tokens):
        sp = SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
        ids = [eng.add_request(None, p, sp) for p in prompts]
        final = {}
        while eng.has_unfinished_requests():
            for o in eng.step():
                if o.finished:
                    final[o.request_id] = o.outputs[0].token_ids
        return [final[i] for i in ids]

    def conn():
        return eng.core.executor.driver.worker.model_runner.kv_connector

    def report(tokens):
        c = conn() if kv is not None else None
        sched = eng.core.scheduler.kv_connector if kv is not None else None
        print("RESULT " + json.dumps({
            "tokens": tokens,
            "device_bytes_sent": getattr(c, "device_bytes_sent", 0),
            "device_bytes_received": getattr(c, "device_bytes_received", 0),
            "exported_bytes": getattr(c, "_exported_bytes", 0),
            "num_loaded_tokens": getattr(sched, "num_loaded_tokens", 0),
            "num_saved_requests": getattr(sched, "num_saved_requests", 0)}), flush=True)

    if role == "producer":
        report(run(PROMPTS, 1))
        for line in sys.stdin:
            cmd = line.strip()
            if cmd == "reap":
                import time
                run([[11, 12, 13]], 2)
                deadline = time.time() + 10
                while conn()._exported_bytes and time.time() < deadline:      # the side thread collects on a poll
                    run([[11, 12, 13]], 2)
                    time.sleep(0.1)
                report([])
            elif cmd == "exit":
                break
    else:
        report(run(PROMPTS, 6))
    eng.shutdown()


if __name__ == "__main__":
    main()
