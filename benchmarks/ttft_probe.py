"""Single-request prefill latency (time to first token) on an idle engine.
Run under ``rocprofv3 --kernel-trace`` and feed the trace to step_breakdown.py
to see the kernels of the last request's prefill step.

    python benchmarks/ttft_probe.py [model] [prompt_len] [requests]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vllm_distributed_amd.engine.arg_utils import EngineArgs  # noqa: E402
from vllm_distributed_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_distributed_amd.sampling_params import SamplingParams  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "llama-3-70b"
plen = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nreq = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng = LLMEngine.from_engine_args(EngineArgs(
    model=model, load_format="dummy", max_model_len=plen + 64, max_num_seqs=64, max_num_batched_tokens=16384,
    enable_prefix_caching=False, hip_graph_sizes=[1, 8], async_scheduling=True, gpu_memory_utilization=0.6))
rng = np.random.default_rng(0)
vocab = eng.model_config.vocab_size
for i in range(nreq):
    ids = rng.integers(0, vocab, size=plen).tolist()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rid = eng.add_request(f"t{i}", ids, SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True,
                                                       detokenize=False))
    got = False
    while not got:
        for out in eng.step():
            got = got or out.request_id == rid
    torch.cuda.synchronize()
    print(f"ttft[{i}] = {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    while eng.has_unfinished_requests():
        eng.step()
eng.shutdown()
