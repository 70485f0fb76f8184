"""cProfile of the host side of decode steps (scheduler + runner launch path) at a large batch."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from vllm_distributed_amd.engine.arg_utils import EngineArgs  # noqa: E402
from vllm_distributed_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_distributed_amd.sampling_params import SamplingParams  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
model = sys.argv[2] if len(sys.argv) > 2 else "llama-3-8b"
eng = LLMEngine.from_engine_args(EngineArgs(
    model=model, load_format="dummy", max_model_len=256, max_num_seqs=batch, max_num_batched_tokens=16384,
    enable_prefix_caching=False, hip_graph_sizes=[batch], max_capture_size=batch, async_scheduling=True))
rng = np.random.default_rng(0)
detok = os.environ.get("DETOK") == "1"
sp = SamplingParams(temperature=0.0, max_tokens=100, ignore_eos=True, detokenize=detok)
for i in range(batch):
    eng.add_request(f"r{i}", rng.integers(0, 1000, size=64).tolist(), sp)
sched = eng.core.scheduler
while sched.waiting or any(r.num_computed_tokens < r.num_prompt_tokens for r in sched.running):
    eng.step()
for _ in range(3):
    eng.step()
import time  # noqa: E402
import torch  # noqa: E402
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.step()
torch.cuda.synchronize()
print(f"batch {batch} detokenize={detok}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step wall (unprofiled)")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    eng.step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
