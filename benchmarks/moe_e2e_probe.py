"""End-to-end decode of a cut-down Qwen3-Coder-480B-A35B (same layer shapes,
fewer layers / experts so it fits a quick run): checks that decode steps with
sparse-MoE layers are captured in hipGraphs and reports tokens/s.

    python benchmarks/moe_e2e_probe.py [awq]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from vllm_distributed_amd.engine.arg_utils import EngineArgs  # noqa: E402
from vllm_distributed_amd.engine.llm_engine import LLMEngine  # noqa: E402
from vllm_distributed_amd.sampling_params import SamplingParams  # noqa: E402

quant = "awq" if len(sys.argv) > 1 and sys.argv[1] == "awq" else None
batch, steps = 64, 32
eng = LLMEngine.from_engine_args(EngineArgs(
    model="qwen3-coder-480b-a35b", load_format="dummy", quantization=quant, max_model_len=512,
    hf_overrides={"num_hidden_layers": 4, "num_experts": 64}, max_num_seqs=batch, hip_graph_sizes=[batch],
    max_capture_size=batch, gpu_memory_utilization=0.5, async_scheduling=True, enable_prefix_caching=False))
runner = eng.core.executor.driver.worker.model_runner
print("hip graphs captured:", sorted(runner.graphs), flush=True)
rng = np.random.default_rng(0)
sp = SamplingParams(temperature=0.0, max_tokens=steps + 8, ignore_eos=True, detokenize=False)
for i in range(batch):
    eng.add_request(f"r{i}", rng.integers(0, 1000, size=64).tolist(), sp)
sched = eng.core.scheduler
while sched.waiting or any(r.num_computed_tokens < r.num_prompt_tokens for r in sched.running):
    eng.step()
for _ in range(4):
    eng.step()
import torch  # noqa: E402
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
for _ in range(steps):
    n += sum(len(o.new_token_ids) for o in eng.step())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"quant={quant} batch={batch}: {1e3 * dt / steps:.2f} ms/step, {n / dt:.0f} tok/s", flush=True)
eng.shutdown()
