"""Check: the reference's documented flags (--max-model-len 262144, Qwen3-Coder-480B shapes, AWQ) on a
cut-down layer count: decode graphs capture, memory for the per-request tables stays small, and short
and long contexts both decode.  Prints the time per decode step at a short and at a 100k-token context."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import LLM, SamplingParams  # noqa: E402

llm = LLM(model="qwen3-coder-480b-a35b", load_format="dummy", quantization="awq", max_model_len=262144,
          hf_overrides={"num_hidden_layers": 4}, gpu_memory_utilization=0.5, max_num_seqs=64,
          max_num_batched_tokens=8192)
runner = llm.engine.core.executor.driver.worker.model_runner
print("graphs:", runner.graph_sizes, "workspace MiB:", runner.decode_workspace.numel() * 4 >> 20,
      "torch allocated GiB:", round(torch.cuda.memory_allocated() / 2**30, 1), flush=True)
for ctx in (128, 100_000):
    t0 = time.perf_counter()
    out = llm.generate([{"prompt_token_ids": [7] * ctx}], SamplingParams(temperature=0.0, max_tokens=33, ignore_eos=True))[0]
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    ttft = out.metrics.first_token_time - out.metrics.arrival_time
    print(f"ctx {ctx}: ttft {ttft * 1e3:.1f} ms, decode {(total - ttft) / 32 * 1e3:.2f} ms/token (4 layers)", flush=True)
llm.shutdown()
