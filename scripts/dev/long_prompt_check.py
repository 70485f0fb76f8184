"""One long prompt through a full-size model with random weights: the first generated tokens and the top log-probabilities
of the first one.  Run once as is and once with VDAMD_PREFILL_LDS=0: the LDS-staged (global -> LDS DMA) and the
fragment-direct prefill kernels must agree to bf16 noise.  usage: long_prompt_check.py [model] [prompt_len]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from vllm_distributed_amd import LLM, SamplingParams  # noqa: E402

if __name__ == "__main__":
    model = sys.argv[1] if len(sys.argv) > 1 else "llama-3-8b"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
    llm = LLM(model=model, load_format="dummy", max_model_len=n + 64, max_num_batched_tokens=4096,
              gpu_memory_utilization=0.5, enable_prefix_caching=False)
    ids = np.random.default_rng(0).integers(0, 32000, size=n).tolist()
    out = llm.generate([{"prompt_token_ids": ids}], SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True,
                                                                   logprobs=5))[0]
    lp = out.outputs[0].logprobs[0]
    print("TOKENS", out.outputs[0].token_ids)
    print("TOP", sorted(((round(v, 3), k) for k, v in lp.items()), reverse=True)[:5])
    llm.shutdown()
