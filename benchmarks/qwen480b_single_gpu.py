"""Qwen3-Coder-480B-A35B (the reference's documented model, which it spreads over
TP=2 x PP=2) as 4-bit AWQ weights on ONE MI355X: 288 GB of HBM hold the ~250 GB of
packed experts plus a KV cache.  Random weights of the real shapes; reports load
time, memory and decode speed.

    python benchmarks/qwen480b_single_gpu.py [--layers 62] [--batch 16] [--steps 32] [--quantization awq|nvfp4]

``--model gpt-oss-120b --quantization none`` runs another bundled sparse architecture the same way (gpt-oss:
experts widened to bf16, 36 layers = 218 GiB of weights); ``--model deepseek-v2-lite --quantization none
--prompt-len 8192 --max-model-len 16384`` a multi-head-latent-attention model at a context where the cache form
matters (``VDAMD_MLA_LATENT=0`` for the expanded per-head cache).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd.engine.arg_utils import EngineArgs       # noqa: E402
from vllm_distributed_amd.engine.llm_engine import LLMEngine       # noqa: E402
from vllm_distributed_amd.sampling_params import SamplingParams    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="qwen3-coder-480b-a35b")
ap.add_argument("--layers", type=int, default=0, help="0: as many as the architecture has")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--prompt-len", type=int, default=128)
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--quantization", default="awq")
ap.add_argument("--gpu-memory-utilization", type=float, default=0.96)
ap.add_argument("--max-batched-tokens", type=int, default=4096)
ap.add_argument("--max-model-len", type=int, default=4096)
args = ap.parse_args()

t0 = time.time()
eng = LLMEngine.from_engine_args(EngineArgs(
    model=args.model, load_format="dummy", quantization=None if args.quantization == "none" else args.quantization,
    hf_overrides={"num_hidden_layers": args.layers} if args.layers else {}, max_model_len=args.max_model_len, max_num_seqs=max(args.batch, 8),
    max_num_batched_tokens=args.max_batched_tokens, gpu_memory_utilization=args.gpu_memory_utilization, enable_prefix_caching=False,
    hip_graph_sizes=[args.batch], max_capture_size=args.batch, async_scheduling=True))
load_s = time.time() - t0
print(f"engine ready in {load_s:.1f}s; torch allocated {torch.cuda.memory_allocated() / 2 ** 30:.1f} GiB, "
      f"reserved {torch.cuda.memory_reserved() / 2 ** 30:.1f} GiB", flush=True)
rng = np.random.default_rng(0)
sp = SamplingParams(temperature=0.0, max_tokens=args.steps + 8, ignore_eos=True, detokenize=False)
t1 = time.time()
for i in range(args.batch):
    eng.add_request(f"r{i}", rng.integers(0, min(150000, eng.model_config.vocab_size), size=args.prompt_len).tolist(), sp)
sched = eng.core.scheduler
while sched.waiting or any(r.num_computed_tokens < r.num_prompt_tokens for r in sched.running):
    eng.step()
for _ in range(3):
    eng.step()
torch.cuda.synchronize()
prefill_s = time.time() - t1
t2 = time.perf_counter()
for _ in range(args.steps):
    eng.step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t2) / args.steps * 1e3
print(json.dumps({"model": args.model, "layers": eng.core.vllm_config.model_config.num_hidden_layers, "quantization": args.quantization,
                  "gpus": 1, "batch": args.batch, "prompt_len": args.prompt_len, "load_s": round(load_s, 1),
                  "prefill_s": round(prefill_s, 2), "decode_ms_per_step": round(ms, 2),
                  "decode_tokens_per_s": round(args.batch / ms * 1e3, 1),
                  "hbm_allocated_gib": round(torch.cuda.memory_allocated() / 2 ** 30, 1),
                  "kv_cache_tokens": int(sched.kv.pool.num_blocks * sched.kv.block_size),
                  "mla_latent_cache": os.environ.get("VDAMD_MLA_LATENT", "1") != "0"}), flush=True)
eng.shutdown()
