"""Sparse-MoE layer latency: grouped HIP GEMMs over expert-sorted rows against
the per-expert library-GEMM loop, for Qwen3-Coder-480B-A35B (TP=2 shard: 160
experts, hidden 6144, expert width 1280, top-8) and Mixtral-8x7B (TP=1) shapes,
dense bf16 and AWQ 4-bit experts.

    python benchmarks/moe_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd.models.layers.linear import QuantSpec, quantized_linears   # noqa: E402
from vllm_distributed_amd.models.layers.moe import FusedMoE                          # noqa: E402
from vllm_distributed_amd.parallel import state as pstate                            # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def build(hidden, inter, experts, top_k, quantized):
    with torch.device(DEV), quantized_linears(QuantSpec("awq", 128) if quantized else None):
        moe = FusedMoE(hidden, inter, experts, top_k, True, torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for name, p in moe.named_parameters():
        if p.dtype == torch.int32:
            p.data.copy_(torch.randint(-(1 << 31), (1 << 31) - 1, p.shape, device=DEV, generator=gen,
                                       dtype=torch.int64).to(torch.int32))
        elif "scales" in name:
            p.data.uniform_(0.002, 0.004)
        else:
            p.data.normal_(0.0, 0.03)
    moe.finalize_weights()
    return moe


def main():
    pstate.ensure_single_process_groups(torch.device(DEV))
    print(f"{'model':22s} {'weights':8s} {'T':>5s} {'grouped us':>11s} {'loop us':>10s} {'speedup':>8s} {'GB/s (grouped)':>15s}")
    for name, (h, i, e, k) in (("qwen3-coder-480b tp2", (6144, 1280, 160, 8)), ("mixtral-8x7b tp1", (4096, 14336, 8, 2))):
        for quantized in (False, True):
            moe = build(h, i, e, k, quantized)
            wbytes = e * 3 * h * i * (0.5 if quantized else 2.0)
            for t in (1, 16, 64, 256, 1024):
                x = torch.randn(t, h, device=DEV).to(torch.bfloat16)
                with torch.inference_mode():
                    FusedMoE._grouped_tile_rows = _orig
                    rows = moe._grouped_tile_rows(x)
                    tg = timeit(lambda: moe(x)) if rows else float("nan")
                    FusedMoE._grouped_tile_rows = lambda self, x: 0
                    tl = timeit(lambda: moe(x), iters=3)
                # experts touched: at most min(E, T*k)
                touched = min(e, t * k) / e
                print(f"{name:22s} {'awq4' if quantized else 'bf16':8s} {t:5d} {tg:11.1f} {tl:10.1f} {tl / tg:8.2f} "
                      f"{wbytes * touched / tg / 1e3:15.0f}", flush=True)
            del moe
            torch.cuda.empty_cache()


_orig = FusedMoE._grouped_tile_rows
if __name__ == "__main__":
    main()
