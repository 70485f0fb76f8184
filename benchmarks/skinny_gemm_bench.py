"""Short-batch GEMM: the weight-streaming kernel (csrc/skinny_gemm.hip) against
the library GEMM (F.linear -> hipBLASLt) on Llama-3-70B shapes.

    python benchmarks/skinny_gemm_bench.py [--sweep]

Weights rotate over several copies so they stream from HBM rather than the
256 MiB last-level cache.  --sweep also times every (ntw, ksplit) choice.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops                      # noqa: E402

SHAPES = [("tp1 qkv", 10240, 8192), ("tp1 o", 8192, 8192), ("tp1 gate_up", 57344, 8192),
          ("tp1 down", 8192, 28672), ("tp1 lm_head", 128256, 8192), ("tp8 qkv", 1280, 8192),
          ("tp8 o", 8192, 1024), ("tp8 gate_up", 7168, 8192), ("tp8 down", 8192, 3584)]


SHAPES_8B = [("8b qkv", 6144, 4096), ("8b o", 4096, 4096), ("8b gate_up", 28672, 4096), ("8b down", 4096, 14336)]


def timeit(fn, ws, iters=30):
    """Per-call time inside a captured graph (how decode steps run): no launch gaps."""
    for i in range(3):
        fn(ws[i % len(ws)])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(ws[i % len(ws)])
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--shapes", choices=["70b", "8b"], default="70b")
    args = ap.parse_args()
    assert ops.hip_available()
    from vllm_distributed_amd.ops import hip
    print(f"{'shape':14s} {'M':>4s} {'skinny us':>10s} {'TB/s':>6s} {'lib us':>9s} {'TB/s':>6s} {'speedup':>8s}  choice")
    for name, n, k in (SHAPES if args.shapes == "70b" else SHAPES_8B):
        nw = max(2, min(8, int(3e9 // (n * k * 2))))
        ws = [torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(nw)]
        for m in (1, 16, 32, 64, 128):
            x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
            ts = timeit(lambda w: hip.skinny_gemm(x, w), ws)
            tl = timeit(lambda w: F.linear(x, w), ws)
            bw = n * k * 2 / 1e6
            print(f"{name:14s} {m:4d} {ts:10.1f} {bw / ts:6.2f} {tl:9.1f} {bw / tl:6.2f} {tl / ts:8.2f}  "
                  f"{hip.skinny_choose(m, n, k)}", flush=True)
            if args.sweep:
                best = []
                for ntw in (1, 2, 4):
                    if m > 64 and ntw == 4:
                        continue
                    for ksplit in (1, 2, 4, 8, 16):
                        if ksplit > k // 512:
                            continue
                        t = timeit(lambda w: hip.skinny_gemm(x, w, ntw, ksplit), ws, iters=10)
                        best.append((t, ntw, ksplit))
                best.sort()
                print("      sweep best:", ", ".join(f"ntw{a} ks{b} {t:.1f}us" for t, a, b in best[:4]), flush=True)
        del ws


if __name__ == "__main__":
    main()
