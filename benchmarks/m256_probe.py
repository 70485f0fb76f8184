"""Probe: M in (128, 256] through the weight-streaming kernel as row tiles of one
group (grouped mode) against the library GEMM, Llama-3-8B and 70B/TP8 shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops                      # noqa: E402

assert ops.hip_available()
from vllm_distributed_amd.ops import hip                   # noqa: E402

SHAPES = [("8b qkv", 6144, 4096), ("8b o", 4096, 4096), ("8b gate_up", 28672, 4096), ("8b down", 4096, 14336),
          ("70b/8 qkv", 1280, 8192), ("70b/8 o", 8192, 1024), ("70b/8 gate_up", 7168, 8192), ("70b/8 down", 8192, 3584)]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, n, k in SHAPES:
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02
    w3 = w.view(1, n, k)
    for m in (128, 192, 256, 384, 512):
        x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
        ref = torch.nn.functional.linear(x, w)
        t_lib = timeit(lambda: torch.nn.functional.linear(x, w))
        best = None
        for rows in (64, 128):
            tiles = torch.tensor([[0, r, min(rows, m - r)] for r in range(0, m, rows)], dtype=torch.int32, device="cuda")
            y = hip.grouped_gemm(x, w3, tiles, rows)
            err = (y.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
            t = timeit(lambda: hip.grouped_gemm(x, w3, tiles, rows))
            if best is None or t < best[0]:
                best = (t, rows, err)
        gbs = n * k * 2 / (min(best[0], t_lib) * 1e-6) / 1e12
        print(f"{name:14s} M={m:4d} lib {t_lib:7.1f} us  tiles {best[0]:7.1f} us (rows {best[1]}, err {best[2]:.4f})  "
              f"ratio {t_lib / best[0]:.2f}  best {gbs:.2f} TB/s", flush=True)
