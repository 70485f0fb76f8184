"""Probe: library bf16 GEMM throughput by operand layout for the Llama-3-70B projections at
decode batch 2048 (one GPU) and 16384 (TP=8 shard shapes): x @ W^T with W [N, K] (the stored
layout), x @ Wt with Wt [K, N], and the swapped product W @ x^T.  In-graph timing."""
import torch
import torch.nn.functional as F


def timeit(fn, iters=10):
    for _ in range(2):
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


cases = [("qkv", 2048, 10240, 8192), ("o", 2048, 8192, 8192), ("gate_up", 2048, 57344, 8192), ("down", 2048, 8192, 28672),
         ("tp8 qkv", 16384, 1280, 8192), ("tp8 o", 16384, 8192, 1024), ("tp8 gate_up", 16384, 7168, 8192),
         ("tp8 down", 16384, 8192, 3584)]
for name, m, n, k in cases:
    x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02
    wt = w.t().contiguous()
    xt = x.t().contiguous()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    outt = torch.empty(n, m, device="cuda", dtype=torch.bfloat16)
    res = {
        "x@W^T (stored)": timeit(lambda: F.linear(x, w)),
        "x@Wt [K,N]": timeit(lambda: torch.mm(x, wt, out=out)),
        "W@x^T -> [N,M]": timeit(lambda: torch.mm(w, x.t(), out=outt)),
        "W@xt(contig)": timeit(lambda: torch.mm(w, xt, out=outt)),
    }
    fl = 2.0 * m * n * k
    print(f"{name:12s} M={m:5d} N={n:5d} K={k:5d}  " + "  ".join(f"{k_}: {v:7.1f}us {fl / v / 1e9:5.2f}PF" for k_, v in res.items()),
          flush=True)
