"""Probe: fused 4-bit GEMMs over (variant, K split) for small batches, timed
inside a captured graph (kernel + split-K reduce), for the attention projections
of Qwen3-Coder-480B and Llama-3-70B."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops                      # noqa: E402
from vllm_distributed_amd.ops import quant                # noqa: E402

assert ops.hip_available()
from vllm_distributed_amd.ops import hip                   # noqa: E402


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


for name, n, k in (("480b qkv", 14336, 6144), ("480b o", 6144, 12288), ("70b qkv", 10240, 8192), ("70b o", 8192, 8192),
                   ("70b gate_up", 57344, 8192), ("70b down", 8192, 28672)):
    gen = torch.Generator(device="cuda").manual_seed(0)
    # several copies so the weights stream from HBM, not the last-level cache
    copies = []
    for c in range(max(2, min(6, int(1.5e9 // (n * k // 2))))):
        q = torch.randint(0, 16, (n, k), device="cuda", generator=gen, dtype=torch.uint8)
        z = torch.randint(0, 16, (n, k // 128), device="cuda", generator=gen).float()
        s = torch.rand((n, k // 128), device="cuda", generator=gen) * 0.01 + 0.005
        copies.append(quant.pack_w4(q, z, s, torch.bfloat16))
        del q
    for m in (1, 16, 32):
        x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
        res = []
        for variant in (0, 1):
            for ks in (1, 2, 3, 4, 6, 8):
                if ks > k // 128 // 4:
                    continue
                state = {"i": 0}

                def call():
                    qw, sz = copies[state["i"] % len(copies)]
                    state["i"] += 1
                    hip.w4a16_gemm(x, qw, sz, variant=variant, ksplit=ks)
                res.append((timeit(call), variant, ks))
        res.sort()
        av = hip.w4_choose_variant(m, n, k)
        ak = hip.w4_choose_ksplit(m, n, k, av)
        auto = [t for t, v, ks in res if v == av and ks == min(ak, 8)]
        print(f"{name:12s} M={m:3d} best " + ", ".join(f"v{v} ks{ks} {t:.1f}us" for t, v, ks in res[:4])
              + f"   auto v{av} ks{ak}" + (f" {auto[0]:.1f}us" if auto else ""), flush=True)
    del copies
