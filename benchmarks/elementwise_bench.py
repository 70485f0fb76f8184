"""Bandwidth of the small decode kernels (fused add + RMSNorm, rotary + KV-cache
write, SiLU-and-multiply) at Llama-3-70B / 8B decode shapes, timed inside a
captured graph over rotating buffers (so the data comes from HBM)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd.ops import hip  # noqa: E402

hip.load()
dev = "cuda"
COPIES = 6


def timeit(fn, iters=24):
    for i in range(3):
        fn(i)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def run(t, h, f, hq, hkv, d=128, bs=64):
    w = torch.ones(h, device=dev, dtype=torch.bfloat16)
    xs = [torch.randn(t, h, device=dev, dtype=torch.bfloat16) for _ in range(COPIES)]
    rs = [torch.randn(t, h, device=dev, dtype=torch.bfloat16) for _ in range(COPIES)]
    us = timeit(lambda i: hip.fused_add_rms_norm(xs[i % COPIES], rs[i % COPIES], w, 1e-5))
    print(f"T={t:5d} fused_add_rms_norm H={h}: {us:7.1f} us  {4 * t * h * 2 / us / 1e6:5.2f} TB/s", flush=True)
    gu = [torch.randn(t, 2 * f, device=dev, dtype=torch.bfloat16) for _ in range(COPIES)]
    us = timeit(lambda i: hip.silu_and_mul(gu[i % COPIES]))
    print(f"T={t:5d} silu_and_mul F={f}: {us:7.1f} us  {3 * t * f * 2 / us / 1e6:5.2f} TB/s", flush=True)
    del gu
    qkv = [torch.randn(t, (hq + 2 * hkv) * d, device=dev, dtype=torch.bfloat16) for _ in range(COPIES)]
    nblk = max((t + bs - 1) // bs * 4, t)          # the decode-like pattern below needs a block per token
    kc = torch.zeros(nblk, hkv, bs, d, device=dev, dtype=torch.bfloat16)
    vc = torch.zeros(nblk, hkv, d // 16, bs // 8, 16, 8, device=dev, dtype=torch.bfloat16)
    pos = torch.randint(0, 4096, (t,), device=dev, dtype=torch.int32)
    slots = (torch.randperm(nblk * bs, device=dev)[:t]).to(torch.int32)
    inv = 1.0 / (500000.0 ** (torch.arange(0, d, 2, device=dev).float() / d))
    ang = torch.arange(8192, device=dev).float()[:, None] * inv[None]
    cs = torch.cat([ang.cos(), ang.sin()], -1).contiguous()
    us = timeit(lambda i: hip.rope_and_cache(pos, qkv[i % COPIES], cs, kc, vc, slots, hq, hkv, d))
    by = t * ((hq + 2 * hkv) * d + (hq + hkv) * d + 2 * hkv * d) * 2
    print(f"T={t:5d} rope_and_cache Hq={hq} Hkv={hkv}: {us:7.1f} us  {by / us / 1e6:5.2f} TB/s", flush=True)
    # a decode step: every sequence writes at (nearly) the same offset inside its own block
    blocks = torch.randperm(nblk, device=dev)[:t]
    for off, label in ((37, "same in-block offset"), (None, "in-block offset spread over 16")):
        o = 37 if off is not None else (torch.arange(t, device=dev) % 16) + 30
        dslots = (blocks * bs + o).to(torch.int32)
        us = timeit(lambda i: hip.rope_and_cache(pos, qkv[i % COPIES], cs, kc, vc, dslots, hq, hkv, d))
        print(f"T={t:5d} rope_and_cache, {label}: {us:7.1f} us", flush=True)


for t in (256, 2048, 8192):
    run(t, 8192, 28672, 64, 8)
run(256, 4096, 14336, 32, 8)
run(2048, 4096, 14336, 32, 8)
