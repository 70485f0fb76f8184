"""W4A16 GEMM micro-benchmark: the fused 4-bit MFMA kernel against the bf16
library GEMM on the same logical shape, small batches (decode regime).

    python benchmarks/w4_gemm_bench.py [--iters 50]

Reports time, effective weight bandwidth (4-bit bytes / time) and the speed-up
over bf16 F.linear.  Shapes: Llama-3-70B per-rank matrices at TP=8 and TP=1,
Qwen3-Coder-480B attention / expert matrices at TP=2.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops                      # noqa: E402
from vllm_distributed_amd.ops import quant                # noqa: E402

SHAPES = [            # (name, N, K)
    ("70b tp8 qkv", 1280, 8192), ("70b tp8 o", 8192, 1024), ("70b tp8 gate_up", 7168, 8192),
    ("70b tp8 down", 8192, 3584), ("70b tp1 qkv", 10240, 8192), ("70b tp1 gate_up", 57344, 8192),
    ("70b tp1 down", 8192, 28672), ("qwen480b tp2 qkv", 5120, 6144), ("qwen480b tp2 expert w13", 2560, 6144),
    ("qwen480b tp2 expert w2", 6144, 1280),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--format", choices=["int4", "fp4"], default="int4",
                    help="int4: AWQ-style groups of 128; fp4: NVFP4 (E2M1, one scale per 16 columns)")
    args = ap.parse_args()
    assert ops.hip_available()
    from vllm_distributed_amd.ops import hip
    dev = "cuda"
    print(f"{'shape':26s} {'M':>4s} {'v0 us':>9s} {'v1 us':>9s} {'best GB/s':>9s} {'bf16 us':>9s} {'speedup':>8s} {'auto':>5s}")
    for name, n, k in SHAPES:
        gen = torch.Generator(device=dev).manual_seed(0)
        # pack on the GPU in row chunks (the packer's int64 temporaries are 8x the weights)
        q = torch.randint(0, 16, (n, k), device=dev, generator=gen, dtype=torch.uint8)
        z = torch.randint(0, 16, (n, k // 128), device=dev, generator=gen).float()
        s = torch.rand((n, k // 128), device=dev, generator=gen) * 0.01 + 0.005
        if args.format == "fp4":
            packed = torch.randint(0, 256, (n, k // 2), device=dev, generator=gen, dtype=torch.int32).to(torch.uint8)
            bs = (0.5 + torch.rand((n, k // 16), device=dev, generator=gen)).to(torch.float8_e4m3fn)
            qw, sz = quant.pack_fp4(packed, bs)
            alpha = torch.full((n,), 0.01, device=dev)
            w = hip.fp4_dequant(qw, sz, alpha, torch.bfloat16)
            gemm = lambda x, v: hip.fp4a16_gemm(x, qw, sz, alpha, variant=v)      # noqa: E731
            dequant = lambda: hip.fp4_dequant(qw, sz, alpha, torch.bfloat16, w)  # noqa: E731
        else:
            qw, sz = quant.pack_w4(q, z, s, torch.bfloat16)
            w = hip.w4_dequant(qw, sz)
            gemm = lambda x, v: hip.w4a16_gemm(x, qw, sz, variant=v)             # noqa: E731
            dequant = lambda: hip.w4_dequant(qw, sz, w)                          # noqa: E731
        del q
        for m in (1, 16, 32, 64, 128, 256):
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            t0 = timeit(lambda: gemm(x, 0), args.iters)
            t1 = timeit(lambda: gemm(x, 1), args.iters)
            tb = timeit(lambda: torch.nn.functional.linear(x, w), args.iters)
            t4 = min(t0, t1)
            gbs = (n * k / 2) / (t4 * 1e-6) / 1e9
            print(f"{name:26s} {m:4d} {t0:9.1f} {t1:9.1f} {gbs:9.0f} {tb:9.1f} {tb / t4:8.2f} "
                  f"{hip.w4_choose_variant(m, n, k):5d}", flush=True)
        td = timeit(dequant, args.iters)
        print(f"{name:26s} dequant to bf16: {td:.1f} us ({n * k * 2.5 / (td * 1e-6) / 1e9:.0f} GB/s moved)")
        del qw, sz, w


if __name__ == "__main__":
    main()
