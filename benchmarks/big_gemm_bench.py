"""Large-batch GEMM: the 256 x 256 LDS-DMA kernel (csrc/big_gemm.hip) against the library GEMM on the
decoder-layer shapes, numerics against an fp32 product, random operands, padded weight rows.

    python benchmarks/big_gemm_bench.py [--m 2048] [--iters 20] [--check]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd.ops import hip  # noqa: E402

SHAPES = {  # name: (N, K)
    "70b qkv": (10240, 8192), "70b o": (8192, 8192), "70b gate_up": (57344, 8192), "70b down": (8192, 28672),
    "70b lm_head": (128256, 8192), "8b qkv": (6144, 4096), "8b gate_up": (28672, 4096), "8b down": (4096, 14336),
    "70b/tp8 gate_up": (7168, 8192), "70b/tp8 down": (8192, 3584),
}


def padded(n, k, pad=128):
    buf = torch.empty((n, k + pad), dtype=torch.bfloat16, device="cuda")
    w = buf[:, :k]
    w.normal_(0, 0.02)
    return w


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def check(m, n, k, silu=False, res=False):
    x = torch.randn((m, k), dtype=torch.bfloat16, device="cuda")
    w = padded(n, k)
    r = torch.randn((m, n), dtype=torch.bfloat16, device="cuda") if res else None
    y = hip.big_gemm(x, w, residual=r, silu_mul=silu)
    ref = x.float() @ w.float().t()
    if res:
        ref = ref + r.float()
    if silu:
        g, u = ref[:, :n // 2], ref[:, n // 2:]
        ref = torch.nn.functional.silu(g) * u
    err = (y.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    return err, scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="*", default=[2048])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--shapes", type=str, nargs="*", default=None)
    ap.add_argument("--variants", type=int, nargs="*", default=[0])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--fused", action="store_true")
    args = ap.parse_args()
    hip.load()
    torch.manual_seed(0)
    if args.check:
        for (m, n, k, silu, res) in [(256, 256, 128, False, False), (256, 256, 1024, False, False),
                                     (512, 512, 256, False, False), (300, 336, 384, False, False),
                                     (2048, 1024, 8192, False, False), (1000, 2048, 2048, False, True),
                                     (777, 1024, 1024, True, False), (2048, 7168, 8192, True, False),
                                     (2048, 8192, 3584, False, True)]:
            err, scale = check(m, n, k, silu, res)
            print(f"check m={m} n={n} k={k} silu={silu} res={res}: max err {err:.4g} (max |ref| {scale:.4g})",
                  flush=True)
            assert err <= 2e-2 * max(scale, 1.0), "numerics"
    for m in args.m:
        x = torch.randn((m, 28672), dtype=torch.bfloat16, device="cuda")
        for name, (n, k) in SHAPES.items():
            if args.shapes and not any(s in name for s in args.shapes):
                continue
            w = padded(n, k)
            xk = x[:, :k]
            out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
            fl = 2.0 * m * n * k
            fns = {"lib": lambda: torch.mm(xk, w.t(), out=out)}
            for v in args.variants:
                fns[f"v{v}"] = (lambda v=v: hip.big_gemm(xk, w, out=out, variant=v))
            best = {kk: 1e30 for kk in fns}
            for _ in range(args.rounds):
                for kk, fn in fns.items():
                    best[kk] = min(best[kk], timeit(fn, args.iters))
            line = f"M={m:5d} {name:16s} N={n:6d} K={k:5d} " + " | ".join(
                f"{kk} {tt:7.1f} us {fl / tt / 1e6:6.0f} TF" for kk, tt in best.items())
            if "gate_up" in name and args.fused:
                act = torch.empty((m, n // 2), dtype=torch.bfloat16, device="cuda")
                from vllm_distributed_amd import ops
                t_f = timeit(lambda: hip.big_gemm(xk, w, out=act, silu_mul=True), args.iters)
                t_u = timeit(lambda: ops.silu_and_mul(torch.mm(xk, w.t(), out=out)), args.iters)
                line += f" | fused silu {t_f:8.1f} us vs lib+act {t_u:8.1f} us ({t_u / t_f:.2f}x)"
            print(line, flush=True)
            del w, out


if __name__ == "__main__":
    main()
