"""Weight-streaming GEMM sweep: achieved HBM bandwidth of the M<=16 kernel per (shape, M, config)
against the library GEMM and the LDS-staged skinny kernel, timed inside captured hipGraphs
(20 launches per replay over rotating weight copies so W never comes from the 256 MB cache).

    PYTHONPATH=. python benchmarks/stream_gemm_sweep.py [--full]
"""
import argparse
import sys

import torch
import torch.nn.functional as F

from vllm_distributed_amd.ops import hip

hip.load()
DEV = "cuda"
SHAPES = {  # name: (N, K)
    "70b.qkv": (10240, 8192), "70b.o": (8192, 8192), "70b.gate_up": (57344, 8192), "70b.down": (8192, 28672),
    "70b.lm_head": (128256, 8192),
    "8b.qkv": (6144, 4096), "8b.o": (4096, 4096), "8b.gate_up": (28672, 4096), "8b.down": (4096, 14336),
    "70b/tp8.qkv": (1280, 8192), "70b/tp8.o": (8192, 1024), "70b/tp8.gate_up": (7168, 8192),
    "70b/tp8.down": (8192, 3584),
}


def timed(fn, copies, reps=3):
    """us per call: a graph of len(copies) calls, each on its own weight copy."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for w in copies[:2]:
            fn(w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for w in copies:
                fn(w)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(reps):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / len(copies))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--shapes", type=str, default="")
    ap.add_argument("--ms", type=str, default="1,8,16")
    ap.add_argument("--variant", type=int, default=-1, help="mid_gemm variant to sweep (-1: both with --full)")
    ap.add_argument("--pad", type=int, default=0, help="row padding of W in elements (W is a [:, :K] view)")
    args = ap.parse_args()
    names = [s for s in args.shapes.split(",") if s] or list(SHAPES)
    for name in names:
        n, k = SHAPES[name]
        nbytes = n * k * 2
        ncopies = max(2, min(12, (600 << 20) // nbytes + 1))
        copies = []
        for _ in range(ncopies):
            buf = torch.empty(n, k + args.pad, device=DEV, dtype=torch.bfloat16)
            buf[:, :k] = (torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16)
            copies.append(buf[:, :k])
        for m in [int(v) for v in args.ms.split(",")]:
            x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
            ref = F.linear(x.float(), copies[0].float())
            res = {}
            res["lib"] = timed(lambda w: F.linear(x, w), copies)
            if m <= 128 and hip.skinny_supported(x, copies[0]):
                res["skinny"] = timed(lambda w: hip.skinny_gemm(x, w), copies)
            cfgs = [(None, None, None, False)]
            if m > 16:
                cfgs = []
            if args.full and m <= 16:
                cfgs = [(ntw, ks, d, nt) for ntw in (1, 2) for ks in (1, 2, 4, 8) for d in (2, 3, 4)
                        for nt in (False, True) if ks <= k // 128 // 4]
            best = None
            for ntw, ks, d, nt in cfgs:
                got = hip.stream_gemm(x, copies[0], ntw=ntw, ksplit=ks, depth=d, nt=nt)
                err = (got.float() - ref).abs().max().item()
                assert err < 0.05 * max(1.0, ref.abs().max().item()), (name, m, ntw, ks, d, nt, err)
                t = timed(lambda w: hip.stream_gemm(x, w, ntw=ntw, ksplit=ks, depth=d, nt=nt), copies)
                if best is None or t < best[0]:
                    best = (t, ntw, ks, d, nt)
                if args.full:
                    print(f"    {name} M={m} ntw={ntw} ks={ks} d={d} nt={int(nt)}: {t:7.1f} us "
                          f"{nbytes / t / 1e6:5.2f} TB/s", flush=True)
            if best is not None:
                res["stream"] = best[0]
            mbest = None
            if m > 16 and hip.mid_supported(x, copies[0]):
                mcfgs = [(None, None, None)]
                if args.full:
                    mcfgs = [(bm, bn, ks) for bm, bn in ((256, 256), (256, 128), (128, 256), (128, 128), (64, 256), (64, 512))
                             for ks in (1, 2, 4, 8, 16) if bm >= min(m, 256) or (bm == 128 and m > 128)]
                if args.variant >= 0:
                    mcfgs = [c + (args.variant,) for c in mcfgs]
                elif args.full:
                    mcfgs = [c + (v,) for c in mcfgs for v in (0, 1) if not (v == 1 and c[1] > 256)]
                else:
                    mcfgs = [c + (None,) for c in mcfgs]
                for bm, bn, ks, var in mcfgs:
                    got = hip.mid_gemm(x, copies[0], bm=bm, bn=bn, ksplit=ks, variant=var)
                    err = (got.float() - ref).abs().max().item()
                    assert err < 0.05 * max(1.0, ref.abs().max().item()), (name, m, bm, bn, ks, var, err)
                    t = timed(lambda w: hip.mid_gemm(x, w, bm=bm, bn=bn, ksplit=ks, variant=var), copies)
                    if mbest is None or t < mbest[0]:
                        mbest = (t, bm, bn, ks, var)
                    if args.full:
                        print(f"    {name} M={m} bm={bm} bn={bn} ks={ks} v={var}: {t:7.1f} us {nbytes / t / 1e6:5.2f} TB/s "
                              f"{2 * m * n * k / t / 1e6:6.0f} TF", flush=True)
                res["mid"] = mbest[0]
            line = f"{name:16s} N={n:6d} K={k:5d} M={m:2d} | " + " | ".join(
                f"{kk} {v:7.1f} us {nbytes / v / 1e6:5.2f} TB/s" for kk, v in res.items())
            if args.full and best is not None:
                line += f" | best ntw={best[1]} ks={best[2]} d={best[3]} nt={int(best[4])}"
            elif best is not None:
                line += f" | cfg {hip.stream_choose(n, k)}"
            if mbest is not None:
                line += f" | mid cfg {mbest[1:] if args.full else hip.mid_choose(m, n, k)}"
            print(line, flush=True)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    sys.exit(main())
