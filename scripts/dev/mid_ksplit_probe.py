"""mid_gemm at K-split counts that are not powers of two: the 70B shapes at 256 and 128 rows, every (bm, bn, ksplit)
that fills the chip differently, timed over rotating weight buffers.  usage: mid_ksplit_probe.py [rows ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from vllm_distributed_amd.ops import hip  # noqa: E402

hip.load()
torch.manual_seed(0)
rows = [int(a) for a in sys.argv[1:]] or [256, 128]
shapes = {"70b.qkv": (10240, 8192), "70b.o": (8192, 8192), "70b.down": (8192, 28672),
          "70b/tp8.qkv": (1280, 8192), "70b/tp8.o": (8192, 1024), "70b/tp8.gate_up": (7168, 8192),
          "70b/tp8.down": (8192, 3584), "70b/tp2.qkv": (5120, 8192), "70b/tp2.o": (8192, 4096),
          "70b/tp2.down": (8192, 14336), "8b.qkv": (6144, 4096), "8b.o": (4096, 4096), "8b.down": (4096, 14336)}


def timeit(fn, iters=24):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for name, (n, k) in shapes.items():
    ws = []
    for _ in range(4):
        buf = torch.empty(n, k + 128, device="cuda", dtype=torch.bfloat16)
        buf[:, :k].normal_(0, 0.02)
        ws.append(buf[:, :k])
    for m in rows:
        x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        chosen = hip.mid_choose(m, n, k)
        res = []
        for bm in (256, 128):
            if bm < min(m, 256):
                continue
            for bn in (128, 256):
                for ks in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
                    if k // 64 // ks < 4:
                        continue
                    try:
                        # partial sums only (what the serving path consumes) when split, the plain output otherwise
                        if ks > 1:
                            us = timeit(lambda i: hip.mid_gemm(x, ws[i % 4], bm=bm, bn=bn, ksplit=ks, partials=True))
                        else:
                            us = timeit(lambda i: hip.mid_gemm(x, ws[i % 4], bm=bm, bn=bn, ksplit=ks))
                    except Exception as e:  # noqa: BLE001
                        continue
                    res.append((us, bm, bn, ks))
        res.sort()
        cur = [r for r in res if r[1:] == chosen]
        print(f"{name} M={m}: chosen {chosen} {cur[0][0] if cur else float('nan'):.1f} us | best "
              + "  ".join(f"({b},{c},{d}) {a:.1f}" for a, b, c, d in res[:6]), flush=True)
