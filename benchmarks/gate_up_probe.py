"""Probe: the 70B gate_up shape (N = 57344, K = 8192; 940 MB of weights) at 16 < M <= 512 through the
library, the split-K register-ring kernel (mid_gemm, K unsplit) and the 256 x 256 LDS-DMA kernel with its
SiLU epilogue.  Random operands, 10 launches per graph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops                      # noqa: E402
from vllm_distributed_amd.ops import hip                   # noqa: E402

assert ops.hip_available()


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


n, k = int(os.environ.get("N", 57344)), int(os.environ.get("K", 8192))
pad = 128
wfull = torch.randn(n, k + pad, device="cuda", dtype=torch.bfloat16) * 0.02
w = wfull[:, :k]                                          # padded rows, as the model runner stores them
# a second, different weight so consecutive launches do not find W in the Infinity Cache
for m in (32, 64, 128, 192, 256, 384, 512):
    x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    row = [f"M={m:4d}"]
    t = timeit(lambda: torch.nn.functional.linear(x, w))
    t2 = timeit(lambda: ops.silu_and_mul(torch.nn.functional.linear(x, w)))
    row.append(f"lib {t:7.1f} us ({n * k * 2 / t / 1e6:.2f} TB/s, {2 * m * n * k / t / 1e6:.0f} TF) +silu {t2:7.1f}")
    if m <= 256:
        for bm, bn in ((256, 256), (256, 128), (128, 256), (128, 128), (64, 256)):
            if bm < min(m, 256):
                continue
            try:
                tm = timeit(lambda: hip.mid_gemm(x, w, ksplit=1, bm=bm, bn=bn))
                row.append(f"mid{bm}x{bn} {tm:7.1f}")
            except Exception as e:  # noqa: BLE001
                row.append(f"mid{bm}x{bn} ERR {str(e)[:30]}")
    if hip.big_supported(x, w, True):
        try:
            tb = timeit(lambda: hip.big_gemm(x, w, silu_mul=True))
            row.append(f"big+silu {tb:7.1f}")
        except Exception as e:  # noqa: BLE001
            row.append(f"big ERR {str(e)[:40]}")
    print("  ".join(row), flush=True)
