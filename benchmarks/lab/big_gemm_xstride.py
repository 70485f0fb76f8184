"""Does the row stride of the activation operand matter to the large-batch GEMM?  x [2048, K] with rows
K, K + 64 and K + 128 elements apart (weights always padded by 128), own kernel and library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_distributed_amd.ops import hip  # noqa: E402

hip.load()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


m = 2048
for name, n, k in (("qkv", 10240, 8192), ("o", 8192, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)):
    wb = torch.empty((n, k + 128), dtype=torch.bfloat16, device="cuda")
    w = wb[:, :k]
    w.normal_(0, 0.02)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    line = f"{name:8s}"
    for pad in (0, 64, 128):
        xb = torch.randn((m, k + pad), dtype=torch.bfloat16, device="cuda")
        x = xb[:, :k]
        tb = min(timeit(lambda: hip.big_gemm(x, w, out=out)) for _ in range(3))
        tl = min(timeit(lambda: torch.mm(x, w.t(), out=out)) for _ in range(3))
        line += f" | x pad {pad:3d}: big {tb:7.1f} us lib {tl:7.1f} us"
    print(line, flush=True)
