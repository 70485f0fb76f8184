"""qkv projection at M=2048 (70B: N = 8192 + 2048, K = 8192): one library GEMM against the q part on the
256x256-tile kernel (256 tiles = one round) plus the kv part on the library, both writing into one buffer."""
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


m, n, k, nq = 2048, 10240, 8192, 8192
wb = torch.empty((n, k + 128), dtype=torch.bfloat16, device="cuda")
w = wb[:, :k]
w.normal_(0, 0.02)
x = torch.randn((m, k), dtype=torch.bfloat16, device="cuda")
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
ref = torch.mm(x, w.t())


def split():
    hip.big_gemm(x, w[:nq], out=out[:, :nq])
    torch.mm(x, w[nq:].t(), out=out[:, nq:])


def split_big():
    hip.big_gemm(x, w[:nq], out=out[:, :nq])
    hip.big_gemm(x, w[nq:], out=out[:, nq:])


split()
print("max diff split vs whole", (out.float() - ref.float()).abs().max().item())
for name, fn in (("library, whole", lambda: torch.mm(x, w.t(), out=out)), ("big q + library kv", split),
                 ("big q + big kv", split_big), ("big q only", lambda: hip.big_gemm(x, w[:nq], out=out[:, :nq])),
                 ("library kv only (strided out)", lambda: torch.mm(x, w[nq:].t(), out=out[:, nq:])),
                 ("big whole", lambda: hip.big_gemm(x, w, out=out))):
    print(f"{name:32s} {min(timeit(fn) for _ in range(3)):8.1f} us", flush=True)
