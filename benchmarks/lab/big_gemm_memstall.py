"""How much of the large-batch GEMM's time is memory stall?  The same launch with row strides of zero
(every tile row reads the same 16 KB line set: all loads hit L2 after the first touch) against the real one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vllm_distributed_amd.ops import hip  # noqa: E402

lib = hip.load()


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
st = torch.cuda.current_stream().cuda_stream
for name, n, k in (("o", 8192, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)):
    wb = torch.empty((n, k + 128), dtype=torch.bfloat16, device="cuda")
    w = wb[:, :k]
    w.normal_(0, 0.02)
    x = torch.randn((m, k), dtype=torch.bfloat16, device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")

    def run(xs, ws):
        rc = lib.vdamd_big_gemm(out.data_ptr(), out.stride(0), x.data_ptr(), xs, w.data_ptr(), ws, None, 0,
                                m, n, k, 0, 0, 0, st)
        assert rc == 0, rc
    t_real = min(timeit(lambda: run(x.stride(0), w.stride(0))) for _ in range(3))
    t_x0 = min(timeit(lambda: run(0, w.stride(0))) for _ in range(3))
    t_w0 = min(timeit(lambda: run(x.stride(0), 0)) for _ in range(3))
    t_00 = min(timeit(lambda: run(0, 0)) for _ in range(3))
    print(f"{name:8s} real {t_real:7.1f} us | x rows aliased {t_x0:7.1f} | w rows aliased {t_w0:7.1f} | "
          f"both aliased {t_00:7.1f} us", flush=True)
