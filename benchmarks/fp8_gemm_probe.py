"""torch._scaled_mm (hipBLASLt fp8, OCP e4m3) on Llama-3-70B shapes: availability of per-tensor and
row-wise scaling on this stack and throughput against the bf16 GEMM."""
import torch
import torch.nn.functional as F


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


f8 = torch.float8_e4m3fn
for m in (16, 256, 2048, 8192):
    for name, n, k in (("qkv", 10240, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)):
        x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02
        xq, wq = x.to(f8), w.to(f8)
        one = torch.ones((), device="cuda", dtype=torch.float32)
        res = {}
        res["bf16"] = timeit(lambda: F.linear(x, w))
        try:
            res["fp8 tensor"] = timeit(lambda: torch._scaled_mm(xq, wq.t(), scale_a=one, scale_b=one,
                                                                 out_dtype=torch.bfloat16))
        except Exception as e:  # noqa: BLE001
            res["fp8 tensor"] = float("nan")
            print("per-tensor failed:", repr(e)[:200])
        try:
            sa = torch.ones(m, 1, device="cuda", dtype=torch.float32)
            sb = torch.ones(1, n, device="cuda", dtype=torch.float32)
            y = torch._scaled_mm(xq, wq.t(), scale_a=sa, scale_b=sb, out_dtype=torch.bfloat16)
            res["fp8 rowwise"] = timeit(lambda: torch._scaled_mm(xq, wq.t(), scale_a=sa, scale_b=sb,
                                                                  out_dtype=torch.bfloat16))
            ref = xq.float() @ wq.float().t()
            res["err"] = ((y.float() - ref).abs().max() / ref.abs().max()).item()
        except Exception as e:  # noqa: BLE001
            res["fp8 rowwise"] = float("nan")
            print("rowwise failed:", repr(e)[:200])
        fl = 2.0 * m * n * k
        print(f"M={m:5d} {name:8s} " + "  ".join(
            f"{k2}: {v:8.1f}us {fl / v / 1e6:6.0f}TF" if k2 != "err" else f"relerr {v:.4f}" for k2, v in res.items()),
            flush=True)
