import sys, torch
sys.path.insert(0, "/root/repo")
from vllm_distributed_amd.ops import hip
hip.load()
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
m = 2048
for n in (8192, 57344):
    for k in (128, 256, 512, 1024, 2048, 4096, 8192):
        x = torch.randn((m, k), dtype=torch.bfloat16, device="cuda")
        w = torch.randn((n, k), dtype=torch.bfloat16, device="cuda") * 0.02
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        t = min(timeit(lambda: hip.big_gemm(x, w, out=out, variant=0)) for _ in range(3))
        tl = min(timeit(lambda: torch.mm(x, w.t(), out=out)) for _ in range(3))
        print(f"N={n} K={k:5d}: big {t:8.1f} us  lib {tl:8.1f} us", flush=True)
