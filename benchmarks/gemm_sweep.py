"""hipBLASLt (torch F.linear) throughput for the Llama-3-70B projection shapes
at decode batch sizes: which per-GPU batch makes the GEMMs efficient."""
import sys
import torch
import torch.nn.functional as F

def bench(m, n, k, iters=20):
    x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    # rotate over several weight copies so the weights stream from HBM, not the 256 MiB LLC
    nw = max(1, min(8, int(2e9 // (n * k * 2))))
    ws = [torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(nw)]
    for i in range(3):
        F.linear(x, ws[i % nw])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        F.linear(x, ws[i % nw])
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    return us, 2.0 * m * n * k / us / 1e6, (n * k * 2) / us / 1e6

tp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, FF, V = 8192, 28672, 128256
shapes = {"qkv": (10240 // tp, H), "o": (H, H // tp), "gate_up": (2 * FF // tp, H), "down": (H, FF // tp)}
ms = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 512, 1024, 2048, 4096, 8192]
for m in ms:
    tot_us, tot_fl = 0.0, 0.0
    line = []
    for name, (n, k) in shapes.items():
        us, tf, tbs = bench(m, n, k)
        tot_us += us
        tot_fl += 2.0 * m * n * k
        line.append(f"{name}: {us:7.1f}us {tf:6.0f}TF {tbs:4.2f}TB/s")
    print(f"TP={tp} M={m:5d} layer {tot_us:8.1f}us  {tot_fl / tot_us / 1e6:6.0f} TF/s | " + " | ".join(line), flush=True)
