# decode-attention bandwidth: bf16 vs e4m3 cache, 70B-shaped heads (Hq 64, Hkv 8, D 128), block 64
import math, sys, torch
from vllm_distributed_amd.ops import hip
hip.load()
dev = "cuda"
hq, hkv, d, bs = 64, 8, 128, 64
def run(B, ctx, fp8):
    nblk = (ctx + bs - 1) // bs
    total = B * nblk
    dt = torch.float8_e4m3fn if fp8 else torch.bfloat16
    kc = (torch.randn(total, hkv, bs, d, device=dev) * 0.5).to(dt)
    vc = (torch.randn(total, hkv, d, bs, device=dev) * 0.5).to(dt)
    bt = torch.randperm(total, device=dev).view(B, nblk).to(torch.int32)
    sl = torch.full((B,), ctx, device=dev, dtype=torch.int32)
    q = torch.randn(B, hq, d, device=dev).to(torch.bfloat16)
    ws = None
    for _ in range(3):
        hip.paged_attention_decode(q, kc, vc, bt, sl, 1 / math.sqrt(d), max_seq_len=ctx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        hip.paged_attention_decode(q, kc, vc, bt, sl, 1 / math.sqrt(d), max_seq_len=ctx)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byts = 2 * B * ctx * hkv * d * (1 if fp8 else 2)
    return ms, byts / ms / 1e9
for B, ctx in [(64, 8192), (256, 2048), (16, 8192), (256, 512), (2048, 256), (8, 32768)]:
    a = run(B, ctx, False); b = run(B, ctx, True)
    print(f"B={B:5d} ctx={ctx:6d}  bf16 {a[0]*1e3:8.1f} us {a[1]:6.2f} TB/s | e4m3 {b[0]*1e3:8.1f} us {b[1]:6.2f} TB/s | time x{b[0]/a[0]:.2f}", flush=True)
