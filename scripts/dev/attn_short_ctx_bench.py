# decode attention at short contexts: waves per sequence 4 (num_splits=1), 2 (num_splits=-2), 1 (num_splits=0);
# 70B-shaped heads (Hq 64, Hkv 8, D 128), block 64, contexts jittered +-25 %
import math, torch
from vllm_distributed_amd.ops import hip
from vllm_distributed_amd import ops
hip.load()
dev = "cuda"
hq, hkv, d, bs = 64, 8, 128, 64
def run(B, ctx, splits):
    torch.manual_seed(0)
    lens = ((torch.rand(B) * 0.5 + 0.75) * ctx).clamp(min=1).int()
    nblk = int((int(lens.max()) + bs - 1) // bs)
    total = B * nblk
    kc = (torch.randn(ops.k_cache_shape(total, hkv, d, bs), device=dev) * 0.5).to(torch.bfloat16)
    vc = (torch.randn(ops.v_cache_shape(total, hkv, d, bs), device=dev) * 0.5).to(torch.bfloat16)
    bt = torch.randperm(total, device=dev).view(B, nblk).to(torch.int32)
    sl = lens.to(dev)
    q = torch.randn(B, hq, d, device=dev).to(torch.bfloat16)
    f = lambda: hip.paged_attention_decode(q, kc, vc, bt, sl, 1 / math.sqrt(d), max_seq_len=int(lens.max()), num_splits=splits)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byts = 2 * int(lens.sum()) * hkv * d * 2
    return ms, byts / ms / 1e9, f()
for B, ctx in [(2048, 64), (2048, 128), (2048, 160), (2048, 256), (2048, 512), (2048, 1024), (512, 160), (512, 1024), (256, 160), (4096, 160), (64, 512)]:
    a = run(B, ctx, 1); b = run(B, ctx, -2); c = run(B, ctx, 0); au = run(B, ctx, None)
    err = (a[2].float() - b[2].float()).abs().max().item()
    print(f"B={B:5d} ctx~{ctx:5d}  4-wave {a[0]*1e3:7.1f} us {a[1]:5.2f} TB/s | 2-wave {b[0]*1e3:7.1f} us {b[1]:5.2f} TB/s | 1-wave {c[0]*1e3:7.1f} us {c[1]:5.2f} TB/s | auto {au[0]*1e3:7.1f} us | max diff {err:.4f}", flush=True)
