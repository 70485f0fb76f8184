# one decode-attention shape, a few launches: the target of a rocprofv3 counter pass
import math, sys, torch
from vllm_distributed_amd.ops import hip
hip.load()
B, ctx = int(sys.argv[1]), int(sys.argv[2])
dev, hq, hkv, d, bs = "cuda", 64, 8, 128, 64
torch.manual_seed(0)
lens = ((torch.rand(B) * 0.5 + 0.75) * ctx).clamp(min=1).int()
nblk = int((int(lens.max()) + bs - 1) // bs)
total = B * nblk
kc = (torch.randn(total, hkv, bs, d, device=dev) * 0.5).to(torch.bfloat16)
vc = (torch.randn(total, hkv, d, bs, device=dev) * 0.5).to(torch.bfloat16)
bt = torch.randperm(total, device=dev).view(B, nblk).to(torch.int32)
sl = lens.to(dev)
q = torch.randn(B, hq, d, device=dev).to(torch.bfloat16)
for _ in range(5):
    hip.paged_attention_decode(q, kc, vc, bt, sl, 1 / math.sqrt(d), max_seq_len=int(lens.max()), num_splits=1)
torch.cuda.synchronize()
print("ideal bytes", 2 * int(lens.sum()) * hkv * d * 2)
