# one mid_gemm configuration, a few launches: the target of a rocprofv3 counter pass
import sys, torch
from vllm_distributed_amd.ops import hip
hip.load()
m, n, k, bm, bn, ks, var = [int(v) for v in sys.argv[1:8]]
x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
buf = torch.empty(n, k + 128, device="cuda", dtype=torch.bfloat16)
buf[:, :k] = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
w = buf[:, :k]
for _ in range(5):
    hip.mid_gemm(x, w, bm=bm, bn=bn, ksplit=ks, variant=var)
torch.cuda.synchronize()
