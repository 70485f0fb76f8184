"""One call of the LDS-staged skinny GEMM on an LM-head shape at two rows (guarded check of a shape the
sweep stopped at)."""
import torch
from vllm_distributed_amd.ops import hip
hip.load()
n, k = 128256, 8192
w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
for m in (1, 2):
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    print("launch", m, hip.skinny_choose(m, n, k), flush=True)
    y = hip.skinny_gemm(x, w)
    torch.cuda.synchronize()
    ref = torch.nn.functional.linear(x.float(), w.float())
    print("done", m, (y.float() - ref).abs().max().item(), flush=True)
