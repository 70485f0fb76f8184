"""One shape of the fused 4-bit GEMMs (variant 1) for counter collection:
    rocprofv3 --pmc ... -- python3 benchmarks/w4_pmc_target.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops                      # noqa: E402
from vllm_distributed_amd.ops import quant                # noqa: E402

assert ops.hip_available()
from vllm_distributed_amd.ops import hip                   # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, k = 57344, 8192
gen = torch.Generator(device="cuda").manual_seed(0)
packed = torch.randint(0, 256, (n, k // 2), device="cuda", generator=gen, dtype=torch.int32).to(torch.uint8)
bs = (0.5 + torch.rand((n, k // 16), device="cuda", generator=gen)).to(torch.float8_e4m3fn)
fq, fs = quant.pack_fp4(packed, bs)
q = torch.randint(0, 16, (n, k), device="cuda", generator=gen, dtype=torch.uint8)
z = torch.randint(0, 16, (n, k // 128), device="cuda", generator=gen).float()
s = torch.rand((n, k // 128), device="cuda", generator=gen) * 0.01 + 0.005
qw, sz = quant.pack_w4(q, z, s, torch.bfloat16)
x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    hip.w4a16_gemm(x, qw, sz, variant=1, ksplit=1)
    hip.fp4a16_gemm(x, fq, fs, None, variant=1, ksplit=1)
torch.cuda.synchronize()
