"""A few launches of the library GEMM and of csrc/big_gemm.hip on two 70B shapes, for rocprofv3 --pmc:

    rocprofv3 --pmc SQ_WAVE_CYCLES ... -d out -o p --output-format csv -- python3 benchmarks/big_gemm_pmc_target.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd.ops import hip  # noqa: E402

hip.load()
torch.manual_seed(0)
variant = int(os.environ.get("VDAMD_BIG_GEMM_VARIANT", "0"))
m = 2048
for (n, k) in ((8192, 8192), (57344, 8192)):
    buf = torch.empty((n, k + 128), dtype=torch.bfloat16, device="cuda")
    w = buf[:, :k]
    w.normal_(0, 0.02)
    x = torch.randn((m, k), dtype=torch.bfloat16, device="cuda")
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    for _ in range(4):
        torch.mm(x, w.t(), out=out)
        hip.big_gemm(x, w, out=out, variant=variant)
    torch.cuda.synchronize()
