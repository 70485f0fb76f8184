"""Probe: sensitivity of the fused 4-bit GEMMs to the K split at medium batch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops                      # noqa: E402
from vllm_distributed_amd.ops import quant                # noqa: E402

assert ops.hip_available()
from vllm_distributed_amd.ops import hip                   # noqa: E402
from w4_gemm_bench import timeit                            # noqa: E402

for name, n, k in (("70b tp1 gate_up", 57344, 8192), ("70b tp1 down", 8192, 28672), ("70b tp1 qkv", 10240, 8192)):
    gen = torch.Generator(device="cuda").manual_seed(0)
    packed = torch.randint(0, 256, (n, k // 2), device="cuda", generator=gen, dtype=torch.int32).to(torch.uint8)
    bs = (0.5 + torch.rand((n, k // 16), device="cuda", generator=gen)).to(torch.float8_e4m3fn)
    fq, fs = quant.pack_fp4(packed, bs)
    q = torch.randint(0, 16, (n, k), device="cuda", generator=gen, dtype=torch.uint8)
    z = torch.randint(0, 16, (n, k // 128), device="cuda", generator=gen).float()
    s = torch.rand((n, k // 128), device="cuda", generator=gen) * 0.01 + 0.005
    qw, sz = quant.pack_w4(q, z, s, torch.bfloat16)
    del q, packed
    for m in (64, 128, 256):
        x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
        row = []
        for ks in (1, 2, 3, 4, 6, 8):
            if ks > k // 128 // 4:
                continue
            ti = timeit(lambda: hip.w4a16_gemm(x, qw, sz, variant=1, ksplit=ks), 20)
            tf = timeit(lambda: hip.fp4a16_gemm(x, fq, fs, None, variant=1, ksplit=ks), 20)
            row.append(f"ks{ks}: {ti:6.1f}/{tf:6.1f}")
        auto = hip.w4_choose_ksplit(m, n, k, 1)
        print(f"{name:16s} M={m:3d} int4/fp4 us  " + "  ".join(row) + f"   auto ks{auto}", flush=True)
