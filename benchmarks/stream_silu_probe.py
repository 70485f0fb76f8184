"""gate_up + SiLU at M <= 8: the weight-streaming kernel with the activation in its epilogue against the
library GEMM followed by the activation kernel (us per call inside a hipGraph, rotating weight copies).

    PYTHONPATH=. python benchmarks/stream_silu_probe.py
"""
import sys

import torch
import torch.nn.functional as F

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import hip
from stream_gemm_sweep import timed

hip.load()
DEV = "cuda"
SHAPES = {"70b.gate_up": (57344, 8192), "8b.gate_up": (28672, 4096), "70b/tp8.gate_up": (7168, 8192),
          "70b.lm_head": (128256, 8192)}


def main():
    for name, (n, k) in SHAPES.items():
        nbytes = n * k * 2
        ncopies = max(2, min(12, (600 << 20) // nbytes + 1))
        copies = []
        for _ in range(ncopies):
            buf = torch.empty(n, k + 128, device=DEV, dtype=torch.bfloat16)
            buf[:, :k] = (torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16)
            copies.append(buf[:, :k])
        for m in (1, 2, 4, 8):
            x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
            print(f"{name} M={m}: ", end="", flush=True)
            if "lm_head" in name:
                t_lib = timed(lambda w: F.linear(x, w), copies)
                print(f"lib {t_lib:7.1f} us ", end="", flush=True)
                t_st = timed(lambda w: hip.stream_gemm(x, w), copies)
                print(f"| stream {t_st:7.1f} us", flush=True)
                continue
            ref = ops.silu_and_mul(F.linear(x, copies[0]))
            got = hip.stream_gemm(x, copies[0], silu_mul=True)
            err = (got.float() - ref.float()).abs().max().item()
            t_lib = timed(lambda w: ops.silu_and_mul(F.linear(x, w)), copies)
            print(f"lib+silu {t_lib:7.1f} us ", end="", flush=True)
            t_st = timed(lambda w: hip.stream_gemm(x, w, silu_mul=True), copies)
            print(f"| stream(silu) {t_st:7.1f} us | max err {err:.4f}", flush=True)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    sys.exit(main())
