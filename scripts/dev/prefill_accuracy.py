"""Error of the prefill attention kernels against an fp32 reference, per head size: max and mean absolute error over
a varlen batch.  Run twice to compare the kernels: as is (LDS-staged) and with VDAMD_PREFILL_LDS=0 (fragment-direct)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from vllm_distributed_amd import ops  # noqa: E402
from vllm_distributed_amd.ops import torch_ref  # noqa: E402

assert ops.hip_available()
torch.manual_seed(0)
dev = "cuda"
for d, hq, hkv, bs in ((64, 8, 2, 16), (128, 8, 2, 64), (256, 8, 4, 32)):
    cases = [(128, 128), (130, 130), (40, 300), (257, 600), (33, 33), (64, 1000)]
    sls = [c[1] for c in cases]
    nb = [(s + bs - 1) // bs for s in sls]
    total = sum(nb)
    kc = torch.randn(total, hkv, bs, d, device=dev).to(torch.bfloat16)
    vc = torch.randn(total, hkv, d // 16, bs // 8, 16, 8, device=dev).to(torch.bfloat16)
    bt = torch.zeros(len(cases), max(nb), dtype=torch.int32, device=dev)
    perm = torch.randperm(total, device=dev).to(torch.int32)
    o = 0
    for i, n in enumerate(nb):
        bt[i, :n] = perm[o:o + n]
        o += n
    qsl = [0]
    for ql, _ in cases:
        qsl.append(qsl[-1] + ql)
    q = torch.randn(qsl[-1], hq, d, device=dev).to(torch.bfloat16)
    qsl_t = torch.tensor(qsl, device=dev, dtype=torch.int32)
    sl = torch.tensor(sls, device=dev, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    ref = torch_ref.paged_attention(q.float(), kc.float(), vc.float(), bt, qsl_t, sl, scale)
    got = ops.paged_attention(q, kc, vc, bt, qsl_t, sl, scale, max(c[0] for c in cases), max(sls)).float()
    err = (got - ref).abs()
    print(f"D={d:3d}: max {err.max().item():.5f} mean {err.mean().item():.6f} (|ref| mean {ref.abs().mean().item():.4f})",
          flush=True)
