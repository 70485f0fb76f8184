"""W4A16 GEMM at one to a few rows: time per (variant, K split) for the attention projections of
Qwen3-Coder-480B (4-bit), inside a hipGraph over rotating weight copies.

    PYTHONPATH=. python benchmarks/w4_ksplit_probe.py
"""
import torch

from vllm_distributed_amd.ops import hip, quant
from stream_gemm_sweep import timed

hip.load()
DEV = "cuda"
SHAPES = {"qwen480b.qkv": (14336, 6144), "qwen480b.o": (6144, 12288)}

for name, (n, k) in SHAPES.items():
    gen = torch.Generator(device=DEV).manual_seed(0)
    copies = []
    for _ in range(12):
        q = torch.randint(0, 16, (n, k), device=DEV, generator=gen, dtype=torch.uint8)
        z = torch.randint(0, 16, (n, k // 128), device=DEV, generator=gen).float()
        s = torch.rand((n, k // 128), device=DEV, generator=gen) * 0.01 + 0.005
        copies.append(quant.pack_w4(q, z, s, torch.bfloat16))
        del q
    nbytes = n * k // 2
    for m in (1, 4, 16):
        x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
        auto_v, auto_ks = hip.w4_choose_variant(m, n, k), None
        auto_ks = hip.w4_choose_ksplit(m, n, k, auto_v)
        res = []
        for variant in (0, 1):
            for ks in (1, 2, 3, 4, 6, 8, 12, 16):
                if ks > k // 128 // 2:
                    continue
                try:
                    t = timed(lambda w: hip.w4a16_gemm(x, w[0], w[1], variant=variant, ksplit=ks), copies)
                    tp = timed(lambda w: hip.w4a16_gemm(x, w[0], w[1], variant=variant, ksplit=ks, partials=True), copies) if ks > 1 else t
                except Exception as e:  # noqa: BLE001
                    continue
                res.append((t, tp, variant, ks))
        best = min(res)
        bestp = min(res, key=lambda r: r[1])
        auto = next(r for r in res if r[2] == auto_v and r[3] == auto_ks)
        print(f"{name} M={m}: auto v{auto_v} ks{auto_ks} {auto[0]:.1f} us (slabs only {auto[1]:.1f}) | best v{best[2]} ks{best[3]} "
              f"{best[0]:.1f} us {nbytes / best[0] / 1e6:.2f} TB/s | best slabs-only v{bestp[2]} ks{bestp[3]} {bestp[1]:.1f} us", flush=True)
        print("   " + "  ".join(f"v{v}k{ks}:{t:.1f}/{tp:.1f}" for t, tp, v, ks in res), flush=True)
