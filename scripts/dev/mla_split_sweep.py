"""KV-split sweep of the latent attention kernel (which split count per batch / context is fastest)."""
import torch
from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import hip as _hip, torch_ref
assert ops.hip_available()
DK, DV, BS = 576, 512, 64
dev = "cuda"
for hq in (16, 32):
    for batch, ctx in ((1, 1024), (1, 4096), (1, 32768), (4, 4096), (8, 4096), (16, 4096), (32, 4096), (64, 4096), (128, 4096), (64, 512), (256, 1024), (32, 32768)):
        nblk = batch * (-(-ctx // BS)) + 1
        kc = (torch.randn(torch_ref.k_cache_shape(nblk, 1, DK, BS), device=dev) * 0.3).bfloat16()
        vc = (torch.randn(torch_ref.v_cache_shape(nblk, 1, DV, BS), device=dev) * 0.3).bfloat16()
        table = (torch.arange(batch * (-(-ctx // BS)), dtype=torch.int32, device=dev) + 1).view(batch, -1)
        q = torch.randn(batch, hq, DK, device=dev).bfloat16()
        lens = torch.full((batch,), ctx, dtype=torch.int32, device=dev)
        out = torch.empty(batch, hq, DV, dtype=torch.bfloat16, device=dev)
        res = []
        for sp in (1, 2, 4, 8, 16, 32, 64):
            if ctx // sp < 64:
                continue
            fn = lambda: _hip.mla_attention(q, kc, vc, table, None, lens, DK ** -0.5, ctx, out, splits=sp)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            b.record()
            torch.cuda.synchronize()
            res.append((sp, a.elapsed_time(b) / 20 * 1e3))
        best = min(res, key=lambda r: r[1])
        print(f"Hq={hq:3d} B={batch:4d} ctx={ctx:6d}: " + "  ".join(f"{sp}:{us:.1f}" for sp, us in res)
              + f"   best {best[0]} chosen {_hip.mla_num_splits(batch, -(-hq // 16), ctx)}", flush=True)
