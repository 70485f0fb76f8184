"""Latent-cache attention (DeepSeek-V3 widths: 576-wide keys, 512-wide values) on one MI355X: microseconds and
bytes of cache streamed per second, for decode batches at several context lengths and head counts per rank
(128 heads = TP 1, 16 = TP 8), and for a prefill chunk.  Run on the GPU box: python benchmarks/mla_attention_bench.py"""
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import hip as _hip
from vllm_distributed_amd.ops import torch_ref

assert ops.hip_available(), "the HIP kernel library is not built"

DK, DV, BS = 576, 512, 64
dev = "cuda"


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def case(batch, ctx, hq, q_len=1):
    nblk = batch * (-(-ctx // BS)) + 1
    kc = (torch.randn(torch_ref.k_cache_shape(nblk, 1, DK, BS), device=dev) * 0.3).bfloat16()
    vc = (torch.randn(torch_ref.v_cache_shape(nblk, 1, DV, BS), device=dev) * 0.3).bfloat16()
    table = (torch.arange(batch * (-(-ctx // BS)), dtype=torch.int32, device=dev) + 1).view(batch, -1)
    t = batch * q_len
    q = torch.randn(t, hq, DK, device=dev).bfloat16()
    if q_len == 1:
        rows, lens = None, torch.full((batch,), ctx, dtype=torch.int32, device=dev)
    else:
        rows = torch.arange(batch, dtype=torch.int32, device=dev).repeat_interleave(q_len)
        lens = (torch.arange(q_len, dtype=torch.int32, device=dev) + ctx - q_len + 1).repeat(batch)
    out = torch.empty(t, hq, DV, dtype=torch.bfloat16, device=dev)
    us = bench(lambda: _hip.mla_attention(q, kc, vc, table, rows, lens, DK ** -0.5, ctx, out))
    unique = batch * ctx * (DK + DV) * 2                      # each cached row once per sequence
    streamed = int(lens.sum().item()) * (DK + DV) * 2 * (-(-hq // 16))    # ... and as the workgroups request it
    flops = 2 * int(lens.sum().item()) * hq * (DK + DV)
    splits = _hip.mla_num_splits(t, -(-hq // 16), ctx)
    print(f"{'decode ' if q_len == 1 else 'prefill'} B={batch:5d} q={q_len:4d} ctx={ctx:6d} Hq={hq:3d}: {us:9.1f} us  "
          f"{unique / us / 1e6:6.2f} TB/s of cache  {streamed / us / 1e6:6.2f} TB/s requested  "
          f"{flops / us / 1e6:7.1f} TF/s  (splits={splits})", flush=True)


if __name__ == "__main__":
    for hq in (16, 128):
        for batch, ctx in ((1, 4096), (1, 32768), (8, 4096), (32, 4096), (128, 4096), (256, 1024), (32, 32768)):
            case(batch, ctx, hq)
    for hq in (16, 128):
        case(1, 2048, hq, q_len=2048)
        case(4, 4096, hq, q_len=512)
