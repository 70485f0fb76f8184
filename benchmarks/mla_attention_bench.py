"""Latent-cache attention (DeepSeek-V3 widths: 576-wide keys, 512-wide values) on one MI355X: microseconds and
bytes of cache streamed per second, for decode batches at several context lengths and head counts per rank
(128 heads = TP 1, 16 = TP 8), and for a prefill chunk.  Run on the GPU box: python benchmarks/mla_attention_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops  # noqa: E402
from vllm_distributed_amd.ops import hip as _hip  # noqa: E402
from vllm_distributed_amd.ops import torch_ref  # noqa: E402

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


def case(batch, ctx, hq, q_len=1, mode=None, kv8=False):
    nblk = batch * (-(-ctx // BS)) + 1
    cdt = torch.float8_e4m3fn if kv8 else torch.bfloat16
    kc = (torch.randn(torch_ref.k_cache_shape(nblk, 1, DK, BS), device=dev) * 0.3).to(cdt)
    vc = (torch.randn(torch_ref.v_cache_shape(nblk, 1, DV, BS), device=dev) * 0.3).to(cdt)
    table = (torch.arange(batch * (-(-ctx // BS)), dtype=torch.int32, device=dev) + 1).view(batch, -1)
    t = batch * q_len
    q = torch.randn(t, hq, DK, device=dev).bfloat16()
    if q_len == 1:
        rows, lens = None, torch.full((batch,), ctx, dtype=torch.int32, device=dev)
    else:
        rows = torch.arange(batch, dtype=torch.int32, device=dev).repeat_interleave(q_len)
        lens = (torch.arange(q_len, dtype=torch.int32, device=dev) + ctx - q_len + 1).repeat(batch)
    out = torch.empty(t, hq, DV, dtype=torch.bfloat16, device=dev)
    us = bench(lambda: _hip.mla_attention(q, kc, vc, table, rows, lens, DK ** -0.5, ctx, out, mode=mode))
    width = 1 if kv8 else 2
    unique = batch * ctx * (DK + DV) * width                  # each cached row once per sequence
    # ... and once per (query token, 16-head tile): what mode 0 requests from the L2 (modes 1 / 2 a quarter of it)
    streamed = int(lens.sum().item()) * (DK + DV) * width * (-(-hq // 16))
    flops = 2 * int(lens.sum().item()) * hq * (DK + DV)
    print(f"{'decode ' if q_len == 1 else 'prefill'} {'e4m3' if kv8 else 'bf16'} mode={'auto' if mode is None else mode} B={batch:5d} q={q_len:4d} ctx={ctx:6d} Hq={hq:3d}: {us:9.1f} us  "
          f"{unique / us / 1e6:6.2f} TB/s of cache  {streamed / us / 1e6:6.2f} TB/s per-16-head-tile  "
          f"{flops / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":      # a single shape, for a counter run: one <prefill|decode> <heads>
        hq = int(sys.argv[3])
        if sys.argv[2] == "prefill":
            case(1, 32768, hq, q_len=2048)
        else:
            case(128, 4096, hq)
        sys.exit(0)
    for hq in (16, 128):
        for batch, ctx in ((1, 4096), (1, 32768), (8, 4096), (32, 4096), (128, 4096), (256, 1024), (32, 32768)):
            case(batch, ctx, hq)
            if hq >= 64:
                case(batch, ctx, hq, mode=0)
    for batch, ctx in ((32, 4096), (128, 4096), (256, 1024), (32, 32768)):
        case(batch, ctx, 16, kv8=True)
    case(1, 2048, 16, q_len=2048, kv8=True)
    for hq in (16, 128):
        for mode in (None, 0):
            case(1, 2048, hq, q_len=2048, mode=mode)
            case(4, 4096, hq, q_len=512, mode=mode)
            case(1, 32768, hq, q_len=2048, mode=mode)
