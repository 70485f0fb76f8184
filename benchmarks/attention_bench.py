"""Paged-attention kernel throughput on synthetic caches: decode in TB/s of KV
read, prefill in TFLOP/s (causal FLOPs = 2 * 2 * H * D * sum_i q_i * (ctx_i + (q_i+1)/2))."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vllm_distributed_amd.ops import hip

hip.load()
dev = "cuda"
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def caches(nblocks, hkv, d, bs):
    k = (torch.randn(nblocks, hkv, bs, d, device=dev) * 0.5).to(torch.bfloat16)
    v = (torch.randn(nblocks, hkv, d // 16, bs // 8, 16, 8, device=dev) * 0.5).to(torch.bfloat16)
    return k, v


def decode(b, ctx, hq, hkv, d=128, bs=64):
    nb = (ctx + bs - 1) // bs
    k, v = caches(b * nb, hkv, d, bs)
    bt = torch.randperm(b * nb, device=dev).to(torch.int32).view(b, nb)
    sl = torch.full((b,), ctx, device=dev, dtype=torch.int32)
    q = torch.randn(b, hq, d, device=dev).to(torch.bfloat16)
    ws = torch.empty(hip.decode_workspace_floats(b, hq, d, hkv, ctx), device=dev)
    out = torch.empty_like(q)
    us = timeit(lambda: hip.paged_attention_decode(q, k, v, bt, sl, d ** -0.5, ctx, out, ws))
    bytes_ = b * ctx * hkv * d * 2 * 2
    print(f"decode  B={b:5d} ctx={ctx:6d} Hq={hq:3d} Hkv={hkv}: {us:9.1f} us  {bytes_ / us / 1e6:5.2f} TB/s "
          f"(splits={hip.choose_num_splits(b, hkv, ctx)})", flush=True)


def prefill(nseq, qlen, ctx, hq, hkv, d=128, bs=64):
    sl_ = ctx + qlen
    nb = (sl_ + bs - 1) // bs
    k, v = caches(nseq * nb, hkv, d, bs)
    bt = torch.randperm(nseq * nb, device=dev).to(torch.int32).view(nseq, nb)
    sl = torch.full((nseq,), sl_, device=dev, dtype=torch.int32)
    qsl = (torch.arange(nseq + 1, device=dev) * qlen).to(torch.int32)
    q = torch.randn(nseq * qlen, hq, d, device=dev).to(torch.bfloat16)
    out = torch.empty_like(q)
    us = timeit(lambda: hip.paged_attention_prefill(q, k, v, bt, qsl, sl, d ** -0.5, qlen, sl_, out), 10)
    flops = 4.0 * hq * d * nseq * qlen * (ctx + (qlen + 1) / 2)
    print(f"prefill n={nseq:4d} q={qlen:6d} ctx={ctx:6d} Hq={hq:3d} Hkv={hkv}: {us:10.1f} us  "
          f"{flops / us / 1e6:7.1f} TF/s", flush=True)


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "decode"):
    for b, ctx in ((1024, 192), (256, 192), (256, 2048), (64, 8192), (8, 32768), (1, 131072), (1, 2048), (16, 512)):
        decode(b, ctx, 64, 8)
    decode(1024, 192, 8, 1)       # 70B TP=8 shard
    decode(256, 2048, 32, 8)      # 8B
if which in ("all", "prefill"):
    for n, ql, ctx in ((128, 128, 0), (16, 1024, 0), (4, 4096, 0), (1, 16384, 0), (8, 512, 4096)):
        prefill(n, ql, ctx, 64, 8)
    prefill(16, 1024, 0, 8, 1)
if which == "decode2048":          # the headline step's decode attention, for a counter run
    decode(2048, 160, 64, 8)
if which == "prefill16k":          # one shape, for a counter run under rocprofv3 --pmc
    prefill(1, 16384, 0, 64, 8)
if which in ("all", "headsize"):
    # head sizes 64 (gpt-oss-120b: 64 / 8 heads) and 256 (Gemma-2-9B: 16 / 8 heads)
    for d, hq, hkv in ((64, 64, 8), (256, 16, 8), (128, 32, 8)):
        print(f"head size {d}", flush=True)
        for n, ql, ctx in ((16, 1024, 0), (4, 4096, 0), (1, 16384, 0), (8, 512, 4096)):
            prefill(n, ql, ctx, hq, hkv, d)
