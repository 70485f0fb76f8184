"""A short run of every hand-written kernel at a representative shape, for
hardware-counter collection:

    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES ... -d out -o pmc --output-format csv -- python3 benchmarks/pmc_kernels.py
    python benchmarks/pmc_kernels.py --summarize out/pmc_counter_collection.csv

``--summarize`` averages each counter per kernel name.
"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarize(path: str) -> None:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            for stop in ("(", "<"):
                if not name.startswith("_Z") and stop in name:
                    name = name[:name.index(stop)]
            agg[name[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[name[:70]]["~dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    counters = sorted({c for v in agg.values() for c in v})
    print("kernel".ljust(70), " ".join(c[-22:].rjust(22) for c in counters))
    for k, v in sorted(agg.items()):
        if not any(s in k for s in ("paged", "rms_norm", "silu", "rope", "w4a16", "w4_dequant", "skinny", "argmax",
                                    "sample", "embedding", "layer_norm", "Cijk")):
            continue
        print(k.ljust(70), " ".join(f"{sum(v[c]) / len(v[c]):22.4g}" if c in v else " " * 22 for c in counters))


def main() -> None:
    import torch

    from vllm_distributed_amd import ops
    from vllm_distributed_amd.ops import hip, quant
    assert ops.hip_available()
    dev = "cuda"
    torch.manual_seed(0)
    bf = torch.bfloat16
    # decode attention: 256 sequences x 2048 tokens, Llama-3-70B heads
    b, ctx, hq, hkv, d, bs = 256, 2048, 64, 8, 128, 64
    nb = ctx // bs
    k = (torch.randn(b * nb, hkv, bs, d, device=dev) * 0.5).to(bf)
    v = (torch.randn(b * nb, hkv, d // 16, bs // 8, 16, 8, device=dev) * 0.5).to(bf)
    bt = torch.randperm(b * nb, device=dev).to(torch.int32).view(b, nb)
    sl = torch.full((b,), ctx, device=dev, dtype=torch.int32)
    q = torch.randn(b, hq, d, device=dev).to(bf)
    for _ in range(3):
        hip.paged_attention_decode(q, k, v, bt, sl, d ** -0.5, ctx)
    # prefill attention: 8 sequences x 2048 tokens
    nseq, qlen = 8, 2048
    bt2 = bt[:nseq]
    sl2 = torch.full((nseq,), qlen, device=dev, dtype=torch.int32)
    qsl = (torch.arange(nseq + 1, device=dev) * qlen).to(torch.int32)
    q2 = torch.randn(nseq * qlen, hq, d, device=dev).to(bf)
    for _ in range(3):
        hip.paged_attention_prefill(q2, k, v, bt2, qsl, sl2, d ** -0.5, qlen, qlen)
    # norms / activation / rotary at 2048 x 8192
    x = torch.randn(2048, 8192, device=dev).to(bf)
    res = torch.randn(2048, 8192, device=dev).to(bf)
    w = torch.ones(8192, device=dev, dtype=bf)
    gu = torch.randn(2048, 57344, device=dev).to(bf)
    for _ in range(3):
        ops.fused_add_rms_norm(x, res, w, 1e-5)
        ops.silu_and_mul(gu)
    # W4A16 (gate_up, 16 rows) and skinny bf16 GEMM (down, 64 rows)
    n, kk = 57344, 8192
    qq = torch.randint(0, 16, (n, kk), device=dev, dtype=torch.uint8)
    z = torch.randint(0, 16, (n, kk // 128), device=dev).float()
    s = torch.rand((n, kk // 128), device=dev) * 0.01 + 0.005
    qw, sz = quant.pack_w4(qq, z, s, bf)
    del qq
    x16 = torch.randn(16, kk, device=dev).to(bf)
    x64 = torch.randn(64, kk, device=dev).to(bf)
    wd = (torch.randn(8192, 28672, device=dev) * 0.02).to(bf)
    xd = torch.randn(64, 28672, device=dev).to(bf)
    for _ in range(3):
        hip.w4a16_gemm(x16, qw, sz)
        hip.w4a16_gemm(x64, qw, sz)
        hip.skinny_gemm(xd, wd)
    # sampling
    logits = torch.randn(256, 128256, device=dev).to(bf)
    for _ in range(3):
        ops.argmax(logits)
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        main()
