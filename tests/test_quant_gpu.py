"""W4A16 kernels (csrc/quant_gemm.hip) against the fp32 definition of the
packed layout (ops/quant.py), and an AWQ model end to end on the GPU."""
import pytest
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import quant

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _packed(n, k, g, dtype, seed=0):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    q = torch.randint(0, 16, (n, k), generator=gen, dtype=torch.uint8)
    z = torch.randint(0, 16, (n, k // g), generator=gen).float()
    s = (0.5 + torch.rand((n, k // g), generator=gen)) * 0.01
    qw, sz = quant.pack_w4(q, z, s, dtype)
    return qw.to(DEV), sz.to(DEV)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,k,g", [(256, 512, 128), (1040, 1024, 128), (1280, 8192, 128), (512, 1024, 256),
                                   (8192, 4096, 128)])     # the last: more workgroups than CUs
@pytest.mark.parametrize("m", [1, 7, 16, 33, 64, 130, 256])
@pytest.mark.parametrize("variant", [0, 1])
def test_w4a16_gemm_matches_definition(dtype, n, k, g, m, variant):
    qw, sz = _packed(n, k, g, dtype, seed=n + k + m)
    x = torch.randn(m, k, device=DEV, dtype=dtype)
    ops.hip_available()
    from vllm_distributed_amd.ops import hip
    y = hip.w4a16_gemm(x, qw, sz, variant=variant)
    torch.cuda.synchronize()
    w = quant.w4_dequant(qw, sz)                         # fp32, exact (q - z) * s
    ref = x.float() @ w.t()
    scale = ref.abs().max().item()
    # fp32 accumulation; only the output rounding to 16-bit separates the two
    tol = scale * (2 ** -7 if dtype == torch.bfloat16 else 2 ** -9)
    assert (y.float() - ref).abs().max().item() <= tol
    # exact zero beyond M must not be touched: output has exactly m rows
    assert y.shape == (m, n)


def test_w4a16_gemm_strided_rows_and_forced_splits():
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    qw, sz = _packed(512, 2048, 128, torch.bfloat16)
    big = torch.randn(20, 4096, device=DEV, dtype=torch.bfloat16)
    x = big[:, 1024:3072]                                # row stride 4096, 16-byte aligned offset
    ref = x.float() @ quant.w4_dequant(qw, sz).t()
    for variant, ksplit in ((0, None), (1, None), (0, 1), (1, 1), (0, 2), (1, 4)):
        y = hip.w4a16_gemm(x, qw, sz, variant=variant, ksplit=ksplit)
        assert (y.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
    assert hip.w4_choose_ksplit(16, 512, 2048) > 1 and hip.w4_choose_ksplit(16, 65536, 2048) == 1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_w4_dequant_and_large_batch_path(dtype):
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    qw, sz = _packed(768, 1024, 128, dtype)
    w = hip.w4_dequant(qw, sz)
    ref = quant.w4_dequant(qw, sz)
    assert torch.equal(w.float(), ref.to(dtype).float())
    x = torch.randn(hip.W4_MAX_FUSED_M + 44, 1024, device=DEV, dtype=dtype)
    bias = torch.randn(768, device=DEV, dtype=dtype)
    y = ops.w4a16_linear(x, qw, sz, bias)
    want = x.float() @ ref.t() + bias.float()
    assert (y.float() - want).abs().max().item() <= want.abs().max().item() * 2 ** -6


def test_awq_model_on_gpu_matches_dequantised_model():
    """Packed 4-bit execution against the same checkpoint dequantised to
    bf16 at load: the two differ only by bf16 rounding of the weights, so the
    prompt logits must agree closely and greedy decoding must start the same."""
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.parallel import state as pstate
    outs = {}
    for q in ("awq", "awq-dequant"):
        llm = LLM(model="llama-tiny", dtype="bfloat16", max_model_len=256, num_gpu_blocks_override=64,
                  max_num_seqs=8, quantization=q, gpu_memory_utilization=0.3)
        model = llm.engine.core.executor.driver.worker.model_runner.model
        layer = next(iter(model.layers.values()))
        assert (layer.self_attn.qkv_proj.qw is not None) == (q == "awq")
        o = llm.generate([[5, 9, 200, 17], list(range(10, 90))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logprobs=3))
        outs[q] = o
        llm.shutdown()
        pstate.destroy_model_parallel()
    for a, b in zip(outs["awq"], outs["awq-dequant"]):
        assert a.outputs[0].token_ids[:2] == b.outputs[0].token_ids[:2]
        la, lb = a.outputs[0].logprobs[0], b.outputs[0].logprobs[0]
        t = a.outputs[0].token_ids[0]
        assert abs(la[t] - lb[t]) < 0.05


@pytest.mark.parametrize("quantized", [False, True])
@pytest.mark.parametrize("tokens", [1, 7, 64, 300, 1100])
def test_fused_moe_grouped_gemm_path_matches_per_expert_loop(quantized, tokens, monkeypatch):
    """Experts as grouped HIP GEMMs over expert-sorted rows (device-side
    routing tables) against the per-expert library-GEMM loop."""
    from vllm_distributed_amd.models.layers.linear import QuantSpec, quantized_linears
    from vllm_distributed_amd.models.layers.moe import FusedMoE
    from vllm_distributed_amd.parallel import state as pstate
    pstate.destroy_model_parallel()
    pstate.ensure_single_process_groups(torch.device(DEV))
    hidden, inter, experts, top_k = 512, 256, 16, 4
    torch.manual_seed(tokens)
    with torch.device(DEV), quantized_linears(QuantSpec("awq", 128) if quantized else None):
        moe = FusedMoE(hidden, inter, experts, top_k, True, torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for name, p in moe.named_parameters():
        if p.dtype == torch.int32:
            p.data.copy_(torch.randint(-(1 << 31), (1 << 31) - 1, p.shape, device=DEV, generator=gen,
                                       dtype=torch.int64).to(torch.int32))
        elif "scales" in name:
            p.data.copy_((0.002 + 0.002 * torch.rand(p.shape, device=DEV, generator=gen)).to(p.dtype))
        else:
            p.data.copy_((torch.randn(p.shape, device=DEV, generator=gen) * 0.05).to(p.dtype))
    moe.finalize_weights()
    assert (moe.qw13 is not None) == quantized
    x = torch.randn(tokens, hidden, device=DEV, generator=gen).to(torch.bfloat16)
    rows = moe._grouped_tile_rows(x)
    avg = tokens * top_k / experts
    assert rows == 0 if avg > (512 if quantized else 1024) else rows in (16, 32, 64, 128)
    with torch.inference_mode():
        monkeypatch.setattr(FusedMoE, "_grouped_tile_rows", lambda self, x: 0)
        ref = moe(x).float()
        for tile_rows in {16, 64, rows or 64}:
            monkeypatch.setattr(FusedMoE, "_grouped_tile_rows", lambda self, x, r=tile_rows: r)
            got = moe(x).float()
            torch.cuda.synchronize()
            err = (got - ref).abs().max().item()
            assert err <= 0.03 * ref.abs().max().item() + 1e-3, (tile_rows, err)
    # the routing tables are built without a host sync: the step can be captured
    if tokens == 64:
        monkeypatch.setattr(FusedMoE, "_grouped_tile_rows", lambda self, x: 32)
        with torch.inference_mode():
            moe(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = moe(x)
            g.replay()
            torch.cuda.synchronize()
            assert (y.float() - ref).abs().max().item() <= 0.03 * ref.abs().max().item() + 1e-3
    pstate.destroy_model_parallel()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile_rows,tokens", [(16, 1), (16, 9), (32, 40)])
def test_grouped_w4a16_gemm_forms_silu_mul_while_loading(dtype, tile_rows, tokens):
    """The second expert GEMM of a decode step reads ``[gate | up]`` and multiplies by
    ``silu(gate) * up`` formed in registers: against the activation kernel followed by the same GEMM
    (same rounding of the activation to 16 bits, so the products agree to accumulation order) and against
    an fp32 reference over the dequantised weights."""
    from vllm_distributed_amd.ops import hip
    experts, top_k, inter, hidden = 8, 2, 384, 256        # second GEMM: [rows, inter] x [hidden, inter]^T
    gen = torch.Generator(device=DEV).manual_seed(tile_rows * 100 + tokens)
    qws, szs, deq = [], [], []
    for _ in range(experts):
        w = torch.randn(hidden, inter, device=DEV, generator=gen) * 0.05
        q, z, sc = quant.quantize_w4(w, 128)
        qw, sz = quant.pack_w4(q, z, sc, dtype)
        qws.append(qw), szs.append(sz)
        deq.append(quant.w4_dequant(qw, sz).t().contiguous())                         # [inter, hidden]
    qw, sz = torch.stack(qws).contiguous(), torch.stack(szs).contiguous()
    logits = torch.randn(tokens, experts, device=DEV, generator=gen)
    topi = logits.topk(top_k, dim=-1).indices.to(torch.int32)
    src_row, inv, tiles = hip.moe_align(topi, experts, tile_rows, 0, -1)
    h = (torch.randn(src_row.numel(), 2 * inter, device=DEV, generator=gen) * 2).to(dtype)
    fused = hip.grouped_w4a16_gemm(h, qw, sz, tiles, tile_rows, silu_mul=True)
    act = ops.silu_and_mul(h)
    plain = hip.grouped_w4a16_gemm(act, qw, sz, tiles, tile_rows)
    torch.cuda.synchronize()
    flat = topi.reshape(-1).long()
    for slot in range(flat.numel()):
        row = int(inv[slot])
        a32 = torch.nn.functional.silu(h[row, :inter].float()) * h[row, inter:].float()
        want = a32 @ deq[int(flat[slot])]
        scale = want.abs().max().item() + 1e-3
        assert (fused[row].float() - want).abs().max().item() <= 0.02 * scale, (slot, row)
        assert (fused[row].float() - plain[row].float()).abs().max().item() <= 0.01 * scale, (slot, row)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k", [(1, 512), (7, 4096), (300, 8192), (16, 28672), (5, 1000)])
def test_dynamic_fp8_quant_matches_torch_conversion(dtype, m, k):
    from vllm_distributed_amd.ops import hip
    assert ops.hip_available()
    x = (torch.randn(m, k, device=DEV) * torch.rand(m, 1, device=DEV) * 5).to(dtype)
    x[0, 0] = 0.0
    xq, scale = hip.dynamic_fp8_quant(x)
    assert xq.dtype == torch.float8_e4m3fn and scale.shape == (m, 1)
    ref_scale = x.float().abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
    torch.testing.assert_close(scale, ref_scale, rtol=1e-6, atol=0)
    ref_q = (x.float() / ref_scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    # the hardware convert and torch both round to nearest even: bit-identical except where the
    # reciprocal multiply (x * (1/s) vs x / s) lands on the other side of a tie - then one code apart
    a, b = xq.view(torch.uint8).int(), ref_q.view(torch.uint8).int()
    assert (a == b).float().mean().item() > 0.99
    assert int((a - b).abs().max()) <= 1
    err = (xq.float() * scale - x.float()).abs().amax(dim=1, keepdim=True)
    assert bool((err <= ref_scale * 16.0 + 1e-6).all())          # half a step at the top binade (step 32)


@pytest.mark.parametrize("m", [1, 16, 200, 2048])
def test_fp8_linear_matches_dequantised_reference(m):
    n, k = 1280, 4096
    w = (torch.randn(n, k, device=DEV) * 0.03).to(torch.bfloat16)
    x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
    bias = torch.randn(n, device=DEV).to(torch.bfloat16)
    wq, ws = ops.fp8_quantize_weight(w)
    y = ops.fp8_linear(x, wq, ws, bias)
    from vllm_distributed_amd.ops import hip
    xq, xs = hip.dynamic_fp8_quant(x)
    ref = (xq.float() * xs) @ (wq.float() * ws.view(-1, 1)).t() + bias.float()
    assert (y.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
    dense = x.float() @ w.float().t() + bias.float()
    rel = (y.float() - dense).norm() / dense.norm()
    assert rel < 0.05, rel                    # the quantisation error itself: ~3 % for gaussian data


def test_fp8_model_on_gpu_tracks_bf16_model():
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.parallel import state as pstate
    outs = {}
    for q in (None, "fp8"):
        pstate.destroy_model_parallel()
        llm = LLM(model="llama-tiny", dtype="bfloat16", max_model_len=256, num_gpu_blocks_override=64,
                  max_num_seqs=8, quantization=q, gpu_memory_utilization=0.3)
        layer = next(iter(llm.engine.core.executor.driver.worker.model_runner.model.layers.values()))
        assert (layer.mlp.down_proj.weight.dtype == torch.float8_e4m3fn) == (q == "fp8")
        outs[q] = llm.generate([[5, 9, 200, 17], list(range(10, 90))],
                               SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True, logprobs=8))
        llm.shutdown()
        pstate.destroy_model_parallel()
    for a, b in zip(outs[None], outs["fp8"]):
        t = a.outputs[0].token_ids[0]
        lb = b.outputs[0].logprobs[0]
        assert t in lb and abs(lb[t] - a.outputs[0].logprobs[0][t]) < 0.25


# ------------------------------------------------------------------ NVFP4
def _fp4_packed(n, k, seed=0, alpha=True):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    packed = torch.randint(0, 256, (n, k // 2), generator=gen, dtype=torch.int32).to(torch.uint8)
    bs = (0.25 + 2.0 * torch.rand((n, k // 16), generator=gen)).to(torch.float8_e4m3fn)
    qw, fs = quant.pack_fp4(packed, bs)
    al = (0.002 + 0.01 * torch.rand(n, generator=gen)).float() if alpha else None
    return qw.to(DEV), fs.to(DEV), None if al is None else al.to(DEV)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,k", [(256, 512), (1040, 1024), (1280, 8192), (8192, 4096)])
@pytest.mark.parametrize("m", [1, 7, 16, 33, 64, 130, 256])
@pytest.mark.parametrize("variant", [0, 1])
def test_fp4a16_gemm_matches_definition(dtype, n, k, m, variant):
    """E2M1 weights x e4m3 block scales x per-row fp32 scale against the fp32
    definition of the layout (ops/quant.py: fp4_dequant)."""
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    qw, fs, alpha = _fp4_packed(n, k, seed=n + k + m, alpha=(m % 2 == 1))
    x = torch.randn(m, k, device=DEV, dtype=dtype)
    y = hip.fp4a16_gemm(x, qw, fs, alpha, variant=variant)
    torch.cuda.synchronize()
    # the kernel rounds each (e2m1 * block scale) product to the activation dtype before the MFMA
    w = quant.fp4_dequant(qw, fs, None).to(dtype).float()
    ref = x.float() @ w.t()
    if alpha is not None:
        ref = ref * alpha.reshape(1, -1)
    tol = ref.abs().max().item() * (2 ** -7 if dtype == torch.bfloat16 else 2 ** -9)
    assert y.shape == (m, n)
    assert (y.float() - ref).abs().max().item() <= tol


def test_fp4_forced_splits_dequant_and_large_batch_path():
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    qw, fs, alpha = _fp4_packed(512, 2048, seed=3)
    big = torch.randn(20, 4096, device=DEV, dtype=torch.bfloat16)
    x = big[:, 1024:3072]
    wref = quant.fp4_dequant(qw, fs, None).to(torch.bfloat16).float()
    ref = (x.float() @ wref.t()) * alpha.reshape(1, -1)
    for variant, ksplit in ((0, None), (1, None), (0, 1), (1, 1), (0, 2), (1, 4)):
        y = hip.fp4a16_gemm(x, qw, fs, alpha, variant=variant, ksplit=ksplit)
        assert (y.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
    # several workgroups per CU, repeatedly: results must not depend on what shares the CU
    qw2, fs2, _ = _fp4_packed(8192, 4096, seed=9, alpha=False)
    w2 = quant.fp4_dequant(qw2, fs2, None).to(torch.bfloat16).float()
    for m in (48, 192):
        x2 = torch.randn(m, 4096, device=DEV, dtype=torch.bfloat16)
        ref2 = x2.float() @ w2.t()
        for variant, ksplit in ((0, 4), (1, 1), (1, 4)):
            for _ in range(3):
                y = hip.fp4a16_gemm(x2, qw2, fs2, None, variant=variant, ksplit=ksplit)
                assert (y.float() - ref2).abs().max().item() <= ref2.abs().max().item() * 2 ** -7
    for dtype in (torch.bfloat16, torch.float16):
        w = hip.fp4_dequant(qw, fs, alpha, dtype)
        want = quant.fp4_dequant(qw, fs, alpha)
        assert (w.float() - want).abs().max().item() <= want.abs().max().item() * 2 ** -8
    xl = torch.randn(hip.W4_MAX_FUSED_M + 44, 2048, device=DEV, dtype=torch.bfloat16)
    bias = torch.randn(512, device=DEV, dtype=torch.bfloat16)
    y = ops.fp4_linear(xl, qw, fs, alpha, bias)
    want = xl.float() @ quant.fp4_dequant(qw, fs, alpha).t() + bias.float()
    assert (y.float() - want).abs().max().item() <= want.abs().max().item() * 2 ** -5


def test_quantize_fp4_round_trip_on_gpu_kernel():
    """Dense weights -> NVFP4 checkpoint tensors -> kernel: the product tracks
    the dense product to fp4 accuracy."""
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    torch.manual_seed(0)
    w = torch.randn(256, 1024) * 0.03
    packed, bs, g = quant.quantize_fp4(w)
    qw, fs = quant.pack_fp4(packed, bs)
    alpha = (1.0 / g).expand(256).contiguous()
    x = torch.randn(8, 1024, device=DEV, dtype=torch.bfloat16)
    y = hip.fp4a16_gemm(x, qw.to(DEV), fs.to(DEV), alpha.to(DEV))
    ref = x.float() @ w.to(DEV).t()
    rel = (y.float() - ref).norm() / ref.norm()
    assert rel < 0.15, rel


@pytest.mark.parametrize("tokens", [1, 64, 300])
def test_fused_moe_nvfp4_grouped_matches_per_expert_loop(tokens, monkeypatch):
    from vllm_distributed_amd.models.layers.linear import QuantSpec, quantized_linears
    from vllm_distributed_amd.models.layers.moe import FusedMoE
    from vllm_distributed_amd.parallel import state as pstate
    pstate.destroy_model_parallel()
    pstate.ensure_single_process_groups(torch.device(DEV))
    hidden, inter, experts, top_k = 512, 256, 16, 4
    with torch.device(DEV), quantized_linears(QuantSpec("nvfp4", 16, 1)):
        moe = FusedMoE(hidden, inter, experts, top_k, True, torch.bfloat16)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for name, p in moe.named_parameters():
        if p.dtype == torch.uint8:
            p.data.copy_(torch.randint(0, 256, p.shape, device=DEV, generator=gen, dtype=torch.int32).to(torch.uint8))
        elif p.dtype == torch.float8_e4m3fn:
            p.data.copy_((0.5 + torch.rand(p.shape, device=DEV, generator=gen)).to(p.dtype))
        elif "alpha" in name:
            p.data.copy_(0.005 + 0.005 * torch.rand(p.shape, device=DEV, generator=gen))
        else:
            p.data.copy_((torch.randn(p.shape, device=DEV, generator=gen) * 0.05).to(p.dtype))
    moe.finalize_weights()
    assert moe.alpha13 is not None and moe.sz13.dtype == torch.int32
    x = torch.randn(tokens, hidden, device=DEV, generator=gen).to(torch.bfloat16)
    with torch.inference_mode():
        monkeypatch.setattr(FusedMoE, "_grouped_tile_rows", lambda self, x: 0)
        ref = moe(x).float()
        for tile_rows in (16, 64):
            monkeypatch.setattr(FusedMoE, "_grouped_tile_rows", lambda self, x, r=tile_rows: r)
            got = moe(x).float()
            torch.cuda.synchronize()
            err = (got - ref).abs().max().item()
            assert err <= 0.03 * ref.abs().max().item() + 1e-3, (tile_rows, err)
    pstate.destroy_model_parallel()


def test_nvfp4_model_on_gpu_matches_cpu_definition(tmp_path):
    """Packed NVFP4 execution on the GPU against the same checkpoint file run
    through the fp32 CPU definition of the format."""
    import json
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    from vllm_distributed_amd.parallel import state as pstate
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy", quantization="nvfp4")
    save_dummy_checkpoint(mc, str(tmp_path))
    (tmp_path / "config.json").write_text(json.dumps(dict(
        mc.hf_config, quantization_config={"quant_method": "compressed-tensors", "format": "nvfp4-pack-quantized"})))
    outs = {}
    for dev in ("cuda", "cpu"):
        kw = dict(dtype="bfloat16", gpu_memory_utilization=0.3) if dev == "cuda" else dict(dtype="float32", device="cpu")
        llm = LLM(model=str(tmp_path), max_model_len=256, num_gpu_blocks_override=64, max_num_seqs=8, **kw)
        if dev == "cuda":
            model = llm.engine.core.executor.driver.worker.model_runner.model
            layer = next(iter(model.layers.values()))
            assert layer.self_attn.qkv_proj.alpha is not None and layer.mlp.down_proj.sz.dtype == torch.int32
        outs[dev] = llm.generate([[5, 9, 200, 17], list(range(10, 90))],
                                 SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True, logprobs=20))
        llm.shutdown()
        pstate.destroy_model_parallel()
    # random fp4 weights give a nearly flat distribution, so the argmax itself is
    # not stable under bf16 rounding: compare the log-probabilities of the
    # tokens both runs rank in their top 20
    for a, b in zip(outs["cuda"], outs["cpu"]):
        la, lb = a.outputs[0].logprobs[0], b.outputs[0].logprobs[0]
        common = set(la) & set(lb)
        assert len(common) >= 8, (sorted(la), sorted(lb))
        assert max(abs(la[t] - lb[t]) for t in common) < 0.15


@pytest.mark.parametrize("method", ["awq", "nvfp4"])
def test_dense_shadow_serves_large_batches(method, monkeypatch):
    """VDAMD_W4_DENSE_SHADOW keeps a dequantised copy next to the 4-bit tiles:
    large batches use it (same numbers as dequantise-per-call), small ones keep
    the fused kernel."""
    from vllm_distributed_amd.models.layers.linear import QuantSpec, ReplicatedLinear, quantized_linears
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    outs = {}
    for shadow in ("0", "1"):
        monkeypatch.setenv("VDAMD_W4_DENSE_SHADOW", shadow)
        spec = QuantSpec("awq", 128) if method == "awq" else QuantSpec("nvfp4", 16, 1)
        with torch.device(DEV), quantized_linears(spec):
            lin = ReplicatedLinear(1024, 512, False, torch.bfloat16)
        gen = torch.Generator(device=DEV).manual_seed(3)
        for name, p in lin.named_parameters():
            if p.dtype == torch.int32:
                p.data.copy_(torch.randint(-(1 << 31), (1 << 31) - 1, p.shape, device=DEV, generator=gen,
                                           dtype=torch.int64).to(torch.int32))
            elif p.dtype == torch.uint8:
                p.data.copy_(torch.randint(0, 256, p.shape, device=DEV, generator=gen, dtype=torch.int32).to(torch.uint8))
            elif p.dtype == torch.float8_e4m3fn:
                p.data.copy_((0.5 + torch.rand(p.shape, device=DEV, generator=gen)).to(p.dtype))
            else:
                p.data.copy_((0.002 + 0.002 * torch.rand(p.shape, device=DEV, generator=gen)).to(p.dtype))
        lin.finalize_weights()
        assert (lin.dense_shadow is not None) == (shadow == "1")
        xs = torch.randn(8, 1024, device=DEV, generator=gen).to(torch.bfloat16)
        xl = torch.randn(hip.W4_MAX_FUSED_M + 64, 1024, device=DEV, generator=gen).to(torch.bfloat16)
        outs[shadow] = (lin(xs).float(), lin(xl).float())
    assert torch.equal(outs["0"][0], outs["1"][0])
    assert (outs["0"][1] - outs["1"][1]).abs().max() <= 1e-6 * outs["0"][1].abs().max() + 1e-6


@pytest.mark.parametrize("tokens", [1, 7, 64, 300])
def test_gpt_oss_expert_block_grouped_path_matches_per_expert_loop(tokens, monkeypatch):
    """gpt-oss experts (biases, clamped SwiGLU, reduction dimensions zero-padded to the GEMM's K tile) as
    grouped HIP GEMMs with the per-row expert biases added on the device, against the per-expert loop of
    library GEMMs on the same weights and the same router logits; and captured in a hipGraph."""
    from vllm_distributed_amd.models.gpt_oss import GptOssMoE
    from vllm_distributed_amd.models.layers.moe import FusedMoE
    from vllm_distributed_amd.parallel import state as pstate
    pstate.destroy_model_parallel()
    pstate.ensure_single_process_groups(torch.device(DEV))
    cfg = dict(hidden_size=192, intermediate_size=160, num_local_experts=8, num_experts_per_tok=4, swiglu_limit=7.0)
    torch.manual_seed(tokens)
    with torch.device(DEV):
        moe = GptOssMoE(cfg, torch.bfloat16, False)
    assert (moe.h_pad, moe.i_pad) == (256, 256)
    gen = torch.Generator(device=DEV).manual_seed(3)
    E, I, H = 8, 160, 192
    moe.load_stacked("gate_up_proj", (torch.randn(E, H, 2 * I, device=DEV, generator=gen) * 0.08).to(torch.bfloat16))
    moe.load_stacked("gate_up_proj_bias", (torch.randn(E, 2 * I, device=DEV, generator=gen) * 0.5).to(torch.bfloat16))
    moe.load_stacked("down_proj", (torch.randn(E, I, H, device=DEV, generator=gen) * 0.08).to(torch.bfloat16))
    moe.load_stacked("down_proj_bias", (torch.randn(E, H, device=DEV, generator=gen) * 0.5).to(torch.bfloat16))
    moe.gate.weight.data.copy_((torch.randn(E, H, device=DEV, generator=gen) * 0.1).to(torch.bfloat16))
    moe.gate.bias.data.copy_((torch.randn(E, device=DEV, generator=gen) * 0.5).to(torch.bfloat16))
    x = (torch.randn(tokens, H, device=DEV, generator=gen) * 2).to(torch.bfloat16)     # large enough to hit the clamp
    with torch.inference_mode():
        rows = moe._grouped_tile_rows(x)
        assert rows in (16, 32, 64, 128)
        got = moe(x).float()
        monkeypatch.setattr(GptOssMoE, "_grouped_tile_rows", lambda self, x: 0)
        ref = moe(x).float()
        monkeypatch.undo()
        torch.cuda.synchronize()
        err = (got - ref).abs().max().item()
        assert err <= 0.03 * ref.abs().max().item() + 1e-3, err
        if tokens == 64:
            g = torch.cuda.CUDAGraph()
            moe(x)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                y = moe(x)
            g.replay()
            torch.cuda.synchronize()
            assert (y.float() - ref).abs().max().item() <= 0.03 * ref.abs().max().item() + 1e-3
    pstate.destroy_model_parallel()
