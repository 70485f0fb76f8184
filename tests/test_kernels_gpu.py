"""HIP kernels vs the plain-PyTorch fp32 reference of the same op (needs an MI355X)."""
import math

import pytest
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import torch_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _lib():
    from vllm_distributed_amd.ops import hip
    hip.load()      # fail loudly if the extension was not built
    yield


def _rand(*shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, device=DEV, dtype=torch.float32) * scale).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,H", [(1, 4096), (7, 8192), (300, 768), (33, 128), (5, 16384)])
def test_rms_norm_and_fused_add(dtype, T, H):
    x, r, w = _rand(T, H, dtype=dtype), _rand(T, H, dtype=dtype), _rand(H, dtype=dtype, scale=0.3) + 1
    ref = torch_ref.rms_norm(x.float(), w.float(), 1e-5)
    got = ops.rms_norm(x, w, 1e-5)
    torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)
    ref_n, ref_r = torch_ref.fused_add_rms_norm(x.clone(), r.clone(), w, 1e-5)
    xx, rr = x.clone(), r.clone()
    got_n, got_r = ops.fused_add_rms_norm(xx, rr, w, 1e-5)
    assert got_n.data_ptr() == xx.data_ptr() and got_r.data_ptr() == rr.data_ptr()   # in place
    torch.testing.assert_close(got_r.float(), ref_r.float(), rtol=0, atol=0)
    torch.testing.assert_close(got_n.float(), ref_n.float(), rtol=2e-2, atol=2e-2)


def test_rms_norm_strided_rows():
    big = _rand(10, 3 * 512)
    x = big[:, 512:1024]
    w = _rand(512) + 1
    torch.testing.assert_close(ops.rms_norm(x, w, 1e-6).float(),
                               torch_ref.rms_norm(x.float(), w.float(), 1e-6), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("T,H", [(3, 768), (64, 4096)])
def test_layer_norm(T, H):
    x, w, b = _rand(T, H, dtype=torch.float16), _rand(H, dtype=torch.float16) + 1, _rand(H, dtype=torch.float16)
    ref = torch_ref.layer_norm(x.float(), w.float(), b.float(), 1e-5)
    torch.testing.assert_close(ops.layer_norm(x, w, b, 1e-5).float(), ref, rtol=1e-2, atol=1e-2)
    ref = torch_ref.layer_norm(x.float(), w.float(), None, 1e-5)
    torch.testing.assert_close(ops.layer_norm(x, w, None, 1e-5).float(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("T,F", [(1, 14336), (257, 3584), (4096, 512)])
def test_silu_and_mul(T, F):
    x = _rand(T, 2 * F)
    torch.testing.assert_close(ops.silu_and_mul(x).float(), torch_ref.silu_and_mul(x.float()),
                               rtol=2e-2, atol=2e-2)


def test_embedding_vocab_shard():
    w = _rand(1000, 256)
    ids = torch.tensor([0, 5, 999, 1000, 2500, 1999, 1001], device=DEV, dtype=torch.int32)
    got = ops.embedding(ids, w, vocab_start=1000, masked=True)
    ref = torch_ref.embedding(ids, w, 1000, True)
    torch.testing.assert_close(got, ref, rtol=0, atol=0)
    assert got[0].abs().sum() == 0 and got[3].abs().sum() > 0


@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (8, 8, 64, 16), (12, 2, 128, 32), (32, 32, 96, 32), (8, 4, 256, 16)])
def test_rope_and_cache_matches_reference(hq, hkv, d, bs):
    T, nb = 37, 12
    qkv = _rand(T, (hq + 2 * hkv) * d)
    cos_sin = torch_ref.build_cos_sin_cache(d, 4096, 500000.0).to(DEV)
    pos = torch.randint(0, 4096, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int32)
    slots[5] = -1                                       # padding token: no cache write
    kc = torch.zeros(nb, hkv, bs, d, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros(nb, hkv, d // 16, bs // 8, 16, 8, device=DEV, dtype=torch.bfloat16)
    kc_ref, vc_ref = kc.float(), vc.float()
    q_ref = torch_ref.rope_and_cache(pos, qkv.float().clone(), cos_sin, kc_ref, vc_ref, slots, hq, hkv, d)
    q = ops.rope_and_cache(pos, qkv, cos_sin, kc, vc, slots, hq, hkv, d)
    torch.testing.assert_close(q.float(), q_ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(kc.float(), kc_ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(vc.float(), vc_ref, rtol=0, atol=0)


@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (8, 8, 64, 16), (12, 1, 128, 32), (8, 4, 256, 16)])
def test_rope_and_cache_with_per_head_rms_norm(hq, hkv, d, bs):
    """Qwen3: q and k heads are RMS-normalised before the rotation, in the same kernel."""
    T, nb = 41, 12
    qkv = _rand(T, (hq + 2 * hkv) * d)
    qw, kw = (1.0 + 0.2 * torch.randn(d, device=DEV)).to(torch.bfloat16), \
        (1.0 + 0.2 * torch.randn(d, device=DEV)).to(torch.bfloat16)
    cos_sin = torch_ref.build_cos_sin_cache(d, 4096, 1e6).to(DEV)
    pos = torch.randint(0, 4096, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int32)
    kc = torch.zeros(nb, hkv, bs, d, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros(nb, hkv, d // 16, bs // 8, 16, 8, device=DEV, dtype=torch.bfloat16)
    kc_ref, vc_ref = kc.float(), vc.float()
    q_ref = torch_ref.rope_and_cache(pos, qkv.float().clone(), cos_sin, kc_ref, vc_ref, slots, hq, hkv, d,
                                     qw.float(), kw.float(), 1e-6)
    q = ops.rope_and_cache(pos, qkv, cos_sin, kc, vc, slots, hq, hkv, d, qw, kw, 1e-6)
    torch.testing.assert_close(q.float(), q_ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(kc.float(), kc_ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(vc.float(), vc_ref, rtol=0, atol=0)


@pytest.mark.parametrize("dtype,kv8", [(torch.bfloat16, False), (torch.float16, False), (torch.bfloat16, True)])
@pytest.mark.parametrize("hq,hkv,d,bs,ks,norm", [(32, 8, 128, 64, 4, False), (12, 1, 128, 32, 7, True),
                                                 (8, 4, 256, 16, 1, True), (8, 8, 64, 16, 9, False)])
def test_rope_and_cache_from_k_split_partial_sums(dtype, kv8, hq, hkv, d, bs, ks, norm):
    """The rotary + cache kernel adding up the projection's fp32 K-split slabs itself: the rotated q and
    both caches against the same kernel fed the summed, rounded qkv (and through it against the fp32
    reference the tests above pin that kernel to)."""
    from vllm_distributed_amd.ops import hip
    T, nb = 19, 12
    n = (hq + 2 * hkv) * d
    slabs = torch.randn(ks, T, n, device=DEV) / math.sqrt(ks)
    qkv = slabs.sum(0).to(dtype)
    qw = kw = None
    if norm:
        qw, kw = (1.0 + 0.2 * torch.randn(d, device=DEV)).to(dtype), (1.0 + 0.2 * torch.randn(d, device=DEV)).to(dtype)
    cos_sin = torch_ref.build_cos_sin_cache(d, 4096, 1e6).to(DEV)
    pos = torch.randint(0, 4096, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int32)
    slots[3] = -1
    cdt = torch.float8_e4m3fn if kv8 else dtype
    scales = (torch.rand(2, hkv, device=DEV) + 0.5) if kv8 else None

    def caches():
        return (torch.zeros(nb, hkv, bs, d, device=DEV, dtype=torch.float32).to(cdt),
                torch.zeros(nb, hkv, d // 16, bs // 8, 16, 8, device=DEV, dtype=torch.float32).to(cdt))
    kc0, vc0 = caches()
    q0 = hip.rope_and_cache(pos, qkv.clone(), cos_sin, kc0, vc0, slots, hq, hkv, d, qw, kw, 1e-6, scales).clone()
    kc1, vc1 = caches()
    q1 = hip.rope_and_cache_partials(pos, slabs, dtype, cos_sin, kc1, vc1, slots, hq, hkv, d, qw, kw, 1e-6, scales)
    torch.cuda.synchronize()
    assert q1.shape == (T, hq * d) and q1.dtype == dtype
    tol = dict(rtol=2e-2, atol=2e-2) if not kv8 else dict(rtol=0.13, atol=0.07)
    torch.testing.assert_close(q1.float(), q0.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(kc1.float(), kc0.float(), **tol)
    torch.testing.assert_close(vc1.float(), vc0.float(), **tol)
    # nearly every element is bit-identical (a different order of the fp32 additions flips a rounding now and then)
    assert (q1 == q0).float().mean().item() > 0.99
    assert (vc1.float() == vc0.float()).float().mean().item() > 0.99
    assert kc0.float().abs().sum().item() > 0 and vc0.float().abs().sum().item() > 0


def test_rotary_and_reshape_unfused():
    T, hq, hkv, d, bs, nb = 19, 4, 2, 64, 16, 8
    q, k, v = _rand(T, hq * d), _rand(T, hkv * d), _rand(T, hkv * d)
    cos_sin = torch_ref.build_cos_sin_cache(d, 512, 10000.0).to(DEV)
    pos = torch.randint(0, 512, (T,), device=DEV, dtype=torch.int32)
    qr, kr = q.float().clone(), k.float().clone()
    torch_ref.rotary_embedding(pos, qr, kr, cos_sin, d)
    ops.rotary_embedding(pos, q, k, cos_sin, d)
    torch.testing.assert_close(q.float(), qr, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(k.float(), kr, rtol=2e-2, atol=2e-2)
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int32)
    kc = torch.zeros(nb, hkv, bs, d, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros(nb, hkv, d // 16, bs // 8, 16, 8, device=DEV, dtype=torch.bfloat16)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    torch_ref.reshape_and_cache(k.view(T, hkv, d), v.view(T, hkv, d), kc_ref, vc_ref, slots)
    ops.reshape_and_cache(k.view(T, hkv, d), v.view(T, hkv, d), kc, vc, slots)
    assert torch.equal(kc, kc_ref) and torch.equal(vc, vc_ref)


def _make_paged(seq_lens, hkv, d, bs, dtype=torch.bfloat16, extra_blocks=3):
    """Random paged caches with a shuffled block table."""
    nblk = [(s + bs - 1) // bs for s in seq_lens]
    total = sum(nblk) + extra_blocks
    perm = torch.randperm(total).tolist()
    maxb = max(nblk) + 1
    bt = torch.zeros(len(seq_lens), maxb, dtype=torch.int32)
    it = iter(perm)
    for i, n in enumerate(nblk):
        for j in range(n):
            bt[i, j] = next(it)
    kc = _rand(total, hkv, bs, d, dtype=dtype)
    vc = _rand(total, hkv, d // 16, bs // 8, 16, 8, dtype=dtype)
    return kc, vc, bt.to(DEV)


@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (64, 8, 128, 64), (12, 12, 64, 16),
                                         (24, 2, 128, 32), (4, 2, 64, 128), (32, 32, 96, 32), (8, 4, 256, 16)])
@pytest.mark.parametrize("splits", [None, 0, 1, 4])          # 0 = the one-wave-per-sequence kernel
def test_paged_attention_decode(hq, hkv, d, bs, splits):
    from vllm_distributed_amd.ops import hip
    seq_lens = [1, 31, 32, 33, 64, 100, 517, 1000, 129, 64 * 7]
    kc, vc, bt = _make_paged(seq_lens, hkv, d, bs)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    q = _rand(len(seq_lens), hq, d)
    scale = 1.0 / math.sqrt(d)
    ref = torch_ref.paged_attention_decode(q.float(), kc.float(), vc.float(), bt, sl, scale)
    got = hip.paged_attention_decode(q, kc, vc, bt, sl, scale, max_seq_len=max(seq_lens), num_splits=splits)
    torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)


def test_paged_attention_decode_strided_q_and_padding_rows():
    from vllm_distributed_amd.ops import hip
    hq, hkv, d, bs = 32, 8, 128, 64
    seq_lens = [200, 0, 77]                     # a zero-length (graph padding) row
    kc, vc, bt = _make_paged([200, 1, 77], hkv, d, bs)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    qkv = _rand(3, (hq + 2 * hkv) * d)
    q = qkv[:, :hq * d].view(3, hq, d)          # row stride = packed qkv width
    got = hip.paged_attention_decode(q, kc, vc, bt, sl, 0.09, max_seq_len=200)
    ref = torch_ref.paged_attention_decode(q.float()[[0, 2]], kc.float(), vc.float(), bt[[0, 2]], sl[[0, 2]], 0.09)
    torch.testing.assert_close(got[[0, 2]].float(), ref, rtol=2e-2, atol=2e-2)
    assert got[1].abs().sum() == 0


def test_decode_softmax_rescale_branch_with_spiked_key():
    """A late key whose score dwarfs all earlier ones forces the running-max
    rescale of everything accumulated so far."""
    from vllm_distributed_amd.ops import hip
    hq, hkv, d, bs = 8, 1, 128, 64
    seq_lens = [700]
    kc, vc, bt = _make_paged(seq_lens, hkv, d, bs)
    q = _rand(1, hq, d)
    blk = int(bt[0, 600 // bs])
    kc[blk, 0, 600 % bs, :] = (q[0, 3].float() * 4).to(kc.dtype)      # huge dot product with head 3
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    ref = torch_ref.paged_attention_decode(q.float(), kc.float(), vc.float(), bt, sl, 0.5)
    for splits in (1, 3):
        got = hip.paged_attention_decode(q, kc, vc, bt, sl, 0.5, max_seq_len=700, num_splits=splits)
        torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (8, 8, 64, 16), (6, 2, 128, 32), (32, 32, 96, 32), (8, 4, 256, 16),
                                         (8, 2, 128, 24), (8, 2, 128, 128), (4, 2, 256, 64), (8, 2, 64, 256)])
def test_paged_attention_prefill_varlen_and_chunked(hq, hkv, d, bs):
    # (block sizes: below, at and above the LDS kernel's 64-key tile - it looks a block up per chunk or per tile -
    # and 24, not a power of two, which only the fragment-direct kernel addresses)
    # (q_len, seq_len): full prompts, a chunk continuing a cached prefix, a single token
    cases = [(5, 5), (128, 128), (130, 130), (40, 300), (1, 77), (257, 600), (33, 33)]
    seq_lens = [c[1] for c in cases]
    kc, vc, bt = _make_paged(seq_lens, hkv, d, bs)
    qsl = [0]
    for ql, _ in cases:
        qsl.append(qsl[-1] + ql)
    q = _rand(qsl[-1], hq, d)
    qsl_t = torch.tensor(qsl, device=DEV, dtype=torch.int32)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    ref = torch_ref.paged_attention(q.float(), kc.float(), vc.float(), bt, qsl_t, sl, scale)
    got = ops.paged_attention(q, kc, vc, bt, qsl_t, sl, scale, max(c[0] for c in cases), max(seq_lens))
    torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("window", [1, 16, 33, 100, 512, 5000])
@pytest.mark.parametrize("kv8", [False, True])
def test_sliding_window_attention_decode_and_prefill(window, kv8):
    """A query sees its last ``window`` keys only: both decode kernels, every split count, the
    fragment-direct and the LDS-staged prefill kernels (full prompts and chunks over a cached prefix),
    bf16 and e4m3 caches, windows shorter than a key group and longer than every context."""
    from vllm_distributed_amd.ops import hip
    hq, hkv, d, bs = 32, 8, 128, 64
    scale = 1.0 / math.sqrt(d)
    seq_lens = [1, 31, 33, 64, 100, 517, 1000, 129]
    kc, vc, bt = _make_paged(seq_lens, hkv, d, bs)
    kv_scales = None
    if kv8:
        kv_scales = (torch.rand(hkv, device=DEV) + 0.5, torch.rand(hkv, device=DEV) + 0.5)
        kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    q = _rand(len(seq_lens), hq, d)
    ref = torch_ref.paged_attention_decode(q.float(), kc, vc, bt, sl, scale, kv_scales, window)
    for splits in (None, 0, 1, 4):
        got = hip.paged_attention_decode(q, kc, vc, bt, sl, scale, max_seq_len=max(seq_lens), num_splits=splits,
                                         kv_scales=kv_scales, window=window)
        torch.testing.assert_close(got.float(), ref, rtol=3e-2, atol=3e-2)
    # prefill: (q_len, seq_len)
    for cases in ([(5, 5), (20, 300), (1, 77), (30, 30)],                   # every chunk <= 32 rows: fragment-direct
                  [(128, 128), (130, 130), (40, 300), (257, 600), (33, 33)]):    # LDS-staged kernel
        sls = [c[1] for c in cases]
        kc2, vc2, bt2 = _make_paged(sls, hkv, d, bs)
        if kv8:
            kc2, vc2 = kc2.to(torch.float8_e4m3fn), vc2.to(torch.float8_e4m3fn)
        qsl = [0]
        for ql, _ in cases:
            qsl.append(qsl[-1] + ql)
        q2 = _rand(qsl[-1], hq, d)
        qsl_t = torch.tensor(qsl, device=DEV, dtype=torch.int32)
        sl2 = torch.tensor(sls, device=DEV, dtype=torch.int32)
        ref2 = torch_ref.paged_attention(q2.float(), kc2, vc2, bt2, qsl_t, sl2, scale, kv_scales, window)
        got2 = ops.paged_attention(q2, kc2, vc2, bt2, qsl_t, sl2, scale, max(c[0] for c in cases), max(sls),
                                   kv_scales=kv_scales, window=window)
        torch.testing.assert_close(got2.float(), ref2, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (64, 8, 64, 16), (8, 8, 64, 16)])
@pytest.mark.parametrize("which", ["sinks", "alibi", "both+window"])
def test_attention_sinks_and_alibi_bias_decode_and_prefill(hq, hkv, d, bs, which):
    """Attention sinks (an extra per-head score in the softmax's denominator, large ones taking most of
    the mass) and ALiBi (``slope * (key - query)`` on the scores): both decode kernels at every split
    count and both prefill kernels against the fp32 reference."""
    from vllm_distributed_amd.ops import hip
    scale = 1.0 / math.sqrt(d)
    sinks = alibi = None
    window = 0
    if which != "alibi":
        sinks = torch.randn(hq, device=DEV) * 2
        sinks[0], sinks[-1] = 9.0, -30.0               # nearly all of the mass / no visible effect
    if which != "sinks":
        alibi = torch.tensor([2.0 ** (-8.0 * (i + 1) / hq) for i in range(hq)], device=DEV)
    if which == "both+window":
        window = 100
    seq_lens = [1, 31, 33, 64, 100, 517, 1000, 129]
    kc, vc, bt = _make_paged(seq_lens, hkv, d, bs)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    q = _rand(len(seq_lens), hq, d)
    ref = torch_ref.paged_attention_decode(q.float(), kc.float(), vc.float(), bt, sl, scale, None, window, 0.0,
                                           sinks, alibi)
    plain = torch_ref.paged_attention_decode(q.float(), kc.float(), vc.float(), bt, sl, scale, None, window)
    assert (ref - plain).abs().max().item() > 0.05           # the feature under test changes the result
    for splits in (None, 0, 1, 4):
        got = hip.paged_attention_decode(q, kc, vc, bt, sl, scale, max_seq_len=max(seq_lens), num_splits=splits,
                                         window=window, sinks=sinks, alibi_slopes=alibi)
        torch.testing.assert_close(got.float(), ref, rtol=3e-2, atol=3e-2)
    for cases in ([(5, 5), (20, 300), (1, 77), (30, 30)],                   # every chunk <= 32 rows: fragment-direct
                  [(128, 128), (130, 130), (40, 300), (257, 600), (33, 33)]):    # LDS-staged kernel where D = 128
        sls = [c[1] for c in cases]
        kc2, vc2, bt2 = _make_paged(sls, hkv, d, bs)
        qsl = [0]
        for ql, _ in cases:
            qsl.append(qsl[-1] + ql)
        q2 = _rand(qsl[-1], hq, d)
        qsl_t = torch.tensor(qsl, device=DEV, dtype=torch.int32)
        sl2 = torch.tensor(sls, device=DEV, dtype=torch.int32)
        ref2 = torch_ref.paged_attention(q2.float(), kc2.float(), vc2.float(), bt2, qsl_t, sl2, scale, None, window,
                                         0.0, sinks, alibi)
        got2 = ops.paged_attention(q2, kc2, vc2, bt2, qsl_t, sl2, scale, max(c[0] for c in cases), max(sls),
                                   window=window, sinks=sinks, alibi_slopes=alibi)
        torch.testing.assert_close(got2.float(), ref2, rtol=3e-2, atol=3e-2)


def test_mfma_operand_maps_with_exact_integer_data():
    """A = I style check with asymmetric data: one-hot queries pick out single
    K dimensions, V holds distinct small integers, so any swapped lane/row map
    in the kernels changes the result by whole numbers."""
    from vllm_distributed_amd.ops import hip
    hq, hkv, d, bs, S = 8, 1, 128, 64, 64
    kc = torch.zeros(1, hkv, bs, d, device=DEV, dtype=torch.bfloat16)
    vc = torch.zeros(1, hkv, d // 16, bs // 8, 16, 8, device=DEV, dtype=torch.bfloat16)
    # key j has a large value only in dimension j -> query one-hot(dim j) selects key j
    for j in range(S):
        kc[0, 0, j, j] = 64.0
    keys = torch.arange(S, device=DEV).float()
    dims = torch.arange(d, device=DEV).float()
    vt = ((dims[:, None] * 2 + keys[None, :] * 3) % 61).to(torch.bfloat16)         # [d, key], asymmetric
    vc[0, 0] = vt.view(d // 16, 16, bs // 8, 8).permute(0, 2, 1, 3)                 # fragment order
    q = torch.zeros(1, hq, d, device=DEV, dtype=torch.bfloat16)
    picks = [3, 17, 40, 63, 0, 9, 31, 32]
    for h, j in enumerate(picks):
        q[0, h, j] = 8.0
    bt = torch.zeros(1, 1, device=DEV, dtype=torch.int32)
    sl = torch.tensor([S], device=DEV, dtype=torch.int32)
    got = hip.paged_attention_decode(q, kc, vc, bt, sl, 1.0, max_seq_len=S, num_splits=1).float()
    for h, j in enumerate(picks):
        torch.testing.assert_close(got[0, h], vt[:, j].float(), rtol=0, atol=1e-2)


@pytest.mark.parametrize("window", [0, 48])
def test_attention_score_soft_cap(window):
    """softcap * tanh(score / softcap) before the softmax (Gemma-2), with scores large enough for the cap
    to bite: both decode kernels, every split count, both prefill kernels, with and without a window."""
    from vllm_distributed_amd.ops import hip
    hq, hkv, d, bs = 16, 4, 128, 32
    scale, cap = 0.5, 6.0
    seq_lens = [1, 31, 100, 517, 129]
    kc, vc, bt = _make_paged(seq_lens, hkv, d, bs)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    q = _rand(len(seq_lens), hq, d)
    ref = torch_ref.paged_attention_decode(q.float(), kc.float(), vc.float(), bt, sl, scale, None, window, cap)
    plain = torch_ref.paged_attention_decode(q.float(), kc.float(), vc.float(), bt, sl, scale, None, window)
    assert (ref - plain).abs().max() > 0.05              # the cap changes the answer
    for splits in (None, 0, 1, 4):
        got = hip.paged_attention_decode(q, kc, vc, bt, sl, scale, max_seq_len=max(seq_lens), num_splits=splits,
                                         window=window, softcap=cap)
        torch.testing.assert_close(got.float(), ref, rtol=3e-2, atol=3e-2)
    for cases in ([(5, 5), (20, 300), (30, 30)], [(128, 128), (257, 600), (40, 300)]):
        sls = [c[1] for c in cases]
        kc2, vc2, bt2 = _make_paged(sls, hkv, d, bs)
        qsl = [0]
        for ql, _ in cases:
            qsl.append(qsl[-1] + ql)
        q2 = _rand(qsl[-1], hq, d)
        qsl_t = torch.tensor(qsl, device=DEV, dtype=torch.int32)
        sl2 = torch.tensor(sls, device=DEV, dtype=torch.int32)
        ref2 = torch_ref.paged_attention(q2.float(), kc2.float(), vc2.float(), bt2, qsl_t, sl2, scale, None, window, cap)
        got2 = ops.paged_attention(q2, kc2, vc2, bt2, qsl_t, sl2, scale, max(c[0] for c in cases), max(sls),
                                   window=window, softcap=cap)
        torch.testing.assert_close(got2.float(), ref2, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("d,hq,hkv,bs", [(64, 16, 2, 16), (256, 8, 4, 32), (256, 4, 4, 64)])
@pytest.mark.parametrize("kv8", [False, True])
def test_lds_prefill_kernel_at_head_sizes_64_and_256(d, hq, hkv, bs, kv8):
    """The LDS-staged prefill kernel is built for head sizes 64, 128 and 256: window + soft cap (Gemma-2's
    combination at 256), sinks + ALiBi (the bias form of the kernel), bf16 and e4m3 caches, full prompts and
    chunks over a cached prefix, several key tiles per query tile and ragged last tiles."""
    scale = 1.0 / math.sqrt(d)
    cases = [(128, 128), (130, 130), (40, 300), (257, 600), (33, 33), (64, 1000)]
    sls = [c[1] for c in cases]
    kc, vc, bt = _make_paged(sls, hkv, d, bs)
    kv_scales = None
    if kv8:
        kv_scales = (torch.rand(hkv, device=DEV) + 0.5, torch.rand(hkv, device=DEV) + 0.5)
        kc, vc = kc.to(torch.float8_e4m3fn), vc.to(torch.float8_e4m3fn)
    qsl = [0]
    for ql, _ in cases:
        qsl.append(qsl[-1] + ql)
    q = _rand(qsl[-1], hq, d)
    qsl_t = torch.tensor(qsl, device=DEV, dtype=torch.int32)
    sl = torch.tensor(sls, device=DEV, dtype=torch.int32)
    sinks = torch.randn(hq, device=DEV) * 2
    alibi = torch.tensor([2.0 ** (-8.0 * (i + 1) / hq) for i in range(hq)], device=DEV)
    plain = torch_ref.paged_attention(q.float(), kc, vc, bt, qsl_t, sl, scale, kv_scales)
    for window, cap, snk, slopes in ((0, 0.0, None, None), (100, 0.0, None, None), (0, 4.0, None, None),
                                     (70, 4.0, None, None), (0, 0.0, sinks, None), (0, 0.0, None, alibi),
                                     (100, 0.0, sinks, alibi)):
        ref = torch_ref.paged_attention(q.float(), kc, vc, bt, qsl_t, sl, scale, kv_scales, window, cap, snk, slopes)
        if window or cap or snk is not None or slopes is not None:
            assert (ref - plain).abs().max().item() > 0.02           # the feature under test changes the result
        got = ops.paged_attention(q, kc, vc, bt, qsl_t, sl, scale, max(c[0] for c in cases), max(sls),
                                  kv_scales=kv_scales, window=window, softcap=cap, sinks=snk, alibi_slopes=slopes)
        torch.testing.assert_close(got.float(), ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_argmax(dtype):
    logits = _rand(9, 128256, dtype=dtype)
    logits[3, 777] = 50.0
    logits[4, 10] = 60.0
    logits[4, 5] = 60.0            # tie -> lowest index
    got = ops.argmax(logits)
    ref = logits.float().argmax(-1)
    assert got[3] == 777 and got[4] == 5
    assert torch.equal(got.long(), torch.where(torch.arange(9, device=DEV) == 4, torch.tensor(5, device=DEV), ref))


def test_sample_matches_reference_on_separated_distributions():
    torch.manual_seed(0)
    B, V = 16, 5000
    # well separated probabilities: random permutation of a geometric-like ladder
    base = torch.linspace(8.0, -30.0, V)
    logits = torch.stack([base[torch.randperm(V)] for _ in range(B)]).to(DEV)
    temperature = torch.tensor([0.0, 1.0, 0.7, 1.3] * 4, device=DEV)
    top_k = torch.tensor([-1, -1, 50, 5, 1, 100, -1, 20] * 2, device=DEV, dtype=torch.int32)
    top_p = torch.tensor([1.0, 0.9, 1.0, 0.5, 1.0, 0.95, 0.3, 0.8] * 2, device=DEV)
    min_p = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.05, 0.0, 0.1] * 2, device=DEV)
    for u in (0.013, 0.37, 0.61, 0.93, 0.999):
        uniform = torch.full((B,), u, device=DEV)
        got = ops.sample(logits, temperature, top_k, top_p, min_p, uniform)
        ref = torch_ref.sample(logits, temperature, top_k, top_p, min_p, uniform)
        assert torch.equal(got.long(), ref.long()), (u, got.tolist(), ref.tolist())


def test_sample_respects_filters_statistically():
    torch.manual_seed(1)
    V = 128256
    logits = _rand(4, V, dtype=torch.float32, scale=3.0)
    temperature = torch.ones(4, device=DEV)
    top_k = torch.tensor([10, -1, 10, -1], device=DEV, dtype=torch.int32)
    top_p = torch.tensor([1.0, 0.5, 0.5, 1.0], device=DEV)
    min_p = torch.tensor([0.0, 0.0, 0.0, 0.2], device=DEV)
    probs = torch.softmax(logits, -1)
    sp, si = probs.sort(-1, descending=True)
    for u in torch.linspace(0.01, 0.99, 25).tolist():
        tok = ops.sample(logits, temperature, top_k, top_p, min_p, torch.full((4,), u, device=DEV)).long()
        rank = [(si[i] == tok[i]).nonzero().item() for i in range(4)]
        assert rank[0] < 10 and rank[2] < 10
        assert sp[1, :rank[1]].sum() < 0.5 + 1e-4 and sp[2, :rank[2]].sum() < 0.5 + 1e-4
        assert probs[3, tok[3]] >= 0.2 * sp[3, 0] - 1e-7


@pytest.mark.parametrize("sp", ["0", "force"])
def test_microbatch_overlap_path_matches_plain_decode(monkeypatch, sp):
    """The two-stream micro-batch schedule (forced on at TP=1, where the
    all-reduce is an identity) must reproduce the plain forward: checks the
    stream / event choreography and the metadata slicing, eagerly and as a
    captured hipGraph.  Hidden states are compared (token ids of a random-init
    model flip on bf16 GEMM tiling noise)."""
    from vllm_distributed_amd import LLM
    from vllm_distributed_amd.models.layers.attention import AttentionMetadata
    from vllm_distributed_amd.parallel import state as pstate
    monkeypatch.setenv("VDAMD_TP_OVERLAP", "force")
    # "force": the sequence-parallel variant (reduce-scatter / shard norm / all-gather chains)
    monkeypatch.setenv("VDAMD_SEQUENCE_PARALLEL", sp)
    pstate.destroy_model_parallel()
    llm = LLM(model="llama-tiny", device="cuda", dtype="bfloat16", max_model_len=512, max_num_seqs=64,
              gpu_memory_utilization=0.3, enable_prefix_caching=False, enforce_eager=True,
              hf_overrides={"head_dim": 128, "num_attention_heads": 4, "num_key_value_heads": 2,
                            "hidden_size": 512, "intermediate_size": 1024, "num_hidden_layers": 6})
    runner = llm.engine.core.executor.driver.worker.model_runner
    assert runner.comm_stream is not None
    bs, B = runner.block_size, 48
    g = torch.Generator(device="cpu").manual_seed(0)
    for k, v in runner.kv_caches:
        k.copy_(_rand(*k.shape) * 0.5)
        v.copy_(_rand(*v.shape) * 0.5)
    seq_lens = torch.randint(1, 400, (B,), generator=g).to(torch.int32)
    nblk = (seq_lens + bs - 1) // bs
    perm = torch.randperm(runner.num_blocks, generator=g)
    bt = torch.zeros(B, int(nblk.max()), dtype=torch.int32)
    off = 0
    for i in range(B):
        bt[i, :nblk[i]] = perm[off:off + nblk[i]]
        off += int(nblk[i])
    pos = (seq_lens - 1)
    slots = torch.tensor([int(bt[i, pos[i] // bs]) * bs + int(pos[i]) % bs for i in range(B)], dtype=torch.int32)
    md = AttentionMetadata(
        num_decodes=B, num_decode_tokens=B, num_prefills=0, num_prefill_tokens=0,
        slot_mapping=slots.to(DEV), block_table=bt.to(DEV), seq_lens=seq_lens.to(DEV),
        query_start_loc=torch.arange(B + 1, dtype=torch.int32, device=DEV), max_query_len=1,
        max_seq_len=int(seq_lens.max()), max_decode_seq_len=int(seq_lens.max()),
        decode_workspace=runner.decode_workspace)
    ids = torch.randint(0, 500, (B,), generator=g).to(torch.int32).to(DEV)
    posd = pos.to(torch.int32).to(DEV)
    with torch.inference_mode():
        stream = runner.comm_stream
        runner.comm_stream = None
        monkeypatch.setenv("VDAMD_SEQUENCE_PARALLEL", "0")
        ref = runner._forward_decode(md, ids, posd).float().clone()
        monkeypatch.setenv("VDAMD_SEQUENCE_PARALLEL", sp)
        if sp == "force":        # plain (single-stream) sequence-parallel forward
            torch.testing.assert_close(runner._forward_decode(md, ids, posd).float(), ref, rtol=3e-2, atol=3e-2)
        runner.comm_stream = stream
        assert runner._overlap_bounds(B) == [(0, 24), (24, 48)]
        for _ in range(3):
            got = runner._forward_decode(md, ids, posd).float()
            torch.testing.assert_close(got, ref, rtol=3e-2, atol=3e-2)
        # the same schedule captured and replayed as a graph
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = runner._forward_decode(md, ids, posd)
        for _ in range(3):
            out.zero_()
            graph.replay()
            torch.cuda.synchronize()
            torch.testing.assert_close(out.float(), ref, rtol=3e-2, atol=3e-2)
    llm.shutdown()
    pstate.destroy_model_parallel()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [(1, 256, 512), (7, 1040, 1024), (16, 1280, 8192), (33, 512, 2048),
                                   (64, 2064, 1024), (100, 768, 3584), (128, 1024, 4096)])
def test_skinny_gemm_matches_fp32_reference(dtype, m, n, k):
    from vllm_distributed_amd.ops import hip
    assert ops.hip_available()
    x = _rand(m, k).to(dtype)
    w = (_rand(n, k) * 0.05).to(dtype)
    ref = x.float() @ w.float().t()
    tol = ref.abs().max().item() * (2 ** -7 if dtype == torch.bfloat16 else 2 ** -9)
    for ntw, ksplit in ((None, None), (1, 1), (2, 2), (4, 1), (1, 4)):
        y = hip.skinny_gemm(x, w, ntw, ksplit)
        assert y.shape == (m, n)
        assert (y.float() - ref).abs().max().item() <= tol, (ntw, ksplit)
    # strided views (a column slice of a fused activation, a row slice of a weight)
    xb = _rand(m, 2 * k).to(dtype)
    y = hip.skinny_gemm(xb[:, k:], w[16:])
    ref = xb[:, k:].float() @ w[16:].float().t()
    assert (y.float() - ref).abs().max().item() <= tol


# ------------------------------------------------------------- MoE routing
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("tokens,experts,k,renorm", [(1, 160, 8, True), (7, 8, 2, True), (300, 64, 4, False),
                                                     (33, 512, 16, True), (5, 128, 1, False)])
def test_moe_topk_matches_softmax_topk(dtype, tokens, experts, k, renorm):
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    torch.manual_seed(tokens + experts)
    logits = (torch.randn(tokens, experts, device=DEV) * 2).to(dtype)
    topw, topi = hip.moe_topk(logits, k, renorm)
    probs = torch.softmax(logits.float(), dim=-1)
    rw, ri = probs.topk(k, dim=-1)
    if renorm:
        rw = rw / rw.sum(-1, keepdim=True)
    # same experts (as sets: equal probabilities may order differently) and the same weights per expert
    assert torch.equal(topi.long().sort(-1).values, ri.sort(-1).values)
    got = torch.zeros_like(probs).scatter_(1, topi.long(), topw)
    want = torch.zeros_like(probs).scatter_(1, ri, rw)
    assert torch.allclose(got, want, atol=1e-6, rtol=1e-5)
    # weights come out in descending order like torch.topk
    assert (topw[:, :-1] >= topw[:, 1:]).all()


@pytest.mark.parametrize("tokens,experts,k,tile_rows,ep", [(1, 160, 8, 16, None), (64, 16, 4, 16, None),
                                                           (1000, 160, 8, 64, None), (300, 16, 4, 32, (4, 8)),
                                                           (9000, 64, 8, 128, None)])
def test_moe_align_and_combine(tokens, experts, k, tile_rows, ep):
    """The slot sort is a permutation grouped by expert, the tile table covers
    exactly the rows of each (local) expert, and combine is the weighted sum."""
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    gen = torch.Generator(device=DEV).manual_seed(tokens)
    probs = torch.rand(tokens, experts, device=DEV, generator=gen)
    topw, topi = probs.topk(k, dim=-1)
    topi = topi.to(torch.int32).contiguous()
    topw = topw.float().contiguous()
    start, local = ep if ep else (0, -1)
    src_row, inv, tiles = hip.moe_align(topi, experts, tile_rows, start, local)
    slots = tokens * k
    flat = topi.reshape(-1).long()
    assert torch.equal(inv.long().sort().values, torch.arange(slots, device=DEV))          # a permutation
    assert torch.equal(src_row.long()[inv.long()], torch.arange(slots, device=DEV) // k)   # rows point at their token
    sorted_experts = torch.empty(slots, dtype=torch.long, device=DEV)
    sorted_experts[inv.long()] = flat
    assert (sorted_experts[1:] >= sorted_experts[:-1]).all()                              # grouped by expert
    t = tiles.cpu().tolist()
    se = sorted_experts.cpu()
    covered = torch.zeros(slots, dtype=torch.int32)
    for g, r0, rows in t:
        assert 0 <= rows <= tile_rows
        if rows == 0:
            continue
        e = g + (start if ep else 0)
        assert (se[r0:r0 + rows] == e).all()
        covered[r0:r0 + rows] += 1
    is_local = torch.ones(slots, dtype=torch.bool) if not ep else (se >= start) & (se < start + local)
    assert torch.equal(covered, is_local.to(torch.int32))
    # combine
    h = 256
    y = torch.randn(slots, h, device=DEV, generator=gen).to(torch.bfloat16)
    out = hip.moe_combine(y, topw, topi, inv, start, local)
    rows = y.float()[inv.long()].view(tokens, k, h)
    w = topw.clone()
    if ep:
        w = torch.where((topi >= start) & (topi < start + local), w, torch.zeros_like(w))
    want = (rows * w.unsqueeze(-1)).sum(1)
    assert (out.float() - want).abs().max().item() <= 2 ** -7 * want.abs().max().item() + 1e-6
    # combine + residual add + RMSNorm in one launch: the same sum (rounded to 16 bits as above) added into
    # the residual, then normalised, against fp32 arithmetic on the separate kernel's output
    for hh in (256, 6144):
        y = torch.randn(slots, hh, device=DEV, generator=gen).to(torch.bfloat16)
        res = torch.randn(tokens, hh, device=DEV, generator=gen).to(torch.bfloat16)
        gamma = (1 + 0.1 * torch.randn(hh, device=DEV, generator=gen)).to(torch.bfloat16)
        summed = hip.moe_combine(y, topw, topi, inv, start, local)
        want_res = (summed.float() + res.float()).to(torch.bfloat16)
        want_out = want_res.float() * torch.rsqrt(want_res.float().pow(2).mean(-1, keepdim=True) + 1e-6) * gamma.float()
        got_out, got_res = hip.moe_combine_add_rms_norm(y, topw, topi, inv, start, local, res, gamma, 1e-6)
        assert got_res.data_ptr() == res.data_ptr()
        assert (got_res.float() - want_res.float()).abs().max().item() <= 2 ** -7 * want_res.abs().max().item()
        assert (got_out.float() - want_out).abs().max().item() <= 2 ** -7 * want_out.abs().max().item()


@pytest.mark.parametrize("ldtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("tokens,experts,k,tile_rows,renorm,ep", [(1, 160, 8, 16, True, None), (16, 160, 8, 16, True, None),
                                                                   (7, 8, 2, 32, False, None), (16, 512, 16, 16, True, None),
                                                                   (5, 16, 4, 16, True, (4, 8)), (3, 64, 1, 64, False, None)])
def test_moe_route_small_is_topk_align_and_gather_in_one_launch(ldtype, tokens, experts, k, tile_rows, renorm, ep):
    """The decode-batch routing kernel: the picks and weights of ``moe_topk``, the tile table of ``moe_align``
    (it depends on the per-expert counts only), a slot sort that is a permutation grouped by expert with the
    rows of one expert in slot order, and ``xs`` = the token rows in that order."""
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    torch.manual_seed(tokens * 1000 + experts)
    logits = (torch.randn(tokens, experts, device=DEV) * 2).to(ldtype)
    x = torch.randn(tokens, 1032, device=DEV).to(torch.bfloat16)[:, :1024]       # strided rows
    start, local = ep if ep else (0, -1)
    topw, topi, inv, tiles, xs = hip.moe_route_small(logits, x, k, renorm, tile_rows, start, local)
    rw, ri = hip.moe_topk(logits, k, renorm)
    assert torch.equal(topi, ri) and torch.equal(topw, rw)
    _, _, rtiles = hip.moe_align(ri, experts, tile_rows, start, local)
    assert torch.equal(tiles, rtiles)
    slots = tokens * k
    flat = topi.reshape(-1).long()
    assert torch.equal(inv.long().sort().values, torch.arange(slots, device=DEV))
    sorted_experts = torch.empty(slots, dtype=torch.long, device=DEV)
    sorted_experts[inv.long()] = flat
    assert (sorted_experts[1:] >= sorted_experts[:-1]).all()
    # slot order within an expert: the sort is the stable one
    assert torch.equal(inv.long(), torch.argsort(torch.argsort(flat, stable=True), stable=True))
    assert torch.equal(xs[inv.long()], x[torch.arange(slots, device=DEV) // k])


@pytest.mark.parametrize("tokens,experts,k", [(1, 160, 8), (16, 160, 8), (4, 64, 2)])
def test_moe_route_small_adds_up_router_gemm_partial_sums(tokens, experts, k):
    """The router GEMM's K-split partial sums go to the routing kernel unreduced: same picks as routing on
    their sum, and ``skinny_gemm_partials`` really is the unreduced GEMM."""
    from vllm_distributed_amd.ops import hip
    ops.hip_available()
    torch.manual_seed(tokens + experts)
    x = torch.randn(tokens, 2048, device=DEV).to(torch.bfloat16)
    w = (torch.randn(experts, 2048, device=DEV) * 0.05).to(torch.bfloat16)
    parts = hip.skinny_gemm_partials(x, w)
    assert parts is not None and parts.dim() == 3 and parts.shape[1:] == (tokens, experts)
    total = parts.sum(0)
    assert (total - x.float() @ w.float().t()).abs().max().item() < 2e-2
    topw, topi, inv, tiles, xs = hip.moe_route_small(parts, x, k, True, 16)
    rw, ri = hip.moe_topk(total.contiguous(), k, True)
    assert torch.equal(topi, ri) and torch.allclose(topw, rw, atol=1e-6)
    assert torch.equal(xs[inv.long()], x[torch.arange(tokens * k, device=DEV) // k])


# ---------------------------------------------------------------- e4m3 KV cache
FP8 = torch.float8_e4m3fn


def _fp8_caches(seq_lens, hkv, d, bs):
    kc, vc, bt = _make_paged(seq_lens, hkv, d, bs)
    ks = torch.rand(hkv, device=DEV) * 0.7 + 0.3            # deliberately not powers of two
    vs = torch.rand(hkv, device=DEV) * 1.5 + 0.5
    kc8 = (kc.float() / ks.view(1, -1, 1, 1)).clamp(-448, 448).to(FP8)
    vc8 = (vc.float() / vs.view(1, -1, 1, 1, 1, 1)).clamp(-448, 448).to(FP8)
    # what the kernels must see: the stored bytes widened exactly, times the head's scale
    kref = kc8.float() * ks.view(1, -1, 1, 1)
    vref = vc8.float() * vs.view(1, -1, 1, 1, 1, 1)
    return kc8, vc8, bt, (ks, vs), kref, vref


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (64, 8, 128, 64), (12, 12, 64, 16), (24, 2, 128, 32), (32, 32, 96, 32), (8, 4, 256, 16)])
@pytest.mark.parametrize("splits", [None, 0, 1, 4])
def test_paged_attention_decode_fp8_cache(dtype, hq, hkv, d, bs, splits):
    from vllm_distributed_amd.ops import hip
    seq_lens = [1, 31, 33, 64, 100, 517, 1000, 64 * 7]
    kc8, vc8, bt, scales, kref, vref = _fp8_caches(seq_lens, hkv, d, bs)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    q = _rand(len(seq_lens), hq, d, dtype=dtype)
    scale = 1.0 / math.sqrt(d)
    ref = torch_ref.paged_attention_decode(q.float(), kref, vref, bt, sl, scale)
    got = hip.paged_attention_decode(q, kc8, vc8, bt, sl, scale, max_seq_len=max(seq_lens), num_splits=splits,
                                     kv_scales=scales)
    torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)
    # the CPU emulation of the same cache agrees too
    emu = torch_ref.paged_attention_decode(q.float(), kc8, vc8, bt, sl, scale, scales)
    torch.testing.assert_close(emu, ref, rtol=1e-4, atol=1e-4)
    # no scales = 1.0
    ref1 = torch_ref.paged_attention_decode(q.float(), kc8.float(), vc8.float(), bt, sl, scale)
    got1 = hip.paged_attention_decode(q, kc8, vc8, bt, sl, scale, max_seq_len=max(seq_lens), num_splits=splits)
    torch.testing.assert_close(got1.float(), ref1, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (8, 8, 64, 16), (6, 2, 128, 32), (32, 32, 96, 32), (8, 4, 256, 16)])
def test_paged_attention_prefill_fp8_cache(hq, hkv, d, bs):
    cases = [(5, 5), (128, 128), (130, 130), (40, 300), (1, 77), (257, 600), (33, 33)]
    seq_lens = [c[1] for c in cases]
    kc8, vc8, bt, scales, kref, vref = _fp8_caches(seq_lens, hkv, d, bs)
    qsl = [0]
    for ql, _ in cases:
        qsl.append(qsl[-1] + ql)
    q = _rand(qsl[-1], hq, d)
    qsl_t = torch.tensor(qsl, device=DEV, dtype=torch.int32)
    sl = torch.tensor(seq_lens, device=DEV, dtype=torch.int32)
    scale = 1.0 / math.sqrt(d)
    ref = torch_ref.paged_attention(q.float(), kref, vref, bt, qsl_t, sl, scale)
    got = ops.paged_attention(q, kc8, vc8, bt, qsl_t, sl, scale, max(c[0] for c in cases), max(seq_lens),
                              kv_scales=scales)
    torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("hq,hkv,d,bs", [(32, 8, 128, 64), (8, 8, 64, 16)])
def test_rope_and_cache_fp8_cache(hq, hkv, d, bs):
    """The write side: rotated K and V land in the cache as e4m3 of value / scale.  A value that sits
    on a rounding boundary may land one e4m3 step (2^-3 relative) from the fp32 reference's."""
    T, nb = 37, 12
    qkv = _rand(T, (hq + 2 * hkv) * d)
    cos_sin = torch_ref.build_cos_sin_cache(d, 4096, 500000.0).to(DEV)
    pos = torch.randint(0, 4096, (T,), device=DEV, dtype=torch.int32)
    slots = torch.randperm(nb * bs, device=DEV)[:T].to(torch.int32)
    slots[5] = -1
    ks = torch.rand(hkv, device=DEV) * 0.7 + 0.3
    vs = torch.rand(hkv, device=DEV) * 1.5 + 0.5
    kc = torch.zeros(nb, hkv, bs, d, device=DEV, dtype=FP8)
    vc = torch.zeros(nb, hkv, d // 16, bs // 8, 16, 8, device=DEV, dtype=FP8)
    kc_ref, vc_ref = torch.zeros_like(kc), torch.zeros_like(vc)
    q_ref = torch_ref.rope_and_cache(pos, qkv.float().clone(), cos_sin, kc_ref, vc_ref, slots, hq, hkv, d,
                                     kv_scales=(ks, vs))
    q = ops.rope_and_cache(pos, qkv, cos_sin, kc, vc, slots, hq, hkv, d, kv_scales=(ks, vs))
    torch.testing.assert_close(q.float(), q_ref, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(kc.float(), kc_ref.float(), rtol=0.13, atol=2e-2)
    torch.testing.assert_close(vc.float(), vc_ref.float(), rtol=0.13, atol=2e-2)
    assert (kc.float() != 0).any() and (vc.float() != 0).any()
    # V is not rotated: bf16 -> e4m3 of the same number must agree bit for bit almost everywhere
    same = (vc.view(torch.uint8) == vc_ref.view(torch.uint8)).float().mean().item()
    assert same > 0.98, same
    # the unfused write path
    k = _rand(T, hkv, d)
    v = _rand(T, hkv, d)
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    kr2, vr2 = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.reshape_and_cache(k, v, kc2, vc2, slots, (ks, vs))
    torch_ref.reshape_and_cache(k.float(), v.float(), kr2, vr2, slots, (ks, vs))
    torch.testing.assert_close(kc2.float(), kr2.float(), rtol=0.13, atol=2e-2)
    torch.testing.assert_close(vc2.float(), vr2.float(), rtol=0.13, atol=2e-2)


def test_fp8_kv_cache_engine_matches_bf16_greedy():
    """End to end: the same prompts through a bf16-cache engine and an e4m3-cache engine (prefill,
    chunk continuation and graph-captured decode all read the narrow cache).  Hidden states of a
    random-init model sit on near-ties, so the check is on the logprob of the bf16 run's tokens."""
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.parallel import state as pstate
    prompts = [{"prompt_token_ids": [(7 * i + 3 * j) % 900 + 10 for j in range(20 + 3 * i)]} for i in range(8)]
    sp = SamplingParams(temperature=0.0, max_tokens=8, logprobs=1)
    outs = {}
    for kvd in ("auto", "fp8"):
        pstate.destroy_model_parallel()
        llm = LLM(model="llama-tiny", device="cuda", dtype="bfloat16", kv_cache_dtype=kvd,
                  gpu_memory_utilization=0.3, max_model_len=256, max_num_seqs=16, seed=0,
                  hf_overrides={"head_dim": 128, "num_attention_heads": 4, "num_key_value_heads": 2,
                                "hidden_size": 512, "intermediate_size": 1024, "num_hidden_layers": 4})
        runner = llm.engine.core.executor.driver.worker.model_runner
        assert runner.kv_caches[0][0].dtype == (FP8 if kvd == "fp8" else torch.bfloat16)
        outs[kvd] = llm.generate(prompts, sp)
        if kvd == "fp8":
            assert (runner.kv_caches[0][0].view(torch.uint8) != 0).any()
        llm.shutdown() if hasattr(llm, "shutdown") else None
        del llm, runner
    first_same = sum(a.outputs[0].token_ids[0] == b.outputs[0].token_ids[0]
                     for a, b in zip(outs["auto"], outs["fp8"]))
    assert first_same >= 6, [(a.outputs[0].token_ids, b.outputs[0].token_ids)
                             for a, b in zip(outs["auto"], outs["fp8"])]
    for a, b in zip(outs["auto"], outs["fp8"]):
        ta, tb = a.outputs[0].token_ids[0], b.outputs[0].token_ids[0]
        if ta == tb:
            la, lb = a.outputs[0].logprobs[0][ta], b.outputs[0].logprobs[0][tb]
            la, lb = getattr(la, "logprob", la), getattr(lb, "logprob", lb)
            assert abs(la - lb) < 0.25, (la, lb)


# ------------------------------------------------------- decode GEMM kernels
def _padded_weight(n, k, pad, dtype=torch.bfloat16):
    buf = torch.zeros(n, k + pad, device=DEV, dtype=dtype)
    buf[:, :k] = (torch.randn(n, k, device=DEV) * 0.05).to(dtype)
    return buf[:, :k]


@pytest.mark.parametrize("m", [1, 3, 8])
def test_wide_outputs_at_a_few_rows_take_the_weight_streaming_kernel(m, monkeypatch):
    """``ops.gate_up_silu`` and ``ops.linear`` hand up to 8 rows into a very wide output to
    ``stream_gemm`` (activation in its epilogue): results against fp32, and the routing itself."""
    from vllm_distributed_amd.ops import hip
    n, k = 28672, 1024 * 2
    x = _rand(m, k)
    buf = _rand(n, k + 128, scale=0.03)
    w = buf[:, :k]                                      # padded rows, as the runner lays dense weights out
    calls = []
    real = hip.stream_gemm
    monkeypatch.setattr(hip, "stream_gemm", lambda *a, **kw: (calls.append(kw.get("silu_mul", False)), real(*a, **kw))[1])
    got = ops.gate_up_silu(x, w)
    assert calls == [True] and got.shape == (m, n // 2)
    y = torch.nn.functional.linear(x.float(), w.float())
    want = torch.nn.functional.silu(y[:, :n // 2]) * y[:, n // 2:]
    assert (got.float() - want).abs().max().item() <= 0.02 * want.abs().max().item() + 1e-3
    got = ops.linear(x, w)
    assert calls == [True, False]
    assert (got.float() - y).abs().max().item() <= 0.02 * y.abs().max().item() + 1e-3
    assert ops.gate_up_silu(_rand(9, k), w) is None                      # beyond 8 rows: not this kernel
    assert not hip.stream_profitable(4, 7168, 8192)                       # a TP=8 shard: the library is ahead



@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,pad", [(1, 4096, 4096, 0), (3, 1024, 8192, 128), (16, 6144, 4096, 128),
                                       (7, 48, 256, 0), (16, 2064, 1152, 8), (2, 28672, 1024, 128)])
def test_stream_gemm_matches_fp32(dtype, m, n, k, pad):
    from vllm_distributed_amd.ops import hip
    x = _rand(m, k, dtype=dtype)
    w = _padded_weight(n, k, pad, dtype)
    ref = x.float() @ w.float().t()
    tol = dict(rtol=2e-2, atol=2e-2 * max(1.0, ref.abs().max().item()) / 4)
    for ntw, ks, depth, nt in [(None, None, None, False), (1, 1, 2, True), (2, 2, 4, False), (2, 1, 3, True)]:
        if ks is not None and ks > k // 128:
            continue
        got = hip.stream_gemm(x, w, ntw=ntw, ksplit=ks, depth=depth, nt=nt)
        torch.testing.assert_close(got.float(), ref, **tol)
    res = _rand(m, n, dtype=dtype)
    got = hip.stream_gemm(x, w, residual=res)
    torch.testing.assert_close(got.float(), ref + res.float(), **tol)
    got = hip.stream_gemm(x, w, residual=res, ksplit=2) if k >= 256 else got
    torch.testing.assert_close(got.float(), ref + res.float(), **tol)
    if n % 32 == 0:
        act = torch_ref.silu_and_mul(ref)
        for ks in (1, 2):
            if ks > k // 128:
                continue
            got = hip.stream_gemm(x, w, silu_mul=True, ksplit=ks)
            torch.testing.assert_close(got.float(), act, rtol=3e-2, atol=3e-2 * max(1.0, act.abs().max().item()) / 4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,pad", [(256, 2048, 2048, 128), (255, 1040, 1024, 0), (100, 4096, 4096, 128),
                                       (17, 512, 8192, 128), (64, 272, 512, 8), (300, 1024, 1024, 128),
                                       (48, 8192, 1024, 0)])
def test_mid_gemm_matches_fp32(dtype, m, n, k, pad):
    from vllm_distributed_amd.ops import hip
    x = _rand(m, k, dtype=dtype)
    w = _padded_weight(n, k, pad, dtype)
    ref = x.float() @ w.float().t()
    tol = dict(rtol=2e-2, atol=2e-2 * max(1.0, ref.abs().max().item()) / 4)
    got = hip.mid_gemm(x, w)
    torch.testing.assert_close(got.float(), ref, **tol)
    for bm, bn in [(256, 256), (256, 128), (128, 256), (128, 128), (64, 256), (64, 512)]:
        for ks in (1, 3, 4):
            if ks > k // 64:
                continue
            for variant in ((0, 1) if bn <= 256 else (0,)):      # register-staged x / LDS-DMA x with deep W prefetch
                got = hip.mid_gemm(x, w, bm=bm, bn=bn, ksplit=ks, variant=variant)
                torch.testing.assert_close(got.float(), ref, **tol)
    # strided output (a slice of a wider buffer), as the fused-collective path uses
    wide = torch.zeros(m, n + 64, device=DEV, dtype=dtype)
    hip.mid_gemm(x, w, out=wide[:, :n])
    torch.testing.assert_close(wide[:, :n].float(), ref, **tol)
    assert wide[:, n:].abs().sum() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k,pad", [(256, 256, 128, 0), (512, 512, 256, 128), (300, 336, 384, 8),
                                       (1000, 2048, 2048, 128), (2048, 1024, 8192, 128),
                                       # more tiles than CUs (9 x 30): workgroups walk several tiles, with a
                                       # ragged last band; and a second, short band (ntm = 9)
                                       (2200, 7680, 256, 128), (2304, 8192, 128, 0),
                                       # fewer tile columns than XCDs
                                       (4096, 1024, 512, 128)])
def test_big_gemm_matches_fp32(dtype, m, n, k, pad):
    """Large-batch 256 x 256-tile GEMM: plain, + residual and SiLU*mul epilogues, both store widths,
    against an fp32 product of the same (padded-row) operands."""
    from vllm_distributed_amd.ops import hip
    x = _rand(m, k, dtype=dtype)
    w = _padded_weight(n, k, pad, dtype)
    ref = x.float() @ w.float().t()
    tol = dict(rtol=2e-2, atol=2e-2 * max(1.0, ref.abs().max().item()) / 4)
    variants = (0, 1) if dtype == torch.bfloat16 else (0,)
    for variant in variants:
        got = hip.big_gemm(x, w, variant=variant)
        torch.testing.assert_close(got.float(), ref, **tol)
        r = _rand(m, n, dtype=dtype)
        got = hip.big_gemm(x, w, residual=r, variant=variant)
        torch.testing.assert_close(got.float(), ref + r.float(), **tol)
        if n % 32 == 0:
            act = torch_ref.silu_and_mul(ref)
            got = hip.big_gemm(x, w, silu_mul=True, variant=variant)
            torch.testing.assert_close(got.float(), act, rtol=3e-2, atol=3e-2 * max(1.0, act.abs().max().item()) / 4)
    # in place on the residual, and a strided output (a slice of a wider buffer)
    r = _rand(m, n, dtype=dtype)
    want = ref + r.float()
    hip.big_gemm(x, w, out=r, residual=r)
    torch.testing.assert_close(r.float(), want, **tol)
    wide = torch.zeros(m, n + 64, device=DEV, dtype=dtype)
    hip.big_gemm(x, w, out=wide[:, :n])
    torch.testing.assert_close(wide[:, :n].float(), ref, **tol)
    assert wide[:, n:].abs().sum() == 0


def test_big_gemm_exact_integers_catch_any_misplaced_tile():
    """Small-integer operands make every output an exact integer: a fragment written to the wrong row,
    column or half-tile cannot hide inside a tolerance."""
    from vllm_distributed_amd.ops import hip
    m, n, k = 768, 1280, 256
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randint(-1, 2, (m, k), generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randint(-2, 3, (n, k), generator=g).to(torch.bfloat16).to(DEV)
    ref = x.float() @ w.float().t()
    assert ref.abs().max() <= 256         # every output exactly representable in bf16
    for variant in (0, 1):
        assert torch.equal(hip.big_gemm(x, w, variant=variant).float(), ref)


def test_large_batch_dispatch_reaches_the_big_gemm(monkeypatch):
    """At decode-batch sizes that fill the chip ops.linear and the gated MLP front half run the
    hand-written large-batch kernel, and agree with library GEMM + activation kernel."""
    from vllm_distributed_amd.ops import hip
    calls = []
    real = hip.big_gemm
    monkeypatch.setattr(hip, "big_gemm", lambda *a, **kw: (calls.append(bool(kw.get("silu_mul"))), real(*a, **kw))[1])
    x = _rand(2048, 4096)
    w = _padded_weight(8192, 4096, 128)
    y = ops.linear(x, w)
    ref = torch.nn.functional.linear(x, w)
    torch.testing.assert_close(y.float(), ref.float(), rtol=2e-2, atol=0.1)
    assert calls == [False]
    wgu = _padded_weight(2 * 7168, 4096, 128)
    act = ops.gate_up_silu(x, wgu)
    assert calls == [False, True] and act is not None and act.shape == (2048, 7168)
    want = ops.silu_and_mul(torch.nn.functional.linear(x, wgu))
    torch.testing.assert_close(act.float(), want.float(), rtol=3e-2, atol=0.1)
    # a qkv-shaped matrix whose q part alone is whole rounds of tiles: split in two GEMMs, one buffer
    wqkv = _padded_weight(8192 + 2048, 4096, 128)
    y = ops.linear_split(x, wqkv, 8192)
    assert y is not None and calls == [False, True, False]
    torch.testing.assert_close(y.float(), torch.nn.functional.linear(x, wqkv).float(), rtol=2e-2, atol=0.1)
    assert ops.linear_split(x, w, 4096) is None              # the whole matrix already tiles the chip
    calls.pop()
    # too few tiles to fill the chip: the library keeps the shape
    assert ops.gate_up_silu(_rand(512, 4096), _padded_weight(2 * 1024, 4096, 128)) is None
    ops.linear(_rand(512, 4096), w)
    assert len(calls) == 2


def test_linear_dispatch_uses_hand_written_kernels_in_their_range(monkeypatch):
    """ops.linear must reach the hand-written kernels where they are profitable, and agree with the
    library on the same padded weight."""
    from vllm_distributed_amd.ops import hip
    calls = []
    real_mid = hip.mid_gemm
    monkeypatch.setattr(hip, "mid_gemm", lambda *a, **kw: (calls.append("mid"), real_mid(*a, **kw))[1])
    w = _padded_weight(8192, 8192, 128)
    for m in (64, 128, 256):
        x = _rand(m, 8192)
        y = ops.linear(x, w)
        ref = torch.nn.functional.linear(x, w)
        torch.testing.assert_close(y.float(), ref.float(), rtol=2e-2, atol=0.1)
    assert calls == ["mid"] * 3
    x = _rand(512, 8192)
    ops.linear(x, w)
    assert len(calls) == 3            # above the range: library


@pytest.mark.parametrize("m", [1, 5, 16, 40, 128])
def test_skinny_gemm_with_silu_mul_operand(m):
    """The down projection fed with [gate | up]: silu(gate) * up is formed inside the GEMM."""
    from vllm_distributed_amd.ops import hip
    k, n = 1024, 512
    gate_up = _rand(m, 2 * k)
    w = _padded_weight(n, k, 128)
    act = torch_ref.silu_and_mul(gate_up.float()).to(torch.bfloat16).float()
    ref = act @ w.float().t()
    for ks in (1, 2):
        got = hip.skinny_gemm(gate_up, w, ksplit=ks, silu_mul=True)
        torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2 * max(1.0, ref.abs().max().item()) / 4)
    # the dispatching op agrees with activation + linear whichever path it takes
    w2 = _padded_weight(8192, 28672, 128)
    gu = _rand(2, 2 * 28672, scale=0.5)
    a = ops.silu_mul_linear(gu, w2)
    b = ops.linear(ops.silu_and_mul(gu), w2)
    torch.testing.assert_close(a.float(), b.float(), rtol=3e-2, atol=0.3)


@pytest.mark.parametrize("m,n,k", [(256, 8192, 8192), (64, 8192, 28672), (1, 8192, 28672), (200, 4096, 14336)])
def test_k_split_partial_sums_reduced_by_the_following_norm(m, n, k):
    """A projection left as fp32 K-split slabs + the kernel that sums them, adds the residual and
    normalises, against the fp32 definition; and against the reduce -> add + norm pair it replaces."""
    torch.manual_seed(m + n)
    x = (torch.randn(m, k, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    res = torch.randn(m, n, device="cuda").to(torch.bfloat16)
    g = (torch.rand(n, device="cuda") + 0.5).to(torch.bfloat16)
    slabs = ops.linear_partials(x, w)
    assert slabs is not None and slabs.dim() == 3 and slabs.shape[1:] == (m, n) and slabs.dtype == torch.float32
    r1 = res.clone()
    normed, r1 = ops.reduce_add_rms_norm(slabs, r1, g, 1e-5)
    y = x.float() @ w.float().t()
    tot = (y + res.float()).to(torch.bfloat16).float()
    want = tot * torch.rsqrt(tot.pow(2).mean(-1, keepdim=True) + 1e-5) * g.float()
    assert (r1.float() - (y + res.float())).abs().max().item() < 0.06
    assert (normed.float() - want).abs().max().item() < 0.06
    r2 = res.clone()
    n2, r2 = ops.fused_add_rms_norm(ops.linear(x, w), r2, g, 1e-5)
    assert (normed.float() - n2.float()).abs().max().item() < 0.13      # two bf16 roundings of values up to ~8


@pytest.mark.parametrize("dk,dv,hq,bs", [(576, 512, 16, 64), (576, 512, 128, 16), (576, 512, 8, 64), (96, 64, 4, 16),
                                         (96, 64, 48, 32)])
@pytest.mark.parametrize("dtype,kv8", [(torch.bfloat16, False), (torch.float16, False), (torch.bfloat16, True),
                                       (torch.float16, True)])
def test_latent_attention_decode_prefill_and_cache_write(dk, dv, hq, bs, dtype, kv8):
    """Multi-head latent attention over the latent cache against an fp32 PyTorch statement of the same sum:
    the cache-write kernel against index assignment, then a decode batch (lengths around the 32-key group
    boundaries, one long enough for several KV splits) and a prefill chunk behind a cached context (rows are
    tokens: each sees keys up to its own position)."""
    from vllm_distributed_amd.ops import hip as _hip
    torch.manual_seed(dk + hq)
    lens = [1, 31, 32, 33, 200, 1500]
    nblk = sum(-(-n // bs) for n in lens) + 4
    # (kv8: the caches hold e4m3 bytes; the reference reads the same bytes, so only the kernels' arithmetic differs)
    cdt = torch.float8_e4m3fn if kv8 else dtype
    kc = torch.zeros(torch_ref.k_cache_shape(nblk, 1, dk, bs), dtype=torch.float32, device=DEV).to(cdt)
    vc = torch.zeros(torch_ref.v_cache_shape(nblk, 1, dv, bs), dtype=torch.float32, device=DEV).to(cdt)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    order = torch.randperm(nblk - 1)[:nblk - 4] + 1           # block 0 stays empty: a clamped read must not matter
    table = torch.zeros((len(lens), -(-max(lens) // bs)), dtype=torch.int32)
    slots, at = [], 0
    for b, n in enumerate(lens):
        nb = -(-n // bs)
        table[b, :nb] = order[at:at + nb].int()
        at += nb
        slots += [int(table[b, i // bs]) * bs + i % bs for i in range(n)]
    latent = (torch.randn(len(slots), dk, device=DEV) * 0.5).to(dtype)
    slot_t = torch.tensor(slots + [-1], dtype=torch.int32, device=DEV)
    extra = torch.cat([latent, latent[:1]])                    # a row with slot -1 is skipped
    _hip.mla_cache_write(extra, kc, vc, slot_t)
    torch_ref.mla_cache_write(extra, kc_ref, vc_ref, slot_t)
    assert torch.equal(kc.view(torch.uint8) if kv8 else kc, kc_ref.view(torch.uint8) if kv8 else kc_ref)
    assert torch.equal(vc.view(torch.uint8) if kv8 else vc, vc_ref.view(torch.uint8) if kv8 else vc_ref)
    table = table.to(DEV)
    scale = dk ** -0.5
    # ---- decode: one query per sequence
    q = torch.randn(len(lens), hq, dk, device=DEV).to(dtype)
    sl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    ref = torch_ref.mla_attention(q.float(), kc if kv8 else kc.float(), vc if kv8 else vc.float(), table, None, sl, scale)
    for mode in (None, 0, 2):             # by shape; a workgroup per 16 heads; four head tiles per workgroup
        got = _hip.mla_attention(q, kc, vc, table, None, sl, scale, max(lens), mode=mode)
        torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)
    got = _hip.mla_attention(q, kc, vc, table, None, sl, scale, max(lens), mode=2, splits=1)
    torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)
    assert _hip.mla_num_splits(len(lens), 1, max(lens)) > 1
    # ---- a mixed step: three decode rows, then the last 41 tokens of sequence 4 and the last 70 of sequence 5,
    # causal (four consecutive tokens of a sequence share a workgroup: chunks start at odd positions here, and
    # single-token rows lead a workgroup of their own)
    tok_seq, tok_len = [0, 1, 3], [lens[0], lens[1], lens[3]]
    # (114 rows; row 111 sits at a position that is a multiple of four: the leader with the highest workgroup id)
    for b, n_q, first_pos in ((4, 41, lens[4] - 41), (5, 70, 1429)):
        tok_seq += [b] * n_q
        tok_len += list(range(first_pos + 1, first_pos + n_q + 1))
    assert len(tok_seq) == 114 and (tok_len[111] - 1) % 4 == 0
    ts = torch.tensor(tok_seq, dtype=torch.int32, device=DEV)
    tl = torch.tensor(tok_len, dtype=torch.int32, device=DEV)
    q = torch.randn(len(tok_seq), hq, dk, device=DEV).to(dtype)
    ref = torch_ref.mla_attention(q.float(), kc if kv8 else kc.float(), vc if kv8 else vc.float(), table, ts, tl, scale)
    for mode, splits in ((None, None), (0, None), (1, 3), (2, None)):
        got = _hip.mla_attention(q, kc, vc, table, ts, tl, scale, max(lens), mode=mode, splits=splits)
        torch.testing.assert_close(got.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("e,k,n_group,topk_group,v3,renorm", [
    (256, 8, 8, 4, True, True), (256, 8, 1, 1, True, True), (160, 6, 8, 3, False, False), (64, 6, 1, 1, False, False),
    (64, 6, 1, 1, False, True), (8, 2, 4, 2, True, True), (128, 1, 1, 1, True, True), (72, 4, 8, 8, False, True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_grouped_routing_kernel_matches_the_deepseek_formulas(e, k, n_group, topk_group, v3, renorm, dtype):
    """DeepSeek's routing as one kernel against the same rules written with PyTorch ops: V3 = sigmoid scores,
    choice on score + bias, groups ranked by the sum of their two best, weights renormalised and scaled; V2 =
    softmax scores, groups ranked by their best, weights renormalised OR scaled."""
    from vllm_distributed_amd.ops import hip as _hip
    torch.manual_seed(e + k)
    tokens, scale = 37, 2.5
    logits = (torch.randn(tokens, e, device=DEV) * 2).to(dtype)
    bias = (torch.randn(e, device=DEV) * 0.3) if v3 else None
    x = logits.float()
    scores = x.sigmoid() if v3 else torch.softmax(x, dim=-1)
    choice = scores + bias if v3 else scores
    if n_group > 1 and topk_group < n_group:
        per_group = choice.view(tokens, n_group, e // n_group)
        gs = per_group.topk(min(2, e // n_group), dim=-1).values.sum(-1) if v3 else per_group.max(-1).values
        keep = torch.zeros_like(gs, dtype=torch.bool).scatter_(1, gs.topk(topk_group, dim=-1).indices, True)
        choice = choice.masked_fill(~keep.unsqueeze(-1).expand(-1, -1, e // n_group).reshape(tokens, e), float("-inf"))
    want_i = choice.topk(k, dim=-1).indices
    want_w = scores.gather(1, want_i)
    if renorm:
        want_w = want_w / (want_w.sum(-1, keepdim=True) + 1e-20)
    if v3 or not renorm:
        want_w = want_w * scale
    got_w, got_i = _hip.moe_topk_grouped(logits, bias, k, n_group, topk_group, v3, v3, renorm,
                                         scale if (v3 or not renorm) else 1.0)
    assert got_i.dtype == torch.int32 and got_w.dtype == torch.float32
    assert torch.equal(got_i.long().sort(-1).values, want_i.sort(-1).values)
    order_g, order_w = got_i.long().argsort(-1), want_i.argsort(-1)
    torch.testing.assert_close(got_w.gather(1, order_g), want_w.gather(1, order_w), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("kv8", [False, True])
@pytest.mark.parametrize("h,nope,rank,rope", [(16, 128, 512, 64), (4, 64, 64, 32), (3, 16, 16, 8)])
def test_latent_attention_prep_kernel_matches_its_parts(h, nope, rank, rope, kv8):
    """Norm of the latent, rotation of the shared key part and of every head's query part, assembly of the
    absorbed query and the write into both cache forms as one kernel, against the separate PyTorch steps."""
    from vllm_distributed_amd.ops import hip as _hip
    torch.manual_seed(rank)
    t, bs, nblk, dtype = 21, 16, 8, torch.bfloat16
    qbuf = torch.randn(t, h * (nope + rope) + 8, device=DEV).to(dtype)
    q = qbuf[:, :h * (nope + rope)].view(t, h, nope + rope)                      # a strided view, as projections give
    q_lat = torch.randn(h, t, rank, device=DEV).to(dtype)
    latent = torch.randn(t, rank + rope, device=DEV).to(dtype)
    w = (1 + 0.1 * torch.randn(rank, device=DEV)).to(dtype)
    cos_sin = torch_ref.build_cos_sin_cache(rope, 512, 10000.0, None).to(DEV)
    positions = torch.randint(0, 500, (t,), dtype=torch.int32, device=DEV)
    slots = torch.randperm(nblk * bs)[:t].int().to(DEV)
    slots[5] = -1
    cdt = torch.float8_e4m3fn if kv8 else dtype
    kc = torch.zeros(torch_ref.k_cache_shape(nblk, 1, rank + rope, bs), dtype=torch.float32, device=DEV).to(cdt)
    vc = torch.zeros(torch_ref.v_cache_shape(nblk, 1, rank, bs), dtype=torch.float32, device=DEV).to(cdt)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    got = _hip.mla_prep(q, nope, q_lat, latent, w, 1e-6, cos_sin, positions, kc, vc, slots)
    c = torch_ref.rms_norm(latent[:, :rank].contiguous(), w, 1e-6)
    q_rot = q[..., nope:].reshape(t, h * rope).clone()
    k_rot = latent[:, rank:].clone()
    torch_ref.rotary_embedding(positions, q_rot, k_rot, cos_sin, rope)
    want = torch.cat([q_lat.transpose(0, 1), q_rot.view(t, h, rope)], dim=-1)
    torch_ref.mla_cache_write(torch.cat([c, k_rot], dim=-1), kc_ref, vc_ref, slots)
    torch.testing.assert_close(got.float(), want.float(), rtol=1e-2, atol=1e-2)
    # (e4m3: the rotated values are rounded to the model type first in the reference, straight from fp32 in the
    # kernel - one e4m3 step apart at most, 1/8 relative)
    tol = dict(rtol=0.13, atol=2e-2) if kv8 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(kc.float(), kc_ref.float(), **tol)
    torch.testing.assert_close(vc.float(), vc_ref.float(), **tol)
    only_q = _hip.mla_prep(q, nope, q_lat, latent, w, 1e-6, cos_sin, positions, None, None, None)
    assert torch.equal(only_q, got)
