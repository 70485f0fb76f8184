"""End-to-end on CPU: tiny random-init models through the whole engine."""
import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.parallel import state as pstate


@pytest.fixture(autouse=True)
def _fresh_groups():
    pstate.destroy_model_parallel()
    yield
    pstate.destroy_model_parallel()


def _hf_style_greedy(llm, prompt_ids, n):
    """Reference: recompute the full sequence from scratch for every token with
    a fresh engine-free forward (no KV cache reuse across steps)."""
    from vllm_distributed_amd.models.layers.attention import AttentionMetadata, set_forward_context
    runner = llm.engine.core.executor.driver.worker.model_runner
    model = runner.model
    bs = runner.block_size
    ids = list(prompt_ids)
    out = []
    for _ in range(n):
        t = len(ids)
        nb = (t + bs - 1) // bs
        # use scratch blocks at the END of the cache so we do not disturb live state
        blocks = list(range(runner.num_blocks - nb, runner.num_blocks))
        pos = torch.arange(t, dtype=torch.int32)
        slots = torch.tensor([blocks[p // bs] * bs + p % bs for p in range(t)], dtype=torch.int32)
        md = AttentionMetadata(
            num_decodes=0, num_decode_tokens=0, num_prefills=1, num_prefill_tokens=t,
            slot_mapping=slots, block_table=torch.tensor([blocks], dtype=torch.int32),
            seq_lens=torch.tensor([t], dtype=torch.int32),
            query_start_loc=torch.tensor([0, t], dtype=torch.int32), max_query_len=t, max_seq_len=t,
            prefill_query_start_loc=torch.tensor([0, t], dtype=torch.int32), max_prefill_query_len=t)
        with torch.inference_mode():
            set_forward_context(md, pos)
            h = model(torch.tensor(ids, dtype=torch.int32), pos, None)
            set_forward_context(None)
            tok = int(model.compute_logits(h[-1:]).argmax(-1))
        ids.append(tok)
        out.append(tok)
    return out


@pytest.mark.parametrize("model", ["llama-tiny", "opt-tiny", "qwen3-moe-tiny"])
def test_greedy_matches_recompute_reference(model):
    llm = LLM(model=model, device="cpu", dtype="float32", max_model_len=128, block_size=16,
              num_gpu_blocks_override=64, max_num_seqs=8)
    prompts = [[5, 9, 200, 17, 3, 3, 8], list(range(10, 45)), [7]]
    sp = SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True)
    outs = llm.generate(prompts, sp)
    for p, o in zip(prompts, outs):
        assert o.finished and len(o.outputs[0].token_ids) == 12
        assert o.outputs[0].token_ids == _hf_style_greedy(llm, p, 12)
    llm.shutdown()


def test_chunked_prefill_and_batching_do_not_change_tokens():
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, block_size=16,
              num_gpu_blocks_override=128, enable_prefix_caching=False)
    prompts = [list(range(3, 100)), list(range(50, 61)), [9, 8, 7, 6, 5] * 9]
    sp = SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True)
    llm = LLM(max_num_batched_tokens=512, **kw)
    ref = [o.outputs[0].token_ids for o in llm.generate(prompts, sp)]
    llm.shutdown()
    pstate.destroy_model_parallel()
    llm = LLM(max_num_batched_tokens=16, **kw)     # forces many prefill chunks
    got = [o.outputs[0].token_ids for o in llm.generate(prompts, sp)]
    llm.shutdown()
    assert got == ref


def test_prefix_cache_hit_and_same_tokens():
    llm = LLM(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, block_size=16,
              num_gpu_blocks_override=128, enable_prefix_caching=True)
    shared = list(range(20, 20 + 70))
    sp = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)
    a = llm.generate([shared + [1, 2, 3]], sp)[0]
    b = llm.generate([shared + [1, 2, 3]], sp)[0]
    assert a.outputs[0].token_ids == b.outputs[0].token_ids
    assert b.metrics.num_cached_tokens == 64     # 4 full blocks of the 73-token prompt
    llm.shutdown()


def test_preemption_recompute_keeps_outputs():
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
              enable_prefix_caching=False)
    prompts = [list(range(2, 40)), list(range(100, 140)), list(range(200, 230))]
    sp = SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True)
    llm = LLM(num_gpu_blocks_override=64, **kw)
    ref = [o.outputs[0].token_ids for o in llm.generate(prompts, sp)]
    llm.shutdown()
    pstate.destroy_model_parallel()
    llm = LLM(num_gpu_blocks_override=11, **kw)    # too few blocks for all three to finish
    got = [o.outputs[0].token_ids for o in llm.generate(prompts, sp)]
    assert llm.engine.core.scheduler.num_preemptions > 0
    llm.shutdown()
    assert got == ref


def test_sampling_seeded_and_stop_conditions():
    llm = LLM(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
              num_gpu_blocks_override=64)
    p = [[4, 5, 6, 7]]
    sp = SamplingParams(temperature=0.9, top_p=0.9, top_k=50, seed=1234, max_tokens=16, ignore_eos=True)
    a = llm.generate(p, sp)[0].outputs[0].token_ids
    b = llm.generate(p, sp)[0].outputs[0].token_ids
    assert a == b and len(a) == 16
    g = llm.generate(p, SamplingParams(temperature=0, max_tokens=8, ignore_eos=True))[0].outputs[0].token_ids
    stop_tok = g[3]
    o = llm.generate(p, SamplingParams(temperature=0, max_tokens=8, ignore_eos=True,
                                       stop_token_ids=[stop_tok]))[0]
    assert o.outputs[0].finish_reason == "stop"
    assert o.outputs[0].token_ids == g[:g.index(stop_tok) + 1]
    lp = llm.generate(p, SamplingParams(temperature=0, max_tokens=3, ignore_eos=True, logprobs=2))[0]
    assert len(lp.outputs[0].logprobs) == 3 and lp.outputs[0].cumulative_logprob < 0
    llm.shutdown()


def _run_tokens(prompts, sps, **kw):
    pstate.destroy_model_parallel()
    llm = LLM(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
              enable_prefix_caching=False, **kw)
    outs = llm.generate(prompts, sps)
    steps = llm.engine.core.timer.n
    sched = llm.engine.core.scheduler
    assert not sched.running and not sched.waiting and sched.kv.num_free_blocks == sched.kv.pool.num_blocks
    llm.shutdown()
    pstate.destroy_model_parallel()
    return [(o.outputs[0].token_ids, o.outputs[0].finish_reason) for o in outs], steps


def test_async_scheduling_matches_sync_including_stops_and_sampling():
    """One step is always scheduled ahead of the sampled tokens (device-resident
    inputs, placeholder accounting); results must be identical to the
    synchronous engine for greedy, seeded sampling, EOS/stop-token stops,
    max_tokens limits and penalties."""
    prompts = [[5, 9, 200, 17, 3], list(range(10, 60)), [7], list(range(100, 131)), [42, 43]]
    base, _ = _run_tokens(prompts, SamplingParams(temperature=0.0, max_tokens=20, ignore_eos=True),
                          num_gpu_blocks_override=64)
    stop_tok = base[1][0][6]
    sps = [SamplingParams(temperature=0.0, max_tokens=20, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=20, ignore_eos=True, stop_token_ids=[stop_tok]),
           SamplingParams(temperature=0.9, top_p=0.8, seed=11, max_tokens=1, ignore_eos=True),
           SamplingParams(temperature=0.7, top_k=20, seed=5, max_tokens=13, ignore_eos=True,
                          repetition_penalty=1.3, presence_penalty=0.5),
           SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True)]
    ref, _ = _run_tokens(prompts, sps, num_gpu_blocks_override=64)
    got, _ = _run_tokens(prompts, sps, num_gpu_blocks_override=64, async_scheduling=True)
    assert got == ref
    assert ref[1][1] == "stop" and ref[0][1] == "length" and len(ref[2][0]) == 1 and len(ref[4][0]) == 2


def test_async_scheduling_with_preemption_and_chunked_prefill():
    prompts = [list(range(2, 40)), list(range(100, 140)), list(range(200, 230))]
    sp = SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True)
    ref, _ = _run_tokens(prompts, sp, num_gpu_blocks_override=64)
    got, _ = _run_tokens(prompts, sp, num_gpu_blocks_override=11, async_scheduling=True,
                         max_num_batched_tokens=24)
    assert got == ref


@pytest.mark.parametrize("connector", ["SharedStorageConnector", "DeviceIpcConnector"])
def test_shared_storage_kv_connector_moves_prefill_between_engines(tmp_path, connector):
    """Disaggregated prefill: a producer engine prefills and writes each prompt's KV blocks to
    shared storage, a consumer engine loads them instead of computing them, and generates what
    an engine that computed everything itself generates.  The producer frees a finished
    request's blocks only once its save has been confirmed.  (The device-buffer connector has no
    device here and falls back to files.)"""
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.parallel import state as pstate
    prompts = [list(range(3, 3 + 70)), [9, 8, 7] * 15, [4, 5]]           # 4, 2 and 0 whole blocks to hand over

    def make(role):
        pstate.destroy_model_parallel()
        kv = None if role is None else {"kv_connector": connector, "kv_role": role,
                                        "kv_connector_extra_config": {"shared_storage_path": str(tmp_path)}}
        return LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, kv_transfer_config=kv, enable_prefix_caching=False))

    def run(eng, max_tokens):
        sp = SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
        ids = [eng.add_request(None, p, sp) for p in prompts]
        final = {}
        while eng.has_unfinished_requests():
            for o in eng.step():
                if o.finished:
                    final[o.request_id] = o.outputs[0].token_ids
        return [final[i] for i in ids]

    eng = make(None)
    ref = run(eng, 6)
    eng.shutdown()
    prod = make("kv_producer")
    first = run(prod, 1)                                   # prefill-only requests
    conn = prod.core.scheduler.kv_connector
    assert [t[0] for t in ref] == [t[0] for t in first]
    assert conn.num_saved_requests == 2 and not conn.sending and not conn.held_until_sent
    assert prod.core.scheduler.kv.pool.num_free_blocks == 64 - 0 or prod.core.scheduler.kv.usage == 0.0
    prod.shutdown()
    cons = make("kv_consumer")
    got = run(cons, 6)
    assert got == ref
    assert cons.core.scheduler.kv_connector.num_loaded_tokens == 64 + 32
    cons.shutdown()
    pstate.destroy_model_parallel()


def test_kv_transfers_run_beside_the_steps_and_an_aborted_load_keeps_its_blocks(tmp_path, monkeypatch):
    """The connector's file I/O and copies happen on a side thread: with a load slowed to 0.4 s another
    request keeps generating (steps are not held up), the parked request is scheduled only once the load
    has been reported, and a request aborted mid-load gives its blocks back only after the copy has ended."""
    import time
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.executor.kv_transfer import WorkerKVConnector
    from vllm_distributed_amd.parallel import state as pstate
    long_a, long_b, other = list(range(3, 3 + 70)), list(range(100, 100 + 50)), [9, 8, 7, 6, 5]

    def make(role):
        pstate.destroy_model_parallel()
        kv = {"kv_connector": "SharedStorageConnector", "kv_role": role,
              "kv_connector_extra_config": {"shared_storage_path": str(tmp_path)}}
        return LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, kv_transfer_config=kv, enable_prefix_caching=False))

    def drain(eng):
        final = {}
        while eng.has_unfinished_requests():
            for o in eng.step():
                if o.finished:
                    final[o.request_id] = o.outputs[0].token_ids
        return final

    prod = make("kv_producer")
    sp1 = SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True)
    ids = [prod.add_request(None, p, sp1) for p in (long_a, long_b)]
    first = drain(prod)
    wconn = prod.core.executor.driver.worker.model_runner.kv_connector
    assert wconn.transfers_started == 2 and wconn.idle() and wconn.max_step_block_s < 0.05
    prod.shutdown()

    real_load = WorkerKVConnector._load

    def slow_load(self, op, path, kv_caches):
        time.sleep(0.4)
        real_load(self, op, path, kv_caches)
    monkeypatch.setattr(WorkerKVConnector, "_load", slow_load)
    cons = make("kv_consumer")
    sched = cons.core.scheduler
    free0 = sched.kv.pool.num_free_blocks
    sp = SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True)
    rid_other = cons.add_request(None, other, sp)
    rid_a = cons.add_request(None, long_a, SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True))
    rid_b = cons.add_request(None, long_b, sp)
    tokens_while_loading = 0
    t0 = time.time()
    final = {}
    aborted = False
    while cons.has_unfinished_requests():
        for o in cons.step():
            if o.request_id == rid_other and sched.kv_loading:
                tokens_while_loading += 1
            if o.finished:
                final[o.request_id] = o.outputs[0].token_ids
        if not aborted and rid_b in sched.kv_loading:
            held = len(sched.kv.req_blocks[rid_b])
            cons.abort_request(rid_b)                         # mid-copy: blocks must stay allocated
            assert rid_b in sched._kv_load_orphans and len(sched.kv.req_blocks[rid_b]) == held
            aborted = True
    assert aborted and time.time() - t0 >= 0.4
    assert tokens_while_loading >= 5, tokens_while_loading      # the slowed copies did not stall decoding
    assert final[rid_a][0] == first[ids[0]][0] and len(final[rid_a]) == 3 and len(final[rid_other]) == 40
    assert not sched.kv_loading and not sched._kv_load_orphans and rid_b not in sched.kv.req_blocks
    assert sched.kv.pool.num_free_blocks == free0
    assert sched.kv_connector.num_loaded_tokens == 64 + 48
    cons.shutdown()
    pstate.destroy_model_parallel()


def test_fp8_kv_cache_reference_ops_and_engine_cpu():
    """The e4m3 KV cache on the CPU reference path: write / read round trip with per-head scales,
    half the bytes per block, and an engine that generates with it."""
    import math
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.ops import torch_ref
    fp8 = torch.float8_e4m3fn
    T, hkv, d, bs, nb = 19, 2, 16, 8, 4
    k, v = torch.randn(T, hkv, d), torch.randn(T, hkv, d)
    ks, vs = torch.tensor([0.5, 0.3]), torch.tensor([1.7, 0.9])
    kc, vc = torch.zeros(nb, hkv, bs, d, dtype=fp8), torch.zeros(nb, hkv, d // 16, bs // 8, 16, 8, dtype=fp8)
    slots = torch.arange(T, dtype=torch.int32)
    torch_ref.reshape_and_cache(k, v, kc, vc, slots, (ks, vs))
    bt = torch.arange(nb, dtype=torch.int32).view(1, -1)
    kk, vv = torch_ref._gather_kv(kc, vc, bt[0], T, (ks, vs))
    # e4m3 keeps 3 mantissa bits: relative error of a stored value is at most 2^-4
    torch.testing.assert_close(kk.transpose(0, 1), k, rtol=0.07, atol=0.02)     # [H, S, D] -> [S, H, D]
    q = torch.randn(1, 4, d)
    out8 = torch_ref.paged_attention_decode(q, kc, vc, bt, torch.tensor([T], dtype=torch.int32),
                                            1 / math.sqrt(d), (ks, vs))
    kc32, vc32 = torch.zeros(nb, hkv, bs, d), torch.zeros(nb, hkv, d // 16, bs // 8, 16, 8)
    torch_ref.reshape_and_cache(k, v, kc32, vc32, slots)
    out32 = torch_ref.paged_attention_decode(q, kc32, vc32, bt, torch.tensor([T], dtype=torch.int32),
                                             1 / math.sqrt(d))
    torch.testing.assert_close(out8, out32, rtol=0.1, atol=0.1)

    llm = LLM(model="llama-tiny", device="cpu", dtype="float32", kv_cache_dtype="fp8", max_model_len=128,
              num_gpu_blocks_override=32, max_num_seqs=4)
    runner = llm.engine.core.executor.driver.worker.model_runner
    assert runner.kv_caches[0][0].dtype == fp8
    assert runner.get_kv_cache_spec().dtype_bytes == 1
    out = llm.generate([{"prompt_token_ids": list(range(5, 40))}], SamplingParams(temperature=0.0, max_tokens=6))
    assert len(out[0].outputs[0].token_ids) == 6
    with pytest.raises(ValueError):
        LLM(model="llama-tiny", device="cpu", kv_cache_dtype="fp8_e5m2")


def test_pooling_requests_last_and_mean_chunked_equals_unchunked():
    """Embedding-style requests: the pooled final hidden state is the same whether the prompt runs as
    one chunk or several (a 16-token step budget), and rides along with ordinary generation."""
    import numpy as np
    from vllm_distributed_amd import LLM, SamplingParams
    prompts = [{"prompt_token_ids": list(range(5, 45))}, {"prompt_token_ids": [7, 8, 9]}]
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, num_gpu_blocks_override=64,
              max_num_seqs=4)
    small = LLM(max_num_batched_tokens=16, **kw)
    whole = LLM(**kw)
    for mode in ("last", "mean"):
        a, b = small.encode(prompts, mode), whole.encode(prompts, mode)
        assert len(a[0]) == 256
        np.testing.assert_allclose(np.array(a), np.array(b), rtol=1e-4, atol=1e-5)
    assert not np.allclose(small.encode(prompts, "last")[0], small.encode(prompts, "mean")[0])
    e = whole.embed(prompts)
    assert abs(sum(x * x for x in e[0]) - 1.0) < 1e-6
    s = whole.score("hello", ["hello", "something else"])
    assert abs(s[0] - 1.0) < 1e-6 and s[1] < 0.9999
    # pooled and generating requests in the same batch
    outs = whole.generate(prompts, [SamplingParams(pooling="mean"), SamplingParams(max_tokens=4, temperature=0.0)])
    assert outs[0].pooled is not None and len(outs[0].outputs[0].token_ids) == 1
    assert outs[1].pooled is None and len(outs[1].outputs[0].token_ids) == 4
    with pytest.raises(ValueError):
        SamplingParams(pooling="max")


def test_ngram_speculative_decoding_reproduces_greedy_output():
    """Prompt-lookup drafts are verified in one step: the text is exactly what one-token-at-a-time
    greedy decoding writes, in fewer steps; requests with logprobs, a seed or logit edits in the same
    batch keep the one-token path; max_tokens cuts an accepted run at the right token."""
    from vllm_distributed_amd import LLM, SamplingParams
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, num_gpu_blocks_override=128,
              max_num_seqs=8)
    base = [3, 9, 4, 7, 11, 5]
    prompts = [{"prompt_token_ids": base * 6 + [3, 9]}, {"prompt_token_ids": list(range(20, 60))},
               {"prompt_token_ids": [5, 6] * 20}, {"prompt_token_ids": base * 4}]
    sps = [SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=23, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True, logprobs=1),
           SamplingParams(temperature=0.0, max_tokens=17, ignore_eos=True, logit_bias={7: 2.0})]
    ref = LLM(**kw)
    want = [o.outputs[0].token_ids for o in ref.generate(prompts, sps)]
    spec = LLM(speculative_config={"method": "ngram", "num_speculative_tokens": 4, "prompt_lookup_max": 3,
                                   "prompt_lookup_min": 1}, **kw)
    assert not spec.engine.vllm_config.scheduler_config.async_scheduling
    outs = spec.generate(prompts, sps)
    got = [o.outputs[0].token_ids for o in outs]
    assert got == want
    assert [len(g) for g in got] == [40, 23, 40, 17]
    assert len(outs[2].outputs[0].logprobs) == 40
    sched = spec.engine.core.scheduler
    assert sched.spec_stats["drafted"] > 0 and sched.spec_stats["accepted"] > 0
    solo_ref, solo_spec = LLM(**kw), LLM(speculative_config={"num_speculative_tokens": 4, "prompt_lookup_min": 1}, **kw)
    a = solo_ref.generate(prompts[:1], sps[0])[0].outputs[0].token_ids
    b = solo_spec.generate(prompts[:1], sps[0])[0].outputs[0].token_ids
    assert a == b and solo_spec.engine.core.scheduler.step_id < solo_ref.engine.core.scheduler.step_id
    # sampling at temperature > 0 also goes through verification (tokens differ run to run; lengths hold)
    hot = spec.generate(prompts[:2], SamplingParams(temperature=0.8, max_tokens=30, ignore_eos=True))
    assert [len(o.outputs[0].token_ids) for o in hot] == [30, 30]
    # prefix-cache hits after a verified run: same answer again
    again = [o.outputs[0].token_ids for o in spec.generate(prompts, sps)]
    assert again == want
    with pytest.raises(ValueError):
        LLM(speculative_config={"method": "eagle"}, **kw)


def test_draft_model_speculative_decoding_reproduces_greedy_output():
    """A second model on the workers proposes, the target verifies: whatever the draft says, the text is
    what one-token greedy decoding writes.  With the target as its own draft every proposal is accepted
    (k + 1 tokens per step); with an unrelated draft nearly all are rejected and the text still holds."""
    from vllm_distributed_amd import LLM, SamplingParams
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, num_gpu_blocks_override=128,
              max_num_seqs=8, load_format="dummy")
    prompts = [{"prompt_token_ids": [3, 9, 4, 7, 11, 5] * 6}, {"prompt_token_ids": list(range(20, 60))},
               {"prompt_token_ids": [5, 6] * 20}, {"prompt_token_ids": list(range(100, 117))}]
    sps = [SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=23, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True, logprobs=1),   # decoded one token at a time
           SamplingParams(temperature=0.0, max_tokens=30, ignore_eos=True)]
    ref = LLM(**kw)
    want = [o.outputs[0].token_ids for o in ref.generate(prompts, sps)]
    twin = LLM(speculative_config={"model": "llama-tiny", "num_speculative_tokens": 3}, **kw)
    assert twin.engine.vllm_config.speculative_config.method == "draft_model"
    runner = twin.engine.core.executor.driver.worker.model_runner
    spec = runner.get_kv_cache_spec()
    assert spec.extra_bytes_per_block == 2 * spec.num_layers * spec.block_size * spec.num_kv_heads * spec.head_dim * \
        spec.dtype_bytes                                   # the draft's caches are budgeted next to the target's
    got = [o.outputs[0].token_ids for o in twin.generate(prompts, sps)]
    assert got == want
    st = twin.engine.core.scheduler.spec_stats
    assert st["drafted"] > 0 and st["accepted"] == st["drafted"]          # same weights: never wrong
    assert st["accepted"] >= 0.6 * (40 + 23 + 30)            # ~3 of every 4 tokens of the three plain requests
    solo = twin.generate(prompts[:1], sps[0])[0].outputs[0].token_ids
    assert solo == want[0] and twin.engine.core.scheduler.step_id - 40 <= 14       # 40 tokens in ~11 steps
    # an unrelated draft model (same vocabulary): rejections roll positions back, text unchanged;
    # a prefix-cache hit on the second pass skips prompt blocks for both models alike
    other = LLM(speculative_config={"method": "draft_model", "model": "phi-tiny", "num_speculative_tokens": 4}, **kw)
    for _ in range(2):
        got = [o.outputs[0].token_ids for o in other.generate(prompts, sps)]
        assert got == want
    st = other.engine.core.scheduler.spec_stats
    assert st["drafted"] > 0 and st["accepted"] < st["drafted"]
    hot = other.generate(prompts[:2], SamplingParams(temperature=0.8, max_tokens=30, ignore_eos=True))
    assert [len(o.outputs[0].token_ids) for o in hot] == [30, 30]
    with pytest.raises(ValueError):
        LLM(speculative_config={"method": "draft_model"}, **kw)          # no model named
    for llm in (ref, twin, other):
        llm.shutdown()


def _record_head_proposals(llm):
    """(request id, position of the request's newest token) -> the head's drafts, for every proposal made."""
    runner = llm.engine.core.executor.driver.worker.model_runner
    seen, last = {}, {}
    inner_propose, inner_step = runner.draft.propose, runner._propose_from_hidden

    def propose(feats, start_pos, *a, **k):
        out = inner_propose(feats, start_pos, *a, **k)
        last["start"], last["out"] = start_pos.copy(), out
        return out

    def step(p, md, input_ids, positions, hidden, sel, *a, **k):
        last.clear()
        res = inner_step(p, md, input_ids, positions, hidden, sel, *a, **k)
        if sel is not None and last:
            for j, i in enumerate(sel.tolist()):
                seen[(p["req_ids"][i], int(last["start"][j]))] = list(last["out"][j])
        return res

    runner.draft.propose, runner._propose_from_hidden = propose, step
    return seen


def test_eagle3_head_reads_three_depths_of_the_residual_stream():
    """EAGLE-3: ``fc`` over the target's residual stream in front of three of its layers, one decoder layer whose
    attention reads ``[input_layernorm(embed) ; hidden_norm(hidden)]``, ``norm`` + an LM head over a reduced draft
    vocabulary, ``d2t`` back to target ids.  Whatever the head proposes, the text is greedy decoding's, and the
    drafts do not depend on how the prompts were chunked.  The wiring is pinned with a silenced layer (zero output
    projections: the feature is ``fc(taps)`` and stays that for every further draft):
      * taps moved to layers (0, 1, 2) and ``fc`` reading the first one through a random matrix: the tap in front
        of layer 0 is the embedding of the token AT the position, so every draft after a step is
        ``d2t-mapped argmax(lm_head(norm(A emb[token at the last kept position])))``, computed on the host;
      * ``fc`` zero and the attention live on one half of its input at a time: with the embedding half of the
        q / k / v weights zeroed nothing reaches the output (every draft is the mapped id of draft 0), with the
        hidden half zeroed the drafts follow the tokens - the concatenation is ``[embeds ; hidden]``.
    No EAGLE-3 fixtures ship with the reference: parity with other implementations is unpinned."""
    import torch
    from vllm_distributed_amd import LLM, SamplingParams
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, num_gpu_blocks_override=128,
              max_num_seqs=8, load_format="dummy")
    prompts = [{"prompt_token_ids": [3, 9, 4, 7, 11, 5] * 6}, {"prompt_token_ids": list(range(20, 60))},
               {"prompt_token_ids": [5, 6] * 20}, {"prompt_token_ids": list(range(100, 117))}]
    sps = [SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=23, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True, logprobs=1),
           SamplingParams(temperature=0.0, max_tokens=30, ignore_eos=True)]
    ref = LLM(**kw)
    want = [o.outputs[0].token_ids for o in ref.generate(prompts, sps)]
    ref.shutdown()
    spec_cfg = {"method": "eagle3", "model": "eagle3-llama-tiny", "num_speculative_tokens": 3}

    logs = []
    for extra in ({}, {"max_num_batched_tokens": 16}):
        llm = LLM(speculative_config=spec_cfg, **kw, **extra)
        runner = llm.engine.core.executor.driver.worker.model_runner
        head, target = runner.draft.model, runner.model
        assert target.aux_hidden_layers == (1, 2, 3) and head.fc.weight.shape == (256, 768)
        assert head.embed_tokens is target.embed_tokens and head.lm_head is not target.lm_head
        assert head.lm_head.weight.shape[0] >= 128 and head.layers["0"].self_attn.qkv_proj.weight.shape[1] == 512
        d2t = head.d2t
        assert d2t.dtype == torch.int64 and bool((d2t[1:] + 1 >= d2t[:-1]).all()) and int(d2t[-1]) + 127 < 512
        assert runner.get_kv_cache_spec().extra_bytes_per_block > 0
        seen = _record_head_proposals(llm)
        outs = llm.generate(prompts, sps)
        assert [o.outputs[0].token_ids for o in outs] == want
        st = llm.engine.core.scheduler.spec_stats
        assert st["drafted"] > 0 and st["accepted"] < st["drafted"]
        targets = set((torch.arange(128) + d2t).tolist())
        assert all(t in targets for d in seen.values() for t in d)          # drafts are ids of the draft vocabulary
        order = {o.request_id: i for i, o in enumerate(outs)}
        logs.append({(order[r], pos): d for (r, pos), d in seen.items()})
        assert target.path_counts["plain"] > 0 and len(target.path_counts) == 1
        llm.shutdown()
    assert len(logs[0]) > 60 and logs[0] == logs[1]

    def silenced(llm):
        runner = llm.engine.core.executor.driver.worker.model_runner
        head, target = runner.draft.model, runner.model
        with torch.no_grad():
            layer = head.layers["0"]
            layer.self_attn.o_proj.weight.zero_()
            layer.mlp.down_proj.weight.zero_()
            head.fc.weight.zero_()
        return head, target

    # ---- tap alignment, fc, norm, LM head and d2t
    llm = LLM(speculative_config=spec_cfg, **kw)
    head, target = silenced(llm)
    target.aux_hidden_layers = (0, 1, 2)
    h = 256
    a = torch.randn(h, h, generator=torch.Generator().manual_seed(5)) / h ** 0.5
    with torch.no_grad():
        head.fc.weight[:, :h] = a
    emb = target.embed_tokens.weight.float()
    feat = emb @ a.T
    normed = feat * torch.rsqrt(feat.pow(2).mean(-1, keepdim=True) + head.norm.eps) * head.norm.weight.float()
    pick = (normed @ head.lm_head.weight.float().T)[:, :128].argmax(-1)
    g = (pick + head.d2t[pick]).tolist()
    seen = _record_head_proposals(llm)
    outs = llm.generate(prompts, sps)
    assert [o.outputs[0].token_ids for o in outs] == want
    full = {o.request_id: list(o.prompt_token_ids) + list(o.outputs[0].token_ids) for o in outs}
    checked = 0
    for (rid, pos), drafts in seen.items():
        if pos >= len(full[rid]):
            continue
        assert drafts == [g[full[rid][pos - 1]]] * len(drafts), (rid, pos, drafts)
        checked += 1
    assert checked > 60
    llm.shutdown()

    # ---- the attention's input is [embeds ; hidden]
    for zero_half in ("embeds", "hidden"):
        llm = LLM(speculative_config=spec_cfg, **kw)
        head, target = silenced(llm)
        with torch.no_grad():
            layer = head.layers["0"]
            torch.nn.init.normal_(layer.self_attn.o_proj.weight, std=0.1, generator=torch.Generator().manual_seed(7))
            w = layer.self_attn.qkv_proj.weight
            (w[:, :h] if zero_half == "embeds" else w[:, h:]).zero_()
        seen = _record_head_proposals(llm)
        outs = llm.generate(prompts, sps)
        assert [o.outputs[0].token_ids for o in outs] == want
        firsts = {d[0] for d in seen.values() if d}
        if zero_half == "embeds":
            assert firsts == {int(head.d2t[0])}, firsts       # fc is zero: nothing but zeros reaches the LM head
        else:
            assert len(firsts) > 8, firsts
        llm.shutdown()


def test_eagle_and_mtp_heads_draft_from_the_targets_hidden_states():
    """Heads that read (next token, target hidden state) pairs.  Whatever they propose, the text is greedy
    decoding's.  The wiring is pinned with heads whose layer is silenced (zero output projections), so the
    feature is ``fc([embed(next token) ; hidden])`` alone:
      * embedding half only -> the drafts after a token ``s`` are ``f(s), f(f(s)), ...`` for a host-computable ``f``
        (the token paired with the last kept position is the one the target has just sampled);
      * hidden half only, as the identity -> the feature at the last kept position is the target's own hidden
        state there, whose best token is the one just sampled: every draft repeats it.
    The head's KV follows the target's slots: the drafts do not depend on how the prompt was chunked.
    No EAGLE / MTP fixtures ship with the reference: parity with other implementations is unpinned."""
    import torch
    from vllm_distributed_amd import LLM, SamplingParams
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, num_gpu_blocks_override=128,
              max_num_seqs=8, load_format="dummy")
    prompts = [{"prompt_token_ids": [3, 9, 4, 7, 11, 5] * 6}, {"prompt_token_ids": list(range(20, 60))},
               {"prompt_token_ids": [5, 6] * 20}, {"prompt_token_ids": list(range(100, 117))}]
    sps = [SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=23, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True, logprobs=1),   # decoded one token at a time
           SamplingParams(temperature=0.0, max_tokens=30, ignore_eos=True)]
    ref = LLM(**kw)
    want = [o.outputs[0].token_ids for o in ref.generate(prompts, sps)]
    spec_cfg = {"method": "eagle", "model": "eagle-llama-tiny", "num_speculative_tokens": 3}

    # ---- a random head: lossless, and the same drafts whether the prompts ran whole or in 16-token chunks
    logs = []
    for extra in ({}, {"max_num_batched_tokens": 16}):
        llm = LLM(speculative_config=spec_cfg, **kw, **extra)
        assert llm.engine.vllm_config.speculative_config.reads_target_hidden
        runner = llm.engine.core.executor.driver.worker.model_runner
        assert runner.draft.model.embed_tokens is runner.model.embed_tokens          # shared, not copied
        assert runner.get_kv_cache_spec().extra_bytes_per_block > 0                  # the head's cache is budgeted
        seen = _record_head_proposals(llm)
        outs = llm.generate(prompts, sps)
        assert [o.outputs[0].token_ids for o in outs] == want
        st = llm.engine.core.scheduler.spec_stats
        assert st["drafted"] > 0 and st["accepted"] < st["drafted"]
        order = {o.request_id: i for i, o in enumerate(outs)}
        logs.append({(order[r], pos): d for (r, pos), d in seen.items()})
        if extra:       # again over cached prefixes: the head's KV of the cached blocks is reused as well
            assert [o.outputs[0].token_ids for o in llm.generate(prompts, sps)] == want
        llm.shutdown()
    assert len(logs[0]) > 60 and logs[0] == logs[1]
    assert all(len(d) == 3 for (i, pos), d in logs[0].items() if pos + 4 < len(prompts[i]["prompt_token_ids"]) + sps[i].max_tokens)

    # ---- silenced layer: the two halves of fc, one at a time
    for half in ("embedding", "hidden"):
        llm = LLM(speculative_config=spec_cfg, **kw)
        runner = llm.engine.core.executor.driver.worker.model_runner
        head, target = runner.draft.model, runner.model
        h = head.cfg["hidden_size"]
        gen = torch.Generator().manual_seed(5)
        a = torch.randn(h, h, generator=gen) / h ** 0.5
        with torch.no_grad():
            layer = head.layers["0"]
            layer.self_attn.o_proj.weight.zero_()
            layer.mlp.down_proj.weight.zero_()
            head.fc.weight.zero_()
            head.fc.bias.zero_()
            if half == "embedding":
                head.fc.weight[:, :h] = a
            else:
                head.fc.weight[:, h:] = torch.eye(h)
        emb, lm = target.embed_tokens.weight.float(), target.lm_head.weight.float()
        f = ((emb @ a.T) @ lm.T).argmax(-1).tolist()
        seen = _record_head_proposals(llm)
        outs = llm.generate(prompts, sps)
        assert [o.outputs[0].token_ids for o in outs] == want
        full = {o.request_id: list(o.prompt_token_ids) + list(o.outputs[0].token_ids) for o in outs}
        checked = 0
        for (rid, pos), drafts in seen.items():
            if pos >= len(full[rid]):
                continue                       # (the step that produced the request's final token)
            s = full[rid][pos]
            if half == "embedding":
                expect, t = [], s
                for _ in drafts:
                    t = f[t]
                    expect.append(t)
            else:
                expect = [s] * len(drafts)
            assert drafts == expect, (half, rid, pos, s, drafts, expect)
            checked += 1
        assert checked > 60
        llm.shutdown()

    # ---- DeepSeek-V3's multi-token-prediction module, loaded from the target's own checkpoint
    kw["model"] = "deepseek-v3-mtp-tiny"
    ref2 = LLM(**kw)
    want2 = [o.outputs[0].token_ids for o in ref2.generate(prompts[:2], sps[:2])]
    mtp = LLM(speculative_config={"method": "deepseek_mtp", "num_speculative_tokens": 2}, **kw)
    assert mtp.engine.vllm_config.speculative_config.method == "mtp"
    runner = mtp.engine.core.executor.driver.worker.model_runner
    assert list(runner.draft.model.layers) == ["3"] and runner.draft.model.lm_head is runner.model.lm_head
    assert [o.outputs[0].token_ids for o in mtp.generate(prompts[:2], sps[:2])] == want2
    assert mtp.engine.core.scheduler.spec_stats["drafted"] > 0
    with pytest.raises(ValueError):
        LLM(speculative_config={"method": "mtp"}, **dict(kw, model="deepseek-v3-tiny"))     # no such module inside
    for llm in (ref, ref2, mtp):
        llm.shutdown()


def test_image_prompts_prefix_cache_preemption_and_refusals():
    """Requests with images next to text requests: blocks over an image are cached under the image's content
    (same picture: hit, other picture behind the same tokens: no hit), a preempted request brings its image
    back with it, a pre-expanded prompt is taken as it is, and malformed inputs are refused at the door."""
    import numpy as np
    from PIL import Image
    from vllm_distributed_amd import LLM, SamplingParams
    rng = np.random.default_rng(5)
    pics = [Image.fromarray((rng.random((36, 48, 3)) * 255).astype("uint8")) for _ in range(4)]
    kw = dict(model="llava-tiny", device="cpu", dtype="float32", max_model_len=256, load_format="dummy", max_num_seqs=8)
    sp = SamplingParams(temperature=0.0, max_tokens=24, ignore_eos=True)
    text = list(range(30, 60))

    def prompt(k, tail=()):
        return {"prompt_token_ids": text + [511] + list(tail), "multi_modal_data": {"image": pics[k]}}

    roomy = LLM(num_gpu_blocks_override=128, **kw)
    want = [o.outputs[0].token_ids for o in roomy.generate([prompt(0), prompt(1), prompt(2, (7, 8))], sp)]
    assert want[0] != want[1]
    kv = roomy.engine.core.scheduler.kv
    hits = kv.prefix_hits
    again = roomy.generate([prompt(1)], sp)[0]                    # 30 text + 16 image tokens: two full blocks cached
    assert again.outputs[0].token_ids == want[1] and kv.prefix_hits - hits >= 32
    hits = kv.prefix_hits
    other = roomy.generate([{"prompt_token_ids": text + [511], "multi_modal_data": {"image": pics[3]}}], sp)[0]
    assert kv.prefix_hits - hits == 16                            # the text-only first block, nothing over the image
    assert other.outputs[0].token_ids not in (want[0], want[1])
    expanded = roomy.generate([{"prompt_token_ids": text + [511] * 16, "multi_modal_data": {"image": pics[0]}}], sp)[0]
    assert expanded.outputs[0].token_ids == want[0]
    with pytest.raises(ValueError, match="placeholder"):
        roomy.generate([{"prompt_token_ids": text, "multi_modal_data": {"image": pics[0]}}], sp)
    with pytest.raises(ValueError, match="placeholder"):
        roomy.generate([{"prompt_token_ids": text + [511, 511], "multi_modal_data": {"image": pics[0]}}], sp)
    with pytest.raises(ValueError, match="cut through"):
        roomy.generate([prompt(0)], SamplingParams(max_tokens=4, truncate_prompt_tokens=20))
    roomy.shutdown()
    # a pool too small for three 70-token sequences: one is preempted and recomputed, image included
    tight = LLM(num_gpu_blocks_override=11, enable_prefix_caching=False, **dict(kw, max_model_len=128))
    got = [o.outputs[0].token_ids for o in tight.generate([prompt(0), prompt(1), prompt(2, (7, 8))], sp)]
    assert tight.engine.core.scheduler.num_preemptions > 0 and got == want
    runner = tight.engine.core.executor.driver.worker.model_runner
    tight.generate([{"prompt_token_ids": [1, 2, 3]}], SamplingParams(max_tokens=2))     # (finished ids travel with the next step)
    assert not runner._mm_inputs and not runner._mm_embeds       # nothing kept once the requests are gone
    tight.shutdown()
    plain = LLM(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, load_format="dummy")
    with pytest.raises(ValueError, match="does not take images"):
        plain.generate([{"prompt_token_ids": [1, 2, 3], "multi_modal_data": {"image": pics[0]}}], sp)
    plain.shutdown()


def test_medusa_heads_draft_every_position_from_one_hidden_state(tmp_path):
    """Medusa: head i reads the target's hidden state at a request's last kept position and names the token
    i + 2 places on.  Pinned with heads whose residual block is silent and whose LM head is the target's with its
    rows permuted: the target's own pick there is the token it just sampled, ``s``, so head i must draft
    ``inverse_permutation_i[s]``.  The checkpoint is written under the Medusa repository's tensor names.
    (No Medusa fixtures ship with the reference: parity with other implementations is unpinned.)"""
    import json
    import torch
    from safetensors.torch import save_file
    from vllm_distributed_amd import LLM, SamplingParams
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, num_gpu_blocks_override=128,
              max_num_seqs=8, load_format="dummy")
    prompts = [{"prompt_token_ids": [3, 9, 4, 7, 11, 5] * 6}, {"prompt_token_ids": list(range(20, 60))},
               {"prompt_token_ids": [5, 6] * 20}]
    sps = [SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=23, ignore_eos=True),
           SamplingParams(temperature=0.0, max_tokens=30, ignore_eos=True, logprobs=1)]      # one token at a time
    ref = LLM(**kw)
    want = [o.outputs[0].token_ids for o in ref.generate(prompts, sps)]
    lm = ref.engine.core.executor.driver.worker.model_runner.model.lm_head.weight[:512].clone()
    gen = torch.Generator().manual_seed(3)
    perms = [torch.randperm(512, generator=gen) for _ in range(4)]
    tensors = {}
    for i, perm in enumerate(perms):
        tensors[f"medusa_head.{i}.0.linear.weight"] = torch.zeros(256, 256)
        tensors[f"medusa_head.{i}.0.linear.bias"] = torch.zeros(256)
        tensors[f"medusa_head.{i}.1.weight"] = lm[perm].contiguous()
    save_file(tensors, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps({"model_type": "medusa", "hidden_size": 256, "vocab_size": 512,
                                                      "num_heads": 4, "num_hidden_layers": 1, "torch_dtype": "float32"}))
    # (the bundled target runs on deterministic random weights; the heads come from the file)
    llm = LLM(speculative_config={"method": "medusa", "model": str(tmp_path), "num_speculative_tokens": 3,
                                  "load_format": "auto"}, **kw)
    runner = llm.engine.core.executor.driver.worker.model_runner
    assert runner.get_kv_cache_spec().extra_bytes_per_block == 0 and runner.draft.k == 3
    seen = _record_head_proposals(llm)
    outs = llm.generate(prompts, sps)
    assert [o.outputs[0].token_ids for o in outs] == want
    inverse = [torch.argsort(p).tolist() for p in perms]
    full = {o.request_id: list(o.prompt_token_ids) + list(o.outputs[0].token_ids) for o in outs}
    checked = 0
    for (rid, pos), drafts in seen.items():
        if pos >= len(full[rid]):
            continue
        s = full[rid][pos]
        assert drafts == [inverse[i][s] for i in range(len(drafts))], (rid, pos, s, drafts)
        checked += 1
    assert checked > 40 and llm.engine.core.scheduler.spec_stats["drafted"] > 0
    # random heads from a bundled description: lossless all the same
    rnd = LLM(speculative_config={"method": "medusa", "model": "medusa-llama-tiny", "num_speculative_tokens": 8}, **kw)
    assert rnd.engine.core.executor.driver.worker.model_runner.draft.k == 4          # as many drafts as heads
    assert [o.outputs[0].token_ids for o in rnd.generate(prompts, sps)] == want
    for x in (ref, llm, rnd):
        x.shutdown()


def test_offline_n_best_of_and_beam_search():
    """``LLM.generate`` fans ``n`` / ``best_of`` out into requests that share the prompt; beam search runs
    on one-token requests: width 1 is greedy decoding, beams come back best first and every beam's
    cumulative log-probability is what scoring its tokens as a prompt gives."""
    from vllm_distributed_amd import BeamSearchParams
    llm = LLM(model="llama-tiny", load_format="dummy", dtype="float32", device="cpu", max_model_len=128,
              num_gpu_blocks_override=128, max_num_seqs=16)
    try:
        prompt = [5, 9, 200, 17, 33]
        sp = SamplingParams(temperature=1.0, seed=7, max_tokens=5, ignore_eos=True, n=4, logprobs=0)
        out = llm.generate([prompt], sp)[0]
        assert [c.index for c in out.outputs] == [0, 1, 2, 3]
        again = llm.generate([prompt], sp)[0]
        assert [c.token_ids for c in again.outputs] == [c.token_ids for c in out.outputs]
        assert len({tuple(c.token_ids) for c in out.outputs}) > 1            # the seed is advanced per sample
        best = llm.generate([prompt], SamplingParams(temperature=1.0, seed=7, max_tokens=5, ignore_eos=True,
                                                     n=2, best_of=4))[0]
        ranked = sorted(out.outputs, key=lambda c: -c.cumulative_logprob)
        assert [list(c.token_ids) for c in best.outputs] == [list(c.token_ids) for c in ranked[:2]]

        greedy = llm.generate([prompt], SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True))[0]
        b1 = llm.beam_search([prompt], BeamSearchParams(beam_width=1, max_tokens=5, ignore_eos=True))[0]
        assert list(b1.outputs[0].token_ids) == list(greedy.outputs[0].token_ids)
        res = llm.beam_search([prompt], BeamSearchParams(beam_width=3, max_tokens=5, ignore_eos=True))[0]
        assert len(res.outputs) == 3
        scores = [c.cumulative_logprob for c in res.outputs]
        assert scores == sorted(scores, reverse=True) and scores[0] >= b1.outputs[0].cumulative_logprob - 1e-6
        for c in res.outputs:
            scored = llm.generate([prompt + list(c.token_ids)],
                                  SamplingParams(temperature=0.0, max_tokens=1, prompt_logprobs=0))[0]
            lp = sum(d[t] for t, d in zip(c.token_ids, scored.prompt_logprobs[len(prompt):]))
            assert abs(lp - c.cumulative_logprob) < 1e-3
    finally:
        llm.shutdown()
        pstate.destroy_model_parallel()


def test_pipeline_depth_deals_running_requests_over_micro_batches():
    """With PP batches in flight the scheduler hands each schedule() call 1/PP of the running requests:
    all of them in the first batch would leave the other stages idle until it has passed through."""
    from vllm_distributed_amd.config import CacheConfig, SchedulerConfig
    from vllm_distributed_amd.engine.request import Request
    from vllm_distributed_amd.engine.scheduler import Scheduler
    from vllm_distributed_amd.engine.types import ModelRunnerOutput
    sc = SchedulerConfig(max_num_seqs=16, max_num_batched_tokens=256, max_model_len=64)
    sched = Scheduler(sc, CacheConfig(block_size=16), num_blocks=64, pipeline_depth=4)
    for i in range(8):
        sched.add_request(Request(f"r{i}", [1, 2, 3], SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)))
    so = sched.schedule()                          # prefill of all eight (waiting requests are not capped)
    assert len(so.num_scheduled_tokens) == 8
    out = ModelRunnerOutput(req_ids=list(so.num_scheduled_tokens), sampled_token_ids=[[5]] * 8)
    sched.update_from_output(so, out)
    batches = [sched.schedule() for _ in range(4)]  # four micro-batches in flight, nothing returned yet
    sizes = [len(b.num_scheduled_tokens) for b in batches]
    assert sizes == [2, 2, 2, 2]
    assert len(set().union(*[set(b.num_scheduled_tokens) for b in batches])) == 8
    assert sched.schedule().is_empty               # everything is in flight


def test_scheduler_step_cost_at_16384_sequences():
    """schedule() + update_from_output() for 16384 plain decode sequences: the short paths keep the pair
    in the low microseconds per sequence (measured 3.5 us on the development box; the bound is 4x that)."""
    import time

    from vllm_distributed_amd.config import CacheConfig, SchedulerConfig
    from vllm_distributed_amd.engine.request import Request
    from vllm_distributed_amd.engine.scheduler import Scheduler
    from vllm_distributed_amd.engine.types import ModelRunnerOutput
    n = 16384
    sc = SchedulerConfig(max_num_seqs=n, max_num_batched_tokens=n * 8, max_model_len=512, async_scheduling=False)
    sched = Scheduler(sc, CacheConfig(block_size=64, enable_prefix_caching=False), num_blocks=n * 8 + 64,
                      include_new_token_ids=False)
    sp = SamplingParams(temperature=0.0, max_tokens=200, ignore_eos=True, detokenize=False)
    for i in range(n):
        sched.add_request(Request(f"r{i}", [1, 2, 3, 4], sp))
    best = None
    for step in range(8):
        t0 = time.perf_counter()
        so = sched.schedule()
        t1 = time.perf_counter()
        ids = list(so.num_scheduled_tokens)
        out = ModelRunnerOutput(step_id=so.step_id, req_ids=ids, sampled_token_ids=[[7]] * len(ids))
        t2 = time.perf_counter()
        res = sched.update_from_output(so, out)
        t3 = time.perf_counter()
        assert len(res) == n and so.total_num_scheduled_tokens == (4 * n if step == 0 else n)
        if step >= 2:
            cost = (t1 - t0) + (t3 - t2)
            best = cost if best is None else min(best, cost)
    assert all(len(r.output_token_ids) == 8 for r in sched.running)
    assert best < n * 14e-6, f"{best * 1e3:.1f} ms for {n} sequences"


def _drive(sched, steps, arrivals=None):
    """Run ``steps`` schedule/update rounds with a stub model; ``arrivals``: step -> requests added before it.
    Returns, per request, the step in which its first token was sampled."""
    from vllm_distributed_amd.engine.types import ModelRunnerOutput
    first = {}
    for step in range(steps):
        for req in (arrivals or {}).get(step, ()):
            sched.add_request(req)
        so = sched.schedule()
        ids = list(so.num_scheduled_tokens)
        toks = []
        for rid in ids:
            req = sched.requests[rid]
            toks.append([7] if req.num_computed_tokens >= req.num_tokens else [])     # a chunk that ends mid-prompt samples nothing
        for o in sched.update_from_output(so, ModelRunnerOutput(step_id=so.step_id, req_ids=ids, sampled_token_ids=toks)):
            first.setdefault(o.request_id, step)
    return first


def test_prefill_admission_next_to_a_full_decode_batch_and_the_long_prompt_cap():
    """What decides time to first token under load.  (a) A request that arrives while a large decode batch
    runs is admitted by the very next schedule() call, in the same step as the decodes (the token budget is
    shared, decodes first).  (b) A burst larger than the budget is admitted in arrival order, budget tokens
    per step: the last request of a burst waits for the burst's prefill, no longer.  (c) Without
    ``long_prefill_token_threshold`` a long prompt at the head of the queue takes the whole budget of every
    step until it is done and short prompts behind it wait; with it the long prompt advances by that many
    tokens per step and the short ones get in beside it."""
    from vllm_distributed_amd.config import CacheConfig, SchedulerConfig
    from vllm_distributed_amd.engine.request import Request
    from vllm_distributed_amd.engine.scheduler import Scheduler
    sp = SamplingParams(temperature=0.0, max_tokens=64, ignore_eos=True)

    def make(threshold=0, budget=1024):
        sc = SchedulerConfig(max_num_seqs=512, max_num_batched_tokens=budget, max_model_len=8192,
                             enable_chunked_prefill=True, long_prefill_token_threshold=threshold,
                             async_scheduling=False)
        return Scheduler(sc, CacheConfig(block_size=16, enable_prefix_caching=False), num_blocks=4096)

    # (a) 256 decodes running; a newcomer gets its whole 128-token prompt in with the next step
    sched = make()
    base = {0: [Request(f"d{i}", [1, 2, 3], sp) for i in range(256)],
            5: [Request("new", list(range(3, 131)), sp)]}
    first = _drive(sched, 8, base)
    assert first["new"] == 5 and all(first[f"d{i}"] == 0 for i in range(256))

    # (b) a burst of 64 x 128-token prompts against a 1024-token budget: eight per step, in order
    sched = make()
    first = _drive(sched, 12, {0: [Request(f"b{i}", list(range(3, 131)), sp) for i in range(64)]})
    assert [first[f"b{i}"] for i in (0, 7, 8, 63)] == [0, 0, 1, 8]     # (decodes of the admitted eat into later budgets)

    # (c) a 4000-token prompt followed by three short ones
    def long_then_short(threshold):
        sched = make(threshold)
        reqs = [Request("long", list(range(3, 4003)), sp)] + [Request(f"s{i}", [5, 6, 7, 8], sp) for i in range(3)]
        return _drive(sched, 12, {0: reqs})
    plain, capped = long_then_short(0), long_then_short(512)
    assert plain["long"] == 3 and all(plain[f"s{i}"] == 3 for i in range(3))        # head-of-line: 4 steps of 1024
    assert all(capped[f"s{i}"] == 0 for i in range(3)) and capped["long"] == 7       # 8 steps of 512 for the long one
