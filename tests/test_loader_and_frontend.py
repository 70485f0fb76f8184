"""Weight loading (safetensors == dummy generator, AWQ dequantisation, TP/PP
sharding), detokeniser stop strings, scheduler invariants."""
import os

import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.config import CacheConfig, ModelConfig, SchedulerConfig
from vllm_distributed_amd.engine.detokenizer import IncrementalDetokenizer
from vllm_distributed_amd.engine.kv_cache_manager import KVCacheManager
from vllm_distributed_amd.engine.request import Request
from vllm_distributed_amd.engine.scheduler import Scheduler
from vllm_distributed_amd.engine.types import ModelRunnerOutput
from vllm_distributed_amd.models.loader import awq_dequantize, save_dummy_checkpoint
from vllm_distributed_amd.models.utils import get_pp_layer_range
from vllm_distributed_amd.parallel import state as pstate
from vllm_distributed_amd.tokenizer import ByteTokenizer


def test_safetensors_checkpoint_loads_to_the_same_model_as_dummy_init(tmp_path):
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy")
    ckpt = str(tmp_path / "ckpt")
    save_dummy_checkpoint(mc, ckpt)
    prompts = [[5, 9, 200, 17, 3, 3, 8], list(range(10, 45))]
    sp = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
    outs = []
    for model, fmt in (("llama-tiny", "dummy"), (ckpt, "safetensors")):
        pstate.destroy_model_parallel()
        llm = LLM(model=model, load_format=fmt, device="cpu", dtype="float32", max_model_len=128,
                  block_size=16, num_gpu_blocks_override=32)
        outs.append([o.outputs[0].token_ids for o in llm.generate(prompts, sp)])
        llm.shutdown()
    pstate.destroy_model_parallel()
    assert outs[0] == outs[1]


def test_awq_dequantize_matches_definition():
    torch.manual_seed(0)
    k, n, g = 64, 32, 16
    w_int = torch.randint(0, 16, (k, n), dtype=torch.int32)
    z_int = torch.randint(0, 16, (k // g, n), dtype=torch.int32)
    scales = torch.rand(k // g, n) * 0.1 + 0.01
    order = [0, 4, 1, 5, 2, 6, 3, 7]

    def pack(x):
        out = torch.zeros(x.shape[0], x.shape[1] // 8, dtype=torch.int32)
        for col in range(x.shape[1]):
            word, pos = col // 8, order.index(col % 8)
            out[:, word] |= (x[:, col] & 0xF) << (4 * pos)
        return out
    got = awq_dequantize(pack(w_int), pack(z_int), scales, torch.float32)
    ref = ((w_int - z_int.repeat_interleave(g, 0)).float() * scales.repeat_interleave(g, 0)).t()
    torch.testing.assert_close(got, ref)


def test_pp_layer_partition():
    assert [get_pp_layer_range(80, r, 2) for r in range(2)] == [(0, 40), (40, 80)]
    assert [get_pp_layer_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 8), (8, 10)]
    assert [get_pp_layer_range(80, r, 2, "30,50") for r in range(2)] == [(0, 30), (30, 80)]
    with pytest.raises(ValueError):
        get_pp_layer_range(80, 0, 2, "30,40")


def test_incremental_detokenizer_stop_strings_and_holdback():
    tok = ByteTokenizer(512)
    sp = SamplingParams(stop=["END"], max_tokens=50)
    d = IncrementalDetokenizer(tok, [1, 2, 3], sp)
    emitted = ""
    hit = None
    for ch in "hello ENx worldENDtrailing":
        hit = d.append([ord(ch)])
        emitted += d.take_delta(hit is not None)
        if hit:
            break
        assert "END" not in emitted and len(emitted) <= max(0, len(d.text) - 2)   # 2-char tail held back
    assert hit == "END" and emitted == "hello ENx world" and d.text == "hello ENx world"
    d2 = IncrementalDetokenizer(tok, [], SamplingParams(stop=["ab"], include_stop_str_in_output=True))
    d2.append([ord(c) for c in "xxaby"])
    assert d2.text == "xxab"
    d3 = IncrementalDetokenizer(tok, [], SamplingParams())
    d3.append(list("héllo".encode()))       # multi-byte characters only appear once complete
    assert d3.text == "héllo"


def test_stop_string_finishes_request_through_the_engine():
    pstate.destroy_model_parallel()
    llm = LLM(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
              num_gpu_blocks_override=32)
    base = llm.generate(["abc"], SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True))[0]
    text = base.outputs[0].text
    printable = [i for i, c in enumerate(text) if c.isprintable() and c != "�" and i >= 2]
    if printable:       # random weights: only meaningful if some decodable text came out
        stop = text[printable[0]]
        o = llm.generate(["abc"], SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True, stop=[stop]))[0]
        assert o.outputs[0].finish_reason == "stop" and stop not in o.outputs[0].text
        assert len(o.outputs[0].token_ids) <= len(base.outputs[0].token_ids)
    assert not llm.engine.core.scheduler.running
    llm.shutdown()
    pstate.destroy_model_parallel()


def _mk_sched(num_blocks=16, **kw):
    sc = SchedulerConfig(max_num_batched_tokens=kw.pop("budget", 64), max_num_seqs=kw.pop("max_seqs", 8),
                         max_model_len=128, **kw)
    return Scheduler(sc, CacheConfig(block_size=16, enable_prefix_caching=True), num_blocks)


def _fake_output(so, token=7):
    ids = list(so.num_scheduled_tokens)
    return ModelRunnerOutput(step_id=so.step_id, req_ids=ids, sampled_token_ids=[[token] for _ in ids])


def test_scheduler_budget_chunking_and_block_accounting():
    s = _mk_sched(num_blocks=32, budget=40)
    for i in range(3):
        s.add_request(Request(f"r{i}", list(range(1, 31)), SamplingParams(max_tokens=4, ignore_eos=True)))
    so = s.schedule()
    assert so.total_num_scheduled_tokens == 40 and so.num_scheduled_tokens == {"r0": 30, "r1": 10}
    assert len(so.scheduled_new_reqs) == 2 and so.scheduled_new_reqs[1].block_ids
    # r1 is only partially prefilled: it must not sample
    out = ModelRunnerOutput(step_id=0, req_ids=["r0", "r1"], sampled_token_ids=[[5], []])
    res = s.update_from_output(so, out)
    assert [r.request_id for r in res] == ["r0"]
    so2 = s.schedule()
    # r2 shares r0's prompt: its first full block (16 tokens) is a prefix-cache hit, 14 remain
    assert so2.num_scheduled_tokens == {"r0": 1, "r1": 20, "r2": 14}
    # run to completion; every block must come back
    s.update_from_output(so2, ModelRunnerOutput(0, ["r0", "r1", "r2"], [[5], [5], [5]]))
    for _ in range(10):
        so = s.schedule()
        if so.is_empty:
            break
        s.update_from_output(so, _fake_output(so))
    assert not s.has_unfinished_requests() and s.kv.num_free_blocks == 32


def test_kv_manager_prefix_sharing_and_eviction():
    kv = KVCacheManager(num_blocks=8, block_size=4, max_model_len=64, enable_caching=True)
    a = Request("a", list(range(10)), SamplingParams())
    assert kv.get_computed_blocks(a) == ([], 0)
    assert len(kv.allocate_slots(a, 10)) == 3
    a.num_computed_tokens = 10
    kv.cache_blocks(a, 10)
    b = Request("b", list(range(8)) + [99, 98, 97], SamplingParams())
    blocks, n = kv.get_computed_blocks(b)
    assert n == 8 and [x.block_id for x in blocks] == kv.get_block_ids("a")[:2]
    assert len(kv.allocate_slots(b, 3, blocks)) == 3          # 2 shared + 1 fresh
    assert kv.num_free_blocks == 8 - 4
    kv.free(a)
    kv.free(b)
    assert kv.num_free_blocks == 8
    # cached blocks survive being freed until the allocator needs them
    c = Request("c", list(range(10)), SamplingParams())
    _, n = kv.get_computed_blocks(c)
    assert n == 8
    big = Request("big", list(range(100, 132)), SamplingParams())
    assert kv.allocate_slots(big, 32) is not None               # evicts everything
    kv.free(big)
    d = Request("d", list(range(10)), SamplingParams())
    assert kv.get_computed_blocks(d)[1] == 0


def test_sequence_parallel_forward_matches_replicated_forward(monkeypatch):
    """The row-sharded residual stream (reduce-scatter / norm / all-gather) is
    the same arithmetic as all-reduce + replicated norm; on one rank the
    collectives are identities, so tokens must be identical."""
    from vllm_distributed_amd.models.llama import LlamaForCausalLM
    calls = {"n": 0}
    orig = LlamaForCausalLM.forward_sequence_parallel

    def counted(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)
    monkeypatch.setattr(LlamaForCausalLM, "forward_sequence_parallel", counted)
    outs = {}
    for mode in ("0", "force"):
        monkeypatch.setenv("VDAMD_SEQUENCE_PARALLEL", mode)
        llm = LLM(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
                  num_gpu_blocks_override=64, max_num_seqs=8)
        o = llm.generate([[5, 9, 200, 17], list(range(10, 50))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        outs[mode] = [x.outputs[0].token_ids for x in o]
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert calls["n"] > 0 and outs["0"] == outs["force"]


def test_phi3_fused_checkpoint_equals_llama_checkpoint(tmp_path):
    """A Phi-3 checkpoint is a Llama checkpoint with q/k/v and gate/up stored
    fused: rewriting one into the other must not change a single token."""
    import json
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy")
    src = tmp_path / "llama"
    save_dummy_checkpoint(mc, str(src))
    sd = load_file(str(src / "model.safetensors"))
    fused = {}
    for name, w in sd.items():
        if ".q_proj." in name:
            k, v = sd[name.replace(".q_proj.", ".k_proj.")], sd[name.replace(".q_proj.", ".v_proj.")]
            fused[name.replace(".q_proj.", ".qkv_proj.")] = torch.cat([w, k, v], 0)
        elif ".gate_proj." in name:
            fused[name.replace(".gate_proj.", ".gate_up_proj.")] = torch.cat(
                [w, sd[name.replace(".gate_proj.", ".up_proj.")]], 0)
        elif not any(t in name for t in (".k_proj.", ".v_proj.", ".up_proj.")):
            fused[name] = w
    dst = tmp_path / "phi3"
    dst.mkdir()
    save_file({k: v.contiguous() for k, v in fused.items()}, str(dst / "model.safetensors"))
    cfg = dict(mc.hf_config, architectures=["Phi3ForCausalLM"], model_type="phi3")
    (dst / "config.json").write_text(json.dumps(cfg))
    outs = []
    for path in (src, dst):
        llm = LLM(model=str(path), device="cpu", dtype="float32", max_model_len=128, block_size=16,
                  num_gpu_blocks_override=64, max_num_seqs=8)
        o = llm.generate([[5, 9, 200, 17], list(range(10, 50))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        outs.append([x.outputs[0].token_ids for x in o])
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs[0] == outs[1]


def test_mixtral_checkpoint_round_trip_and_quantised_experts(tmp_path):
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    outs = {}
    mc = ModelConfig(model="mixtral-tiny", dtype="float32", load_format="dummy")
    save_dummy_checkpoint(mc, str(tmp_path / "mixtral"))
    for tag, kw in (("dummy", dict(model="mixtral-tiny")), ("file", dict(model=str(tmp_path / "mixtral"))),
                    ("awq", dict(model="mixtral-tiny", quantization="awq")),
                    ("awq-dequant", dict(model="mixtral-tiny", quantization="awq-dequant"))):
        llm = LLM(device="cpu", dtype="float32", max_model_len=128, block_size=16, num_gpu_blocks_override=64,
                  max_num_seqs=8, **kw)
        model = llm.engine.core.executor.driver.worker.model_runner.model
        moe = next(iter(model.layers.values())).mlp
        assert type(moe).__name__ == "FusedMoE" and moe.num_experts == 4 and moe.top_k == 2
        if tag == "awq":
            assert moe.qw13 is not None
        o = llm.generate([[5, 9, 200, 17], list(range(10, 50))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        outs[tag] = [x.outputs[0].token_ids for x in o]
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs["dummy"] == outs["file"] and outs["awq"] == outs["awq-dequant"]


def test_nvfp4_checkpoint_spellings_and_dense_equivalent(tmp_path):
    """NVFP4 tensors (packed E2M1 bytes, e4m3 block scales, per-matrix global
    scale) load from safetensors exactly as the dummy stream yields them, the
    modelopt spelling of the same tensors is accepted, and the packed model
    matches the dense model built from the dequantised definition."""
    import json
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.layers.linear import nvfp4_dequantize
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy", quantization="nvfp4")
    src = tmp_path / "ct"
    save_dummy_checkpoint(mc, str(src))
    cfg = dict(mc.hf_config, quantization_config={"quant_method": "compressed-tensors",
                                                  "format": "nvfp4-pack-quantized"})
    (src / "config.json").write_text(json.dumps(cfg))
    assert ModelConfig(model=str(src), dtype="float32").quantization == "nvfp4"
    sd = load_file(str(src / "model.safetensors"))
    assert any(k.endswith(".weight_packed") for k in sd)
    # modelopt spelling; dense equivalent
    mo, dense = {}, {}
    for k, v in sd.items():
        if k.endswith(".weight_packed"):
            base = k[:-len(".weight_packed")]
            mo[base + ".weight"] = v
            alpha = sd[base + ".weight_global_scale"].reciprocal()
            dense[base + ".weight"] = nvfp4_dequantize(v, sd[base + ".weight_scale"],
                                                       alpha.expand(v.shape[0]), torch.float32)
        elif k.endswith(".weight_global_scale"):
            mo[k[:-len("weight_global_scale")] + "weight_scale_2"] = v.reciprocal()
            mo[k[:-len("weight_global_scale")] + "input_scale"] = torch.ones(1)
        elif k.endswith(".weight_scale"):
            mo[k] = v
        else:
            mo[k] = dense[k] = v
    for tag, tensors, qc in (("mo", mo, {"quant_method": "modelopt", "quant_algo": "NVFP4"}), ("dense", dense, None)):
        d = tmp_path / tag
        d.mkdir()
        save_file({k: v.contiguous() for k, v in tensors.items()}, str(d / "model.safetensors"))
        c = dict(mc.hf_config)
        if qc:
            c["quantization_config"] = qc
        (d / "config.json").write_text(json.dumps(c))
    outs = {}
    for tag, kw in (("dummy", dict(model="llama-tiny", quantization="nvfp4")), ("ct", dict(model=str(src))),
                    ("mo", dict(model=str(tmp_path / "mo"))), ("dense", dict(model=str(tmp_path / "dense")))):
        llm = LLM(device="cpu", dtype="float32", max_model_len=128, block_size=16, num_gpu_blocks_override=64,
                  max_num_seqs=8, **kw)
        model = llm.engine.core.executor.driver.worker.model_runner.model
        layer = next(iter(model.layers.values()))
        assert (layer.mlp.down_proj.alpha is not None) == (tag != "dense")
        o = llm.generate([[5, 9, 200, 17], list(range(10, 50))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        outs[tag] = [x.outputs[0].token_ids for x in o]
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs["dummy"] == outs["ct"] == outs["mo"] == outs["dense"]


@pytest.mark.parametrize("scaling", [
    {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
     "original_max_position_embeddings": 8192},
    {"rope_type": "linear", "factor": 4.0},
    {"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 4096},
    {"rope_type": "yarn", "factor": 8.0, "original_max_position_embeddings": 2048, "beta_fast": 16, "beta_slow": 2,
     "attention_factor": 1.3},
])
def test_rope_scaling_tables_match_transformers(scaling):
    """cos/sin tables against the rotary initialisers that ship with the
    installed ``transformers`` (the definition checkpoints are trained with)."""
    from transformers import LlamaConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    from vllm_distributed_amd.ops import build_cos_sin_cache
    cfg = LlamaConfig(hidden_size=1024, num_attention_heads=8, max_position_embeddings=16384, rope_theta=500000.0,
                      rope_scaling=dict(scaling))
    inv_freq, attn = ROPE_INIT_FUNCTIONS[scaling["rope_type"]](cfg, "cpu")
    pos = torch.tensor([0, 1, 17, 4095, 4096, 9000, 16383])
    want = torch.outer(pos.double(), inv_freq.double())
    want = torch.cat([want.cos(), want.sin()], dim=-1) * attn
    got = build_cos_sin_cache(128, 16384, 500000.0, dict(scaling))[pos]
    # transformers keeps inv_freq in fp32: at position 16383 that alone moves the angle by ~1e-3
    assert torch.allclose(got.double(), want, atol=3e-3), (got.double() - want).abs().max()


def test_unknown_rope_scaling_is_rejected():
    from vllm_distributed_amd.ops import build_cos_sin_cache
    with pytest.raises(ValueError, match="not supported"):
        build_cos_sin_cache(128, 64, 10000.0, {"rope_type": "ntk-by-parts", "factor": 2.0})
    with pytest.raises(ValueError, match="short_factor"):        # LongRoPE without its factor lists
        build_cos_sin_cache(128, 64, 10000.0, {"rope_type": "longrope", "factor": 2.0})


def test_auto_load_format_refuses_a_checkpoint_that_is_not_there(tmp_path, monkeypatch):
    """`serve <hub id>` with nothing on disk must fail, not serve random weights; bundled
    architecture names and an explicit --load-format dummy keep working."""
    from vllm_distributed_amd.config import ModelConfig
    monkeypatch.setenv("HUGGINGFACE_HUB_CACHE", str(tmp_path))
    with pytest.raises(FileNotFoundError, match="no weights found"):
        ModelConfig(model="meta-llama/Meta-Llama-3-8B")
    assert ModelConfig(model="meta-llama/Meta-Llama-3-8B", load_format="dummy").load_format == "dummy"
    assert ModelConfig(model="llama-tiny").load_format == "dummy"
    assert ModelConfig(model="llama-3-70b").load_format == "dummy"
    with pytest.raises(FileNotFoundError):
        ModelConfig(model=str(tmp_path / "typo-in-the-path"))


def test_lazy_tensor_reads_only_the_requested_slice(tmp_path):
    import torch
    from safetensors import safe_open
    from safetensors.torch import save_file
    from vllm_distributed_amd.models.utils import LazyTensor, materialize
    w = torch.arange(24 * 10, dtype=torch.float32).reshape(24, 10).to(torch.bfloat16)
    v = torch.arange(7, dtype=torch.int32)
    save_file({"w": w, "v": v}, str(tmp_path / "m.safetensors"))
    with safe_open(str(tmp_path / "m.safetensors"), framework="pt", device="cpu") as f:
        lw = LazyTensor(f, "w")
        assert lw.shape == w.shape and lw.dtype == torch.bfloat16 and lw.dim() == 2
        part = lw.narrow(0, 6, 12).narrow(1, 2, 5).narrow(0, 3, 4)        # ranges compose
        assert part.shape == (4, 5)
        assert torch.equal(materialize(part), w[9:13, 2:7])
        assert torch.equal(materialize(lw[4:9]), w[4:9])
        assert torch.equal(lw.t(), w.t())                                  # anything else reads the whole tensor
        assert torch.equal(materialize(LazyTensor(f, "v").narrow(0, 2, 3)), v[2:5])
        with pytest.raises(IndexError):
            lw.narrow(1, 8, 5)
    assert materialize(w) is w


@pytest.mark.slow
def test_sharded_state_round_trip_tp2(tmp_path, monkeypatch):
    """TP=2 workers load their slices of a safetensors checkpoint (slice reads), generate,
    save their shards; a second engine started with --load-format sharded_state from that
    directory produces the same tokens, and a different TP degree is refused."""
    import os
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    from vllm_distributed_amd.parallel import state as pstate
    from vllm_distributed_amd.utils import get_open_port
    ckpt, shards = str(tmp_path / "ckpt"), str(tmp_path / "shards")
    save_dummy_checkpoint(ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy"), ckpt)
    for k, v in dict(VLLM_RPC_SECRET="s", VLLM_HOST_IP="127.0.0.1", GLOO_SOCKET_IFNAME="lo",
                     VLLM_HEAD_LOCAL_DEVICES="2", OMP_NUM_THREADS="2").items():
        monkeypatch.setenv(k, v)

    def run(model, load_format, tp=2):
        pstate.destroy_model_parallel()
        monkeypatch.setenv("VLLM_SERVER_PORT", str(get_open_port()))
        eng = LLMEngine.from_engine_args(EngineArgs(
            model=model, device="cpu", dtype="float32", max_model_len=128, block_size=16, load_format=load_format,
            num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=tp,
            distributed_executor_backend="distributed" if tp > 1 else "uni"))
        try:
            sp = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
            ids = [eng.add_request(None, p, sp) for p in ([5, 9, 200, 17], list(range(10, 40)))]
            final = {}
            while eng.has_unfinished_requests():
                for o in eng.step():
                    if o.finished:
                        final[o.request_id] = o.outputs[0].token_ids
            if load_format == "auto" and tp > 1:
                eng.core.executor.save_sharded_state(shards)
            return [final[i] for i in ids]
        finally:
            eng.shutdown()
            pstate.destroy_model_parallel()

    ref = run(ckpt, "auto", tp=1)
    assert run(ckpt, "auto") == ref
    assert sorted(os.listdir(shards)) == ["config.json", "model-rank-0.safetensors", "model-rank-1.safetensors",
                                          "sharded_state.json"]
    assert run(shards, "sharded_state") == ref
    with pytest.raises(Exception):
        run(shards, "sharded_state", tp=1)


def test_sliding_window_is_resolved_per_layer():
    from vllm_distributed_amd.config import layer_sliding_window
    base = dict(sliding_window=4096, num_hidden_layers=4)
    assert [layer_sliding_window(base, i) for i in range(4)] == [4096] * 4          # Mistral, StarCoder2, Phi-3
    assert layer_sliding_window(dict(base, sliding_window=None), 0) is None
    assert layer_sliding_window(dict(base, use_sliding_window=False), 0) is None    # Qwen2 default
    q = dict(base, use_sliding_window=True, max_window_layers=2)
    assert [layer_sliding_window(q, i) for i in range(4)] == [None, None, 4096, 4096]
    t = dict(base, layer_types=["sliding_attention", "full_attention", "sliding_attention", "full_attention"])
    assert [layer_sliding_window(t, i) for i in range(4)] == [4096, None, 4096, None]


def test_blocks_behind_the_sliding_window_return_to_the_pool():
    """With every layer windowed, a sequence holds blocks for its window (plus the margin for steps in
    flight), not for its length: four long generations fit a pool that their full lengths would
    overflow, without a preemption and with the tokens of a roomy run."""
    from vllm_distributed_amd.engine.kv_cache_manager import KVCacheManager
    from vllm_distributed_amd.engine.request import Request
    kv = KVCacheManager(num_blocks=64, block_size=16, max_model_len=1024, enable_caching=True, sliding_window=32)
    req = Request("r", list(range(200)), SamplingParams(max_tokens=400), None)
    assert kv.allocate_slots(req, 200) is not None
    req.num_computed_tokens = 200
    # keys below 200 - 32 - margin(128) = 40 are dead: blocks 0 and 1
    assert kv.release_out_of_window(req) == 2 and kv.release_out_of_window(req) == 0
    assert kv.get_block_ids("r")[:2] == [0, 0] and len(kv.get_block_ids("r")) == 13
    assert kv.num_free_blocks == 64 - 11
    kv.free(req)
    assert kv.num_free_blocks == 64
    kv_all = KVCacheManager(num_blocks=64, block_size=16, max_model_len=1024)      # a global-attention model
    assert kv_all.allocate_slots(req, 200) is not None and kv_all.release_out_of_window(req) == 0

    prompts = [[3 + (7 * i + j) % 400 for j in range(40)] for i in range(4)]
    sp = SamplingParams(temperature=0.0, max_tokens=300, ignore_eos=True)
    outs = {}
    for blocks in (200, 56):        # 4 x 340 tokens = 88 blocks of 16 at full length; (32 + 128) / 16 + 2 each
        llm = LLM(model="mistral-window-tiny", device="cpu", dtype="float32", max_model_len=512, block_size=16,
                  num_gpu_blocks_override=blocks, max_num_seqs=4, load_format="dummy")
        sched = llm.engine.core.scheduler
        assert sched.kv.window == 32
        outs[blocks] = [o.outputs[0].token_ids for o in llm.generate(prompts, sp)]
        assert sched.num_preemptions == 0
        assert sched.kv.num_free_blocks == blocks
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs[56] == outs[200]
    # a model with global layers keeps everything
    llm = LLM(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
              num_gpu_blocks_override=32, load_format="dummy")
    assert llm.engine.core.scheduler.kv.window is None
    llm.shutdown()
    pstate.destroy_model_parallel()


def test_fp8_block_scaled_checkpoint_loads_as_its_dequantised_twin(tmp_path):
    """An e4m3 checkpoint with 128 x 128 block scales (``weight_scale_inv``), per-channel and per-tensor
    scales (``weight_scale``) loads to exactly the model its widened weights describe, and the config's
    ``quant_method: fp8`` selects the W8A8 path on its own."""
    import json
    from safetensors.torch import save_file
    from vllm_distributed_amd.models.loader import dummy_weights_iterator, fp8_checkpoint_dequantize
    fp8 = torch.float8_e4m3fn
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy")
    q8, wide = {}, {}
    for i, (name, w) in enumerate(dummy_weights_iterator(mc, torch.device("cpu"))):
        if not (name.endswith("_proj.weight") and w.dim() == 2):
            q8[name] = wide[name] = w.contiguous()
            continue
        n, k = w.shape
        kind = i % 3
        if kind == 0:        # blocks of 128 x 128 (ragged at the edges)
            gn, gk = -(-n // 128), -(-k // 128)
            pad = torch.zeros(gn * 128, gk * 128)
            pad[:n, :k] = w
            amax = pad.view(gn, 128, gk, 128).abs().amax(dim=(1, 3)).clamp(min=1e-8)
            scale = amax / 448.0
            full = scale.repeat_interleave(128, 0)[:n].repeat_interleave(128, 1)[:, :k]
            sname = name + "_scale_inv"
        elif kind == 1:      # one scale per output channel
            scale = (w.abs().amax(dim=1, keepdim=True).clamp(min=1e-8) / 448.0)
            full, sname = scale, name + "_scale"
        else:                # one scale for the tensor
            scale = (w.abs().amax().clamp(min=1e-8) / 448.0).reshape(1)
            full, sname = scale, name + "_scale"
        q = (w / full).clamp(-448, 448).to(fp8)
        q8[name], q8[sname] = q, scale.float().contiguous()
        wide[name] = fp8_checkpoint_dequantize(q, scale, torch.float32)
        torch.testing.assert_close(wide[name], w, rtol=0.07, atol=1e-3)
    outs = {}
    for tag, tensors, extra in (("fp8", q8, {"quantization_config": {"quant_method": "fp8", "weight_block_size": [128, 128]}}),
                                ("wide", wide, {})):
        d = tmp_path / tag
        d.mkdir()
        save_file({k: v.contiguous() for k, v in tensors.items()}, str(d / "model.safetensors"))
        (d / "config.json").write_text(json.dumps(dict(mc.hf_config, **extra)))
        llm = LLM(model=str(d), device="cpu", dtype="float32", max_model_len=128, block_size=16,
                  num_gpu_blocks_override=32, quantization=None if tag == "fp8" else "fp8")
        assert llm.engine.vllm_config.model_config.quantization == "fp8"
        outs[tag] = [o.outputs[0].token_ids for o in llm.generate(
            [[5, 9, 200, 17, 33], list(range(10, 40))], SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True))]
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs["fp8"] == outs["wide"]


def test_gptq_checkpoint_is_the_same_model_as_its_awq_twin(tmp_path):
    """A 4-bit GPTQ checkpoint holds the integers, zero points and scales of an AWQ one in another
    packing: writing the dummy AWQ model's tensors in GPTQ form (rows packed eight to a word, zero
    points stored minus one, ``g_idx`` alongside) must load to exactly the same generations; act-order
    checkpoints are refused."""
    import json
    from safetensors.torch import save_file
    from vllm_distributed_amd.models.loader import dummy_weights_iterator
    from vllm_distributed_amd.ops import quant
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy", quantization="awq")
    g = mc.quant_group_size
    shifts = torch.arange(0, 32, 4, dtype=torch.int64)

    def wrap(x):          # [.., 8] nibbles -> int32 words, low nibble first
        w = (x.to(torch.int64) << shifts).sum(-1)
        return torch.where(w >= (1 << 31), w - (1 << 32), w).to(torch.int32)

    def write(dirname, desc_act):
        tensors = {}
        for name, w in dummy_weights_iterator(mc, torch.device("cpu")):
            if name.endswith(".qweight"):
                vals = quant.awq_unpack(w)                                     # [in, out]
                tensors[name] = wrap(vals.view(vals.shape[0] // 8, 8, -1).transpose(1, 2)).contiguous()
                idx = torch.arange(vals.shape[0], dtype=torch.int32) // g
                tensors[name[:-len("qweight")] + "g_idx"] = idx.flip(0).contiguous() if desc_act else idx
            elif name.endswith(".qzeros"):
                z = quant.awq_unpack(w).clamp(min=1)          # GPTQ stores zero - 1: a zero of 0 has no GPTQ form
                tensors[name] = wrap((z - 1).view(z.shape[0], -1, 8)).contiguous()
                tensors[name + ".awq"] = quant.awq_pack(z)
            else:
                tensors[name] = w.contiguous()
        awq = {k[:-4] if k.endswith(".awq") else k: v for k, v in tensors.items()
               if not k.endswith((".g_idx",)) and not (k.endswith(".qzeros") and k + ".awq" in tensors)
               and not k.endswith(".qweight")}
        awq.update({k: v for k, v in dummy_weights_iterator(mc, torch.device("cpu")) if k.endswith(".qweight")})
        for sub, tens, method in (("gptq", {k: v for k, v in tensors.items() if not k.endswith(".awq")}, "gptq"),
                                  ("awq", awq, "awq")):
            d = tmp_path / dirname / sub
            d.mkdir(parents=True)
            save_file(tens, str(d / "model.safetensors"))
            cfg = dict(mc.hf_config, quantization_config={"quant_method": method, "bits": 4, "group_size": g,
                                                          "desc_act": desc_act, "sym": False})
            (d / "config.json").write_text(json.dumps(cfg))
        return str(tmp_path / dirname)

    root = write("plain", False)
    outs = {}
    for sub in ("gptq", "awq"):
        llm = LLM(model=f"{root}/{sub}", device="cpu", dtype="float32", max_model_len=128, block_size=16,
                  num_gpu_blocks_override=64, max_num_seqs=8)
        assert llm.engine.model_config.quantization == "awq"
        o = llm.generate([[5, 9, 200, 17], list(range(10, 50))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logprobs=3))
        outs[sub] = [(list(x.outputs[0].token_ids), x.outputs[0].logprobs[0]) for x in o]
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs["gptq"] == outs["awq"]
    bad = write("actorder", True)
    with pytest.raises(NotImplementedError, match="desc_act"):
        LLM(model=f"{bad}/gptq", device="cpu", dtype="float32", max_model_len=128, num_gpu_blocks_override=64)
    pstate.destroy_model_parallel()


def test_a_checkpoint_that_lacks_a_parameter_is_refused(tmp_path):
    """Every parameter must come from the checkpoint: a file without one layer's
    norm weight must fail the load by name, not run on uninitialised memory."""
    import json
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy")
    src = tmp_path / "full"
    save_dummy_checkpoint(mc, str(src))
    sd = load_file(str(src / "model.safetensors"))
    gone = "model.layers.1.post_attention_layernorm.weight"
    assert gone in sd
    dst = tmp_path / "holed"
    dst.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items() if k != gone}, str(dst / "model.safetensors"))
    (dst / "config.json").write_text(json.dumps(mc.hf_config))
    with pytest.raises(Exception, match="post_attention_layernorm"):
        LLM(model=str(dst), device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8)
    pstate.destroy_model_parallel()


def test_packed_quantisation_is_refused_for_families_without_packed_layers():
    """OPT / Phi / StarCoder2 / GPT-2 style models build plain linear layers: asking for the AWQ kernels must
    fail at load with the reason, not leave the packed tensors unplaced."""
    with pytest.raises(Exception, match="no packed 4-bit linear layers"):
        LLM(model="opt-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, quantization="awq")
    pstate.destroy_model_parallel()


@pytest.mark.parametrize("model", ["glm-tiny", "helium-tiny", "stablelm-tiny", "granite-tiny"])
def test_awq_q_k_projections_follow_the_rope_row_order(model):
    """Families whose rotary embedding is folded into the row order of q_proj / k_proj: with packed AWQ weights
    the same re-ordering is applied to the nibbles, zeros and scales, so the packed model generates what its
    dequantised twin (dense rows re-ordered) does.  Granite: the residual multiplier folded into o_proj /
    down_proj goes into the AWQ scales, never into the packed words."""
    outs = []
    for q in ("awq-dequant", "awq"):
        llm = LLM(model=model, device="cpu", dtype="float32", max_model_len=128, block_size=16,
                  num_gpu_blocks_override=64, max_num_seqs=8, quantization=q)
        o = llm.generate([[5, 9, 200, 17], list(range(10, 40))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        outs.append([x.outputs[0].token_ids for x in o])
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("model", ["glm-tiny", "granite-tiny"])
def test_nvfp4_q_k_projections_follow_the_rope_row_order(tmp_path, model):
    """GLM (adjacent-pair rotary over half of each head) with NVFP4 weights: packed bytes and block scales of
    q_proj / k_proj are re-ordered row-wise like dense rows, so the packed model generates what the dense
    checkpoint built from the dequantised tensors does.  Granite: the residual multiplier of o_proj / down_proj
    goes into the per-matrix scale of the packed tensors."""
    import json
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.layers.linear import nvfp4_dequantize
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    mc = ModelConfig(model=model, dtype="float32", load_format="dummy", quantization="nvfp4")
    src = tmp_path / "packed"
    save_dummy_checkpoint(mc, str(src))
    sd = load_file(str(src / "model.safetensors"))
    dense = {}
    for k, v in sd.items():
        if k.endswith(".weight_packed"):
            base = k[:-len(".weight_packed")]
            alpha = sd[base + ".weight_global_scale"].reciprocal()
            dense[base + ".weight"] = nvfp4_dequantize(v, sd[base + ".weight_scale"], alpha.expand(v.shape[0]),
                                                       torch.float32)
        elif not k.endswith((".weight_global_scale", ".weight_scale")):
            dense[k] = v
    d = tmp_path / "dense"
    d.mkdir()
    save_file({k: v.contiguous() for k, v in dense.items()}, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps(dict(mc.hf_config)))
    outs = []
    for kw in (dict(model=model, quantization="nvfp4"), dict(model=str(d))):
        llm = LLM(device="cpu", dtype="float32", max_model_len=128, block_size=16, num_gpu_blocks_override=64,
                  max_num_seqs=8, **kw)
        o = llm.generate([[5, 9, 200, 17], list(range(10, 50))],
                         SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        outs.append([x.outputs[0].token_ids for x in o])
        llm.shutdown()
        pstate.destroy_model_parallel()
    assert outs[0] == outs[1]


def test_gpt_oss_mxfp4_checkpoint_equals_its_widened_twin(tmp_path):
    """The original gpt-oss release stores the expert matrices as MXFP4 (``*_blocks`` / ``*_scales``, output
    channels in rows, gate and up interleaved).  A checkpoint written that way from quantised values must load
    to the model its widened (dense ``[E, in, out]``) twin loads to; encoder and decoder follow the format's
    definition (E2M1 values, one power-of-two scale per 32; parity with a released file is unpinned)."""
    from safetensors.torch import load_file, save_file
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.gpt_oss import mxfp4_dequantize, mxfp4_quantize
    from vllm_distributed_amd.models.loader import save_dummy_checkpoint
    w = torch.randn(6, 64) * torch.logspace(-3, 2, 6).unsqueeze(1)
    blocks, scales = mxfp4_quantize(w)
    back = mxfp4_dequantize(blocks, scales, torch.float32)
    assert blocks.shape == (6, 2, 16) and scales.shape == (6, 2)
    assert ((back - w).abs() <= 0.26 * w.abs().reshape(6, 2, 32).amax(-1).repeat_interleave(32, 1) + 1e-9).all()
    b2, s2 = mxfp4_quantize(back)
    assert torch.equal(mxfp4_dequantize(b2, s2, torch.float32), back)            # a fixed point

    mc = ModelConfig(model="gpt-oss-tiny", dtype="float32", load_format="dummy")
    dense_dir, mx_dir = tmp_path / "dense", tmp_path / "mxfp4"
    save_dummy_checkpoint(mc, str(dense_dir))
    tensors = load_file(str(dense_dir / "model.safetensors"))
    dense, packed = {}, {}
    for name, t in tensors.items():
        if name.endswith((".experts.gate_up_proj", ".experts.down_proj")):
            rows = t.transpose(1, 2).contiguous()                     # [E, out, in]: the release's orientation
            blk, sc = mxfp4_quantize(rows)
            packed[name + "_blocks"], packed[name + "_scales"] = blk, sc
            dense[name] = mxfp4_dequantize(blk, sc, torch.float32).transpose(1, 2).contiguous()
        else:
            dense[name] = packed[name] = t
    save_file(dense, str(dense_dir / "model.safetensors"))
    mx_dir.mkdir()
    save_file(packed, str(mx_dir / "model.safetensors"))
    cfg = dict(mc.hf_config, quantization_config={"quant_method": "mxfp4"})
    import json
    (mx_dir / "config.json").write_text(json.dumps(cfg))
    outs = {}
    for tag, d in (("dense", dense_dir), ("mxfp4", mx_dir)):
        llm = LLM(model=str(d), device="cpu", dtype="float32", max_model_len=128, block_size=16,
                  num_gpu_blocks_override=64, max_num_seqs=8)
        o = llm.generate([[5, 9, 200, 17], list(range(10, 50))],
                         SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True, logprobs=3))
        outs[tag] = [(x.outputs[0].token_ids, x.outputs[0].logprobs[0]) for x in o]
        llm.shutdown()
        pstate.destroy_model_parallel()
    for (ta, la), (tb, lb) in zip(outs["dense"], outs["mxfp4"]):
        assert ta == tb and all(abs(la[t] - lb[t]) < 1e-5 for t in la)


def test_hot_attention_kernels_fit_the_register_file(tmp_path):
    """The attention kernels are written against exact register budgets (four waves per SIMD = 128 registers
    in decode, the whole file in the LDS-staged prefill kernel).  A few spilled registers - reloaded inside the
    key loop - halve their bandwidth, and any change near them can tip the allocator: the compiled kernels the
    serving paths use must have no scratch segment."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = tmp_path / "attention.s"
    subprocess.run([hipcc, "-S", "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17",
                    "-I" + os.path.join(root, "csrc"), os.path.join(root, "csrc", "attention.hip"), "-o", str(asm)],
                   check=True, capture_output=True, timeout=600)
    spilled = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm.read_text(), re.S):
        scratch = int(re.search(r"private_segment_fixed_size (\d+)", m.group(2)).group(1))
        if scratch:
            spilled[m.group(1)] = scratch
    hot = [k for k in spilled if ("paged_decode_kernel" in k or "paged_prefill_kernel" in k
                                  or "paged_prefill_lds_kernel" in k or "paged_prefill_lds64" in k)]
    assert not hot, {k: spilled[k] for k in hot}
    # known and accepted: the one-wave decode kernel at head size 256 (not a default path), the LDS prefill
    # kernel's sinks / ALiBi instantiation, and the LDS prefill kernel at head size 256 (it holds the whole
    # 512-register file and parks a few loop-invariant values; 1.2 - 2.1x the fragment-direct kernel regardless)
    assert all("wave_kernel" in k and "Li256E" in k or "lds_bias_kernel" in k or "prefill_lds256" in k
               for k in spilled), spilled
    assert all(v <= 160 for k, v in spilled.items() if "prefill_lds" in k), spilled


def test_eagle_head_checkpoint_in_the_authors_layout(tmp_path):
    """EAGLE's published head checkpoints name their tensors without a ``model.`` prefix, carry a copy of the
    embedding, ``fc`` with a bias, one decoder layer without an input norm, and no LM head: the file loads, the
    embedding in it is used, the LM head is the target's.  (Synthetic tensors of those names and shapes: no
    EAGLE checkpoint ships with the reference.)"""
    import json
    import torch
    from safetensors.torch import save_file
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import dummy_weights_iterator, save_dummy_checkpoint
    target_dir, head_dir = str(tmp_path / "target"), tmp_path / "head"
    save_dummy_checkpoint(ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy"), target_dir)
    hmc = ModelConfig(model="eagle-llama-tiny", dtype="float32", load_format="dummy")
    hmc.hf_config = dict(hmc.hf_config, architectures=["EagleLlamaForCausalLM"])
    tensors = {n[len("model."):]: w.contiguous() for n, w in dummy_weights_iterator(hmc, torch.device("cpu"))}
    assert "fc.bias" in tensors and "layers.0.input_layernorm.weight" not in tensors and "lm_head.weight" not in tensors
    tensors["embed_tokens.weight"] = torch.randn(512, 256) * 0.05
    head_dir.mkdir()
    save_file(tensors, str(head_dir / "model.safetensors"))
    (head_dir / "config.json").write_text(json.dumps(dict(hmc.hf_config, architectures=["LlamaForCausalLM"])))
    kw = dict(model=target_dir, device="cpu", dtype="float32", max_model_len=128, num_gpu_blocks_override=64,
              max_num_seqs=4)
    prompts = [{"prompt_token_ids": list(range(30, 50))}, {"prompt_token_ids": [7, 8, 9] * 5}]
    sp = SamplingParams(temperature=0.0, max_tokens=24, ignore_eos=True)
    ref = LLM(**kw)
    want = [o.outputs[0].token_ids for o in ref.generate(prompts, sp)]
    llm = LLM(speculative_config={"method": "eagle", "model": str(head_dir), "num_speculative_tokens": 2}, **kw)
    runner = llm.engine.core.executor.driver.worker.model_runner
    head = runner.draft.model
    assert head.embed_tokens is not runner.model.embed_tokens and head.lm_head is runner.model.lm_head
    assert torch.equal(head.embed_tokens.weight[:512], tensors["embed_tokens.weight"])
    assert head.layers["0"].input_layernorm is None and head.norm is None
    assert [o.outputs[0].token_ids for o in llm.generate(prompts, sp)] == want
    assert llm.engine.core.scheduler.spec_stats["drafted"] > 0
    ref.shutdown()
    llm.shutdown()


def test_eagle3_head_checkpoint_in_the_authors_layout(tmp_path):
    """Published EAGLE-3 heads: ``fc.weight`` (3 H wide), the layer under ``midlayer.`` with ``hidden_norm`` and a
    q / k / v input of 2 H, ``norm.weight``, an LM head over the draft vocabulary, the int64 ``d2t`` offsets and a
    boolean ``t2d`` mask, no embedding: the file loads through the safetensors reader (``d2t`` keeps its integers),
    the embedding is the target's, and decoding with it is lossless.  (Synthetic tensors of those names and
    shapes: no EAGLE-3 checkpoint ships with the reference.)"""
    import json
    import torch
    from safetensors.torch import save_file
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.config import ModelConfig
    from vllm_distributed_amd.models.loader import dummy_weights_iterator, save_dummy_checkpoint
    target_dir, head_dir = str(tmp_path / "target"), tmp_path / "head"
    save_dummy_checkpoint(ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy"), target_dir)
    hmc = ModelConfig(model="eagle3-llama-tiny", dtype="float32", load_format="dummy")
    tensors = {n: w.contiguous() for n, w in dummy_weights_iterator(hmc, torch.device("cpu"))}
    assert {"fc.weight", "d2t", "norm.weight", "lm_head.weight", "midlayer.hidden_norm.weight",
            "midlayer.self_attn.q_proj.weight"} <= set(tensors) and "embed_tokens.weight" not in tensors
    assert tensors["d2t"].dtype == torch.int64 and tensors["midlayer.self_attn.q_proj.weight"].shape == (256, 512)
    t2d = torch.zeros(512, dtype=torch.bool)
    t2d[torch.arange(128) + tensors["d2t"]] = True
    tensors["t2d"] = t2d
    head_dir.mkdir()
    save_file(tensors, str(head_dir / "model.safetensors"))
    (head_dir / "config.json").write_text(json.dumps(hmc.hf_config))
    kw = dict(model=target_dir, device="cpu", dtype="float32", max_model_len=128, num_gpu_blocks_override=64,
              max_num_seqs=4)
    prompts = [{"prompt_token_ids": list(range(30, 50))}, {"prompt_token_ids": [7, 8, 9] * 5}]
    sp = SamplingParams(temperature=0.0, max_tokens=24, ignore_eos=True)
    ref = LLM(**kw)
    want = [o.outputs[0].token_ids for o in ref.generate(prompts, sp)]
    llm = LLM(speculative_config={"method": "eagle3", "model": str(head_dir), "num_speculative_tokens": 2}, **kw)
    runner = llm.engine.core.executor.driver.worker.model_runner
    head = runner.draft.model
    assert head.embed_tokens is runner.model.embed_tokens and head.lm_head is not runner.model.lm_head
    assert torch.equal(head.d2t, tensors["d2t"]) and int(head.d2t[-1]) > 0
    assert torch.equal(head.layers["0"].hidden_norm.weight, tensors["midlayer.hidden_norm.weight"])
    assert [o.outputs[0].token_ids for o in llm.generate(prompts, sp)] == want
    assert llm.engine.core.scheduler.spec_stats["drafted"] > 0
    ref.shutdown()
    llm.shutdown()
