"""LoRA adapters: a PEFT directory applied at run time must give the tokens of
the same checkpoint with ``W + scaling * B A`` merged into the weights - for
mixed batches (base + two adapters in one step), under TP=2, and with slot
eviction when more adapters than slots take turns."""
import json
import os
import subprocess
import sys

import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.config import ModelConfig
from vllm_distributed_amd.lora import LoRARequest
from vllm_distributed_amd.models.loader import save_dummy_checkpoint
from vllm_distributed_amd.parallel import state as pstate
from vllm_distributed_amd.utils import get_open_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROMPTS = [[5, 9, 200, 17], list(range(10, 50)), [7, 7, 300]]
KW = dict(device="cpu", dtype="float32", max_model_len=128, block_size=16, num_gpu_blocks_override=64, max_num_seqs=8)


def make_adapter(base_sd, path, seed, rank=4, alpha=8.0, targets=("q_proj", "v_proj", "o_proj", "gate_proj", "down_proj")):
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(seed)
    tensors, merged = {}, dict(base_sd)
    for name, w in base_sd.items():
        proj = name.split(".")[-2]
        if not name.endswith(".weight") or proj not in targets or ".layers." not in name:
            continue
        out_f, in_f = w.shape
        a = torch.randn(rank, in_f, generator=g) * 0.3
        b = torch.randn(out_f, rank, generator=g) * 0.3
        mod = name[:-len(".weight")]
        tensors[f"base_model.model.{mod}.lora_A.weight"] = a
        tensors[f"base_model.model.{mod}.lora_B.weight"] = b
        merged[name] = w + (alpha / rank) * (b @ a)
    os.makedirs(path, exist_ok=True)
    save_file(tensors, os.path.join(path, "adapter_model.safetensors"))
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump({"r": rank, "lora_alpha": alpha, "target_modules": list(targets), "peft_type": "LORA"}, f)
    return merged


@pytest.fixture(scope="module")
def fixtures(tmp_path_factory):
    from safetensors.torch import load_file, save_file
    root = tmp_path_factory.mktemp("lora")
    mc = ModelConfig(model="llama-tiny", dtype="float32", load_format="dummy")
    base = root / "base"
    save_dummy_checkpoint(mc, str(base))
    sd = load_file(str(base / "model.safetensors"))
    out = {"base": str(base)}
    for i, tag in enumerate(("a", "b", "c")):
        merged = make_adapter(sd, str(root / f"adapter_{tag}"), seed=10 + i)
        d = root / f"merged_{tag}"
        d.mkdir()
        save_file({k: v.contiguous() for k, v in merged.items()}, str(d / "model.safetensors"))
        (d / "config.json").write_text(json.dumps(mc.hf_config))
        out[f"adapter_{tag}"] = str(root / f"adapter_{tag}")
        out[f"merged_{tag}"] = str(d)
    return out


def _gen(model, prompts, loras=None, **kw):
    llm = LLM(model=model, **dict(KW, **kw))
    try:
        o = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), lora_request=loras)
        return [x.outputs[0].token_ids for x in o], llm
    finally:
        llm.shutdown()
        pstate.destroy_model_parallel()


def test_mixed_batch_matches_merged_checkpoints_and_slots_are_recycled(fixtures):
    ref = {t: _gen(fixtures[k], PROMPTS)[0] for t, k in (("base", "base"), ("a", "merged_a"), ("b", "merged_b"),
                                                         ("c", "merged_c"))}
    assert ref["a"] != ref["base"] and ref["a"] != ref["b"]
    la, lb, lc = (LoRARequest(t, i + 1, fixtures[f"adapter_{t}"]) for i, t in enumerate("abc"))
    # one step mixes base rows and two adapters
    prompts = PROMPTS * 3
    loras = [None] * 3 + [la] * 3 + [lb] * 3
    got, _ = _gen(fixtures["base"], prompts, loras, enable_lora=True, max_loras=2, max_lora_rank=8)
    assert got[:3] == ref["base"] and got[3:6] == ref["a"] and got[6:] == ref["b"]
    # three adapters through two slots: the scheduler holds the third back until a slot is free
    got, llm = _gen(fixtures["base"], PROMPTS * 3, [la] * 3 + [lb] * 3 + [lc] * 3, enable_lora=True, max_loras=2,
                    max_lora_rank=8)
    assert got[:3] == ref["a"] and got[3:6] == ref["b"] and got[6:] == ref["c"]


def test_lora_errors_are_reported(fixtures):
    la = LoRARequest("a", 1, fixtures["adapter_a"])
    llm = LLM(model=fixtures["base"], **KW)
    with pytest.raises(ValueError, match="enable-lora"):
        llm.generate([PROMPTS[0]], SamplingParams(max_tokens=2), lora_request=la)
    llm.shutdown()
    pstate.destroy_model_parallel()
    llm = LLM(model=fixtures["base"], enable_lora=True, max_lora_rank=2, **KW)
    core = llm.engine.core
    with pytest.raises(Exception, match="rank 4 > max_lora_rank"):
        core.add_lora(la)
    llm.shutdown()
    pstate.destroy_model_parallel()
    llm = LLM(model=fixtures["base"], enable_lora=True, max_lora_rank=4, **KW)
    core = llm.engine.core
    assert core.add_lora(la) and core.list_loras() == [1] and core.pin_lora(1)
    assert core.remove_lora(1) and core.list_loras() == []
    llm.shutdown()
    pstate.destroy_model_parallel()


SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {root!r})
from vllm_distributed_amd import SamplingParams
from vllm_distributed_amd.engine.arg_utils import EngineArgs
from vllm_distributed_amd.engine.llm_engine import LLMEngine
from vllm_distributed_amd.lora import LoRARequest
tp = int(os.environ["WORLD_SIZE"])
eng = LLMEngine.from_engine_args(EngineArgs(
    model={base!r}, device="cpu", dtype="float32", max_model_len=128, block_size=16, num_gpu_blocks_override=64,
    max_num_seqs=8, tensor_parallel_size=tp, distributed_executor_backend="spmd", enable_lora=True, max_loras=2,
    max_lora_rank=8))
g = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)
la = LoRARequest("a", 1, {adapter!r})
ids = [eng.add_request(None, p, g, lora_request=l) for p, l in (({p0!r}, None), ({p1!r}, la), ({p2!r}, la))]
final = {{}}
while eng.has_unfinished_requests():
    for o in eng.step():
        if o.finished:
            final[o.request_id] = o.outputs[0].token_ids
print("RESULT " + json.dumps([final[i] for i in ids]), flush=True)
eng.shutdown()
"""


@pytest.mark.slow
def test_lora_under_tensor_parallel_shards(fixtures):
    ref_base = _gen(fixtures["base"], PROMPTS[:1])[0]
    ref_a = _gen(fixtures["merged_a"], PROMPTS[1:])[0]
    port = get_open_port()
    src = SCRIPT.format(root=ROOT, base=fixtures["base"], adapter=fixtures["adapter_a"], p0=PROMPTS[0],
                        p1=PROMPTS[1], p2=PROMPTS[2])
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VLLM_LOGGING_LEVEL="WARNING", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, "-c", src], env=env, stdout=subprocess.PIPE, text=True))
    for p in procs:
        so, _ = p.communicate(timeout=240)
        assert p.returncode == 0, so
        got = json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
        assert got[:1] == ref_base and got[1:] == ref_a


def test_openai_server_routes_adapter_names(fixtures):
    """``--lora-modules name=path`` serves the adapter under its own model name;
    adapters can be loaded and unloaded at run time."""
    import asyncio
    import httpx
    from vllm_distributed_amd.entrypoints.cli import build_parser
    from vllm_distributed_amd.entrypoints.openai import api_server
    pstate.destroy_model_parallel()
    args = build_parser().parse_args(
        (f"serve {fixtures['base']} --device cpu --dtype float32 --max-model-len 128 --block-size 16 "
         f"--num-gpu-blocks-override 64 --served-model-name tiny --port 0 --enable-lora --max-loras 2 "
         f"--max-lora-rank 8 --lora-modules ada={fixtures['adapter_a']} --disable-frontend-multiprocessing").split())
    args.model = args.model_tag
    def lp(model):
        llm = LLM(model=model, **KW)
        o = llm.generate([PROMPTS[1]], SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logprobs=0))
        llm.shutdown()
        pstate.destroy_model_parallel()
        return [round(next(iter(d.values())), 3) for d in o[0].outputs[0].logprobs]
    ref_base, ref_a, ref_b = lp(fixtures["base"]), lp(fixtures["merged_a"]), lp(fixtures["merged_b"])
    assert ref_base != ref_a != ref_b

    async def main():
        async with api_server.build_async_engine_client(args) as engine:
            app = api_server.build_app(args)
            await api_server.init_app_state(engine, await engine.get_vllm_config(), app.state, args)
            async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
                names = [m["id"] for m in (await c.get("/v1/models")).json()["data"]]
                assert names == ["tiny", "ada"]

                async def complete(model):
                    r = await c.post("/v1/completions", json={"model": model, "prompt": PROMPTS[1], "max_tokens": 6,
                                                              "temperature": 0, "ignore_eos": True, "logprobs": 0})
                    assert r.status_code == 200, r.text
                    return [round(x, 3) for x in r.json()["choices"][0]["logprobs"]["token_logprobs"]]
                assert (await complete("tiny")) == ref_base
                assert (await complete("ada")) == ref_a
                r = await c.post("/v1/load_lora_adapter", json={"lora_name": "adb", "lora_path": fixtures["adapter_b"]})
                assert r.status_code == 200, r.text
                assert (await complete("adb")) == ref_b
                r = await c.post("/v1/load_lora_adapter", json={"lora_name": "bad", "lora_path": "/nonexistent"})
                assert r.status_code == 400
                assert (await c.post("/v1/unload_lora_adapter", json={"lora_name": "adb"})).status_code == 200
                assert (await c.post("/v1/completions", json={"model": "adb", "prompt": "x"})).status_code == 404
    asyncio.run(main())
    pstate.destroy_model_parallel()


@pytest.mark.gpu
def test_lora_on_gpu_tracks_merged_checkpoint(fixtures):
    """bf16 on the GPU: base rows keep running through captured decode graphs,
    rows with an adapter take the eager path and track the merged weights."""
    kw = dict(dtype="bfloat16", max_model_len=128, num_gpu_blocks_override=64, max_num_seqs=8,
              gpu_memory_utilization=0.3)
    sp = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True, logprobs=20)

    def run(model, loras=None, **extra):
        llm = LLM(model=model, **dict(kw, **extra))
        o = llm.generate(PROMPTS, sp, lora_request=loras)
        llm.shutdown()
        pstate.destroy_model_parallel()
        return o
    la = LoRARequest("a", 1, fixtures["adapter_a"])
    merged = run(fixtures["merged_a"])
    base = run(fixtures["base"])
    mixed = run(fixtures["base"], [la, None, la], enable_lora=True, max_lora_rank=8)
    for got, want in ((mixed[0], merged[0]), (mixed[1], base[1]), (mixed[2], merged[2])):
        lg, lw = got.outputs[0].logprobs[0], want.outputs[0].logprobs[0]
        common = set(lg) & set(lw)
        assert len(common) >= 10
        assert max(abs(lg[t] - lw[t]) for t in common) < 0.12
    # the adapter changes the distribution by far more than that tolerance
    lb, lm = base[0].outputs[0].logprobs[0], merged[0].outputs[0].logprobs[0]
    common = set(lb) & set(lm)
    assert not common or max(abs(lb[t] - lm[t]) for t in common) > 0.3
