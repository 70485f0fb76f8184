"""SPMD (externally launched ranks, the layout bench.py uses under torchrun):
two gloo ranks must produce the single-process tokens, identically on both."""
import json
import os
import subprocess
import sys

import pytest

from vllm_distributed_amd.utils import get_open_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {root!r})
from vllm_distributed_amd import SamplingParams
from vllm_distributed_amd.engine.arg_utils import EngineArgs
from vllm_distributed_amd.engine.llm_engine import LLMEngine
tp = int(os.environ.get("WORLD_SIZE", "1"))
eng = LLMEngine.from_engine_args(EngineArgs(
    model=os.environ.get("MODEL") or "llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
    num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=tp,
    distributed_executor_backend="spmd" if tp > 1 else "uni",
    async_scheduling=os.environ.get("ASYNC") == "1", quantization=os.environ.get("QUANT") or None,
    speculative_config=json.loads(os.environ.get("SPEC") or "null"),
    enable_expert_parallel=os.environ.get("EP") == "1"))
sp = SamplingParams(temperature=0.8, top_p=0.9, seed=7, max_tokens=8, ignore_eos=True)
g = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
prompts = [([5, 9, 200, 17], g), (list(range(10, 50)), sp), ([7, 7], g)]
if os.environ.get("IMAGES") == "1":        # an image in front of the first two prompts (placeholder id 511)
    import numpy as np
    from PIL import Image
    pics = [Image.fromarray((np.random.default_rng(k).random((32, 40, 3)) * 255).astype("uint8")) for k in (1, 2)]
    prompts = [({{"prompt_token_ids": [511] + prompts[k][0], "multi_modal_data": {{"image": pics[k]}}}}, prompts[k][1])
               for k in (0, 1)] + prompts[2:]
ids = [eng.add_request(None, p, s) for p, s in prompts]
final = {{}}
while eng.has_unfinished_requests():
    for o in eng.step():
        if o.finished:
            final[o.request_id] = o.outputs[0].token_ids
print("RESULT " + json.dumps([final[i] for i in ids]), flush=True)
eng.shutdown()
"""


def _run(world, async_sched=False, sp="0", quant="", model="", spec="", ep=False, images=False):
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VLLM_LOGGING_LEVEL="WARNING",
                   GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2", ASYNC="1" if async_sched else "0",
                   VDAMD_SEQUENCE_PARALLEL=sp, QUANT=quant, MODEL=model, SPEC=spec, EP="1" if ep else "0",
                   IMAGES="1" if images else "0")
        procs.append(subprocess.Popen([sys.executable, "-c", SCRIPT.format(root=ROOT)], env=env,
                                      stdout=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, _ = p.communicate(timeout=240)
        assert p.returncode == 0, so
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[len("RESULT "):]))
    return outs


@pytest.mark.slow
def test_spmd_tp2_matches_single_process_on_every_rank():
    ref = _run(1)[0]
    got = _run(2)
    assert got[0] == ref and got[1] == ref
    got = _run(2, async_sched=True)        # one step scheduled ahead on every rank
    assert got[0] == ref and got[1] == ref
    # sequence-parallel residual stream wherever the token count divides by TP
    got = _run(2, sp="force")
    assert got[0] == ref and got[1] == ref


@pytest.mark.slow
@pytest.mark.parametrize("model", ["qwen2-moe-tiny", "gemma3-tiny", "phi-tiny", "olmo2-tiny", "cohere-tiny",
                                   "glm4-tiny", "nemotron-tiny", "deepseek-v3-tiny", "falcon-tiny", "gptj-tiny",
                                   "gpt-bigcode-tiny", "smollm3-tiny", "exaone4-tiny", "stablelm-tiny", "arcee-tiny",
                                   "glm4-moe-tiny", "helium-tiny", "gemma2-tiny", "starcoder2-tiny", "gpt-neox-tiny",
                                   "gpt2-tiny", "opt-tiny", "granite-tiny", "mixtral-tiny"])
def test_spmd_tp2_other_families_match_single_process(model):
    """Shared-expert MoE (routed and shared halves under one all-reduce), Gemma-3 (norms on the branch
    outputs: must stay on the plain layer loop even when the sharded residual stream is forced) and Phi
    (parallel branches, biased vocab-parallel LM head) at TP=2 reproduce TP=1."""
    ref = _run(1, model=model)[0]
    got = _run(2, model=model, sp="force")
    assert got[0] == ref and got[1] == ref


@pytest.mark.slow
@pytest.mark.parametrize("model", ["qwen3-moe-tiny", "qwen2-moe-tiny", "mixtral-tiny", "deepseek-v3-tiny",
                                   "glm4-moe-tiny"])
def test_spmd_tp2_expert_parallel_matches_single_process(model):
    """Whole experts per rank (``enable_expert_parallel``) instead of column shards of every expert; shared
    experts and attention stay tensor-parallel."""
    ref = _run(1, model=model)[0]
    got = _run(2, model=model, ep=True)
    assert got[0] == ref and got[1] == ref


@pytest.mark.slow
def test_spmd_tp2_draft_model_speculation_matches_single_process():
    """Target and draft model both sharded over the TP group: every rank proposes and verifies the
    same tokens, and the greedy request reads as without speculation."""
    ref = _run(1)[0]
    got = _run(2, spec='{"model": "phi-tiny", "num_speculative_tokens": 3}')
    assert got[0][0] == ref[0] and got[0][2] == ref[2] and got[1] == got[0]


@pytest.mark.slow
@pytest.mark.parametrize("name", ["llava-tiny", "llava-next-tiny"])
def test_spmd_tp2_llava_with_images_matches_single_process(name):
    """The vision tower is replicated, the decoder sharded: steps that carry image rows leave the fused /
    sequence-parallel forms of the embedding (forced here) for the plain one, and TP=2 reproduces TP=1."""
    ref = _run(1, model=name, images=True)[0]
    plain = _run(1, model=name)[0]
    assert ref[2] == plain[2] and ref[0] != plain[0]
    got = _run(2, model=name, images=True, sp="force")
    assert got[0] == ref and got[1] == ref


@pytest.mark.slow
@pytest.mark.parametrize("method", ["eagle", "eagle3"])
def test_spmd_tp2_eagle_head_speculation_matches_single_process(method):
    """An EAGLE head sharded like the target, sharing its vocabulary-parallel embedding (and, EAGLE-1, its LM head;
    an EAGLE-3 head's own one over the draft vocabulary is sharded by rows, and the target runs the tapped plain
    layer loop on every rank)."""
    ref = _run(1)[0]
    got = _run(2, spec='{"method": "%s", "model": "%s-llama-tiny", "num_speculative_tokens": 3}' % (method, method))
    assert got[0][0] == ref[0] and got[0][2] == ref[2] and got[1] == got[0]


@pytest.mark.slow
@pytest.mark.parametrize("quant", ["awq", "nvfp4"])
def test_spmd_tp2_quantised_shards_match_single_process(quant):
    """Packed 4-bit tensors (integer groups / fp4 blocks with per-matrix
    scalars) are cut per rank before repacking: TP=2 reproduces TP=1."""
    ref = _run(1, quant=quant)[0]
    got = _run(2, quant=quant)
    assert got[0] == ref and got[1] == ref


BENCH_ARGS = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--device", "cpu", "--model", "llama-tiny",
              "--dtype", "float32", "--batch-per-gpu", "4", "--prompt-len", "12", "--ttft-requests", "1",
              "--strong-batch", "4"]


def _bench_json(cmd, env):
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.slow
@pytest.mark.parametrize("executor", ["auto", "spmd"])
def test_bench_script_multi_rank_rehearsal_on_cpu(executor):
    """bench.py exactly as the driver launches it for N=2 (torch.distributed.run,
    one rank per device), on gloo with a tiny model: one JSON line from rank 0, every
    launched process exits 0.  By default rank 0 is the head and rank 1 registers
    with it as a worker; ``--executor spmd`` is the launcher-only lower bound."""
    port = get_open_port()
    env = dict(os.environ, VLLM_LOGGING_LEVEL="ERROR", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           *BENCH_ARGS, "--executor", executor]
    j = _bench_json(cmd, env)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["config"]["global_batch"] == 8
    assert j["config"]["parallelism"] == "tp2" and j["value"] > 0
    if executor == "auto":
        assert j["config"]["executor"] == "distributed"
        assert "head + 1 registered workers" in j["config"]["launch"]
        assert j["rpc_overhead_ms"] > 0
    else:
        assert j["config"]["executor"] == "spmd"


@pytest.mark.slow
def test_bench_script_without_a_launcher_spawns_its_workers():
    """``python bench.py --gpus 2`` alone: the head spawns both workers and drives them
    over the RPC control plane; weak- and strong-scaling points and the RPC overhead are reported."""
    env = dict(os.environ, VLLM_LOGGING_LEVEL="ERROR", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    j = _bench_json([sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS[:-2], "--strong-batch", "2"], env)
    assert j["n_gpus"] == 2 and j["config"]["executor"] == "distributed"
    assert j["config"]["launch"] == "head spawns every worker"
    assert j["scaling"] == "weak" and j["config"]["global_batch"] == 8
    assert j["strong_scaling"]["scaling"] == "strong" and j["strong_scaling"]["global_batch"] == 2
    assert j["strong_scaling"]["tokens_per_s"] > 0
    bd = j["config"]["step_breakdown_ms"]
    assert bd["steps"] == 3 and bd["rpc_overhead_ms"] == pytest.approx(bd["rpc_in_ms"] + bd["rpc_out_ms"], abs=0.01)


@pytest.mark.slow
def test_bench_reports_the_pipeline_layout_next_to_the_headline():
    """``bench.py --gpus 2`` measures TP = 2 (the headline layout) and then, in a child process, PP = 2
    with the running sequences dealt out over two micro-batches; the child's numbers arrive as
    ``alt_parallelism`` and a failure there would only leave an ``error`` field."""
    env = dict(os.environ, VLLM_LOGGING_LEVEL="ERROR", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    j = _bench_json([sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS[:-2], "--strong-batch", "0",
                     "--alt-parallelism", "pp"], env)
    assert j["config"]["parallelism"] == "tp2" and "TP over N GPUs" in j["metric"]
    alt = j["alt_parallelism"]
    assert alt is not None and "error" not in alt, alt
    assert alt["parallelism"] == "pp2" and alt["tokens_per_s"] > 0 and alt["global_batch"] == 8
    # asked for directly, the pipeline layout is the reported one and no second run happens
    j = _bench_json([sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS[:-2], "--strong-batch", "0",
                     "--parallelism", "pp"], env)
    assert j["config"]["parallelism"] == "pp2" and j["alt_parallelism"] is None and "PP over N GPUs" in j["metric"]
