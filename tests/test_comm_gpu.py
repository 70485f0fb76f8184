"""Data-plane communication on a real GPU: the ctypes RCCL binding (single rank
- RCCL refuses several ranks on one device) and the peer-mapped direct
all-reduce between several processes sharing GPU 0 over hipIpc handles."""
import json
import os
import subprocess
import sys

import pytest
import torch

from vllm_distributed_amd.utils import get_open_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CAR_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {root!r})
import torch
from vllm_distributed_amd.parallel import state as pstate
from vllm_distributed_amd.parallel.custom_all_reduce import CustomAllReduce
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
pstate.init_distributed_environment(world, rank, "tcp://127.0.0.1:" + os.environ["MASTER_PORT"], 0, "gloo", dev)
pstate.initialize_model_parallel(world, 1)
g = pstate.get_tp_group()
car = CustomAllReduce(g, max_bytes=4 << 20)
ok = True
worst = 0.0
for it in range(6):
    for n, mode in ((8, 1), (4096, 1), (8192 * 7, 1), (8192 * 7, 2), (1 << 20, 2), (2 << 20, 0), (1000 * 8, 0)):
        for dtype in (torch.bfloat16, torch.float16):
            torch.manual_seed(1000 * it + n + rank)
            x = torch.randn(n, device=dev).to(dtype)
            out = car.all_reduce(x, mode)
            torch.cuda.synchronize()
            parts = g.all_gather_object(x.float().cpu())
            ref = sum(parts).to(dtype).float()
            err = (out.float().cpu() - ref).abs().max().item()
            worst = max(worst, err)
            tol = 0.05 * world
            if not err <= tol:
                ok = False
            bits = g.all_gather_object(out.cpu().view(torch.int16).sum().item())
            if len(set(bits)) != 1:       # every rank must hold identical bits
                ok = False
# graph capture + replay
x = torch.ones(4096, device=dev, dtype=torch.bfloat16) * (rank + 1)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    car.all_reduce(x)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    y = car.all_reduce(x)
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
ok = ok and bool((y.float() == sum(range(1, world + 1))).all().item())
ok = ok and not car.timed_out()
print("RESULT " + json.dumps(dict(ok=ok, worst=worst)), flush=True)
g.barrier()
car.close()
"""


def test_rccl_binding_single_rank_and_graph_capture():
    from vllm_distributed_amd.parallel.rccl import RcclCommunicator
    torch.cuda.set_device(0)
    comm = RcclCommunicator(rank=0, world_size=1)
    x = torch.randn(1 << 16, device="cuda", dtype=torch.bfloat16)
    ref = x.clone()
    comm.all_reduce(x)
    out = torch.empty_like(x)
    comm.all_gather(out, x)
    comm.broadcast(x, 0)
    rs = torch.empty_like(x)
    comm.reduce_scatter(rs, x)
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and torch.equal(out, ref) and torch.equal(rs, ref)
    g = torch.cuda.CUDAGraph()
    y = torch.full((1024,), 3.0, device="cuda", dtype=torch.bfloat16)
    with torch.cuda.graph(g):
        comm.all_reduce(y)
    g.replay()
    torch.cuda.synchronize()
    assert bool((y == 3.0).all())
    comm.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_direct_all_reduce_between_processes_sharing_one_gpu(world):
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT=str(port),
                   VLLM_LOGGING_LEVEL="WARNING", GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, "-c", CAR_SCRIPT.format(root=ROOT)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    results = []
    for p in procs:
        so, _ = p.communicate(timeout=110)
        assert p.returncode == 0, so[-3000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][-1]
        results.append(json.loads(line[7:]))
    assert all(r["ok"] for r in results), results


NCCL_SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from vllm_distributed_amd.parallel import state as pstate
from vllm_distributed_amd.worker.worker import Worker
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = os.environ["PORT"]
Worker._default_loopback_interfaces(type("W", (), dict(distributed_init_method="env://"))())
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
# exactly the calls a TP worker makes, on a world of one rank
pstate.init_distributed_environment(1, 0, "env://", 0, "nccl", dev, force=True)
w = pstate.get_world_group()
assert w.cpu_group is not None and dist.get_backend() == "nccl"
g = dist.new_group([0], backend="nccl")
x = torch.arange(1024, device=dev, dtype=torch.bfloat16)
dist.all_reduce(x, group=g)
out = torch.empty(1024, device=dev, dtype=torch.bfloat16)
dist.all_gather_into_tensor(out, x, group=g)
assert torch.equal(out, x)
assert w.all_gather_object(7) == [7]
w.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
pstate.destroy_model_parallel()
print("RESULT ok", flush=True)
"""


def test_nccl_process_group_bringup_calls_on_one_rank():
    """The torch.distributed call sequence of a GPU worker (env:// rendezvous on
    loopback, nccl + gloo groups, device_id binding) on a one-rank world."""
    env = dict(os.environ, PORT=str(get_open_port()), VLLM_LOGGING_LEVEL="WARNING")
    env.pop("GLOO_SOCKET_IFNAME", None)
    p = subprocess.run([sys.executable, "-c", NCCL_SCRIPT.format(root=ROOT)], env=env, capture_output=True,
                       text=True, timeout=110)
    assert p.returncode == 0 and "RESULT ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


SPMD_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {root!r})
from vllm_distributed_amd import SamplingParams
from vllm_distributed_amd.engine.arg_utils import EngineArgs
from vllm_distributed_amd.engine.llm_engine import LLMEngine
tp = int(os.environ.get("WORLD_SIZE", "1"))
eng = LLMEngine.from_engine_args(EngineArgs(
    model=os.environ.get("MODEL", "llama-tiny"), device="cuda", dtype="bfloat16", max_model_len=256,
    num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=tp,
    distributed_executor_backend="spmd" if tp > 1 else "uni", enable_expert_parallel=os.environ.get("EP") == "1",
    async_scheduling=os.environ.get("ASYNC") == "1", gpu_memory_utilization=0.2,
    hf_overrides=json.loads(os.environ.get("OVERRIDES", "{{}}"))))
g = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logprobs=8)
ids = [eng.add_request(None, p, g) for p in ([5, 9, 200, 17], list(range(10, 90)), [7, 7])]
final, first_lp = {{}}, {{}}
while eng.has_unfinished_requests():
    for o in eng.step():
        if o.finished:
            final[o.request_id] = o.outputs[0].token_ids
            first_lp[o.request_id] = {{str(k): v for k, v in o.outputs[0].logprobs[0].items()}}
from vllm_distributed_amd.parallel import state as pstate
print("RESULT " + json.dumps(dict(tokens=[final[i] for i in ids], first_lp=[first_lp[i] for i in ids],
                                  direct_ar=pstate.get_tp_group().custom_ar is not None)), flush=True)
eng.shutdown()
"""


def _run_spmd_shared(world, async_sched, sp="0", overrides=None, model="llama-tiny", ep=False):
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VLLM_LOGGING_LEVEL="WARNING", GLOO_SOCKET_IFNAME="lo",
                   VDAMD_SHARE_DEVICE="1", ASYNC="1" if async_sched else "0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   VDAMD_SEQUENCE_PARALLEL=sp, OVERRIDES=json.dumps(overrides or {}), MODEL=model,
                   EP="1" if ep else "0")
        procs.append(subprocess.Popen([sys.executable, "-c", SPMD_SCRIPT.format(root=ROOT)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        so, _ = p.communicate(timeout=110)
        assert p.returncode == 0, so[-3000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[len("RESULT "):]))
    return outs


def test_spmd_tensor_parallel_ranks_sharing_one_gpu_match_single_rank():
    """The multi-rank GPU path end to end (SPMD executor, sharded HIP-kernel
    model, peer-mapped all-reduce, vocab-parallel argmax) with ranks mapped
    onto one device.  bf16 partial sums are reduced in a different order than
    the unsharded GEMM, so the first tokens (largest margins) are compared."""
    ref = _run_spmd_shared(1, False)[0]["tokens"]
    for async_sched, sp in ((False, "0"), (True, "0"), (False, "force")):
        got = _run_spmd_shared(2, async_sched, sp)
        assert got[0]["tokens"] == got[1]["tokens"]
        assert got[0]["direct_ar"] and got[1]["direct_ar"]
        assert [t[:2] for t in got[0]["tokens"]] == [t[:2] for t in ref]


@pytest.mark.parametrize("tp,pp", [(1, 2), (2, 1)])
def test_head_plus_remote_worker_sharing_one_gpu(monkeypatch, tp, pp):
    """The distributed executor on the GPU: the head spawns rank 0, a `remote`
    worker process registers over TCP and becomes rank 1 (second pipeline stage,
    or the other half of the TP group); both are mapped onto the one device.
    Exercises placement, the RPC fan-out, (hidden, residual) send/recv between
    stages and sharded HIP-kernel execution under the real worker processes."""
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.parallel import state as pstate

    prompts = [[5, 9, 200, 17, 3, 3, 8], list(range(10, 90)), [7, 7, 7]]

    def generate(engine):
        sp = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)
        ids = [engine.add_request(None, p, sp) for p in prompts]
        final = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                if o.finished:
                    final[o.request_id] = o.outputs[0].token_ids
        return [final[i] for i in ids]

    common = dict(model="llama-tiny", device="cuda", dtype="bfloat16", max_model_len=256,
                  num_gpu_blocks_override=64, max_num_seqs=8, gpu_memory_utilization=0.2)
    pstate.destroy_model_parallel()
    eng = LLMEngine.from_engine_args(EngineArgs(**common))
    ref = generate(eng)
    eng.shutdown()
    pstate.destroy_model_parallel()

    port = get_open_port()
    env = dict(VLLM_SERVER_PORT=str(port), VLLM_RPC_SECRET="test-secret", VLLM_HOST_IP="127.0.0.1",
               VLLM_LOGGING_LEVEL="WARNING", VLLM_REMOTE_RETRY_SECONDS="0.5", GLOO_SOCKET_IFNAME="lo",
               PYTHONPATH=ROOT, VDAMD_SHARE_DEVICE="1", VLLM_HEAD_LOCAL_DEVICES="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    remote = subprocess.Popen([sys.executable, os.path.join(ROOT, "launch.py"), "remote", "127.0.0.1",
                               "--num-devices", "1", "--device-offset", "1"], env=dict(os.environ, **env))
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            tensor_parallel_size=tp, pipeline_parallel_size=pp, distributed_executor_backend="distributed",
            **common))
        assert [a.node_id is None for a in eng.core.executor.assignments] == [True, False]
        got = generate(eng)
        assert [t[:2] for t in got] == [t[:2] for t in ref]
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=20)


@pytest.mark.parametrize("world", [4, 8])
def test_spmd_wide_tensor_parallel_sharing_one_gpu(world):
    """TP=4 and TP=8 sharding (one KV head and 16-32 columns per rank at TP=8)
    through the HIP kernels, the peer-mapped all-reduce between 4 / 8 processes
    and the vocab-parallel argmax, against the unsharded model."""
    ov = {"num_attention_heads": 8, "num_key_value_heads": 8, "head_dim": 64, "intermediate_size": 1024}
    ref = _run_spmd_shared(1, False, overrides=ov)[0]
    got = _run_spmd_shared(world, False, overrides=ov)
    assert all(g["tokens"] == got[0]["tokens"] for g in got)
    assert all(g["direct_ar"] for g in got)
    # bf16 partial sums are added in a different order than in the unsharded GEMM, which can
    # flip near-ties of a random-init model: compare the first-token distributions instead
    for lp_ref, lp_tp, toks in zip(ref["first_lp"], got[0]["first_lp"], ref["tokens"]):
        top = str(toks[0])
        assert top in lp_tp and abs(lp_tp[top] - lp_ref[top]) < 0.1, (lp_ref, lp_tp)


@pytest.mark.parametrize("ep", [False, True])
def test_spmd_moe_tensor_and_expert_parallel_sharing_one_gpu(ep):
    """A cut-down Qwen3-Coder (head_dim 128, 128-aligned experts: HIP routing kernels and grouped
    expert GEMMs) at TP=2 with tensor-sharded and with expert-parallel experts, against one rank."""
    ov = dict(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, moe_intermediate_size=256,
              num_experts=16, num_experts_per_tok=4, num_hidden_layers=2, vocab_size=1024, max_position_embeddings=512)
    ref = _run_spmd_shared(1, False, overrides=ov, model="qwen3-coder-480b-a35b")[0]
    got = _run_spmd_shared(2, False, overrides=ov, model="qwen3-coder-480b-a35b", ep=ep)
    assert got[0]["tokens"] == got[1]["tokens"]
    for lp_ref, lp_tp, toks in zip(ref["first_lp"], got[0]["first_lp"], ref["tokens"]):
        top = str(toks[0])
        assert top in lp_tp and abs(lp_tp[top] - lp_ref[top]) < 0.15, (lp_ref, lp_tp)
