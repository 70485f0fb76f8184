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
car = CustomAllReduce(g, max_bytes=4 << 20, ranks_per_device=world)
car.fence_start = os.environ.get("FENCE_START") == "1"
ok, why = True, []
worst = 0.0

def fail(msg):
    global ok
    ok = False
    why.append(msg)

def gathered(x):
    return g.all_gather_object(x.float().cpu())

# ---- all-reduce: one-shot / two-shot, from the stage area and from ordinary memory
for it in range(4):
    for n, two in ((8, False), (4096, False), (8192 * 7, False), (8192 * 7, True), (1 << 20, True),
                   (2 << 20, None), (1000 * 8, None), (40, True)):
        for dtype in (torch.bfloat16, torch.float16):
            torch.manual_seed(1000 * it + n + rank)
            x = torch.randn(n, device=dev).to(dtype)
            src = x
            if it % 2:
                src = car.stage_view((n,), dtype)
                src.copy_(x)
            out = car.all_reduce(src, two_shot=two)
            torch.cuda.synchronize()
            ref = sum(gathered(x)).to(dtype).float()
            err = (out.float().cpu() - ref).abs().max().item()
            worst = max(worst, err)
            if not err <= 0.05 * world:
                fail(f"all_reduce n={{n}} two={{two}} err={{err}}")
            bits = g.all_gather_object(out.cpu().view(torch.int16).sum().item())
            if len(set(bits)) != 1:       # every rank must hold identical bits
                fail(f"all_reduce n={{n}}: ranks disagree")

# ---- all-reduce + residual add + RMSNorm, residual replicated and row-sharded
def norm_ref(parts, res, w, eps):
    tot = (sum(parts) + res.float().cpu()).to(res.dtype).float()
    return tot, tot * torch.rsqrt(tot.pow(2).mean(-1, keepdim=True) + eps) * w.float().cpu()

for it, (rows, h) in enumerate(((1, 512), (5, 512), (64, 1024), (200, 8192), (37, 4096), (3, 8192))):
    for sharded in (False, True):
        torch.manual_seed(77 * it + rank)
        x = torch.randn(rows, h, device=dev).to(torch.bfloat16)
        torch.manual_seed(5 * it)                       # same residual and weight on every rank
        res = torch.randn(rows, h, device=dev).to(torch.bfloat16)
        w = (torch.rand(h, device=dev) + 0.5).to(torch.bfloat16)
        per, mine = car.shard_rows(rows)
        lo = min(rank * per, rows)
        my_res = torch.zeros(per, h, dtype=res.dtype, device=dev) if sharded else res.clone()
        if sharded:
            my_res[:mine] = res[lo:lo + mine]
        src = x
        if it % 2 == 0:
            src = car.stage_view((rows, h), torch.bfloat16)
            src.copy_(x)
        got = car.all_reduce_add_rms_norm(src, my_res, w, 1e-5, sharded)
        torch.cuda.synchronize()
        tot, want = norm_ref(gathered(x), res, w, 1e-5)
        e1 = (got.float().cpu() - want).abs().max().item()
        got_res = my_res[:mine].float().cpu() if sharded else my_res.float().cpu()
        e2 = (got_res - (tot[lo:lo + mine] if sharded else tot)).abs().max().item() if got_res.numel() else 0.0
        worst = max(worst, e1)
        if not (e1 <= 0.06 and e2 <= 0.05 * world):
            fail(f"fused norm rows={{rows}} h={{h}} sharded={{sharded}} e1={{e1}} e2={{e2}}")
        bits = g.all_gather_object(got.cpu().view(torch.int16).sum().item())
        if len(set(bits)) != 1:
            fail(f"fused norm rows={{rows}}: ranks disagree")

# ---- all-gather of arbitrary 16-byte-multiple pieces
for dtype, shape in ((torch.float32, (3, 8)), (torch.bfloat16, (40, 16)), (torch.int32, (4,))):
    piece = (torch.arange(int(torch.tensor(shape).prod()), device=dev).reshape(shape) + 1000 * rank).to(dtype)
    out = car.all_gather(piece)
    want = torch.cat([(torch.arange(piece.numel(), device=dev).reshape(shape) + 1000 * q).to(dtype) for q in range(world)])
    if not torch.equal(out, want):
        fail(f"all_gather {{dtype}}")
car.end_step()
if car.parity != 0:
    fail("parity after end_step")

# ---- one captured graph with every form, replayed 100 times on changing data
rows, h = 48, 1024
xin = torch.zeros(rows, h, device=dev, dtype=torch.bfloat16)
res_r = torch.zeros(rows, h, device=dev, dtype=torch.bfloat16)
per, mine = car.shard_rows(rows)
lo = min(rank * per, rows)
res_s = torch.zeros(per, h, device=dev, dtype=torch.bfloat16)
w = torch.ones(h, device=dev, dtype=torch.bfloat16)

def body():
    a = car.stage_view((rows, h), torch.bfloat16)
    a.copy_(xin)
    y0 = car.all_reduce(a, two_shot=False)
    b = car.stage_view((rows, h), torch.bfloat16)
    b.copy_(xin)
    y1 = car.all_reduce_add_rms_norm(b, res_r, w, 1e-5, False)
    y2 = car.all_reduce_add_rms_norm(xin, res_s, w, 1e-5, True)      # input copied into the stage area
    y3 = car.all_gather(y0[:2])
    y4 = car.all_reduce(xin, two_shot=True)
    car.end_step()
    return y0, y1, y2, y3, y4

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    outs = body()
tri = world * (world + 1) / 2
for rep in range(100):
    val = float(rep % 7 + 1)
    xin.fill_(val * (rank + 1) / 8)
    res_r.zero_()
    res_s.zero_()
    gr.replay()
    if rep % 10 == 9 or rep < 3:
        torch.cuda.synchronize()
        tot = val * tri / 8
        y0, y1, y2, y3, y4 = outs
        good = (bool((y0.float() == tot).all()) and bool((y4.float() == tot).all())
                and bool((res_r.float() == tot).all()) and bool((res_s[:mine].float() == tot).all())
                and bool((y1.float() - 1.0).abs().max() < 0.02) and bool((y2.float() - 1.0).abs().max() < 0.02)
                and bool((y3.float() == tot).all()) and y3.shape[0] == 2 * world)
        if not good:
            fail(f"graph replay {{rep}}")
            break
torch.cuda.synchronize()
if car.timed_out():
    fail("timed out")
print("RESULT " + json.dumps(dict(ok=ok, worst=worst, why=why[:6])), flush=True)
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
    # grouped point-to-point: a send to and a receive from this same rank is a real device copy
    src = torch.arange(4096, device="cuda", dtype=torch.float32)
    dst = torch.zeros_like(src)
    comm.send_recv(src, 0, dst, 0)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)
    # ... and, captured, a graph that is NOT empty: every replay moves the current contents
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.send_recv(src, 0, dst, 0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        comm.send_recv(src, 0, dst, 0)
    for k in range(3):
        src.fill_(float(k + 5))
        dst.zero_()
        g2.replay()
        torch.cuda.synchronize()
        assert bool((dst == float(k + 5)).all()), "captured RCCL send/recv did not run on replay"
    del g, g2             # RCCL waits for graphs that captured its calls before it lets go of the communicator
    comm.close()


@pytest.mark.parametrize("world,fence_start", [(2, False), (4, False), (8, False), (4, True)])
def test_direct_collectives_between_processes_sharing_one_gpu(world, fence_start):
    """All-reduce (one- and two-shot, staged and copied input), all-reduce + add + RMSNorm with
    replicated and row-sharded residual, all-gather, and one hipGraph holding all of them
    replayed 100 times on changing data, between 2 / 4 / 8 processes over hipIpc mappings."""
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT=str(port),
                   VLLM_LOGGING_LEVEL="WARNING", GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   FENCE_START="1" if fence_start else "0")
        procs.append(subprocess.Popen([sys.executable, "-c", CAR_SCRIPT.format(root=ROOT)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    results = []
    for p in procs:
        so, _ = p.communicate(timeout=115)
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
    speculative_config=json.loads(os.environ.get("SPEC", "null")),
    hf_overrides=json.loads(os.environ.get("OVERRIDES", "{{}}"))))
g = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logprobs=int(os.environ.get("LOGPROBS", "8")))
ids = [eng.add_request(None, p, g) for p in ([5, 9, 200, 17], list(range(10, 90)), [7, 7])]
final, first_lp = {{}}, {{}}
while eng.has_unfinished_requests():
    for o in eng.step():
        if o.finished:
            final[o.request_id] = o.outputs[0].token_ids
            first_lp[o.request_id] = {{str(k): v for k, v in o.outputs[0].logprobs[0].items()}}
from vllm_distributed_amd.parallel import state as pstate
runner = eng.core.executor.driver.worker.model_runner
print("RESULT " + json.dumps(dict(tokens=[final[i] for i in ids], first_lp=[first_lp[i] for i in ids],
                                  direct_ar=pstate.get_tp_group().custom_ar is not None,
                                  rccl=pstate.get_tp_group().rccl is not None,
                                  graphs=list(runner.graph_sizes), fallback=runner.graph_fallback_reason,
                                  paths=dict(getattr(runner.model, "path_counts", {{}})))), flush=True)
eng.shutdown()
"""


def _assert_distributions_close(ref_lps, got_lps, atol=0.08, rtol=2e-2, min_entries=512):
    """Full-vocabulary log-probabilities of the first generated token (fp32, one vector per
    prompt) of a sharded run against the unsharded model.  A mis-sharded weight or a wrong
    reduction shifts logits by their own order of magnitude; bf16 summation order by ~1e-2."""
    worst = 0.0
    for ref, got in zip(ref_lps, got_lps):
        assert len(ref) >= min_entries and set(ref) == set(got)
        for tok, v in ref.items():
            d = abs(got[tok] - v)
            worst = max(worst, d)
            assert d <= atol + rtol * abs(v), (tok, v, got[tok])
    return worst


def _run_spmd_shared(world, async_sched, sp="0", overrides=None, model="llama-tiny", ep=False, logprobs=8,
                     extra_env=None, share=True):
    """``share``: every rank on GPU 0 (gloo, host-staged); otherwise rank r runs on GPU r over RCCL."""
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VLLM_LOGGING_LEVEL="WARNING", GLOO_SOCKET_IFNAME="lo",
                   VDAMD_SHARE_DEVICE="1" if share else "0", ASYNC="1" if async_sched else "0",
                   HSA_ENABLE_IPC_MODE_LEGACY="0",
                   VDAMD_SEQUENCE_PARALLEL=sp, OVERRIDES=json.dumps(overrides or {}), MODEL=model,
                   EP="1" if ep else "0", LOGPROBS=str(logprobs), **(extra_env or {}))
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
    ref = _run_spmd_shared(1, False, logprobs=512)[0]
    # fused all-reduce + norm inside captured decode graphs (default), the same eagerly, the
    # unfused peer-mapped all-reduce, and the sequence-parallel form over host-staged gloo
    for async_sched, sp, extra in ((False, "0", {}), (True, "0", {}), (False, "0", {"VDAMD_USE_HIP_GRAPH": "0"}),
                                   (False, "0", {"VDAMD_FUSED_ALL_REDUCE_NORM": "0"}),
                                   (False, "force", {"VDAMD_FUSED_ALL_REDUCE_NORM": "0"})):
        got = _run_spmd_shared(2, async_sched, sp, logprobs=512, extra_env=extra)
        assert got[0]["tokens"] == got[1]["tokens"]
        assert got[0]["direct_ar"] and got[1]["direct_ar"]
        if extra.get("VDAMD_USE_HIP_GRAPH") != "0" and sp != "force":
            # the decode graph must really hold the step (peer-mapped collectives are capturable;
            # the host-staged gloo reduce-scatter of the forced sequence-parallel form is not, and
            # that run exercises the fall-back to eager decode after a failed capture)
            assert got[0]["graphs"], got[0]["fallback"]
        _assert_distributions_close(ref["first_lp"], got[0]["first_lp"])


@pytest.mark.parametrize("method", ["eagle", "eagle3"])
def test_spmd_speculative_heads_tensor_parallel_sharing_one_gpu(method):
    """EAGLE / EAGLE-3 heads next to a sharded target over the peer-mapped collectives, decode from captured
    graphs: both ranks produce the same tokens, and the greedy text starts as the single-rank run's does.  (The
    EAGLE-3 target runs the tapped plain layer loop - its all-reduces are the unfused peer-mapped ones - and its
    graphs carry the three taps.)"""
    spec = json.dumps({"method": method, "model": f"{method}-llama-tiny", "num_speculative_tokens": 3})
    ref = _run_spmd_shared(1, False)[0]
    got = _run_spmd_shared(2, False, extra_env={"SPEC": spec})
    assert got[0]["tokens"] == got[1]["tokens"]
    assert got[0]["direct_ar"] and got[0]["graphs"], got[0]["fallback"]
    # decode-sized steps keep the fused all-reduce + norm (EAGLE-3: the taps are copies of the replicated residual
    # stream); an EAGLE-3 target's larger, row-sharded steps take the plain loop
    assert got[0]["paths"].get("fused_replicated", 0) > 0, got[0]["paths"]
    for a, b in zip(got[0]["tokens"], ref["tokens"]):
        assert len(a) == len(b) and a[:2] == b[:2], (a, b)       # (bf16 sums in another order: the first tokens)


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
        sp = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logprobs=512)
        ids = [engine.add_request(None, p, sp) for p in prompts]
        final = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                if o.finished:
                    final[o.request_id] = o.outputs[0].logprobs[0]
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
        _assert_distributions_close(ref, got)
        # decode runs from captured graphs on every rank: per pipeline stage (the transfers stay
        # outside the graph), and under TP with the peer-mapped collectives inside it
        stats = eng.core.executor.collective_rpc("get_stats")
        assert all(st["hip_graph_sizes"] for st in stats), [st["hip_graph_fallback"] for st in stats]
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
    ref = _run_spmd_shared(1, False, overrides=ov, logprobs=512)[0]
    got = _run_spmd_shared(world, False, overrides=ov, logprobs=512)
    assert all(g["tokens"] == got[0]["tokens"] for g in got)
    assert all(g["direct_ar"] for g in got)
    assert got[0]["graphs"], got[0]["fallback"]
    # bf16 partial sums are added in a different order than in the unsharded GEMM, which can
    # flip near-ties of a random-init model: compare the whole first-token distributions
    _assert_distributions_close(ref["first_lp"], got[0]["first_lp"])


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


@pytest.mark.parametrize("ep", [False, True])
def test_spmd_deepseek_latent_cache_tensor_parallel_sharing_one_gpu(ep):
    """DeepSeek-V3 routing + multi-head latent attention at TP=2 on the HIP kernels: every rank caches the same
    latent rows and attends with its half of the heads (absorbed up-projections cut per rank), the grouped
    routing kernel feeds tensor-sharded or expert-parallel experts; against one rank."""
    ref = _run_spmd_shared(1, False, model="deepseek-v3-tiny")[0]
    got = _run_spmd_shared(2, False, model="deepseek-v3-tiny", ep=ep)
    assert got[0]["tokens"] == got[1]["tokens"]
    for lp_ref, lp_tp, toks in zip(ref["first_lp"], got[0]["first_lp"], ref["tokens"]):
        top = str(toks[0])
        assert top in lp_tp and abs(lp_tp[top] - lp_ref[top]) < 0.2, (lp_ref, lp_tp)


def _shared_gpu_env(monkeypatch, **extra):
    env = dict(VLLM_SERVER_PORT=str(get_open_port()), VLLM_RPC_SECRET="test-secret", VLLM_HOST_IP="127.0.0.1",
               VLLM_LOGGING_LEVEL="WARNING", GLOO_SOCKET_IFNAME="lo", PYTHONPATH=ROOT, VDAMD_SHARE_DEVICE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


def test_tp2_x_pp2_sharing_one_gpu_runs_the_fused_collectives_on_both_stages(monkeypatch):
    """The reference's documented deployment shape (`-tp 2 -pp 2`, reference ``.env.server:11``) with the
    peer-mapped fused all-reduce + add + RMSNorm inside BOTH pipeline stages, in both of its forms: the
    replicated one (small messages) and the row-sharded one (the residual shards are gathered once at the
    stage boundary).  Every rank must report the fused path and nothing else, decode must run from
    graphs, and the first-token distributions must match the unsharded model."""
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.parallel import state as pstate

    prompts = [[5, 9, 200, 17, 3, 3, 8], list(range(10, 90)), [7, 7, 7], [11, 12, 13, 14]]

    def generate(engine):
        sp = SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logprobs=512)
        ids = [engine.add_request(None, p, sp) for p in prompts]
        final = {}
        while engine.has_unfinished_requests():
            for o in engine.step():
                if o.finished:
                    final[o.request_id] = o.outputs[0].logprobs[0]
        return [final[i] for i in ids]

    common = dict(model="llama-tiny", device="cuda", dtype="bfloat16", max_model_len=256,
                  num_gpu_blocks_override=64, max_num_seqs=8, gpu_memory_utilization=0.15)
    pstate.destroy_model_parallel()
    eng = LLMEngine.from_engine_args(EngineArgs(**common))
    ref = generate(eng)
    eng.shutdown()
    pstate.destroy_model_parallel()
    for one_shot, want in (("0", "fused_sharded"), (str(1 << 30), "fused_replicated")):
        _shared_gpu_env(monkeypatch, VLLM_HEAD_LOCAL_DEVICES="4", VDAMD_CUSTOM_ALL_REDUCE_ONE_SHOT_BYTES=one_shot)
        eng = None
        try:
            eng = LLMEngine.from_engine_args(EngineArgs(
                tensor_parallel_size=2, pipeline_parallel_size=2, distributed_executor_backend="distributed", **common))
            got = generate(eng)
            stats = eng.core.executor.collective_rpc("get_stats")
            assert len(stats) == 4
            for st in stats:
                assert st["tp_direct_all_reduce"], st
                assert set(st["tp_forward_paths"]) == {want}, st["tp_forward_paths"]
                assert st["hip_graph_sizes"], st["hip_graph_fallback"]
                assert st["tp_direct_timed_out"] is False
            _assert_distributions_close(ref, got)
        finally:
            if eng is not None:
                eng.shutdown()


def test_peer_that_never_arrives_is_reported_and_fails_the_executor(monkeypatch):
    """Rank 1 of a TP=2 group stops issuing work at its 6th step (a hung device).  Rank 0's peer-mapped
    kernels give up waiting after the configured bound (they never hang the GPU) and set the timeout
    word; whichever comes first - the worker reading that word after the step, or the step's enforced
    RPC timeout when a host-staged collective of the step blocks first - fails the executor (a serving
    head then exits non-zero), instead of serving sums that miss a rank."""
    import time

    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    from vllm_distributed_amd.engine.llm_engine import LLMEngine
    from vllm_distributed_amd.parallel import state as pstate
    pstate.destroy_model_parallel()
    _shared_gpu_env(monkeypatch, VLLM_HEAD_LOCAL_DEVICES="2", VDAMD_CUSTOM_ALL_REDUCE_TIMEOUT_MS="300",
                    VDAMD_FAULT_STALL_AT="1:6", VLLM_EXECUTE_MODEL_TIMEOUT_SECONDS="8")
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cuda", dtype="bfloat16", max_model_len=256, num_gpu_blocks_override=64,
            max_num_seqs=8, gpu_memory_utilization=0.15, tensor_parallel_size=2,
            distributed_executor_backend="distributed"))
        ex = eng.core.executor
        failed = []
        ex.register_failure_callback(lambda: failed.append(True))
        eng.add_request(None, [5, 9, 200, 17], SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True))
        t0 = time.time()
        with pytest.raises(Exception) as err:
            while eng.has_unfinished_requests():
                eng.step()
        assert time.time() - t0 < 40
        deadline = time.time() + 10
        while not ex.is_failed and time.time() < deadline:
            time.sleep(0.05)
        assert ex.is_failed and failed
        assert any(k in str(err.value) for k in ("gave up waiting", "Executor failed", "timed out")), str(err.value)
    finally:
        if eng is not None:
            eng.shutdown()


def test_device_to_device_kv_hand_over_between_two_engines(tmp_path):
    """Disaggregated prefill on one machine: a producer engine (its own process) prefills and hands the
    prompts' KV blocks over as hipIpc device buffers, a consumer engine (another process) maps them and
    copies device to device, and generates what an engine that computed everything itself generates.
    No KV byte is written to the mailbox directory, and the producer frees each buffer once the consumer
    has acknowledged it."""
    import json
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kv_handover_proc.py")
    env = dict(os.environ, VLLM_LOGGING_LEVEL="WARNING", HSA_ENABLE_IPC_MODE_LEGACY="0")

    def result(line_source):
        for line in line_source:
            if line.startswith("RESULT "):
                return json.loads(line[7:])
        raise AssertionError("child ended without a result")

    def run(role):
        r = subprocess.run([sys.executable, script, role, str(tmp_path)], env=env, capture_output=True, text=True,
                           timeout=170)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return result(r.stdout.splitlines())

    ref = run("reference")
    prod = subprocess.Popen([sys.executable, script, "producer", str(tmp_path)], env=env, stdin=subprocess.PIPE,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        first = result(iter(prod.stdout.readline, ""))
        assert [t[0] for t in first["tokens"]] == [t[0] for t in ref["tokens"]]
        assert first["num_saved_requests"] == 2 and first["device_bytes_sent"] > 0
        assert first["exported_bytes"] == first["device_bytes_sent"]
        files = [f for _, _, fs in os.walk(str(tmp_path)) for f in fs]
        assert files and not [f for f in files if f.endswith(".safetensors")], files
        got = run("consumer")
        assert got["tokens"] == ref["tokens"]
        assert got["num_loaded_tokens"] == 64 + 32 and got["device_bytes_received"] == first["device_bytes_sent"]
        prod.stdin.write("reap\n")
        prod.stdin.flush()
        after = result(iter(prod.stdout.readline, ""))
        assert after["exported_bytes"] == 0                        # acknowledged buffers were freed
        assert not [f for _, _, fs in os.walk(str(tmp_path)) for f in fs]
        prod.stdin.write("exit\n")
        prod.stdin.flush()
        assert prod.wait(timeout=60) == 0, prod.stderr.read()[-3000:]
    finally:
        if prod.poll() is None:
            prod.kill()
