"""Head + remote workers on localhost over gloo (CPU): placement, TP/PP numerics
vs a single-process run, fail-stop behaviour."""
import os
import subprocess
import sys
import time

import pytest
from hypothesis import given, settings, strategies as st

from vllm_distributed_amd import SamplingParams
from vllm_distributed_amd.engine.arg_utils import EngineArgs
from vllm_distributed_amd.engine.llm_engine import LLMEngine
from vllm_distributed_amd.executor.placement import (NodeInfo, PlacementError, max_concurrent_batches,
                                                     output_rank, plan_placement)
from vllm_distributed_amd.parallel import state as pstate
from vllm_distributed_amd.utils import get_open_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------- placement
def test_placement_local_then_remote():
    plan = plan_placement(2, 2, 2, [NodeInfo("n1", (1, 0))])
    assert [(a.rank, a.node_id, a.local_rank) for a in plan] == [
        (0, None, 0), (1, None, 1), (2, "n1", 0), (3, "n1", 1)]


def test_placement_skips_small_node_not_the_stage():
    plan = plan_placement(2, 2, 0, [NodeInfo("small", (0,)), NodeInfo("big", (0, 1, 2, 3))])
    assert [a.node_id for a in plan] == ["big"] * 4
    assert [a.local_rank for a in plan] == [0, 1, 2, 3]      # a node can host several stages


def test_placement_same_host_split_tp_group():
    # head keeps 1 GPU, 7 worker processes on the same host offer GPUs 1..7 -> one TP=8 group
    plan = plan_placement(8, 1, 1, [NodeInfo("w", tuple(range(1, 8)), same_host=True)])
    assert [(a.node_id, a.local_rank) for a in plan] == [(None, 0)] + [("w", i) for i in range(1, 8)]
    with pytest.raises(PlacementError):       # TP groups never span hosts
        plan_placement(8, 1, 1, [NodeInfo("w", tuple(range(1, 8)), same_host=False)])


def test_placement_waits_when_short():
    with pytest.raises(PlacementError):
        plan_placement(1, 2, 1, [])


@given(tp=st.integers(1, 4), pp=st.integers(1, 4), local=st.integers(0, 8),
       nodes=st.lists(st.integers(0, 8), max_size=4))
@settings(max_examples=200, deadline=None)
def test_placement_properties(tp, pp, local, nodes):
    infos = [NodeInfo(f"n{i}", tuple(range(n))) for i, n in enumerate(nodes)]
    try:
        plan = plan_placement(tp, pp, local, infos)
    except PlacementError:
        # must really be impossible: count stages each host can take
        cap = local // tp + sum(n // tp for n in nodes)
        assert cap < pp
        return
    assert [a.rank for a in plan] == list(range(tp * pp))
    assert len({(a.node_id, a.local_rank) for a in plan}) == tp * pp        # no device used twice
    for a in plan:
        assert a.rank == a.pp_rank * tp + a.tp_rank
    for s in range(pp):                                                      # TP group on one node
        assert len({a.node_id for a in plan if a.pp_rank == s}) == 1


def test_output_rank_and_concurrency():
    assert output_rank(32, 8) == 24 and output_rank(4, 4) == 0
    assert max_concurrent_batches(4, False) == 4 and max_concurrent_batches(4, True) == 2
    assert max_concurrent_batches(1, False) == 1


# ------------------------------------------------------- multi-process runs
def _env(port, **extra):
    env = dict(os.environ)
    env.update(VLLM_SERVER_PORT=str(port), VLLM_RPC_SECRET="test-secret", VLLM_HOST_IP="127.0.0.1",
               VLLM_LOGGING_LEVEL="WARNING", VLLM_REMOTE_RETRY_SECONDS="0.5",
               GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2", PYTHONPATH=ROOT)
    env.update(extra)
    return env


def _start_remote(port, num_devices=1, device_offset=0, **extra):
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "launch.py"), "remote", "127.0.0.1",
                             "--num-devices", str(num_devices), "--device-offset", str(device_offset)],
                            env=_env(port, **extra))


PROMPTS = [[5, 9, 200, 17, 3, 3, 8], list(range(10, 60)), [7, 7, 7]]


def _generate(engine, max_tokens=10):
    sp = SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
    ids = [engine.add_request(None, p, sp) for p in PROMPTS]
    final = {}
    while engine.has_unfinished_requests():
        for o in engine.step():
            if o.finished:
                final[o.request_id] = o.outputs[0].token_ids
    return [final[i] for i in ids]


def _reference(model, **kw):
    pstate.destroy_model_parallel()
    eng = LLMEngine.from_engine_args(EngineArgs(
        model=model, device="cpu", dtype="float32", max_model_len=128, block_size=16,
        num_gpu_blocks_override=64, max_num_seqs=8, **kw))
    out = _generate(eng)
    eng.shutdown()
    pstate.destroy_model_parallel()
    return out


@pytest.mark.slow
@pytest.mark.parametrize("model,tp,pp,async_sched", [("opt-tiny", 1, 2, False), ("llama-tiny", 2, 1, False),
                                                     ("llama-tiny", 1, 2, False), ("llama-tiny", 2, 1, True),
                                                     # stage boundaries of the other layer shapes: the whole stream in
                                                     # the residual slot, tied heads on the last stage, per-layer rope
                                                     # tables, MLA + expert layers split across stages
                                                     ("gemma3-tiny", 1, 2, False), ("cohere-tiny", 1, 2, False),
                                                     ("smollm3-tiny", 1, 2, False), ("deepseek-v3-tiny", 1, 2, False),
                                                     ("falcon-tiny", 1, 2, False), ("gptj-tiny", 1, 2, False),
                                                     ("olmo2-tiny", 1, 2, False), ("qwen3-moe-tiny", 1, 2, False),
                                                     ("mixtral-tiny", 1, 2, False),
                                                     # expert biases added once across a sharded expert block,
                                                     # attention sinks cut to each rank's heads
                                                     ("gpt-oss-tiny", 2, 1, False)])
def test_head_plus_remote_worker_matches_single_process(monkeypatch, model, tp, pp, async_sched):
    """OPT greedy with head + 1 worker on localhost is the plumbing configuration;
    the TP=2 case assembles one TP group from the head's device and a same-host worker."""
    ref = _reference(model)
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="1").items():
        monkeypatch.setenv(k, v)
    remote = _start_remote(port, 1, 1)
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model=model, device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=tp,
            pipeline_parallel_size=pp, distributed_executor_backend="distributed",
            async_scheduling=async_sched))
        ex = eng.core.executor
        assert [a.node_id is None for a in ex.assignments] == [True, False]
        assert ex.max_concurrent_batches == (2 if async_sched else pp)
        ex.check_health()
        got = _generate(eng)
        assert got == ref
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=10)


@pytest.mark.slow
@pytest.mark.parametrize("tp,pp", [(2, 1), (1, 2)])
def test_request_level_features_through_remote_workers(monkeypatch, tp, pp):
    """Logit edits, JSON mode and prompt logprobs travel with the request to every worker
    (pickled sampling parameters, per-worker grammar state and vocabulary table) and give
    what a single process gives."""
    import json

    def run(eng):
        sps = [SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, logit_bias={42: 100.0}),
               SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True, allowed_token_ids=[3, 17, 99]),
               SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True, prompt_logprobs=0),
               SamplingParams(temperature=1.0, seed=1, max_tokens=60, json_object=True,
                              logit_bias={ord('"'): 4.0, ord("}"): 5.0, ord(":"): 3.0})]
        ids = [eng.add_request(None, PROMPTS[0], sp) for sp in sps]
        final = {}
        while eng.has_unfinished_requests():
            for o in eng.step():
                if o.finished:
                    final[o.request_id] = o
        return [final[i] for i in ids]

    pstate.destroy_model_parallel()
    single = LLMEngine.from_engine_args(EngineArgs(
        model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
        num_gpu_blocks_override=64, max_num_seqs=8))
    ref = run(single)
    single.shutdown()
    pstate.destroy_model_parallel()
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="1").items():
        monkeypatch.setenv(k, v)
    remote = _start_remote(port, 1, 1)
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=tp, pipeline_parallel_size=pp,
            distributed_executor_backend="distributed"))
        got = run(eng)
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=10)
    assert got[0].outputs[0].token_ids == [42] * 6
    assert set(got[1].outputs[0].token_ids) <= {3, 17, 99}
    assert got[1].outputs[0].token_ids == ref[1].outputs[0].token_ids
    for a, b in zip(got[2].prompt_logprobs[1:], ref[2].prompt_logprobs[1:]):
        (t, lp), = a.items()
        assert abs(lp - b[t]) < 1e-3
    assert got[3].outputs[0].token_ids == ref[3].outputs[0].token_ids
    if got[3].outputs[0].finish_reason == "stop":
        assert isinstance(json.loads(got[3].outputs[0].text), dict)


@pytest.mark.slow
def test_worker_death_fails_fast_and_unused_worker_is_tolerated(monkeypatch):
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="1").items():
        monkeypatch.setenv(k, v)
    # a worker that connects and leaves before being assigned must be tolerated
    early = _start_remote(port, 1, 1)
    time.sleep(0.5)
    early.terminate()
    early.wait(timeout=10)
    # this one is told (test-only fault hook) to die at its 4th execute_model
    remote = _start_remote(port, 1, 1, VDAMD_FAULT_DIE_AT="1:4")
    eng = None
    failed = []
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="opt-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, pipeline_parallel_size=2,
            distributed_executor_backend="distributed"))
        ex = eng.core.executor
        ex.register_failure_callback(lambda: failed.append(True))
        with pytest.raises(Exception):
            _generate(eng, max_tokens=50)
        deadline = time.time() + 10
        while not ex.is_failed and time.time() < deadline:
            time.sleep(0.05)
        assert ex.is_failed and failed
        with pytest.raises(RuntimeError, match="Executor failed"):
            ex.collective_rpc("check_health")
        late = []
        ex.register_failure_callback(lambda: late.append(True))   # registered late: fires at once
        assert late
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=10)


def test_kv_output_aggregator_counts_every_rank():
    import concurrent.futures
    from vllm_distributed_amd.engine.types import ModelRunnerOutput
    from vllm_distributed_amd.executor.kv_transfer import KVOutputAggregator
    agg = KVOutputAggregator(world_size=3)

    def out(tokens=None, sent=None, recv=None):
        o = ModelRunnerOutput(step_id=0, req_ids=["a"] if tokens else [], sampled_token_ids=[tokens] if tokens else [])
        o.finished_sending, o.finished_recving = sent, recv
        return o
    # a transfer is finished only when all three ranks have reported it
    r = agg.aggregate([out(sent={"x"}), out([7], sent={"x"}), out()], output_rank=1)
    assert r.sampled_token_ids == [[7]] and r.finished_sending is None
    r = agg.aggregate([out(recv={"y"}), out([8], recv={"y"}), out(sent={"x"}, recv={"y"})], output_rank=1)
    assert r.finished_sending == {"x"} and r.finished_recving == {"y"}
    futs = [concurrent.futures.Future() for _ in range(3)]
    res = agg.async_aggregate(futs, output_rank=2)
    for i, f in enumerate(futs):
        assert not res.done()
        f.set_result(out([i], sent={"z"}))
    assert res.result(timeout=1).sampled_token_ids == [[2]] and res.result().finished_sending == {"z"}
    bad = [concurrent.futures.Future() for _ in range(2)]
    res = KVOutputAggregator(2).async_aggregate(bad, 0)
    bad[1].set_exception(RuntimeError("worker died"))
    with pytest.raises(RuntimeError):
        res.result(timeout=1)


@pytest.mark.slow
def test_executor_contract_with_stub_workers(monkeypatch):
    """TP=2 x PP=2 over the head's one device and a three-device worker node,
    with a worker class that has no model: placement, rank order, the reply
    rank, per-rank replies of ``collective_rpc``, callables by value,
    ``non_block`` futures and the enforced timeout."""
    import cloudpickle  # noqa: F401  (callables travel by value)
    port = get_open_port()
    tests_dir = os.path.join(ROOT, "tests")
    env = _env(port, VLLM_HEAD_LOCAL_DEVICES="1", PYTHONPATH=ROOT + os.pathsep + tests_dir,
               VLLM_HEARTBEAT_SECONDS="0.2")            # health checks interleave with everything below
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.syspath_prepend(tests_dir)
    remote = subprocess.Popen([sys.executable, os.path.join(ROOT, "launch.py"), "remote", "127.0.0.1",
                               "--num-devices", "3", "--device-offset", "1"], env=env)
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=2, pipeline_parallel_size=2,
            distributed_executor_backend="distributed", worker_cls="stub_worker.StubWorker",
            kv_transfer_config={"kv_connector": "null", "kv_role": "kv_both"}))   # all ranks reply, aggregated
        ex = eng.core.executor
        assert ex.has_connector
        who = ex.collective_rpc("whoami")
        assert [w["rank"] for w in who] == [0, 1, 2, 3]
        assert [w["driver"] for w in who] == [True, False, True, False]
        assert len({w["pid"] for w in who}) == 4
        assert ex.output_rank == 2 and ex.max_concurrent_batches == 2
        # one TP group never spans nodes that are not the same host; every device used once
        assert len({(a.node_id, a.local_rank) for a in ex.assignments}) == 4
        assert ex.collective_rpc("nap", args=(0.0,), unique_reply_rank=3) == [3]
        assert ex.collective_rpc(lambda w: w.rank * 10) == [0, 10, 20, 30]
        futs = ex.collective_rpc("nap", args=(0.05,), non_block=True)
        assert [f.result(timeout=10) for f in futs] == [0, 1, 2, 3]
        with pytest.raises(Exception):
            ex.collective_rpc("nap", args=(3.0,), timeout=0.3)
        time.sleep(3.2)          # let the slow call drain before the next one (FIFO worker thread)
        # tokens come from the reply rank (2): each step adds rank + 1 = 3
        rid = eng.add_request(None, [5, 9, 100], SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True))
        final = None
        while eng.has_unfinished_requests():
            for o in eng.step():
                if o.request_id == rid and o.finished:
                    final = o.outputs[0].token_ids
        assert final == [103, 106, 109, 112]
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=10)


@pytest.mark.slow
def test_step_breakdown_adds_up_through_the_distributed_executor(monkeypatch):
    """With one step in flight at a time, the time the head spends blocked in
    ``execute_model`` is the sum of what the output rank reports: control-plane
    transit in, queueing, input preparation, device time (forward + sampling) and
    transit out.  Two workers spawned by the head, TP=2 over gloo."""
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="2").items():
        monkeypatch.setenv(k, v)
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, block_size=16,
            num_gpu_blocks_override=256, max_num_seqs=16, tensor_parallel_size=2,
            distributed_executor_backend="distributed", async_scheduling=False))
        sp = SamplingParams(temperature=0.0, max_tokens=48, ignore_eos=True)
        for i in range(16):
            eng.add_request(None, [3 + i, 9, 27, 81], sp)
        for _ in range(6):               # prefill + settle
            eng.step()
        timer = eng.core.timer
        timer.reset()
        t0 = time.perf_counter()
        for _ in range(30):
            eng.step()
        wall = (time.perf_counter() - t0) / 30 * 1e3
        s = timer.summary()
        assert s["steps"] == 30
        head = s["schedule_ms"] + s["execute_ms"] + s["update_ms"]
        assert head == pytest.approx(wall, rel=0.1)
        parts = (s["rpc_in_ms"] + s["worker_queued_ms"] + s["worker_prepare_ms"] + s["worker_forward_ms"]
                 + s["worker_sample_ms"] + s["rpc_out_ms"])
        assert parts == pytest.approx(s["execute_ms"], rel=0.2), s
        assert s["rpc_overhead_ms"] == pytest.approx(s["rpc_in_ms"] + s["rpc_out_ms"])
        # the device-side period (busy + idle between steps) is the step period
        assert s["device_period_ms"] == pytest.approx(wall, rel=0.2), s
    finally:
        if eng is not None:
            eng.shutdown()


def test_registration_channel_needs_a_real_secret_off_loopback(monkeypatch):
    """The head refuses to listen beyond loopback, and a worker refuses to connect to another
    host, with an empty or placeholder secret (pickles travel on that channel)."""
    from vllm_distributed_amd.executor.distributed import check_rpc_secret
    monkeypatch.delenv("VLLM_RPC_INSECURE", raising=False)
    assert check_rpc_secret("", "127.0.0.1", "bind") == ""
    assert check_rpc_secret("s3cr3t-9f2c", "0.0.0.0", "bind") == "s3cr3t-9f2c"
    for weak in ("", "change-me", "Change-Me"):
        with pytest.raises(RuntimeError, match="VLLM_RPC_SECRET"):
            check_rpc_secret(weak, "0.0.0.0", "bind")
        with pytest.raises(RuntimeError, match="VLLM_RPC_SECRET"):
            check_rpc_secret(weak, "192.168.2.145", "connect to")
    monkeypatch.setenv("VLLM_RPC_INSECURE", "1")
    assert check_rpc_secret("", "0.0.0.0", "bind") == ""
    monkeypatch.delenv("VLLM_RPC_INSECURE")
    # end to end: an explicit public bind without a secret fails engine start-up
    monkeypatch.setenv("VLLM_RPC_BIND", "0.0.0.0")
    monkeypatch.setenv("VLLM_RPC_SECRET", "")
    monkeypatch.setenv("VLLM_SERVER_PORT", str(get_open_port()))
    monkeypatch.setenv("VLLM_HEAD_LOCAL_DEVICES", "2")
    with pytest.raises(RuntimeError, match="VLLM_RPC_SECRET"):
        LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=2,
            distributed_executor_backend="distributed"))
    # remote role, other host, no secret
    from vllm_distributed_amd.launch import remote_main
    with pytest.raises(RuntimeError, match="VLLM_RPC_SECRET"):
        remote_main("192.168.2.145", num_devices=1)


@pytest.mark.slow
def test_tp2_x_pp2_head_plus_three_remote_workers_two_batches_in_flight_then_a_kill(monkeypatch):
    """The reference's documented deployment shape (`-tp 2 -pp 2`): the head's own device plus a
    three-device worker node, real model, gloo.  Two micro-batches are in flight (one per stage),
    tokens match a single process; then a worker of the second stage dies mid-run and the head
    fails fast."""
    ref = _reference("llama-tiny")
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="1").items():
        monkeypatch.setenv(k, v)
    common = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
                  num_gpu_blocks_override=64, max_num_seqs=2, tensor_parallel_size=2, pipeline_parallel_size=2,
                  distributed_executor_backend="distributed")

    def run(eng, watch):
        sp = SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True)
        ids = [eng.add_request(None, p, sp) for p in PROMPTS + [[11, 12, 13, 14]]]
        final = {}
        while eng.has_unfinished_requests():
            outs = eng.step()
            watch.append(len(eng.core.batch_queue))
            for o in outs:
                if o.finished:
                    final[o.request_id] = o.outputs[0].token_ids
        return [final[i] for i in ids]

    remote = _start_remote(port, 3, 1)
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(**common))
        ex = eng.core.executor
        assert ex.max_concurrent_batches == 2 and ex.output_rank == 2
        assert sorted(a.node_id is None for a in ex.assignments) == [False, False, False, True]
        depth = []
        got = run(eng, depth)
        assert got[:3] == ref                       # the first three prompts are the reference's
        assert max(depth) >= 1 and eng.core.max_concurrent == 2
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=10)

    # same shape, rank 3 (second stage) exits at its 5th step
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="1").items():
        monkeypatch.setenv(k, v)
    remote = _start_remote(port, 3, 1, VDAMD_FAULT_DIE_AT="3:5")
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(**common))
        with pytest.raises(Exception):
            run(eng, [])
        deadline = time.time() + 10
        while not eng.core.executor.is_failed and time.time() < deadline:
            time.sleep(0.05)
        assert eng.core.executor.is_failed
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=10)


@pytest.mark.slow
def test_stalled_rank_is_detected_and_the_serving_head_exits_nonzero(tmp_path):
    """A rank that stops issuing work (a hung device: its process and its connection stay up, so no
    socket closes) leaves its TP peer waiting inside the all-reduce.  The step's enforced timeout
    fails the executor, and a serving head (``launch.py serve``) exits non-zero so that the
    supervisor restarts the deployment (reference ``src/launch.py:142-144``, ``docker-compose.yml:8``)."""
    port, http = get_open_port(), get_open_port()
    env = _env(port, VLLM_HEAD_LOCAL_DEVICES="1", VLLM_EXECUTE_MODEL_TIMEOUT_SECONDS="4",
               VDAMD_FAULT_STALL_AT="1:3", VLLM_LOGGING_LEVEL="INFO")
    remote = _start_remote(port, 1, 1, VDAMD_FAULT_STALL_AT="1:3")
    log = open(tmp_path / "head.log", "w")
    head = subprocess.Popen([sys.executable, os.path.join(ROOT, "launch.py"), "serve", "llama-tiny", "--device", "cpu",
                             "--dtype", "float32", "--max-model-len", "128", "--block-size", "16",
                             "--num-gpu-blocks-override", "64", "--max-num-seqs", "4", "-tp", "2", "--host", "127.0.0.1",
                             "--port", str(http), "--disable-frontend-multiprocessing"],
                            env=env, stdout=log, stderr=subprocess.STDOUT)
    try:
        import urllib.request
        deadline = time.time() + 120
        up = False
        while time.time() < deadline and head.poll() is None and not up:
            try:
                up = urllib.request.urlopen(f"http://127.0.0.1:{http}/health", timeout=2).status == 200
            except Exception:  # noqa: BLE001
                time.sleep(0.5)
        assert up, open(tmp_path / "head.log").read()[-3000:]
        req = urllib.request.Request(f"http://127.0.0.1:{http}/v1/completions", method="POST",
                                     headers={"Content-Type": "application/json"},
                                     data=b'{"model": "llama-tiny", "prompt": [5, 9, 200, 17], "max_tokens": 20}')
        try:
            urllib.request.urlopen(req, timeout=60).read()
        except Exception:  # noqa: BLE001 - the request cannot complete: rank 1 never runs its third step
            pass
        rc = head.wait(timeout=60)
        text = open(tmp_path / "head.log").read()
        assert rc != 0, text[-3000:]
        assert "execute_model failed on rank" in text, text[-3000:]
    finally:
        for p in (head, remote):
            if p.poll() is None:
                p.kill()
            p.wait(timeout=10)
        log.close()


@pytest.mark.slow
def test_awq_expert_model_at_tp2_x_pp2_matches_single_process(monkeypatch):
    """The model of the reference's example deployment in miniature (``.env.server:11``: an AWQ
    Qwen3-MoE checkpoint served with ``-tp 2 -pp 2``): packed 4-bit experts cut over the TP group,
    expert layers spread over two stages, head + a three-device worker node."""
    # groups of 32 input channels, so that an expert's half of the intermediate size still holds whole groups
    over = {"quantization_config": {"quant_method": "awq", "bits": 4, "group_size": 32}}
    ref = _reference("qwen3-moe-tiny", quantization="awq", hf_overrides=over)
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="1").items():
        monkeypatch.setenv(k, v)
    remote = _start_remote(port, 3, 1)
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="qwen3-moe-tiny", quantization="awq", hf_overrides=over, device="cpu", dtype="float32", max_model_len=128,
            block_size=16, num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=2,
            pipeline_parallel_size=2, distributed_executor_backend="distributed"))
        assert _generate(eng) == ref
    finally:
        if eng is not None:
            eng.shutdown()
        remote.terminate()
        remote.wait(timeout=10)


@pytest.mark.slow
def test_kv_connector_reports_are_aggregated_over_tp_ranks(monkeypatch, tmp_path):
    """TP=2 through the distributed executor with the shared-storage KV connector: both ranks
    write / read their halves of the KV heads, the executor counts a transfer as finished when
    both have reported, and a second engine generates from the stored prefill."""
    ref = _reference("llama-tiny")
    kv = {"kv_connector": "SharedStorageConnector", "kv_role": "kv_both",
          "kv_connector_extra_config": {"shared_storage_path": str(tmp_path)}}
    outs = []
    for _ in range(2):
        port = get_open_port()
        for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="2").items():
            monkeypatch.setenv(k, v)
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=2,
            distributed_executor_backend="distributed", kv_transfer_config=kv, enable_prefix_caching=False))
        try:
            assert eng.core.executor.has_connector
            outs.append((_generate(eng), eng.core.scheduler.kv_connector))
            assert not outs[-1][1].sending and not outs[-1][1].held_until_sent
        finally:
            eng.shutdown()
    (first, c1), (second, c2) = outs
    assert first == ref and second == ref
    assert c1.num_saved_requests == 1 and c1.num_loaded_tokens == 0      # only the 50-token prompt spans whole blocks
    assert c2.num_saved_requests == 0 and c2.num_loaded_tokens == 48
    assert sorted(os.listdir(os.path.join(str(tmp_path), os.listdir(str(tmp_path))[0]))) == [
        "meta.json", "rank0.safetensors", "rank1.safetensors"]


def test_speculative_decoding_through_the_distributed_executor(monkeypatch):
    """Draft verification on workers behind the RPC (their replies are read by a reply thread): the
    accepted-token accounting of scheduler and workers stays in step; TP=2 over gloo."""
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="2").items():
        monkeypatch.setenv(k, v)
    base = [3, 9, 4, 7, 11, 5]
    prompts = [base * 6 + [3, 9], list(range(20, 60)), [5, 6] * 20]
    sp = SamplingParams(temperature=0.0, max_tokens=32, ignore_eos=True)
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=256, block_size=16,
              num_gpu_blocks_override=256, max_num_seqs=8)

    def run(eng):
        ids = [eng.add_request(None, p, sp) for p in prompts]
        done = {}
        while eng.has_unfinished_requests():
            for o in eng.step():
                if o.finished:
                    done[o.request_id] = o.outputs[0].token_ids
        return [done[i] for i in ids]
    ref = LLMEngine.from_engine_args(EngineArgs(**kw))
    want = run(ref)
    ref.shutdown()
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            tensor_parallel_size=2, distributed_executor_backend="distributed",
            speculative_config={"method": "ngram", "num_speculative_tokens": 3, "prompt_lookup_min": 1}, **kw))
        got = run(eng)
        assert got == want
        assert eng.core.scheduler.spec_stats["accepted"] > 0
    finally:
        if eng is not None:
            eng.shutdown()


def test_bootstrap_blobs_travel_over_the_control_plane(monkeypatch):
    """Workers behind the RPC exchange their bring-up values (what carries the RCCL unique id and
    the IPC handles on a GPU box) through the head's rendezvous, not through gloo."""
    import asyncio
    from vllm_distributed_amd.parallel.bootstrap import HeadRendezvous

    async def unit():
        rv = HeadRendezvous()
        got = await asyncio.gather(rv.exchange("k#0", 1, 3, b"one"), rv.exchange("k#0", 0, 3, None),
                                   rv.exchange("k#0", 2, 3, [b"two", 2]))
        assert got[0] == got[1] == got[2] == [None, b"one", [b"two", 2]]
        assert not rv._slots
        with pytest.raises(ValueError):
            await asyncio.gather(rv.exchange("j#0", 0, 2, 1), rv.exchange("j#0", 0, 2, 1))
    asyncio.run(unit())

    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="2").items():
        monkeypatch.setenv(k, v)
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(
            model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=4, tensor_parallel_size=2,
            distributed_executor_backend="distributed"))

        def probe(self):
            from vllm_distributed_amd.parallel import bootstrap
            from vllm_distributed_amd.parallel.state import get_tp_group
            tp = get_tp_group()
            vals = tp.bootstrap_all_gather(("blob", tp.rank_in_group, b"\\x00\\x01"), "test")
            return bootstrap.available(), vals
        res = eng.core.executor.collective_rpc(probe)
        for avail, vals in res:
            assert avail
            assert [list(v) for v in vals] == [["blob", 0, b"\\x00\\x01"], ["blob", 1, b"\\x00\\x01"]]
    finally:
        if eng is not None:
            eng.shutdown()


def test_head_spawns_eight_workers_tp8_matches_single_process(monkeypatch):
    """The headline topology in miniature: one head, eight workers it spawns itself, TP=8 (gloo here,
    RCCL on a GPU node), bring-up blobs relayed by the head's rendezvous.  Greedy text equals the
    single-process engine's."""
    port = get_open_port()
    for k, v in _env(port, VLLM_HEAD_LOCAL_DEVICES="8").items():
        monkeypatch.setenv(k, v)
    kw = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
              num_gpu_blocks_override=128, max_num_seqs=8,
              hf_overrides={"num_attention_heads": 8, "num_key_value_heads": 8})
    sp = SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True)
    prompts = [[3 + i, 9, 27, 81, 5] for i in range(6)]

    def run(eng):
        ids = [eng.add_request(None, p, sp) for p in prompts]
        done = {}
        while eng.has_unfinished_requests():
            for o in eng.step():
                if o.finished:
                    done[o.request_id] = o.outputs[0].token_ids
        return [done[i] for i in ids]
    ref = LLMEngine.from_engine_args(EngineArgs(**kw))
    want = run(ref)
    ref.shutdown()
    eng = None
    try:
        eng = LLMEngine.from_engine_args(EngineArgs(tensor_parallel_size=8,
                                                    distributed_executor_backend="distributed", **kw))
        assert eng.core.executor.world_size == 8
        assert run(eng) == want
    finally:
        if eng is not None:
            eng.shutdown()


@pytest.mark.slow
def test_control_plane_round_trip_and_head_cost_at_world_8():
    """``benchmarks/rpc_fanout_bench.py``: eight stub workers behind the real executor, 2048 sequences.
    The bounds are loose (a shared CI box): what they pin is the order of magnitude - a step's fan-out
    and reply in a few milliseconds, the head's own scheduling code in microseconds per sequence."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "rpc_fanout_bench.py"), "--world", "8",
                        "--batches", "2048", "--steps", "10", "--sync-scheduling"], capture_output=True, text=True,
                       timeout=280, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    row = j["results"][0]
    assert j["world"] == 8 and row["batch"] == 2048 and row["tokens_per_step_call"] == 2048
    assert j["noop_one_reply_ms_p50"] < 15.0, j
    assert row["execute_rtt_ms_p50"] < 40.0, row
    assert row["schedule_ms"] + row["update_ms"] < 2048 * 0.02, row          # < 20 us per sequence
