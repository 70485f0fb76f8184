"""The multi-device data plane: what the reference exists for (its workers sit one per GPU, possibly on
other machines: reference ``src/launch.py:149-296,322-384``).

The tests that need two or more physical GPUs are gated on ``torch.cuda.device_count()`` and skip on a
one-GPU box; on a multi-GPU box they exercise, in this order of depth,

  (a) ``RcclCommunicator`` over 2 and min(8, n) devices - all-reduce, all-gather, reduce-scatter,
      broadcast, grouped send/recv, eagerly and (b) as nodes of one hipGraph replayed 100 times;
  (c) ``CustomAllReduce`` with one rank per device (cross-device ``hipIpcOpenMemHandle``, uncached peer
      loads over xGMI): every form, graph replay, no bounded spin ran out, with and without the start fence;
  (d) ``PipelineLink`` edge communicators along 2 and min(4, n) stages with two batches in flight;
  (e) ``bench.py --gpus 2`` through the bare launcher (head spawns the workers) and as the driver starts
      it (``torch.distributed.run``), asserting RCCL + peer-mapped collectives + hipGraph decode are live;
  (f) full-vocabulary first-token distributions of TP = 2 / n against TP = 1.

Each failure names the layer that broke, so the first run on new hardware localises a fault in minutes.
The bodies live in ``tests/multirank_bodies.py``; the same bodies run here on the CPU over gloo and, on
any GPU box, with all ranks sharing device 0, so the test code itself is known to be right before it
meets a second GPU.  Children are always freshly spawned processes.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from vllm_distributed_amd.utils import get_open_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BODIES = os.path.join(ROOT, "tests", "multirank_bodies.py")
NDEV = torch.cuda.device_count()            # does not create a HIP context
WIDE = min(8, NDEV)

needs2 = pytest.mark.skipif(NDEV < 2, reason=f"needs >= 2 GPUs (this box has {NDEV})")
needs_wide = pytest.mark.skipif(NDEV < 4, reason=f"needs >= 4 GPUs (this box has {NDEV})")


def run_body(body, world, mode, extra_env=None, timeout=170):
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT=str(port), MODE=mode,
                   VLLM_LOGGING_LEVEL="WARNING", GLOO_SOCKET_IFNAME="lo", NCCL_SOCKET_IFNAME="lo",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, BODIES, body], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    results = []
    try:
        for p in procs:
            so, _ = p.communicate(timeout=timeout)
            assert p.returncode == 0, so[-3000:]
            line = [l for l in so.splitlines() if l.startswith("RESULT ")][-1]
            results.append(json.loads(line[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(r["ok"] for r in results), results
    return results


# ---------------------------------------------------------------- the bodies themselves, no second GPU needed
def test_bodies_rccl_interface_over_gloo_on_cpu():
    got = run_body("rccl", 2, "cpu")
    assert not got[0]["real_rccl"]


@pytest.mark.parametrize("world", [2, 3])
def test_bodies_pipeline_link_over_gloo_on_cpu(world):
    got = run_body("pipeline", world, "cpu")
    assert got[-1]["batches_checked"] == 6 and not got[0]["accelerated"]


@pytest.mark.gpu
@pytest.mark.parametrize("body,world", [("car", 2), ("car", 4), ("car_timeout", 2), ("pipeline", 2), ("rccl", 2)])
def test_bodies_on_ranks_sharing_one_gpu(body, world):
    run_body(body, world, "shared")


# ------------------------------------------------------------------------------ one rank per physical GPU
@pytest.mark.gpu
@needs2
@pytest.mark.parametrize("world", sorted({2, WIDE} - {0, 1}))
def test_rccl_communicator_across_devices_eager_and_in_a_graph(world):
    got = run_body("rccl", world, "devices", timeout=280)
    assert all(r["real_rccl"] and r["replays"] == 100 for r in got)


@pytest.mark.gpu
@needs2
@pytest.mark.parametrize("world,fence_start", [(2, False), (2, True)] + ([(WIDE, False), (WIDE, True)] if WIDE > 2 else []))
def test_peer_mapped_collectives_one_rank_per_device(world, fence_start):
    got = run_body("car", world, "devices", {"FENCE_START": "1" if fence_start else "0"}, timeout=280)
    assert all(r["timed_out"] is False and r["fence_start"] == fence_start for r in got)


@pytest.mark.gpu
@needs2
def test_peer_mapped_collectives_bounded_wait_across_devices():
    got = run_body("car_timeout", 2, "devices")
    assert 0.15 <= got[0]["waited"] < 5.0


@pytest.mark.gpu
@needs2
@pytest.mark.parametrize("world", sorted({2, min(4, NDEV)} - {0, 1}))
def test_pipeline_edge_communicators_across_devices(world):
    got = run_body("pipeline", world, "devices", timeout=280)
    assert all(r["accelerated"] for r in got) and got[-1]["batches_checked"] == 6


def _bench(cmd, timeout=560):
    env = dict(os.environ, VLLM_LOGGING_LEVEL="WARNING", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


BENCH_ARGS = ["--gpus", "2", "--model", "llama-3-8b", "--steps", "4", "--warmup", "2", "--batch-per-gpu", "256",
              "--strong-batch", "64", "--ttft-requests", "2"]


@pytest.mark.gpu
@needs2
@pytest.mark.parametrize("launcher", ["bare", "torchrun"])
def test_bench_two_gpus_runs_the_fast_data_plane(launcher):
    """The product path on two devices: head + workers over the RPC control plane, RCCL and the
    peer-mapped collectives both live, decode replayed from hipGraphs that hold the collectives."""
    if launcher == "bare":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(get_open_port()), os.path.join(ROOT, "bench.py"), *BENCH_ARGS]
    j = _bench(cmd)
    cfg = j["config"]
    assert j["n_gpus"] == 2 and cfg["parallelism"] == "tp2" and cfg["executor"] == "distributed"
    assert cfg["tp_collectives"]["backend"] == "nccl", cfg
    assert cfg["tp_collectives"]["direct_rccl"], cfg
    assert cfg["tp_collectives"]["direct_all_reduce"], cfg
    assert cfg["hip_graph"], cfg["hip_graph_fallback"]
    assert j["value"] > 0 and j["strong_scaling"]["tokens_per_s"] > 0


@pytest.mark.gpu
@needs2
@pytest.mark.parametrize("world", sorted({2, WIDE} - {0, 1}))
def test_first_token_distribution_tensor_parallel_across_devices(world):
    """Full-vocabulary log-probabilities of the first generated token at TP = world (one rank per device:
    RCCL process groups, peer-mapped fused all-reduce + norm inside the decode graphs, vocab-parallel
    argmax) against the unsharded model."""
    from test_comm_gpu import _assert_distributions_close, _run_spmd_shared
    ov = {"num_attention_heads": 8, "num_key_value_heads": 8, "head_dim": 64, "intermediate_size": 1024}
    ref = _run_spmd_shared(1, False, overrides=ov, logprobs=512)[0]
    got = _run_spmd_shared(world, False, overrides=ov, logprobs=512, share=False)
    assert all(g["tokens"] == got[0]["tokens"] for g in got)
    assert all(g["direct_ar"] and g["rccl"] for g in got), got
    assert got[0]["graphs"], got[0]["fallback"]
    _assert_distributions_close(ref["first_lp"], got[0]["first_lp"])
    # and with the peer-mapped path off: the same numbers over RCCL alone
    got = _run_spmd_shared(world, False, overrides=ov, logprobs=512, share=False,
                           extra_env={"VDAMD_CUSTOM_ALL_REDUCE": "0"})
    assert all(g["rccl"] and not g["direct_ar"] for g in got), got
    _assert_distributions_close(ref["first_lp"], got[0]["first_lp"])


@pytest.mark.gpu
@needs2
def test_preflight_reports_both_paths():
    """``scripts/preflight_multigpu.py`` on two devices: latency and bandwidth of RCCL and of the
    peer-mapped path per message size, and thresholds derived from them."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "preflight_multigpu.py"), "--gpus", "2",
                        "--max-bytes", str(16 << 20), "--iters", "10"], capture_output=True, text=True, timeout=560,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["world"] == 2 and j["mode"] == "devices"
    assert {"rccl_all_reduce", "direct_one_shot", "direct_two_shot"} <= set(j["all_reduce"][0])
    assert j["recommend"]["VDAMD_CUSTOM_ALL_REDUCE_ONE_SHOT_BYTES"] >= 0
