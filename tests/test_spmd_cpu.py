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
    model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
    num_gpu_blocks_override=64, max_num_seqs=8, tensor_parallel_size=tp,
    distributed_executor_backend="spmd" if tp > 1 else "uni"))
sp = SamplingParams(temperature=0.8, top_p=0.9, seed=7, max_tokens=8, ignore_eos=True)
g = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
ids = [eng.add_request(None, p, s) for p, s in (([5, 9, 200, 17], g), (list(range(10, 50)), sp), ([7, 7], g))]
final = {{}}
while eng.has_unfinished_requests():
    for o in eng.step():
        if o.finished:
            final[o.request_id] = o.outputs[0].token_ids
print("RESULT " + json.dumps([final[i] for i in ids]), flush=True)
eng.shutdown()
"""


def _run(world):
    port = get_open_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VLLM_LOGGING_LEVEL="WARNING",
                   GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
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
