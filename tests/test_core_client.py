"""Engine core in a child process: same tokens as the in-process core, stop
strings finish the request in the core, and a dead core is reported."""
import pytest

from vllm_distributed_amd import SamplingParams
from vllm_distributed_amd.engine.arg_utils import EngineArgs
from vllm_distributed_amd.engine.core import EngineDeadError
from vllm_distributed_amd.engine.llm_engine import LLMEngine
from vllm_distributed_amd.parallel import state as pstate

ARGS = dict(model="llama-tiny", device="cpu", dtype="float32", max_model_len=128, block_size=16,
            num_gpu_blocks_override=64, max_num_seqs=8)
PROMPTS = [[5, 9, 200, 17], list(range(10, 50)), "hello there"]


def _run(engine, params):
    ids = [engine.add_request(None, p, params) for p in PROMPTS]
    final = {}
    while engine.has_unfinished_requests():
        for o in engine.step():
            if o.finished:
                final[o.request_id] = o
    return [final[i] for i in ids]


@pytest.mark.slow
def test_multiprocess_core_matches_in_process_core():
    pstate.destroy_model_parallel()
    cfg = EngineArgs(**ARGS).create_engine_config()
    local = LLMEngine(cfg)
    greedy = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True, logprobs=2)
    ref = _run(local, greedy)
    base_text = ref[2].outputs[0].text
    local.shutdown()
    pstate.destroy_model_parallel()

    remote = LLMEngine(EngineArgs(**ARGS, async_scheduling=True).create_engine_config(), multiprocess=True)
    try:
        got = _run(remote, greedy)
        for a, b in zip(got, ref):
            assert a.outputs[0].token_ids == b.outputs[0].token_ids
            assert a.outputs[0].text == b.outputs[0].text
            assert a.outputs[0].logprobs is not None and len(a.outputs[0].logprobs) == 8
            assert a.metrics.first_token_time is not None and a.metrics.num_generation_tokens == 8
        # a stop string is detected by the front-end's detokenizer and must free the request in the core
        stop = next((base_text[i:i + 2] for i in range(2, len(base_text) - 2) if base_text[i:i + 2].strip()), None)
        if stop:
            rid = remote.add_request(None, "hello there", SamplingParams(temperature=0.0, max_tokens=8,
                                                                         ignore_eos=True, stop=[stop]))
            outs = []
            while remote.has_unfinished_requests():
                outs += [o for o in remote.step() if o.request_id == rid]
            assert outs[-1].finished and outs[-1].outputs[0].finish_reason == "stop"
            assert stop not in outs[-1].outputs[0].text
        assert remote.core.reset_prefix_cache() in (True, False)
        assert remote.core.stats().num_running == 0
        # losing the core process is an error, not a hang
        remote.add_request(None, [1, 2, 3], SamplingParams(max_tokens=64, ignore_eos=True))
        remote.core._proc.terminate()
        with pytest.raises(EngineDeadError):
            for _ in range(100):
                remote.step()
    finally:
        remote.shutdown()
