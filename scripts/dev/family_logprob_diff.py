import json, os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.config import ModelConfig
from vllm_distributed_amd.models.loader import save_dummy_checkpoint
from vllm_distributed_amd.parallel import state as pstate
PROMPTS = [[5, 9, 200, 17], list(range(10, 90)), [7, 7, 300, 2, 2, 41]]
if __name__ == "__main__":
    for model in sys.argv[1:]:
        tmp = tempfile.mkdtemp()
        mc = ModelConfig(model=model, dtype="float32", load_format="dummy")
        save_dummy_checkpoint(mc, tmp)
        open(os.path.join(tmp, "config.json"), "w").write(json.dumps(mc.hf_config))
        outs = {}
        for dev in ("cuda", "cpu"):
            kw = dict(dtype=os.environ.get("GPU_DTYPE", "bfloat16"), gpu_memory_utilization=0.3) if dev == "cuda" else dict(dtype="float32", device="cpu")
            llm = LLM(model=tmp, max_model_len=256, num_gpu_blocks_override=64, max_num_seqs=8, **kw)
            outs[dev] = llm.generate(PROMPTS, SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True, logprobs=20))
            llm.shutdown()
            pstate.destroy_model_parallel()
        for i, (a, b) in enumerate(zip(outs["cuda"], outs["cpu"])):
            for step in range(2):
                la, lb = a.outputs[0].logprobs[step], b.outputs[0].logprobs[step]
                common = set(la) & set(lb)
                print(model, "prompt", i, "step", step, "common", len(common),
                      "max diff %.4f" % max(abs(la[t] - lb[t]) for t in common), flush=True)
