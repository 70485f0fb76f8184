"""Every bundled tiny architecture x {dense, fp8, awq, nvfp4} weights x {16-bit, e4m3} KV cache: build the
engine, generate a few tokens, print one line per combination (ok / refused with the reason / FAIL).
Refusals are deliberate ``ValueError`` / ``NotImplementedError`` / ``AssertionError`` messages; anything
else counts as a failure and sets the exit code.

    python scripts/family_matrix.py [--device cuda|cpu] [--only NAME_SUBSTRING]
"""
import argparse
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vllm_distributed_amd import LLM, SamplingParams  # noqa: E402
from vllm_distributed_amd.parallel import state as pstate  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vllm_distributed_amd", "models", "configs")
    names = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(root, "*-tiny.json")))
    names = [n for n in names if args.only in n]
    gpu = args.device == "cuda"
    failures = 0
    for kv in ("auto", "fp8"):
        for quant in (None, "fp8", "awq", "nvfp4"):
            if kv == "fp8" and quant not in (None, "awq"):
                continue
            for name in names:
                tag = f"{name:22s} weights={quant or 'dense':6s} kv={kv:4s}"
                llm = None
                try:
                    llm = LLM(model=name, device=args.device, dtype="bfloat16" if gpu else "float32", max_model_len=256,
                              block_size=64 if gpu else 16, num_gpu_blocks_override=32, max_num_seqs=8,
                              quantization=quant, kv_cache_dtype=kv)
                    outs = llm.generate([[5, 9, 200, 17], list(range(10, 90))],
                                        SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
                    print(tag, "ok", [o.outputs[0].token_ids for o in outs][0], flush=True)
                except (ValueError, NotImplementedError, AssertionError) as e:
                    print(tag, "refused:", str(e).splitlines()[0][:110], flush=True)
                except Exception as e:  # noqa: BLE001
                    failures += 1
                    print(tag, "FAIL", type(e).__name__, str(e).splitlines()[0][:160], flush=True)
                finally:
                    if llm is not None:
                        llm.shutdown()
                    pstate.destroy_model_parallel()
    print(f"{failures} failure(s)")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
