"""HTTP serving benchmark (same as ``python launch.py bench serve``); see
``vllm_distributed_amd/entrypoints/bench_serve.py``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd.entrypoints.bench_serve import main  # noqa: E402

if __name__ == "__main__":
    main()
