#!/usr/bin/env python3
"""``python launch.py serve <model> ...`` / ``python launch.py remote <head_ip>``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from vllm_distributed_amd.launch import main  # noqa: E402

if __name__ == "__main__":
    main()
