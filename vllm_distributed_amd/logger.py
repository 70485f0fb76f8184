"""Logging setup (the reference obtains loggers through ``init_logger``,
reference ``src/launch.py:40,54``)."""
from __future__ import annotations

import logging
import os
import sys

_FORMAT = "%(levelname)s %(asctime)s [%(name)s:%(lineno)d] %(message)s"
_configured = False


def _configure_root() -> None:
    global _configured
    if _configured:
        return
    root = logging.getLogger("vllm_distributed_amd")
    level = os.environ.get("VLLM_LOGGING_LEVEL", "INFO").upper()
    root.setLevel(getattr(logging, level, logging.INFO))
    if not root.handlers:
        h = logging.StreamHandler(sys.stderr)
        h.setFormatter(logging.Formatter(_FORMAT, datefmt="%m-%d %H:%M:%S"))
        root.addHandler(h)
    root.propagate = False
    _configured = True


def init_logger(name: str) -> logging.Logger:
    _configure_root()
    if not name.startswith("vllm_distributed_amd"):
        name = "vllm_distributed_amd." + name
    return logging.getLogger(name)
