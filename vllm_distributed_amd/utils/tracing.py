"""roctx ranges for rocprofv3 (``rocprofv3 --marker-trace --kernel-trace``).

``VDAMD_TRACE=1`` brackets the phases of a worker step (prepare inputs,
forward, sampling) and every collective with named ranges; without it
``trace_range`` is a no-op with no measurable cost.  Worker-side counterpart
of the ``profile`` RPC every worker exposes (reference ``src/launch.py:351-374``
forwards any method name, which is how vLLM's profiler start/stop reaches the
workers).
"""
from __future__ import annotations

import contextlib
import ctypes
from typing import Iterator, Optional

from .. import envs

_lib: Optional[ctypes.CDLL] = None
_tried = False


def _load() -> Optional[ctypes.CDLL]:
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                 "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            lib.roctxMarkA.argtypes = [ctypes.c_char_p]
            _lib = lib
            break
        except (OSError, AttributeError):
            continue
    return _lib


_ENABLED: Optional[bool] = None


def enabled() -> bool:
    global _ENABLED
    if _ENABLED is None:
        _ENABLED = bool(envs.VDAMD_TRACE) and _load() is not None
    return _ENABLED


@contextlib.contextmanager
def _range(name: str) -> Iterator[None]:
    _lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        _lib.roctxRangePop()


_NULL = contextlib.nullcontext()


def trace_range(name: str):
    """Context manager: a named roctx range when tracing is on, else nothing."""
    return _range(name) if enabled() else _NULL


def mark(name: str) -> None:
    if enabled():
        _lib.roctxMarkA(name.encode())
