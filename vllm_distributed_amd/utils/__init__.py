"""Network / device / dispatch helpers.

Covers the utility surface the reference imports from ``vllm.utils`` and
``vllm.platforms`` (reference ``src/launch.py:41-44``): ``get_ip``,
``get_open_port``, ``get_distributed_init_method``, a stateless device count,
``current_platform`` and ``run_method``.
"""
from __future__ import annotations

import argparse
import ipaddress
import os
import socket
import sys
from typing import Any, Callable, Optional, Sequence, Union

import cloudpickle

from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)


# --------------------------------------------------------------------------
# network
# --------------------------------------------------------------------------
def get_ip() -> str:
    """Address other nodes use to reach this one.

    ``VLLM_HOST_IP`` wins; otherwise the source address of a UDP "connection"
    to a public IP (no packet is sent), then IPv6, then loopback.
    """
    host_ip = envs.VLLM_HOST_IP
    if host_ip:
        return host_ip
    for family, probe in ((socket.AF_INET, ("8.8.8.8", 80)),
                          (socket.AF_INET6, ("2001:4860:4860::8888", 80))):
        try:
            with socket.socket(family, socket.SOCK_DGRAM) as s:
                s.connect(probe)
                return s.getsockname()[0]
        except OSError:
            continue
    return "127.0.0.1"


def is_valid_ipv6_address(address: str) -> bool:
    try:
        ipaddress.IPv6Address(address)
        return True
    except ValueError:
        return False


def get_open_port(host: str = "") -> int:
    """Ask the kernel for a currently-free TCP port."""
    for family in (socket.AF_INET, socket.AF_INET6):
        try:
            with socket.socket(family, socket.SOCK_STREAM) as s:
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                s.bind((host, 0))
                return s.getsockname()[1]
        except OSError:
            continue
    raise RuntimeError("no free TCP port")


def get_distributed_init_method(ip: str, port: int) -> str:
    return f"tcp://[{ip}]:{port}" if is_valid_ipv6_address(ip) else f"tcp://{ip}:{port}"


# --------------------------------------------------------------------------
# platform / devices
# --------------------------------------------------------------------------
def _visible_device_env() -> Optional[str]:
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(name)
        if v is not None:
            return v
    return None


def hip_device_count_stateless() -> int:
    """Number of visible AMD GPUs without initialising the HIP runtime.

    The head process must stay GPU-free (it spawns workers, and a process that
    has touched the GPU must not fork/exec), so this only enumerates KFD
    topology nodes and applies ``HIP_VISIBLE_DEVICES``.
    """
    vis = _visible_device_env()
    if vis is not None:
        vis = vis.strip()
        if vis in ("", "-1"):
            return 0
        n_vis = len([x for x in vis.split(",") if x.strip() != ""])
    else:
        n_vis = None
    total = 0
    topo = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(topo):
            try:
                with open(os.path.join(topo, node, "properties")) as f:
                    props = dict(line.split(None, 1) for line in f if " " in line)
                if int(props.get("simd_count", "0").strip()) > 0:
                    total += 1
            except (OSError, ValueError):
                continue
    except OSError:
        total = 0
    if total == 0:
        try:  # device_count() does not create a HIP context
            import torch
            total = int(torch.cuda.device_count())
        except Exception:
            total = 0
        return total
    return min(total, n_vis) if n_vis is not None else total


class _Platform:
    """Answers "is this a GPU box" without creating a HIP context."""

    def is_rocm(self) -> bool:
        return hip_device_count_stateless() > 0

    # the reference asks ``current_platform.is_cuda()`` (src/launch.py:194,610)
    is_cuda = is_rocm

    def device_count(self) -> int:
        return hip_device_count_stateless()

    @property
    def device_type(self) -> str:
        return "cuda" if self.is_rocm() else "cpu"


current_platform = _Platform()


# --------------------------------------------------------------------------
# dispatch by name / callable / pickled callable
# --------------------------------------------------------------------------
def run_method(obj: Any, method: Union[str, bytes, Callable], args: Sequence[Any],
               kwargs: Optional[dict]) -> Any:
    """Invoke ``method`` on ``obj``.

    ``method`` may be an attribute name, a callable taking ``obj`` first, or a
    cloudpickled callable (reference ``src/launch.py:529`` via vLLM's
    ``run_method``).
    """
    kwargs = kwargs or {}
    if isinstance(method, bytes):
        func = cloudpickle.loads(method)
        return func(obj, *args, **kwargs)
    if isinstance(method, str):
        try:
            func = getattr(obj, method)
        except AttributeError:
            raise NotImplementedError(
                f"method {method!r} is not implemented by {type(obj).__name__}") from None
        return func(*args, **kwargs)
    return method(obj, *args, **kwargs)


# --------------------------------------------------------------------------
# argparse that accepts --foo_bar for --foo-bar and "-tp 2" style aliases
# --------------------------------------------------------------------------
class FlexibleArgumentParser(argparse.ArgumentParser):
    """ArgumentParser that treats ``--a_b`` and ``--a-b`` alike and accepts
    ``--key=value`` pairs (the reference builds its CLI on vLLM's class of the
    same name, ``src/launch.py:42,475``)."""

    def parse_args(self, args=None, namespace=None):  # type: ignore[override]
        if args is None:
            args = sys.argv[1:]
        processed = []
        for a in args:
            if a.startswith("--"):
                if "=" in a:
                    k, v = a.split("=", 1)
                    processed.append("--" + k[2:].replace("_", "-") + "=" + v)
                else:
                    processed.append("--" + a[2:].replace("_", "-"))
            else:
                processed.append(a)
        return super().parse_args(processed, namespace)


def cdiv(a: int, b: int) -> int:
    return -(-a // b)


def round_up(a: int, b: int) -> int:
    return cdiv(a, b) * b


__all__ = [
    "get_ip", "get_open_port", "get_distributed_init_method",
    "hip_device_count_stateless", "current_platform", "run_method",
    "FlexibleArgumentParser", "cdiv", "round_up", "is_valid_ipv6_address",
]
