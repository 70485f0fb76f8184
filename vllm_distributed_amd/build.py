"""Build the native libraries in-tree.

``python -m vllm_distributed_amd.build`` compiles

* ``csrc/*.hip``       -> ``vllm_distributed_amd/_C/libvdamd_ops.so``  (hipcc, gfx950)
* ``csrc/core/*.cpp``  -> ``vllm_distributed_amd/_C/libvdamd_core.so`` (g++, host runtime)

Plain C ABI, loaded with ctypes: no torch headers, so a full rebuild takes
seconds and the libraries do not depend on the torch ABI.  hipcc
cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "csrc")
OUT = os.path.join(ROOT, "vllm_distributed_amd", "_C")
BUILD = os.path.join(ROOT, "build")
ARCH = os.environ.get("VDAMD_GPU_ARCH", "gfx950")

HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffast-math",
             "-fno-gpu-rdc", "-Wno-unused-result", "-DNDEBUG"]
CXX_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-DNDEBUG"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(paths: List[str], extra: str) -> str:
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    return h.hexdigest()


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} {cmd[-1]}")


def _build_lib(name: str, sources: List[str], headers: List[str], compiler: str, flags: List[str],
               link_flags: List[str], force: bool) -> str:
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(BUILD, exist_ok=True)
    lib = os.path.join(OUT, name)
    stamp = os.path.join(OUT, name + ".sha256")
    digest = _digest(sources + headers, " ".join([compiler] + flags + link_flags))
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return lib
    objs = [os.path.join(BUILD, name + "." + os.path.basename(s) + ".o") for s in sources]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        list(ex.map(lambda so: _run([compiler] + flags + ["-c", so[0], "-o", so[1]]), zip(sources, objs)))
    _run([compiler, "-shared", "-fPIC"] + ([f"--offload-arch={ARCH}"] if compiler.endswith("hipcc") else [])
         + objs + link_flags + ["-o", lib])
    with open(stamp, "w") as f:
        f.write(digest)
    return lib


def build_sanitized(verbose: bool = True) -> str:
    """Debug preset: the kernel library with AddressSanitizer and UBSan on the
    *host* side of every translation unit (argument checking, launch code, the
    direct all-reduce's handle exchange).  Device code is built as usual - GPU
    sanitizers are not available for gfx950 here.  Load the result with
    ``VDAMD_OPS_LIB=<path> LD_PRELOAD=$(clang -print-file-name=libclang_rt.asan-x86_64.so)``."""
    hip_src = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wno-unused-result",
             "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
             "-Xarch_host", "-fno-omit-frame-pointer"]
    global OUT
    saved, OUT = OUT, os.path.join(OUT, "debug")
    try:
        lib = _build_lib("libvdamd_ops_asan.so", hip_src, headers, _hipcc(), flags,
                         ["-fsanitize=address", "-fsanitize=undefined", "-shared-libsan"], force=False)
    finally:
        OUT = saved
    if verbose:
        print("built", os.path.relpath(lib, ROOT))
    return lib


def build(force: bool = False, verbose: bool = True) -> List[str]:
    built = []
    hip_src = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc")))
    if hip_src:
        built.append(_build_lib("libvdamd_ops.so", hip_src, headers, _hipcc(), HIP_FLAGS, [], force))
    core_dir = os.path.join(CSRC, "core")
    if os.path.isdir(core_dir):
        cpp = sorted(os.path.join(core_dir, f) for f in os.listdir(core_dir) if f.endswith(".cpp"))
        hdr = sorted(os.path.join(core_dir, f) for f in os.listdir(core_dir) if f.endswith(".h"))
        if cpp:
            built.append(_build_lib("libvdamd_core.so", cpp, hdr, shutil.which("g++") or "g++",
                                    CXX_FLAGS, ["-lpthread"], force))
    if verbose:
        for b in built:
            print("built", os.path.relpath(b, ROOT))
    return built


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        build_sanitized()
    else:
        build(force="--force" in sys.argv)
