"""Build libsglk.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m sglang_security_amd.build [--force]

hipcc cross-compiles without a GPU; the built library travels with the tree.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libsglk.so")
TORCH_SRC = os.path.join(CSRC, "torch", "allreduce_ops.cc")
TORCH_LIB = os.path.join(PKG, "sgl_kernel", "common_ops.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _deps_mtime() -> float:
    hs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _deps_mtime())):
        return obj
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", src, "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x")
        cmd.insert(2, "hip")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
    return obj


def build(force: bool = False, jobs: int = 6) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    build_torch_ops(force)
    return LIB


def build_torch_ops(force: bool = False) -> str:
    """sgl_kernel/common_ops.so: the torch.ops.sgl_kernel.* registrations that have to come from C++ (the all-reduce
    op set, whose `str[]` arguments carry raw handle bytes).  Host code only: g++ against the installed torch's
    headers, linked to libsglk.so beside it."""
    import torch
    from torch.utils import cpp_extension as ce

    deps = [TORCH_SRC, os.path.join(ROOT, "include", "sglk.h"), LIB]
    if not force and os.path.exists(TORCH_LIB) and os.path.getmtime(TORCH_LIB) >= max(os.path.getmtime(d) for d in deps):
        return TORCH_LIB
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(rocm, "include")] + ["-I" + d for d in ce.include_paths()] + [
        TORCH_SRC, "-o", TORCH_LIB, "-L" + PKG, "-l:libsglk.so", "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch",
        "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + tlib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch op library failed to build:\n" + r.stderr[-4000:])
    return TORCH_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
