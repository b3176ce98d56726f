"""Build the gfx950 shared library (vaq_amd/lib/libvaqhip.so) with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part of
the numerics contract (vaq_kernels.hip header): only explicit fmaf() fuses.
"""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vaq_amd", "csrc")
LIBDIR = os.path.join(ROOT, "vaq_amd", "lib")
LIB = os.path.join(LIBDIR, "libvaqhip.so")
SOURCES = ["vaq_kernels.hip", "vaqhip_api.cpp"]
HEADERS = [os.path.join(CSRC, "vaq_kernels.h"), os.path.join(ROOT, "include", "vaqhip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-Wall",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    cmd += os.environ.get("VAQ_EXTRA_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def build_demo(force: bool = False) -> str:
    """examples/demo_vaqhip.cpp: plain g++ against the C ABI (no hipcc needed)."""
    lib = build_lib()
    out_dir = os.path.join(ROOT, "examples", "bin")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "demo_vaqhip")
    src = os.path.join(ROOT, "examples", "demo_vaqhip.cpp")
    hdrs = [os.path.join(ROOT, "include", h) for h in ("vaqhip.h", "vaqhip.hpp", "vaqhip_io.hpp")]
    if not force and os.path.exists(exe) and all(
            os.path.getmtime(f) <= os.path.getmtime(exe) for f in [src, lib] + hdrs):
        return exe
    cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
           src, "-o", exe, "-L" + LIBDIR, "-lvaqhip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
