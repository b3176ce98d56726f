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
SOURCES = ["vaq_kernels.hip", "vaq_scan_bytes.hip", "vaq_scan_bits.hip", "vaq_scan_bf.hip", "vaq_scan_bm.hip", "vaq_exact.hip", "vaq_ti.hip", "vaqhip_api.cpp", "vaqhip_multi.cpp"]
KERNEL_HEADER = os.path.join(CSRC, "vaq_kernels.h")
API_HEADER = os.path.join(ROOT, "include", "vaqhip.h")


SCAN_HEADER = os.path.join(CSRC, "vaq_scan.h")
SCAN_BF_HEADER = os.path.join(CSRC, "vaq_scan_bf.h")


def _deps(src: str):
    # only the host file sees the public C header; the scan bodies live in vaq_scan.h
    deps = [os.path.join(CSRC, src), KERNEL_HEADER]
    if src.endswith(".cpp"):
        deps.append(API_HEADER)
    if src in ("vaq_kernels.hip", "vaq_scan_bytes.hip", "vaq_scan_bits.hip", "vaq_scan_bf.hip", "vaq_scan_bm.hip", "vaq_exact.hip"):
        deps.append(SCAN_HEADER)
    if src in ("vaq_kernels.hip", "vaq_scan_bf.hip", "vaq_scan_bm.hip"):
        deps.append(SCAN_BF_HEADER)
    return deps
OBJDIR = os.path.join(LIBDIR, "obj")
# Experiment builds: VAQ_VARIANT=name compiles (with VAQ_EXTRA_FLAGS) into vaq_amd/lib/variants/name/
# and VAQHIP_LIB=<path> makes the package load that library instead (tools/ sweeps).
_VARIANT = os.environ.get("VAQ_VARIANT", "")
# VAQ_VARIANT_ONLY="a.hip b.hip": only these units are compiled with the variant's flags, the
# others are linked from the main build's objects (which must be up to date)
_VARIANT_ONLY = os.environ.get("VAQ_VARIANT_ONLY", "").split()
_MAIN_OBJDIR = OBJDIR
if _VARIANT:
    LIBDIR = os.path.join(LIBDIR, "variants", _VARIANT)
    LIB = os.path.join(LIBDIR, "libvaqhip.so")
    OBJDIR = os.path.join(LIBDIR, "obj")


def _flags_tag() -> str:
    return os.environ.get("VAQ_EXTRA_FLAGS", "")


def _obj(src: str) -> str:
    d = OBJDIR
    if _VARIANT and _VARIANT_ONLY and src not in _VARIANT_ONLY:
        d = _MAIN_OBJDIR
    return os.path.join(d, os.path.splitext(src)[0] + ".o")


def _obj_stale(src: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    tag = o + ".flags"
    if not os.path.exists(tag) or open(tag).read() != _flags_tag():
        return True
    return any(os.path.getmtime(d) > t for d in _deps(src))


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile each translation unit to an object (in parallel, only the stale
    ones) and link them into libvaqhip.so."""
    if os.environ.get("VAQHIP_LIB") and not _VARIANT:
        return os.environ["VAQHIP_LIB"]  # a prebuilt experiment library was named: leave it alone
    if os.environ.get("VAQ_NO_BUILD"):
        # under rocprofv3 a child process (hipcc -> clang) is the exec-after-GPU-init hop the pool
        # forbids: the library must have been built beforehand (tools/profile_*.sh do that)
        if not os.path.exists(LIB):
            raise FileNotFoundError(LIB + " is missing and VAQ_NO_BUILD is set: run `python -m vaq_amd.build` first")
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
              "-ffp-contract=off", "-fno-fast-math", "-Wall",
              "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    extra = _flags_tag().split()
    todo = [s for s in SOURCES if (force or _obj_stale(s)) and
            not (_VARIANT and _VARIANT_ONLY and s not in _VARIANT_ONLY)]
    procs = []
    for s in todo:
        cmd = common + extra + ["-c", os.path.join(CSRC, s), "-o", _obj(s)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, cmd, subprocess.Popen(cmd)))
    failed = [s for s, _, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    for s in todo:
        with open(_obj(s) + ".flags", "w") as f:
            f.write(_flags_tag())
    objs = [_obj(s) for s in SOURCES]
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


def source_hash() -> str:
    """Identity of the library's sources (csrc + include): profile files under profiles/ carry it, and
    bench.py only attaches their counter figures to a run of the SAME build."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp")))
    files += sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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
