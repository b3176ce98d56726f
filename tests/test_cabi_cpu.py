"""CPU tests of the drop-in boundary: the shared library builds for gfx950,
loads, exports every symbol include/vaqhip.h declares, and fails loudly (no
CPU fallback) when there is no GPU.  No compute is attempted here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "vaqhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vaqhip_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported(vaqlib):
    from vaq_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(vaqlib, s), f"{s} declared in include/vaqhip.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms


def test_version_and_error_string(vaqlib):
    assert vaqlib.vaqhip_version() >= 100
    assert isinstance(vaqlib.vaqhip_last_error(), bytes)


def _no_gpu(vaqlib):
    return vaqlib.vaqhip_device_count() <= 0


def test_no_cpu_fallback_without_gpu(vaqlib):
    if not _no_gpu(vaqlib):
        pytest.skip("a GPU is present")
    import vaq_amd
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = [8] * 4
    v.mCentroidsPerSubs = [np.zeros((256, 2), np.float32)] * 4
    v.mCodebook = np.zeros((10, 4), np.uint16)
    with pytest.raises(vaq_amd.VaqHipError) as e:
        v.search(np.zeros((1, 8), np.float32), 1)
    assert e.value.code == -3  # VAQHIP_ENODEVICE


def test_argument_validation_does_not_need_gpu(vaqlib):
    h = C.c_void_p()
    bits = (C.c_int * 6)(*[8] * 6)
    arr = (C.POINTER(C.c_float) * 6)()
    rc = vaqlib.vaqhip_index_create(C.byref(h), 12, 6, bits, arr, None, 0)
    assert rc == -1 and b"4 codes per step" in vaqlib.vaqhip_last_error()
    rc = vaqlib.vaqhip_index_create(C.byref(h), 13, 4, bits, arr, None, 0)
    assert rc == -1
    rc = vaqlib.vaqhip_index_create(None, 8, 4, bits, arr, None, 0)
    assert rc == -1
    assert vaqlib.vaqhip_search(None, None, 1, 1, None, None) == -1


def test_parse_method_string():
    import vaq_amd
    v = vaq_amd.VaqHip()
    v.parseMethodString("VAQ256m32min7max8var1,HEAP")
    assert (v.mBitBudget, v.mSubspaceNum, v.mMinBitsPerSubs, v.mMaxBitsPerSubs) == (256, 32, 7, 8)
    assert v.mMethods == vaq_amd.NNMethod.Heap
    v.parseMethodString("VAQ64m16min3max6var0.99,EA")
    assert v.mMethods == vaq_amd.NNMethod.EA and abs(v.mPercentVarExplained - 0.99) < 1e-6
    v.parseMethodString("VAQ128m32min6max9var0.95,EA_TI200")
    assert v.mMethods == (vaq_amd.NNMethod.EA | vaq_amd.NNMethod.TI)
    assert (v.mTIClusterNum, v.mTISegmentNum) == (200, -1)
    v.parseMethodString("VAQ256m32min7max10var1,EA_TI1000m16")
    assert (v.mTIClusterNum, v.mTISegmentNum) == (1000, 16)
    v.parseMethodString("VAQ256m32min7max10var1,TI500var0.9")
    assert v.mMethods == vaq_amd.NNMethod.TI and v.mTIClusterNum == 500 and abs(v.mTIVariance - 0.9) < 1e-6
    with pytest.raises(vaq_amd.VaqHipError):
        v.parseMethodString("VAQ128m32min6max9var0.95,SORT")
    with pytest.raises(vaq_amd.VaqHipError):
        v.parseMethodString("VAQ128m32min2max4var1,FAST")


def test_bench_plan_defaults():
    """What `bench.py --gpus N` runs by default: one GPU = c2 (the metric's configuration);
    several GPUs = the north-star path (c5, strong scaling, row shards + all-gather + merge);
    replicas only by name."""
    from vaq_amd import sharding
    p1 = sharding.bench_plan(1)
    assert (p1["workload"], p1["rows"], p1["nq"], p1["scaling"], p1["replicas"]) == ("c2", 1_000_000, 10_000, "strong", False)
    for w in (2, 4, 8):
        p = sharding.bench_plan(w)
        assert (p["workload"], p["rows"], p["nq"]) == ("c5", 1_000_000_000, 10_000)
        assert p["mode"] == "rows" and p["scaling"] == "strong" and not p["replicas"]
        assert p["bits"] == [8] * 16
        lo, hi = sharding.shard_bounds(p["rows"], w, w - 1)
        assert hi == p["rows"] and hi - lo == p["rows"] // w
    p = sharding.bench_plan(8, "c2")
    assert p["mode"] == "rows" and p["scaling"] == "strong"
    p = sharding.bench_plan(8, "c2", "replicas")
    assert p["replicas"] and p["scaling"] == "weak" and p["mode"] == "queries"
    assert sharding.bench_plan(8, "c2", "weak")["replicas"]
    p = sharding.bench_plan(4, "c4", "auto", "queries", rows=1000, nq=64)
    assert (p["mode"], p["rows"], p["nq"]) == ("queries", 1000, 64)
    assert not sharding.bench_plan(1, "c2", "replicas")["replicas"]  # one GPU has nothing to replicate


def test_member_identity_tokens():
    """index.py re-uploads a member when it is REPLACED; the token is a weak reference, so an
    array allocated at a recycled id() is never mistaken for the one that was uploaded."""
    from vaq_amd.index import _same, _wref
    a = np.zeros((4, 2), np.float32)
    r = _wref(a)
    assert _same(r, a) and not _same(r, a.copy()) and not _same(r, None)
    assert _same(None, None) and not _same(None, a)
    del a
    b = np.zeros((4, 2), np.float32)  # may reuse the freed id
    assert not _same(r, b)
    lst = [1, 2, 3]
    assert _same(_wref(lst), lst)


@pytest.mark.parametrize("san", ["thread", "address"])
def test_job_pool_under_sanitizers(tmp_path, san):
    """The host-side worker protocol of the multi-device index (vaq_amd/csrc/job_pool.h: persistent
    threads, phases, the exchange only after every shard succeeded) on the CPU build under
    ThreadSanitizer and AddressSanitizer: G = 2..8 workers, concurrent callers, failing shards."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = shutil.which("g++")
    assert cxx
    exe = str(tmp_path / f"job_pool_{san}")
    subprocess.check_call([cxx, "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-pthread",
                           "-I" + os.path.join(root, "vaq_amd", "csrc"), os.path.join(root, "tests", "cpp", "job_pool_test.cpp"),
                           "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "job_pool_test: ok" in out.stdout
