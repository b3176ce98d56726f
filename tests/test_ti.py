"""Triangle-inequality cluster pruning: VAQ::clusterTI +
VAQ::searchTriangleInequality.

CPU part: properties of the oracle's restatement (oracle/vaq_oracle.c,
vo_cluster_ti / vo_search_ti; PARITY UNPINNED, see its header).
GPU part: libvaqhip.so's TI form against that restatement, through the C ABI.
"""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import assert_topk_matches, make_case


def ti_case(seed, D, bits, N, nq, T, seg, dup_frac=0.0, integer=False):
    c = make_case(seed, D, bits, N, nq, dup_frac=dup_frac, rotate=False, integer=integer)
    rng = np.random.default_rng(seed + 99)
    L = c["L"]
    pick = rng.integers(0, max(N, 1), size=T)
    if N > 0:
        cl = np.concatenate([c["cents"][s][c["codes"][pick, s].astype(np.int64)] for s in range(seg)], axis=1)
    else:
        cl = rng.normal(size=(T, seg * L))
    c["clusters"] = np.ascontiguousarray(cl, dtype=np.float32)
    c["seg"] = seg
    c["T"] = T
    return c


# ------------------------------------------------------------------ oracle --
def test_cluster_ti_grouping_invariants():
    c = ti_case(5, 32, [8] * 8, 6000, 4, 37, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], c["seg"])
    member, start, xcc = ti["member"], ti["start"], ti["code2cc"]
    assert start[0] == 0 and start[-1] == 6000 and np.all(np.diff(start) >= 0)
    assert np.array_equal(np.sort(member), np.arange(6000))
    assert np.array_equal(ti["grouped"], c["codes"][member])
    # nearest centre, first minimum, distance = sqrt of the sequential sum
    X = np.concatenate([c["cents"][s][c["codes"][:, s].astype(np.int64)] for s in range(4)], axis=1)
    for t in range(37):
        rows = member[start[t]:start[t + 1]]
        d = xcc[rows]
        assert np.all(d[:-1] >= d[1:])                       # farthest first
        eq = d[:-1] == d[1:]
        assert np.all(rows[:-1][eq] < rows[1:][eq])          # ties: ascending row
    some = np.random.default_rng(0).integers(0, 6000, 200)
    for r in some:
        dist = np.sqrt(np.array([po.ref_l2sqr_ny(X[r], c["clusters"][t:t + 1])[0] if po.have_ref()
                                 else np.float32(((X[r] - c["clusters"][t]) ** 2).sum())
                                 for t in range(37)], dtype=np.float32))
        t_best = int(np.argmin(dist))
        if po.have_ref():
            assert xcc[r] == dist[t_best]
            assert start[t_best] <= np.nonzero(member == r)[0][0] < start[t_best + 1]


@pytest.mark.parametrize("bits,D", [([8] * 8, 32), ([12, 10, 9, 8, 8, 7, 6, 4], 64)])
def test_ti_visit_all_equals_exhaustive(bits, D):
    """Visiting every cluster, TI|EA prunes losslessly: same rows as HEAP, sqrt'ed distances."""
    c = ti_case(7, D, bits, 20000, 24, 40, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], c["seg"])
    for k in (1, 10, 100):
        l1, d1, pruned = po.search_ti(c["X"], c["cents"], ti, k, visit=1.0, projected=True)
        l0, d0 = po.search(c["X"], c["cents"], c["codes"], k, projected=True)
        assert np.array_equal(d1, np.sqrt(d0))
        assert np.array_equal(np.sort(l1, 1), np.sort(l0, 1))
        assert pruned > 0


def test_ti_partial_visit_is_topk_of_visited_rows():
    c = ti_case(11, 32, [8] * 8, 30000, 16, 64, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], c["seg"])
    k, visit = 20, 0.25
    l, d, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=visit, projected=True)
    for q in range(16):
        qcc, order = po.ti_query_order(c["X"][q, :16], c["clusters"])
        rows = np.concatenate([ti["member"][ti["start"][t]:ti["start"][t + 1]] for t in order[:16]])
        lut = po.create_lut(c["X"][q], c["cents"], 8)
        dd = np.sqrt(po.all_dists(lut, c["codes"][rows]))
        best = np.sort(dd)[:k]
        assert np.array_equal(d[q], best)


def test_ti_without_ea_returns_first_k_rows_of_the_visiting_order():
    """VAQ.cpp:1617-1686 never updates bsfKSquared: restated as written."""
    c = ti_case(13, 32, [8] * 8, 5000, 8, 20, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], c["seg"])
    k = 50
    l, d, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=0.5, ea=False, projected=True)
    for q in range(8):
        qcc, order = po.ti_query_order(c["X"][q, :16], c["clusters"])
        rows = np.concatenate([ti["member"][ti["start"][t]:ti["start"][t + 1]] for t in order])[:k]
        assert np.array_equal(np.sort(l[q]), np.sort(rows))


# --------------------------------------------------------------------- GPU --
def visited_dists(c, ti, visit, k, max_bits=None):
    """sqrt'ed distance of every row a TI search may return (inf elsewhere): the
    rows of the first max(int(T*visit), shortest prefix with k rows) clusters."""
    T, seg, L = c["T"], c["seg"], c["L"]
    mb = max_bits if max_bits is not None else max(c["bits"])
    out = np.full((c["X"].shape[0], c["codes"].shape[0]), np.inf, dtype=np.float32)
    sizes = np.diff(ti["start"])
    for q in range(c["X"].shape[0]):
        qcc, order = po.ti_query_order(c["X"][q, :seg * L], c["clusters"])
        nv = int(np.float32(T) * np.float32(visit)) if visit < 1 else T
        cum = np.cumsum(sizes[order])
        enough = int(np.searchsorted(cum, k) + 1) if cum[-1] >= k else T
        nv = min(T, max(nv, enough))
        rows = np.concatenate([ti["member"][ti["start"][t]:ti["start"][t + 1]] for t in order[:nv]])
        lut = po.create_lut(c["X"][q], c["cents"], mb)
        out[q, rows] = np.sqrt(po.all_dists(lut, c["codes"][rows]))
    return out


def _gpu_index(c, order="clusters_first", methods=None, visit=1.0):
    from vaq_amd.index import NNMethod, VaqHip
    v = VaqHip()
    v.mBitsAlloc = c["bits"]
    v.mCentroidsPerSubs = c["cents"]
    v.mEigenVectors = c.get("eig")
    v.mMethods = methods if methods is not None else (NNMethod.TI | NNMethod.EA)
    v.mVisit = visit
    v.mTISegmentNum = c["seg"]
    v.mTIClusterNum = c["T"]
    if order == "clusters_first":
        v.mTIClusters = c["clusters"]
        v.mCodebook = c["codes"]
    else:  # the reference's order: encode, then clusterTI regroups what is already there
        v.mMethods = NNMethod.Heap
        v.mCodebook = c["codes"]
        v.search(c["X"][:1], 1, projected=True)
        v.mMethods = methods if methods is not None else (NNMethod.TI | NNMethod.EA)
        v.mTIClusters = c["clusters"]
    return v


TI_CONFIGS = [
    # seed, D, bits, N, nq, T, seg
    (201, 32, [8] * 8, 20000, 33, 50, 4),
    (202, 128, [8] * 16, 30000, 17, 100, 8),
    (203, 64, [12, 10, 9, 8, 8, 7, 6, 4], 25000, 20, 64, 4),
    (204, 32, [8] * 8, 3000, 9, 200, 8),      # tiny clusters, some empty
    (205, 64, [4, 4, 4, 4], 10000, 12, 16, 2),
    (206, 256, [8] * 32, 8000, 6, 30, 16),
]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", TI_CONFIGS, ids=lambda c: f"s{c[0]}")
@pytest.mark.parametrize("visit", [1.0, 0.25, 0.05])
def test_gpu_ti_ea_matches_oracle(cfg, visit):
    seed, D, bits, N, nq, T, seg = cfg
    c = ti_case(seed, D, bits, N, nq, T, seg)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], seg)
    v = _gpu_index(c, visit=visit)
    for k in (1, 10, 100):
        ans = v.search(c["X"], k, projected=True)
        ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=visit, projected=True)
        alld = visited_dists(c, ti, visit, k) if max(bits) <= 4 else None  # 4-bit codes: exact ties abound
        assert_topk_matches(ans.labels.reshape(nq, k), ans.distances.reshape(nq, k), ol, od, all_dists=alld,
                            what=f"TI|EA s{seed} visit={visit} k={k}")
    v.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", TI_CONFIGS[:4], ids=lambda c: f"s{c[0]}")
def test_gpu_ti_without_ea_matches_oracle(cfg):
    """Exposes the grouping order itself: the result is the first k rows visited."""
    from vaq_amd.index import NNMethod
    seed, D, bits, N, nq, T, seg = cfg
    c = ti_case(seed, D, bits, N, nq, T, seg)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], seg)
    v = _gpu_index(c, methods=NNMethod.TI, visit=0.3)
    for k in (1, 7, 100):
        ans = v.search(c["X"], k, projected=True)
        ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=0.3, ea=False, projected=True)
        assert_topk_matches(ans.labels.reshape(nq, k), ans.distances.reshape(nq, k), ol, od,
                            what=f"TI s{seed} k={k}")
    v.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ea", [True, False])
def test_gpu_ti_long_visiting_list_is_staged_in_chunks(ea):
    """visit asks for 10 clusters (the kernel stages 64 list entries at a time) but k = 1000
    rows need ~70 of these 15-row clusters: the until-k-rows rule (VAQ.cpp:1555, :1611)."""
    from vaq_amd.index import NNMethod
    seed, D, bits, N, nq, T, seg = TI_CONFIGS[3]
    c = ti_case(seed, D, bits, N, nq, T, seg)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], seg)
    v = _gpu_index(c, methods=NNMethod.TI | (NNMethod.EA if ea else 0), visit=0.05)
    k = 1000
    ans = v.search(c["X"], k, projected=True)
    ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=0.05, ea=ea, projected=True)
    assert_topk_matches(ans.labels.reshape(nq, k), ans.distances.reshape(nq, k), ol, od, what=f"chunks ea={ea}")
    v.close()


@pytest.mark.gpu
def test_gpu_ti_regroup_after_codes_and_back():
    """clusterTI after encode (the reference's call order), then back to HEAP."""
    from vaq_amd.index import NNMethod
    c = ti_case(210, 64, [9, 8, 8, 7, 8, 8, 8, 8], 40000, 16, 80, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], 4)
    v = _gpu_index(c, order="codes_first", visit=0.2)
    k = 25
    ans = v.search(c["X"], k, projected=True)
    ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=0.2, projected=True)
    assert_topk_matches(ans.labels.reshape(16, k), ans.distances.reshape(16, k), ol, od, what="regrouped")
    info = v.info()
    assert info["ti_clusters"] == 80 and info["ti_segments"] == 4
    v.mMethods = NNMethod.Heap
    ans = v.search(c["X"], k, projected=True)
    ol, od = po.search(c["X"], c["cents"], c["codes"], k, projected=True)
    assert_topk_matches(ans.labels.reshape(16, k), ans.distances.reshape(16, k), ol, od, what="back to HEAP")
    assert v.info()["ti_clusters"] == 0
    v.close()


@pytest.mark.gpu
def test_gpu_ti_few_queries_many_rows_splits_and_merges():
    """Few queries over many rows: each query's units are spread over several
    workgroups and merged (sqrt applied by the merge)."""
    c = ti_case(220, 32, [8] * 8, 1_500_000, 3, 64, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], 4, nthreads=8)
    v = _gpu_index(c, visit=0.5)
    v.set_option("timing", 1)
    k = 100
    ans = v.search(c["X"], k, projected=True)
    assert v.last_timing()["slices"] > 1
    ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=0.5, nthreads=3, projected=True)
    assert_topk_matches(ans.labels.reshape(3, k), ans.distances.reshape(3, k), ol, od, what="split")
    v.close()


@pytest.mark.gpu
def test_gpu_ti_ties_and_short_index():
    # integer data: massive exact ties; k > rows visited; N < k
    c = ti_case(230, 32, [8] * 8, 4000, 10, 25, 4, dup_frac=0.3, integer=True)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], 4)
    v = _gpu_index(c, visit=1.0)
    k = 64
    ans = v.search(c["X"], k, projected=True)
    ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=1.0, projected=True)
    # all distances for the boundary-tie clause (sqrt'ed, as returned)
    alld = np.stack([np.sqrt(po.all_dists(po.create_lut(c["X"][q], c["cents"], 8), c["codes"]))
                     for q in range(10)])
    assert_topk_matches(ans.labels.reshape(10, k), ans.distances.reshape(10, k), ol, od, all_dists=alld,
                        what="ties")
    v.close()
    c2 = ti_case(231, 32, [8] * 8, 40, 5, 8, 4)
    ti2 = po.cluster_ti(c2["codes"], c2["cents"], c2["clusters"], 4)
    v2 = _gpu_index(c2, visit=0.1)
    ans = v2.search(c2["X"], 100, projected=True)
    ol, od, _ = po.search_ti(c2["X"], c2["cents"], ti2, 100, visit=0.1, projected=True)
    assert_topk_matches(ans.labels.reshape(5, 100), ans.distances.reshape(5, 100), ol, od, what="N<k")
    v2.close()


@pytest.mark.gpu
def test_gpu_ti_state_errors():
    from vaq_amd import _lib
    from vaq_amd.index import NNMethod
    c = ti_case(240, 32, [8] * 8, 2000, 2, 10, 4)
    v = _gpu_index(c, methods=NNMethod.TI | NNMethod.EA)
    v.mTIClusters = None
    with pytest.raises(_lib.VaqHipError):
        v.search(c["X"], 5, projected=True)
    v.close()


@pytest.mark.gpu
def test_gpu_ti_cpp_demo_driver(tmp_path):
    """examples/demo_vaqhip.cpp with --method ...,EA_TI<T>m<seg> --visit-cluster: the
    C++ adapter's TI members (mTIClusters, mVisit, clusterTI) over the C ABI."""
    import subprocess
    from vaq_amd import build, io
    exe = build.build_demo()
    c = ti_case(250, 128, [8] * 8, 30000, 20, 60, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], 4)
    io.save_centroids(c["cents"], str(tmp_path / "c.bin"))
    io.save_codebook(c["codes"], str(tmp_path / "cb.bin"))
    io.write_vecs(str(tmp_path / "q.fvecs"), c["X"])
    c["clusters"].tofile(str(tmp_path / "ti.f32"))
    k = 50
    ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=0.2, projected=True)
    r = subprocess.run([exe, "--centroids", str(tmp_path / "c.bin"), "--codebook", str(tmp_path / "cb.bin"),
                        "--queries", str(tmp_path / "q.fvecs"), "--timeseries-size", "128", "--k", str(k),
                        "--method", "VAQ64m8min8max8var1,EA_TI60m4", "--visit-cluster", "0.2",
                        "--ti-clusters", str(tmp_path / "ti.f32"), "--result", str(tmp_path / "out.csv")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.loadtxt(str(tmp_path / "out.csv"), delimiter=",", dtype=np.int64)
    alld = visited_dists(c, ti, 0.2, k)
    d_got = np.take_along_axis(alld, got, axis=1).astype(np.float32)
    assert_topk_matches(got.astype(np.int32), d_got, ol, od, all_dists=alld, what="cpp demo TI")


@pytest.mark.gpu
def test_gpu_ti_full_size_properties():
    """BASELINE size (1M rows x 8 B, k=100), size-independent properties instead of the oracle:
    visiting every cluster, TI|EA returns the exhaustive HEAP answer with sqrt'ed distances;
    visiting fewer, every rank is no better than the exhaustive one and the rows returned
    really carry the distances reported."""
    from vaq_amd.index import NNMethod
    c = ti_case(260, 128, [8] * 8, 1_000_000, 64, 256, 4)
    nq, k = 64, 100
    v = _gpu_index(c, methods=NNMethod.Heap)
    v.mTIClusters = None
    heap = v.search(c["X"], k, projected=True)
    hl, hd = heap.labels.reshape(nq, k), heap.distances.reshape(nq, k)
    v.mTIClusters = c["clusters"]
    v.mMethods = NNMethod.TI | NNMethod.EA
    v.mVisit = 1.0
    a = v.search(c["X"], k, projected=True)
    al, ad = a.labels.reshape(nq, k), a.distances.reshape(nq, k)
    assert np.array_equal(ad, np.sqrt(hd))
    for q in range(nq):  # same rows wherever the sqrt'ed distances are distinct
        u, cnt = np.unique(ad[q], return_counts=True)
        single = np.isin(ad[q], u[cnt == 1])
        assert np.array_equal(al[q][single], hl[q][single])
    v.mVisit = 0.1
    b = v.search(c["X"], k, projected=True)
    bl, bd = b.labels.reshape(nq, k), b.distances.reshape(nq, k)
    assert np.all(bd >= ad) and np.all(np.diff(bd, axis=1) >= 0) and np.all(bl >= 0)
    for q in range(0, nq, 8):  # reported distance = the row's own ADC distance
        lut = po.create_lut(c["X"][q], c["cents"], 8)
        assert np.array_equal(np.sqrt(po.all_dists(lut, c["codes"][bl[q]])), bd[q])
        assert len(set(bl[q].tolist())) == k
    v.close()


@pytest.mark.gpu
def test_gpu_ti_append_and_method_errors():
    """Rows appended to a TI-grouped index are regrouped with the rest; set_method rejects
    the methods that are not on this path."""
    import ctypes as C
    from vaq_amd import _lib
    c = ti_case(270, 64, [8] * 8, 12000, 6, 40, 4)
    ti = po.cluster_ti(c["codes"], c["cents"], c["clusters"], 4)
    v = _gpu_index(dict(c, codes=c["codes"][:5000]), visit=0.3)
    k = 20
    v.search(c["X"], k, projected=True)
    v.add_codes(c["codes"][5000:])
    ans = v.search(c["X"], k, projected=True)
    ol, od, _ = po.search_ti(c["X"], c["cents"], ti, k, visit=0.3, projected=True)
    assert_topk_matches(ans.labels.reshape(6, k), ans.distances.reshape(6, k), ol, od, what="TI append")
    L = _lib.load()
    assert L.vaqhip_index_set_method(v._h, 0x08, C.c_float(1.0)) == -2      # FAST
    assert L.vaqhip_index_set_method(v._h, 0x01, C.c_float(1.0)) == -2      # SORT
    assert L.vaqhip_index_set_method(v._h, 0x06, C.c_float(0.0)) == -1      # visit must be > 0
    assert L.vaqhip_index_set_method(v._h, 0x80, C.c_float(1.0)) == 0       # HEAP on a TI-grouped index ...
    lab = np.empty((6, k), np.int32)
    dis = np.empty((6, k), np.float32)
    rc = L.vaqhip_search_projected(v._h, c["X"].ctypes.data_as(C.c_void_p), 6, k,
                                   lab.ctypes.data_as(C.c_void_p), dis.ctypes.data_as(C.c_void_p))
    assert rc == -7 and b"TI" in L.vaqhip_last_error()                      # ... is a state error at search
    v.close()


@pytest.mark.gpu
def test_cpp_from_reference_after_cluster_ti(tmp_path):
    """include/vaqhip.hpp VaqHip::fromReference on a stand-in for the reference object AFTER its
    clusterTI() (regrouped mCodebook + mTIClustersMember): labels must be original rows."""
    import os
    import subprocess
    from vaq_amd import build
    lib = build.build_lib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "from_reference_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "from_reference_test.cpp"), "-o", exe,
                           "-L" + os.path.dirname(lib), "-lvaqhip", "-Wl,-rpath," + os.path.dirname(lib),
                           "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "from_reference ok" in r.stdout, r.stdout + r.stderr
