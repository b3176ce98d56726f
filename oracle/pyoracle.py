"""ctypes bindings for the CPU checker (oracle/liboracle.so) and, when built,
the real-reference harness (oracle/_ref/libvaqref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from vaq_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libvaqref.so")

METHOD_EA = 0x02
METHOD_HEAP = 0x80

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_u16p = C.POINTER(C.c_uint16)


def build(ref: bool = True) -> None:
    """Compile the checker (and oracle/_ref when /root/reference is present).
    VAQ_NO_BUILD: only check that it exists (no child process under a profiler)."""
    if os.environ.get("VAQ_NO_BUILD"):
        if not os.path.exists(os.path.join(_HERE, "liboracle.so")):
            raise FileNotFoundError("oracle/liboracle.so is missing and VAQ_NO_BUILD is set: run `make -C oracle` first")
        return
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference/bitvecengine"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _ip(a):
    return a.ctypes.data_as(_i32p)


def _up(a):
    return a.ctypes.data_as(_u16p)


class _VoIndex(C.Structure):
    _fields_ = [
        ("D", C.c_int), ("M", C.c_int), ("L", C.c_int), ("max_bits", C.c_int),
        ("ncent", _i32p), ("cent", C.POINTER(_f32p)), ("eig", _f32p),
        ("codes", _u16p), ("N", C.c_int64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build(ref=False)
        L = C.CDLL(_LIB)
        L.vo_heap_reorder.restype = C.c_size_t
        L.vo_avg_recall.restype = C.c_double
        L.vo_recall_at_r.restype = C.c_double
        L.vo_search.restype = C.c_int
        L.vo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def have_ref() -> bool:
    return os.path.exists(_REF)


_ref = None


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        R.ref_heap_reorder.restype = C.c_size_t
        R.ref_avg_recall.restype = C.c_double
        R.ref_recall_at_r.restype = C.c_double
        R.ref_load_codebook.restype = C.c_int
        R.ref_load_centroids.restype = C.c_int
        R.ref_encode_dist.restype = C.c_float
        _ref = R
    return _ref


def _cent_array(cent):
    arr = (_f32p * len(cent))()
    keep = []
    for i, c in enumerate(cent):
        c = np.ascontiguousarray(c, dtype=np.float32)
        keep.append(c)
        arr[i] = _fp(c)
    return arr, keep


def max_threads() -> int:
    return int(lib().vo_max_threads())


# ------------------------------------------------------------------ heap ---
def topk_from_dists(dist, k, ids=None):
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    out_ids = np.empty(k, dtype=np.int32)
    out_val = np.empty(k, dtype=np.float32)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        idp = _ip(ids)
    lib().vo_topk_from_dists(_fp(dist), idp, C.c_int64(dist.size), C.c_int(k),
                             _ip(out_ids), _fp(out_val))
    return out_ids, out_val


def ref_topk_from_dists(dist, k, ids=None):
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    out_ids = np.empty(k, dtype=np.int32)
    out_val = np.empty(k, dtype=np.float32)
    if ids is None:
        ref().ref_topk_from_dists(_fp(dist), C.c_int64(dist.size), C.c_int(k),
                                  _ip(out_ids), _fp(out_val))
    else:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        ref().ref_topk_from_dists_ids(_fp(dist), _ip(ids), C.c_int64(dist.size),
                                      C.c_int(k), _ip(out_ids), _fp(out_val))
    return out_ids, out_val


# ------------------------------------------------------------------- LUT ---
def project(X, E):
    X = np.ascontiguousarray(X, dtype=np.float32)
    E = np.ascontiguousarray(E, dtype=np.float32)
    out = np.empty_like(X)
    lib().vo_project(_fp(X), C.c_int64(X.shape[0]), C.c_int(X.shape[1]), _fp(E), _fp(out))
    return out


def create_lut(qproj, cent, max_bits):
    """qproj: (D,) projected query; cent: list of (K_s, L) arrays.
    Returns the reference LUTType as an (M, ksub) array: lut[s, c]."""
    M = len(cent)
    L = cent[0].shape[1]
    ksub = 1 << max_bits
    ncent = np.array([c.shape[0] for c in cent], dtype=np.int32)
    arr, keep = _cent_array(cent)
    q = np.ascontiguousarray(qproj, dtype=np.float32)
    lut = np.empty((M, ksub), dtype=np.float32)
    lib().vo_create_lut(_fp(q), C.c_int(M), C.c_int(L), _ip(ncent), arr,
                        C.c_int(ksub), _fp(lut))
    return lut


def ref_lut_column_fma(qsub, cent_s):
    """One LUT column through the reference's fma(); cent_s: (K, L), K % 8 == 0."""
    K, L = cent_s.shape
    cm = np.asfortranarray(cent_s.astype(np.float32))  # column-major K x L
    flat = np.ascontiguousarray(cm.ravel(order="F"))
    q = np.ascontiguousarray(qsub, dtype=np.float32)
    out = np.empty(K, dtype=np.float32)
    ref().ref_lut_column_fma(_fp(q), _fp(flat), C.c_int(K), C.c_int(L), _fp(out))
    return out


def ref_l2sqr_ny(x, y):
    """y: (ny, d) row-major."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty(y.shape[0], dtype=np.float32)
    ref().ref_l2sqr_ny(_fp(out), _fp(x), _fp(y), C.c_size_t(y.shape[1]),
                       C.c_size_t(y.shape[0]))
    return out


# ------------------------------------------------------------------ scan ---
def all_dists(lut, codes):
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    M, ksub = lut.shape
    out = np.empty(codes.shape[0], dtype=np.float32)
    lib().vo_all_dists(_fp(lut), C.c_int(ksub), _up(codes), C.c_int64(codes.shape[0]),
                       C.c_int(M), _fp(out))
    return out


def search_heap(lut, codes, k, ea=False):
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    M, ksub = lut.shape
    ids = np.empty(k, dtype=np.int32)
    dis = np.empty(k, dtype=np.float32)
    fn = lib().vo_search_ea if ea else lib().vo_search_heap
    fn(_fp(lut), C.c_int(ksub), _up(codes), C.c_int64(codes.shape[0]), C.c_int(M),
       C.c_int(k), _ip(ids), _fp(dis))
    return ids, dis


def search(X, cent, codes, k, eig=None, max_bits=None, method=METHOD_HEAP,
           nthreads=1, projected=False):
    """VAQ::search restatement.  Returns (labels (nq,k) int32, distances (nq,k) f32)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    nq, D = X.shape
    M = len(cent)
    L = cent[0].shape[1]
    assert M * L == D
    ncent = np.array([c.shape[0] for c in cent], dtype=np.int32)
    if max_bits is None:
        max_bits = int(np.log2(ncent.max()))
    arr, keep = _cent_array(cent)
    ix = _VoIndex()
    ix.D, ix.M, ix.L, ix.max_bits = D, M, L, max_bits
    ix.ncent = _ip(ncent)
    ix.cent = arr
    if eig is not None:
        eig = np.ascontiguousarray(eig, dtype=np.float32)
        ix.eig = _fp(eig)
    else:
        ix.eig = None
    ix.codes = _up(codes)
    ix.N = codes.shape[0]
    labels = np.empty((nq, k), dtype=np.int32)
    dists = np.empty((nq, k), dtype=np.float32)
    rc = lib().vo_search(C.byref(ix), _fp(X), C.c_int(nq), C.c_int(k), C.c_uint(method),
                         C.c_int(nthreads), C.c_int(1 if projected else 0),
                         _ip(labels), _fp(dists))
    if rc != 0:
        raise ValueError(f"vo_search failed rc={rc}")
    return labels, dists


def encode(Xproj, cent, nthreads=1):
    Xproj = np.ascontiguousarray(Xproj, dtype=np.float32)
    n, D = Xproj.shape
    M = len(cent)
    L = cent[0].shape[1]
    ncent = np.array([c.shape[0] for c in cent], dtype=np.int32)
    arr, keep = _cent_array(cent)
    codes = np.empty((n, M), dtype=np.uint16)
    lib().vo_encode(_fp(Xproj), C.c_int64(n), C.c_int(M), C.c_int(L), _ip(ncent), arr,
                    C.c_int(nthreads), _up(codes))
    return codes


def ref_encode(Xproj, cent):
    """VAQ::encodeImpl's statements (VAQ.cpp:736-745) compiled from the reference's matrix types and
    vendored Eigen (oracle/ref_harness.cpp:ref_encode_column): what Eigen's reduction really sums."""
    Xproj = np.ascontiguousarray(Xproj, dtype=np.float32)
    n, D = Xproj.shape
    M = len(cent)
    L = cent[0].shape[1]
    codes = np.empty((n, M), dtype=np.uint16)
    for s in range(M):
        c = np.ascontiguousarray(cent[s], dtype=np.float32)
        col = np.empty(n, dtype=np.uint16)
        ref().ref_encode_column(_fp(Xproj), C.c_int64(n), C.c_int(D), C.c_int(s), C.c_int(L), _fp(c),
                                C.c_int(c.shape[0]), _up(col))
        codes[:, s] = col
    return codes


def ref_project(X, E):
    """VAQ::ProjectOnEigenVectors (VAQ.hpp:198-201) through Eigen's own (complex) GEMM."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    E = np.ascontiguousarray(E, dtype=np.float32)
    out = np.empty_like(X)
    ref().ref_project(_fp(X), C.c_int64(X.shape[0]), C.c_int(X.shape[1]), _fp(E), _fp(out))
    return out


def refine(Xq, Xtrain, labels_in, k):
    Xq = np.ascontiguousarray(Xq, dtype=np.float32)
    Xtrain = np.ascontiguousarray(Xtrain, dtype=np.float32)
    labels_in = np.ascontiguousarray(labels_in, dtype=np.int32)
    nq, D = Xq.shape
    R = labels_in.shape[1]
    labels = np.empty((nq, k), dtype=np.int32)
    dists = np.empty((nq, k), dtype=np.float32)
    lib().vo_refine(_fp(Xq), C.c_int(nq), C.c_int(D), _fp(Xtrain), _ip(labels_in),
                    C.c_int(R), C.c_int(k), _ip(labels), _fp(dists))
    return labels, dists


def avg_recall(labels, topnn, K=None):
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    topnn = np.ascontiguousarray(topnn, dtype=np.int32)
    nq = labels.shape[0]
    K = K or labels.shape[1]
    return float(lib().vo_avg_recall(_ip(labels), C.c_int(nq), C.c_int(K), _ip(topnn),
                                     C.c_int(topnn.shape[1])))


def recall_at_r(labels, topnn, K=None):
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    topnn = np.ascontiguousarray(topnn, dtype=np.int32)
    nq = labels.shape[0]
    K = K or labels.shape[1]
    return float(lib().vo_recall_at_r(_ip(labels), C.c_int(nq), C.c_int(K), _ip(topnn),
                                      C.c_int(topnn.shape[1])))


def query_lut_1d(qproj, bits, cent_colmajor, codes, k):
    q = np.ascontiguousarray(qproj, dtype=np.float32)
    bits = np.ascontiguousarray(bits, dtype=np.int32)
    cm = np.asfortranarray(cent_colmajor.astype(np.float32))
    flat = np.ascontiguousarray(cm.ravel(order="F"))
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    ids = np.empty(k, dtype=np.int32)
    dis = np.empty(k, dtype=np.float32)
    lib().vo_query_lut_1d(_fp(q), C.c_int(bits.size), _ip(bits), _fp(flat),
                          C.c_int(cm.shape[0]), _up(codes), C.c_int64(codes.shape[0]),
                          C.c_int(codes.shape[1]), C.c_int(k), _ip(ids), _fp(dis))
    return ids, dis


class _VoTi(C.Structure):
    _fields_ = [
        ("clusters", _f32p), ("T", C.c_int), ("seg", C.c_int), ("visit", C.c_float),
        ("member", _i32p), ("start", _i32p), ("code2cc", _f32p), ("grouped", _u16p),
    ]


METHOD_TI = 0x04


def cluster_ti(codes, cent, clusters, seg, nthreads=1):
    """VAQ::clusterTI after mTIClusters exists.  Returns dict(member, start,
    code2cc, grouped)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    clusters = np.ascontiguousarray(clusters, dtype=np.float32)
    N, M = codes.shape
    L = cent[0].shape[1]
    T = clusters.shape[0]
    assert clusters.shape[1] == seg * L
    arr, keep = _cent_array(cent)
    member = np.empty(N, dtype=np.int32)
    start = np.empty(T + 1, dtype=np.int32)
    code2cc = np.empty(N, dtype=np.float32)
    grouped = np.empty((N, M), dtype=np.uint16)
    lib().vo_cluster_ti(_up(codes), C.c_int64(N), C.c_int(M), C.c_int(L), arr, _fp(clusters),
                        C.c_int(T), C.c_int(seg), C.c_int(nthreads), _ip(member), _ip(start),
                        _fp(code2cc), _up(grouped))
    return dict(member=member, start=start, code2cc=code2cc, grouped=grouped,
                clusters=clusters, seg=seg)


def ti_query_order(qproj, clusters):
    clusters = np.ascontiguousarray(clusters, dtype=np.float32)
    T, d = clusters.shape
    qproj = np.ascontiguousarray(qproj, dtype=np.float32)
    qcc = np.empty(T, dtype=np.float32)
    order = np.empty(T, dtype=np.int32)
    lib().vo_ti_query_order(_fp(qproj), _fp(clusters), C.c_int(T), C.c_int(d), _fp(qcc), _ip(order))
    return qcc, order


def search_ti(X, cent, ti, k, visit=1.0, eig=None, max_bits=None, ea=True, nthreads=1,
              projected=False):
    """VAQ::search with NNMethod::TI (| EA).  `ti` = cluster_ti(...) output.
    Returns (labels, distances (sqrt'ed, as the reference stores them), pruned)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    nq, D = X.shape
    M = len(cent)
    L = cent[0].shape[1]
    ncent = np.array([c.shape[0] for c in cent], dtype=np.int32)
    if max_bits is None:
        max_bits = int(np.log2(ncent.max()))
    arr, keep = _cent_array(cent)
    ix = _VoIndex()
    ix.D, ix.M, ix.L, ix.max_bits = D, M, L, max_bits
    ix.ncent = _ip(ncent)
    ix.cent = arr
    if eig is not None:
        eig = np.ascontiguousarray(eig, dtype=np.float32)
        ix.eig = _fp(eig)
    else:
        ix.eig = None
    ix.codes = _up(ti["grouped"])
    ix.N = ti["grouped"].shape[0]
    t = _VoTi()
    t.clusters = _fp(ti["clusters"])
    t.T = ti["clusters"].shape[0]
    t.seg = ti["seg"]
    t.visit = visit
    t.member = _ip(ti["member"])
    t.start = _ip(ti["start"])
    t.code2cc = _fp(ti["code2cc"])
    t.grouped = _up(ti["grouped"])
    labels = np.empty((nq, k), dtype=np.int32)
    dists = np.empty((nq, k), dtype=np.float32)
    pruned = C.c_long(0)
    method = METHOD_TI | (METHOD_EA if ea else 0)
    rc = lib().vo_search_ti_all(C.byref(ix), C.byref(t), _fp(X), C.c_int(nq), C.c_int(k),
                                C.c_uint(method), C.c_int(nthreads),
                                C.c_int(1 if projected else 0), _ip(labels), _fp(dists),
                                C.byref(pruned))
    if rc != 0:
        raise ValueError(f"vo_search_ti_all failed rc={rc}")
    return labels, dists, pruned.value
