"""Python mirror of the reference's `class VAQ` search interface
(bitvecengine/VAQ.hpp:36-113) over the C ABI of include/vaqhip.h.

Member names follow the reference (mBitsAlloc, mCentroidsPerSubs, mCodebook,
mEigenVectors, mMethods ...) so harness code reads like the reference's
drivers (examples/demo_vaq.cpp:58-345).  All compute goes through
libvaqhip.so; nothing here has a NumPy fallback.
"""
from __future__ import annotations

import ctypes as C
import re
import weakref
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


class NNMethod:
    """VAQ::NNMethod bit flags (VAQ.hpp:38-49)."""
    Sort = 0x01
    EA = 0x02
    TI = 0x04
    Fast = 0x08
    Fast2 = 0x10
    Fast3 = 0x20
    Fast4 = 0x40
    Heap = 0x80


@dataclass
class LabelDistVec:
    """utils/Types.hpp:98-104: flat nq*k labels / distances."""
    labels: np.ndarray = field(default_factory=lambda: np.empty(0, np.int32))
    distances: np.ndarray = field(default_factory=lambda: np.empty(0, np.float32))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _wref(obj):
    """Identity token of a member that was uploaded: a weak reference (an id() can be recycled
    by a new array once the old one is freed; a dead weak reference never compares alive)."""
    if obj is None:
        return None
    try:
        return weakref.ref(obj)
    except TypeError:  # plain lists / tuples: fall back to the object itself (small)
        return lambda o=obj: o


def _same(ref, obj) -> bool:
    if ref is None or obj is None:
        return ref is None and obj is None
    return ref() is obj


class VaqHip:
    """Drop-in for the search half of `class VAQ`.

    Typical use (mirrors demo_vaq.cpp:58-345)::

        vaq = VaqHip()
        vaq.parseMethodString("VAQ64m8min8max8var1,HEAP")
        vaq.mBitsAlloc = [8] * 8
        vaq.mCentroidsPerSubs = [...]          # K_s x L float32 each
        vaq.mEigenVectors = E                  # D x D (real part), or None
        vaq.mCodebook = codes                  # N x M uint16 (CodebookType)
        ans = vaq.search(XTest, 100)           # LabelDistVec
    """

    def __init__(self, device: int = 0, sequential_sum: bool = False):
        """sequential_sum=True: BitVecEngine::queryLUT's arithmetic (one scalar
        quantiser per dimension, columns summed one by one; M need not be a
        multiple of 4) instead of VAQ::searchHeap's groups of four."""
        self.device = device
        self.sequential_sum = sequential_sum
        # VAQ.hpp:51-55
        self.mBitBudget = 0
        self.mSubspaceNum = 0
        self.mPercentVarExplained = 1.0
        self.mMinBitsPerSubs = 0
        self.mMaxBitsPerSubs = 0
        self.mMethods = NNMethod.Heap
        # VAQ.hpp:57-75
        self.mEigenVectors: Optional[np.ndarray] = None
        self.mCentroidsPerSubs: List[np.ndarray] = []
        self.mBitsAlloc: List[int] = []
        self.mCodebook: Optional[np.ndarray] = None
        # VAQ.hpp:77-84: triangle-inequality clusters
        self.mTIClusterNum = 0
        self.mTISegmentNum = -1
        self.mTIVariance = 1.0
        self.mVisit = 1.0
        self.mTIClusters: Optional[np.ndarray] = None  # T x (mTISegmentNum * mSubsLen)
        self.id_base = 0
        self._h = C.c_void_p()
        self._sig = None
        self._codes_sig = None
        self._ti_sig = None
        self._method_sig = None

    # ------------------------------------------------------------ parsing --
    def parseMethodString(self, methodString: str) -> None:
        """VAQ::parseMethodString (VAQ.cpp:1189-1267).  HEAP, EA and TI<T>[m<seg>]
        select the scans on this path; the other tokens the reference accepts
        (SORT, FAST*) are outside it and raise."""
        for token in methodString.split(","):
            if token.startswith("VAQ"):
                m = re.match(r"VAQ(\d+)m(\d+)min(\d+)max(\d+)var([0-9.]+)", token)
                if m:
                    self.mBitBudget = int(m.group(1))
                    self.mSubspaceNum = int(m.group(2))
                    self.mMinBitsPerSubs = int(m.group(3))
                    self.mMaxBitsPerSubs = int(m.group(4))
                    self.mPercentVarExplained = float(m.group(5))
            elif any(t in token for t in ("SORT", "HEAP", "EA", "TI", "FAST")):
                methods = 0
                for t in token.split("_"):
                    if "SORT" in t:
                        methods |= NNMethod.Sort
                    elif "HEAP" in t:
                        methods |= NNMethod.Heap
                    elif "EA" in t:
                        methods |= NNMethod.EA
                    elif "TI" in t:
                        # VAQ.cpp:1236-1251: TI<T>var<v> | TI<T>m<seg> | TI<T>
                        mv = re.match(r"TI(\d+)var([0-9.]+)", t)
                        mm = re.match(r"TI(\d+)m(\d+)", t)
                        m1 = re.match(r"TI(\d+)", t)
                        if mv:
                            methods |= NNMethod.TI
                            self.mTIClusterNum = int(mv.group(1))
                            self.mTIVariance = float(mv.group(2))
                        elif mm:
                            methods |= NNMethod.TI
                            self.mTIClusterNum = int(mm.group(1))
                            self.mTISegmentNum = int(mm.group(2))
                        elif m1:
                            methods |= NNMethod.TI
                            self.mTIClusterNum = int(m1.group(1))
                    elif "FAST3" in t:
                        methods |= NNMethod.Fast3
                    elif "FAST2" in t:
                        methods |= NNMethod.Fast2
                    elif "FAST" in t:
                        methods |= NNMethod.Fast
                unsupported = methods & ~(NNMethod.Heap | NNMethod.EA | NNMethod.TI)
                if unsupported:
                    raise _lib.VaqHipError(-2, f"search method bits 0x{unsupported:02x} in "
                                               f"'{token}' are outside the HEAP/EA/TI path")
                self.mMethods = methods

    def searchMethod(self) -> int:
        return self.mMethods

    # ------------------------------------------------------- derived state --
    @property
    def mHighestSubs(self) -> int:
        return len(self.mBitsAlloc)

    @property
    def mCentroidsNum(self) -> List[int]:
        return [1 << b for b in self.mBitsAlloc]

    @property
    def mTotalDim(self) -> int:
        return sum(c.shape[1] for c in self.mCentroidsPerSubs)

    @property
    def mSubsLen(self) -> int:
        return self.mCentroidsPerSubs[0].shape[1]

    # --------------------------------------------------------------- sync --
    def _ensure_index(self):
        L = _lib.load()
        M = len(self.mBitsAlloc)
        if M == 0 or len(self.mCentroidsPerSubs) != M:
            raise _lib.VaqHipError(-1, "mBitsAlloc / mCentroidsPerSubs not set consistently")
        cents = [np.ascontiguousarray(c, dtype=np.float32) for c in self.mCentroidsPerSubs]
        for b, c in zip(self.mBitsAlloc, cents):
            if c.ndim != 2 or c.shape[0] != (1 << b):
                raise _lib.VaqHipError(-1, f"centroid matrix {c.shape} does not match {b} bits")
        D = sum(c.shape[1] for c in cents)
        eig = None
        if self.mEigenVectors is not None:
            eig = np.ascontiguousarray(np.real(self.mEigenVectors), dtype=np.float32)
            if eig.shape != (D, D):
                raise _lib.VaqHipError(-1, f"mEigenVectors {eig.shape} is not {D}x{D}")
        plain = (tuple(self.mBitsAlloc), self.device, self.sequential_sum)
        if (self._h and self._sig is not None and self._sig[0] == plain
                and len(self._sig[1]) == M and all(_same(r, c) for r, c in zip(self._sig[1], self.mCentroidsPerSubs))
                and _same(self._sig[2], self.mEigenVectors)):
            return
        sig = (plain, tuple(_wref(c) for c in self.mCentroidsPerSubs), _wref(self.mEigenVectors))
        self.close()
        bits = (C.c_int * M)(*self.mBitsAlloc)
        arr = (C.POINTER(C.c_float) * M)()
        for i, c in enumerate(cents):
            arr[i] = c.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        _lib.check(L.vaqhip_index_create_ex(C.byref(h), D, M, bits, arr,
                                            _ptr(eig) if eig is not None else None, self.device,
                                            1 if self.sequential_sum else 0))
        self._h = h
        self._sig = sig
        self._codes_sig = None
        self._ti_sig = None
        self._method_sig = None

    # ---------------------------------------------------------------- TI --
    def decodeFirstSegments(self, rows: np.ndarray, seg: int) -> np.ndarray:
        """Rows of mCodebook as the vectors clusterTI works on: the centroids of
        their first `seg` codes side by side (VAQ.cpp:926-933)."""
        cb = np.asarray(self.mCodebook)[rows]
        return np.concatenate([np.asarray(self.mCentroidsPerSubs[s], np.float32)[cb[:, s].astype(np.int64)]
                               for s in range(seg)], axis=1)

    def clusterTI(self, useKMeans: bool = False, verbose: bool = False, seed: int = 13517106) -> None:
        """VAQ::clusterTI (VAQ.hpp:106, VAQ.cpp:878-999).  Makes mTIClusters when it
        is not set yet -- useKMeans=False: mTIClusterNum random code rows, decoded
        (VAQ.cpp:901-911); True: k-means over decoded code rows (:897-900, at most
        256 rows per centre as KMeans::staticFitCodebook samples) -- and leaves the
        grouping itself (:913-996) to the GPU at the next search."""
        if self.mTIVariance < 1:
            raise _lib.VaqHipError(-2, "TI<T>var<v> needs train()'s variance profile; use TI<T>m<seg>")
        seg = self.mTISegmentNum if self.mTISegmentNum != -1 else self.mHighestSubs  # :890-892
        self.mTISegmentNum = seg
        if self.mTIClusters is None:
            if self.mCodebook is None or hasattr(self.mCodebook, "data_ptr"):
                raise _lib.VaqHipError(-7, "clusterTI needs a host mCodebook (or set mTIClusters)")
            T = self.mTIClusterNum
            N = self.mCodebook.shape[0]
            rng = np.random.default_rng(seed)
            if not useKMeans:
                self.mTIClusters = self.decodeFirstSegments(rng.integers(0, N, size=T), seg)
            else:
                import torch
                from . import harness
                n = min(N, 256 * T)
                X = self.decodeFirstSegments(rng.permutation(N)[:n], seg)
                dev = "cuda" if torch.cuda.is_available() else "cpu"
                self.mTIClusters = harness.kmeans(torch.from_numpy(X).to(dev), T, iters=50,
                                                  seed=seed).cpu().numpy()
        self.mMethods |= NNMethod.TI

    def _ensure_ti(self):
        L = _lib.load()
        if self.mMethods & NNMethod.TI:
            if self.mTIClusters is None:
                raise _lib.VaqHipError(-7, "method TI: call clusterTI() or set mTIClusters first")
            cl = np.ascontiguousarray(self.mTIClusters, dtype=np.float32)
            seg = self.mTISegmentNum if self.mTISegmentNum != -1 else self.mHighestSubs
            if cl.ndim != 2 or cl.shape[1] != seg * self.mSubsLen:
                raise _lib.VaqHipError(-1, f"mTIClusters {cl.shape} is not T x {seg * self.mSubsLen}")
            if self._ti_sig is None or self._ti_sig[1] != seg or not _same(self._ti_sig[0], self.mTIClusters):
                _lib.check(L.vaqhip_index_set_ti_clusters(self._h, _ptr(cl), cl.shape[0], seg))
                self._ti_sig = (_wref(self.mTIClusters), seg)
        elif self._ti_sig is not None:
            _lib.check(L.vaqhip_index_set_ti_clusters(self._h, None, 0, 0))
            self._ti_sig = None
        msig = (self.mMethods, float(self.mVisit))
        if msig != self._method_sig:
            _lib.check(L.vaqhip_index_set_method(self._h, self.mMethods, float(self.mVisit)))
            self._method_sig = msig

    def _ensure_codes(self):
        self._ensure_index()
        self._ensure_ti()  # before the codes: they are then grouped once, not twice
        if self.mCodebook is None:
            if self._codes_sig is not None:
                return  # the packed copy already lives on the device (host copy was dropped)
            raise _lib.VaqHipError(-7, "mCodebook is not set")
        if (self._codes_sig is not None and self._codes_sig[1] == self.id_base
                and _same(self._codes_sig[0], self.mCodebook)):
            return
        sig = (_wref(self.mCodebook), self.id_base)
        cb = self.mCodebook
        if hasattr(cb, "data_ptr"):  # torch tensor already on the device: N x M int16/uint16
            if cb.dim() != 2 or cb.shape[1] != len(self.mBitsAlloc) or cb.element_size() != 2:
                raise _lib.VaqHipError(-1, "device mCodebook must be N x M 16-bit")
            if not cb.is_cuda or cb.device.index != self.device or not cb.is_contiguous():
                raise _lib.VaqHipError(-1, f"device mCodebook must be a contiguous tensor on cuda:{self.device} "
                                           f"(got {cb.device}, contiguous={cb.is_contiguous()})")
            import torch
            st = torch.cuda.current_stream(cb.device).cuda_stream
            _lib.check(_lib.load().vaqhip_index_set_codes_u16_device(
                self._h, C.c_void_p(cb.data_ptr()), cb.shape[0], self.id_base, C.c_void_p(st)))
            torch.cuda.current_stream(cb.device).synchronize()
        else:
            cb = np.ascontiguousarray(cb, dtype=np.uint16)
            if cb.ndim != 2 or cb.shape[1] != len(self.mBitsAlloc):
                raise _lib.VaqHipError(-1, f"mCodebook {cb.shape} is not N x {len(self.mBitsAlloc)}")
            _lib.check(_lib.load().vaqhip_index_set_codes_u16(self._h, _ptr(cb), cb.shape[0],
                                                              self.id_base))
        self._codes_sig = sig

    def add_codes(self, codes: np.ndarray) -> None:
        """Append rows to the index on the device (vaqhip_index_add_codes_u16); the host
        mCodebook, if still attached, grows with them so the two stay the same matrix."""
        self._ensure_codes()
        cb = np.ascontiguousarray(codes, dtype=np.uint16)
        if cb.ndim != 2 or cb.shape[1] != len(self.mBitsAlloc):
            raise _lib.VaqHipError(-1, f"codes {cb.shape} is not n x {len(self.mBitsAlloc)}")
        _lib.check(_lib.load().vaqhip_index_add_codes_u16(self._h, _ptr(cb), cb.shape[0]))
        if self.mCodebook is not None and not hasattr(self.mCodebook, "data_ptr"):
            self.mCodebook = np.concatenate([np.asarray(self.mCodebook, dtype=np.uint16), cb])
            self._codes_sig = (_wref(self.mCodebook), self.id_base)

    # ------------------------------------------------------------- search --
    def search(self, XTest: np.ndarray, k: int, verbose: bool = False,
               projected: bool = False) -> LabelDistVec:
        """VAQ::search (VAQ.cpp:776-847): flat labels / distances, ascending per
        query; squared for HEAP / EA, square roots with TI (VAQ.cpp:1583)."""
        if not (self.mMethods & (NNMethod.Heap | NNMethod.EA | NNMethod.TI)):
            raise _lib.VaqHipError(-2, "only HEAP / EA / TI are implemented on this path")
        self._ensure_codes()
        X = np.ascontiguousarray(XTest, dtype=np.float32)
        if X.ndim != 2 or X.shape[1] != self.mTotalDim:
            raise _lib.VaqHipError(-1, f"XTest {X.shape} is not nq x {self.mTotalDim}")
        nq = X.shape[0]
        ret = LabelDistVec(np.empty(nq * k, np.int32), np.empty(nq * k, np.float32))
        fn = _lib.load().vaqhip_search_projected if projected else _lib.load().vaqhip_search
        _lib.check(fn(self._h, _ptr(X), nq, k, _ptr(ret.labels), _ptr(ret.distances)))
        return ret

    def search_device(self, d_queries, k: int, projected: bool = False, out=None):
        """Device-resident variant: torch CUDA tensors in and out, enqueued on
        torch's current stream (no host copies, no synchronisation).  out =
        (labels int32 [nq,k], distances float32 [nq,k]) reuses caller buffers, so a
        steady-state loop performs no allocation at all."""
        import torch
        self._ensure_codes()
        q = d_queries.contiguous()
        assert q.is_cuda and q.dtype == torch.float32 and q.shape[1] == self.mTotalDim
        nq = q.shape[0]
        if out is not None:
            labels, dists = out
            assert labels.shape == (nq, k) and labels.dtype == torch.int32 and labels.is_contiguous()
            assert dists.shape == (nq, k) and dists.dtype == torch.float32 and dists.is_contiguous()
        else:
            labels = torch.empty((nq, k), dtype=torch.int32, device=q.device)
            dists = torch.empty((nq, k), dtype=torch.float32, device=q.device)
        st = torch.cuda.current_stream(q.device).cuda_stream
        _lib.check(_lib.load().vaqhip_search_device(
            self._h, C.c_void_p(q.data_ptr()), nq, k, 1 if projected else 0,
            C.c_void_p(labels.data_ptr()), C.c_void_p(dists.data_ptr()), C.c_void_p(st)))
        return labels, dists

    # ----------------------------------------------- staged search (sharded hosts) --
    def staged_supported(self, nq: int, k: int) -> bool:
        """vaqhip_search_staged_supported: would a search of nq queries run the bucket-major rounds,
        i.e. can it be split around a threshold exchange between shards?"""
        self._ensure_codes()
        return bool(_lib.load().vaqhip_search_staged_supported(self._h, int(nq), int(k)))

    def search_begin_device(self, d_queries, k: int, out, thr_out, projected: bool = False) -> None:
        """vaqhip_search_begin_device: first rounds; thr_out (int32 [nq], CUDA) receives the thresholds
        (distance bits) to be MIN-reduced over the shards; out = (labels, dists) as for search_device."""
        import torch
        self._ensure_codes()
        q = d_queries.contiguous()
        labels, dists = out
        nq = q.shape[0]
        assert thr_out.is_cuda and thr_out.dtype == torch.int32 and thr_out.numel() == nq and thr_out.is_contiguous()
        st = torch.cuda.current_stream(q.device).cuda_stream
        _lib.check(_lib.load().vaqhip_search_begin_device(
            self._h, C.c_void_p(q.data_ptr()), nq, k, 1 if projected else 0, C.c_void_p(labels.data_ptr()),
            C.c_void_p(dists.data_ptr()), C.c_void_p(thr_out.data_ptr()), C.c_void_p(st)))

    def search_finish_device(self, thr_in=None) -> None:
        """vaqhip_search_finish_device: the rest of the search under the exchanged thresholds."""
        import torch
        dev = thr_in.device if thr_in is not None else torch.device("cuda", self.device)
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().vaqhip_search_finish_device(
            self._h, C.c_void_p(thr_in.data_ptr()) if thr_in is not None else None, C.c_void_p(st)))

    # ----------------------------------------------------- encode / refine --
    def encode(self, XTrain: np.ndarray, projected: bool = True) -> None:
        """VAQ::encode (VAQ.cpp:663-748): fills mCodebook (N x M uint16).  Like
        the reference it expects rows already in PCA space (train() projects the
        dataset in place); projected=False applies mEigenVectors first."""
        self._ensure_index()
        X = np.ascontiguousarray(XTrain, dtype=np.float32)
        codes = np.empty((X.shape[0], len(self.mBitsAlloc)), np.uint16)
        _lib.check(_lib.load().vaqhip_encode(self._h, _ptr(X), X.shape[0], 1 if projected else 0,
                                             _ptr(codes)))
        self.mCodebook = codes

    def encode_device(self, d_X, projected: bool = True):
        """torch CUDA tensor in, N x M int16 CUDA tensor (uint16 codes) out, on
        torch's current stream."""
        import torch
        self._ensure_index()
        x = d_X.contiguous()
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[1] == self.mTotalDim
        codes = torch.empty((x.shape[0], len(self.mBitsAlloc)), dtype=torch.int16, device=x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(_lib.load().vaqhip_encode_device(self._h, C.c_void_p(x.data_ptr()), x.shape[0],
                                                    1 if projected else 0, C.c_void_p(codes.data_ptr()),
                                                    C.c_void_p(st)))
        return codes

    def refine(self, XTest: np.ndarray, answersIn: LabelDistVec, XTrain: np.ndarray, k: int) -> LabelDistVec:
        """VAQ::refine (VAQ.cpp:849-876): exact re-rank of the candidates in
        answersIn against the raw dataset XTrain."""
        _lib.load()
        Xq = np.ascontiguousarray(XTest, dtype=np.float32)
        Xt = np.ascontiguousarray(XTrain, dtype=np.float32)
        nq = Xq.shape[0]
        lab = np.ascontiguousarray(answersIn.labels, dtype=np.int32)
        R = lab.size // max(nq, 1)
        ret = LabelDistVec(np.empty(nq * k, np.int32), np.empty(nq * k, np.float32))
        _lib.check(_lib.load().vaqhip_refine(self.device, _ptr(Xq), nq, Xq.shape[1], _ptr(Xt), Xt.shape[0],
                                             _ptr(lab), R, k, _ptr(ret.labels), _ptr(ret.distances)))
        return ret

    # -------------------------------------------------------- test hooks ---
    def build_lut(self, XTest: np.ndarray, projected: bool = False) -> np.ndarray:
        """CreateLUT for every query in the reference's LUTType layout:
        returns (nq, M, ksub) with lut[q, s, c] (column-major ksub x M per query)."""
        self._ensure_index()
        X = np.ascontiguousarray(XTest, dtype=np.float32)
        nq = X.shape[0]
        ksub = 1 << max(self.mBitsAlloc)
        out = np.empty((nq, len(self.mBitsAlloc), ksub), np.float32)
        _lib.check(_lib.load().vaqhip_build_lut(self._h, _ptr(X), nq, 1 if projected else 0,
                                                _ptr(out)))
        return out

    def project(self, X: np.ndarray) -> np.ndarray:
        """VAQ::ProjectOnEigenVectors (VAQ.hpp:198-201)."""
        self._ensure_index()
        X = np.ascontiguousarray(X, dtype=np.float32)
        out = np.empty_like(X)
        _lib.check(_lib.load().vaqhip_project(self._h, _ptr(X), X.shape[0], _ptr(out)))
        return out

    def invalidate(self) -> None:
        """Members are re-uploaded when they are REPLACED (a different object); after editing
        mCodebook, the centroid matrices, mEigenVectors or mTIClusters IN PLACE call this so
        that the next search rebuilds the device index from the current contents."""
        self.close()
        self._ti_sig = None
        self._method_sig = None

    def set_option(self, key: str, value: int) -> None:
        self._ensure_index()
        _lib.check(_lib.load().vaqhip_set_option(self._h, key.encode(), int(value)))

    def info(self) -> dict:
        self._ensure_index()
        inf = _lib.Info()
        _lib.check(_lib.load().vaqhip_index_info(self._h, C.byref(inf)))
        return {f: getattr(inf, f) for f, _ in _lib.Info._fields_}

    def last_timing(self) -> dict:
        t = _lib.Timing()
        _lib.check(_lib.load().vaqhip_last_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in _lib.Timing._fields_}

    def close(self) -> None:
        if self._h:
            _lib.load().vaqhip_index_destroy(self._h)
            self._h = C.c_void_p()
            self._sig = None
            self._codes_sig = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merge_topk_device(dist_lists, label_lists, k: int, out=None):
    """Multi-GPU exchange step: [n_lists, nq, k] CUDA tensors (labels global,
    empty slots -1 / FLT_MAX) -> per-query k smallest by (distance, label)."""
    import torch
    d = dist_lists.contiguous()
    l = label_lists.contiguous()
    n_lists, nq, kk = d.shape
    assert kk == k and l.shape == d.shape and l.dtype == torch.int32 and d.dtype == torch.float32
    if out is not None:
        out_l, out_d = out
    else:
        out_l = torch.empty((nq, k), dtype=torch.int32, device=d.device)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=d.device)
    st = torch.cuda.current_stream(d.device).cuda_stream
    dev = d.device.index if d.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.load().vaqhip_merge_topk_device(
        dev, C.c_void_p(d.data_ptr()), C.c_void_p(l.data_ptr()), n_lists, nq, k,
        C.c_void_p(out_l.data_ptr()), C.c_void_p(out_d.data_ptr()), C.c_void_p(st)))
    return out_l, out_d


def merge_topk_packed_device(packed, world: int, nq: int, k: int, out=None):
    """packed: int32 CUDA tensor [world, 2, nq, k] from ONE all-gather -- plane 0 holds
    labels, plane 1 the float32 distance bits.  Returns the merged (labels, distances)."""
    import torch
    assert packed.is_contiguous() and packed.dtype == torch.int32 and packed.numel() == world * 2 * nq * k
    if out is not None:
        out_l, out_d = out
    else:
        out_l = torch.empty((nq, k), dtype=torch.int32, device=packed.device)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=packed.device)
    st = torch.cuda.current_stream(packed.device).cuda_stream
    dev = packed.device.index if packed.device.index is not None else torch.cuda.current_device()
    base = packed.data_ptr()
    _lib.check(_lib.load().vaqhip_merge_topk_strided_device(
        dev, C.c_void_p(base + nq * k * 4), C.c_void_p(base), world, 2 * nq * k, k, nq, k,
        C.c_void_p(out_l.data_ptr()), C.c_void_p(out_d.data_ptr()), C.c_void_p(st)))
    return out_l, out_d


class VaqHipMulti:
    """One process, several GPUs (include/vaqhip.h "multi-device"): the rows are sharded
    contiguously over `devices`, every device answers all queries on its shard, one RCCL
    all-gather + merge finishes the search.  Naming a device several times gives logical shards
    on that GPU (gather by device copies) -- the form a one-GPU box can test."""

    def __init__(self, devices: Sequence[int], bits: Sequence[int], cents: Sequence[np.ndarray],
                 eig: Optional[np.ndarray] = None, sequential_sum: bool = False):
        L = _lib.load()
        M = len(bits)
        self._cents = [np.ascontiguousarray(c, dtype=np.float32) for c in cents]
        D = sum(c.shape[1] for c in self._cents)
        self._eig = None if eig is None else np.ascontiguousarray(np.real(eig), dtype=np.float32)
        arr = (C.POINTER(C.c_float) * M)()
        for i, c in enumerate(self._cents):
            arr[i] = c.ctypes.data_as(C.POINTER(C.c_float))
        devs = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        self.D, self.M = D, M
        _lib.check_multi(L.vaqhip_multi_create(C.byref(self._h), D, M, (C.c_int * M)(*bits), arr,
                                               _ptr(self._eig) if self._eig is not None else None,
                                               len(devices), devs, 1 if sequential_sum else 0))

    def set_codes(self, codes: np.ndarray, id_base: int = 0) -> None:
        cb = np.ascontiguousarray(codes, dtype=np.uint16)
        assert cb.ndim == 2 and cb.shape[1] == self.M
        _lib.check_multi(_lib.load().vaqhip_multi_set_codes_u16(self._h, _ptr(cb), cb.shape[0], id_base))

    def add_codes(self, codes: np.ndarray) -> None:
        cb = np.ascontiguousarray(codes, dtype=np.uint16)
        _lib.check_multi(_lib.load().vaqhip_multi_add_codes_u16(self._h, _ptr(cb), cb.shape[0]))

    def set_option(self, key: str, value: int) -> None:
        _lib.check_multi(_lib.load().vaqhip_multi_set_option(self._h, key.encode(), int(value)))

    def set_method(self, methods: int, visit: float = 1.0) -> None:
        _lib.check_multi(_lib.load().vaqhip_multi_set_method(self._h, methods, float(visit)))

    def set_ti_clusters(self, clusters: Optional[np.ndarray], seg: int = 0) -> None:
        if clusters is None:
            _lib.check_multi(_lib.load().vaqhip_multi_set_ti_clusters(self._h, None, 0, 0))
            return
        cl = np.ascontiguousarray(clusters, dtype=np.float32)
        _lib.check_multi(_lib.load().vaqhip_multi_set_ti_clusters(self._h, _ptr(cl), cl.shape[0], seg))

    def search(self, XTest: np.ndarray, k: int, projected: bool = False) -> LabelDistVec:
        X = np.ascontiguousarray(XTest, dtype=np.float32)
        nq = X.shape[0]
        ret = LabelDistVec(np.empty(nq * k, np.int32), np.empty(nq * k, np.float32))
        _lib.check_multi(_lib.load().vaqhip_multi_search(self._h, _ptr(X), nq, k, 1 if projected else 0,
                                                         _ptr(ret.labels), _ptr(ret.distances)))
        return ret

    def search_device(self, d_queries, k: int, projected: bool = False, out=None):
        """vaqhip_multi_search_device: torch CUDA tensors on the FIRST device of the list in and out,
        enqueued behind torch's current stream of that device; nothing is synchronised."""
        import torch
        q = d_queries.contiguous()
        assert q.is_cuda and q.dtype == torch.float32 and q.shape[1] == self.D
        nq = q.shape[0]
        if out is None:
            out = (torch.empty((nq, k), dtype=torch.int32, device=q.device),
                   torch.empty((nq, k), dtype=torch.float32, device=q.device))
        labels, dists = out
        stream = torch.cuda.current_stream(q.device).cuda_stream
        _lib.check_multi(_lib.load().vaqhip_multi_search_device(
            self._h, C.c_void_p(q.data_ptr()), nq, k, 1 if projected else 0, C.c_void_p(labels.data_ptr()),
            C.c_void_p(dists.data_ptr()), C.c_void_p(stream)))
        return labels, dists

    def shard(self, g: int):
        """vaqhip_multi_shard: the raw handle of shard g's single-device index (owned by the multi index)."""
        return _lib.load().vaqhip_multi_shard(self._h, g)

    def info(self) -> dict:
        inf = _lib.MultiInfo()
        _lib.check_multi(_lib.load().vaqhip_multi_get_info(self._h, C.byref(inf)))
        n = inf.n_devices
        return dict(n_devices=n, exchange=inf.exchange, N=inf.N, id_base=inf.id_base,
                    device_ids=list(inf.device_ids)[:n], shard_rows=list(inf.shard_rows)[:n],
                    last_search_ms=inf.last_search_ms, last_exchange_ms=inf.last_exchange_ms,
                    last_merge_ms=inf.last_merge_ms)

    def close(self) -> None:
        if self._h:
            _lib.load().vaqhip_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
