"""Option "exact_ties": labels and distances identical to VAQ::search's, slot for slot, also where
rows tie -- the reference's choice among equal distances comes from its heap (VAQ.cpp:1750-1757,
utils/Heap.hpp:115-169, 322-349), replayed on the GPU (vaq_amd/csrc/vaq_exact.hip).

Checked with plain array_equal (no tie contract) against
  * the golden label lists, every one of which was produced by the compiled reference heap
    (tests/golden/make_golden.py), and
  * oracle.search, whose heap is pinned against the reference's (tests/test_oracle_golden.py)."""
import json
import os

import numpy as np
import pytest

from helpers import make_case

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(json.load(open(os.path.join(GOLD, "manifest.json"))).keys())


def make_index(c, **opts):
    import vaq_amd
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = list(c["bits"])
    v.mCentroidsPerSubs = c["cents"]
    v.mEigenVectors = c["eig"]
    v._ensure_index()
    for key, val in opts.items():
        v.set_option(key, val)
    v.mCodebook = c["codes"]
    v.set_option("exact_ties", 1)
    return v


def same(a, k, o_lab, o_dis, what):
    nq = o_lab.shape[0]
    lab, dis = a.labels.reshape(nq, k), a.distances.reshape(nq, k)
    assert np.array_equal(dis.view(np.uint32), o_dis.view(np.uint32)), f"{what}: distances differ"
    bad = np.nonzero((lab != o_lab).any(axis=1))[0]
    assert bad.size == 0, f"{what}: labels differ for queries {bad[:8]}: {lab[bad[0]]} vs {o_lab[bad[0]]}"


@pytest.mark.parametrize("name", CASES)
def test_golden_labels_exactly(vaqlib, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    bits = z["bits"].tolist()
    c = dict(bits=bits, cents=[z[f"cent{s}"] for s in range(len(bits))], eig=z["eig"], codes=z["codes"])
    v = make_index(c)
    seen = 0
    for key in z.files:
        if not key.startswith("labels_k"):
            continue
        k = int(key[len("labels_k"):])
        if k >= 1024:
            continue
        for bf in (1, 0):
            v.set_option("best_first", bf)
            same(v.search(z["X"], k), k, z[key], z[f"dists_k{k}"], f"{name} k={k} bf={bf}")
            seen += 1
    assert seen >= 2
    v.close()


CONFIGS = [
    # seed, D, bits, N, nq, k, make_case kwargs
    (301, 16, [3] * 4, 5000, 16, 100, {"integer": True}),                    # one dword per row, massive ties
    (302, 64, [4] * 8, 30000, 12, 100, {"integer": True, "rotate": False}),
    (303, 128, [8] * 8, 40000, 9, 100, {"dup_frac": 0.3}),
    (304, 128, [8] * 16, 20000, 5, 37, {"dup_frac": 0.5}),
    (305, 128, [12, 10, 9, 8, 8, 7, 6, 4], 20000, 6, 100, {"dup_frac": 0.2}),
    (306, 64, [4] * 8, 90, 5, 100, {"integer": True, "rotate": False}),       # N < k
    (307, 64, [4] * 8, 100, 5, 100, {"integer": True, "rotate": False}),      # N == k
    (308, 64, [4] * 8, 3000, 7, 1, {"integer": True, "rotate": False}),       # k = 1
    (309, 64, [4] * 8, 70000, 4, 1000, {"integer": True, "rotate": False}),   # k near the maximum
    (310, 96, [5, 6, 7, 9, 11, 13, 3, 2, 1, 4, 8, 10], 8000, 3, 50, {"dup_frac": 0.4}),  # fields straddle dwords
]


@pytest.mark.parametrize("seed,D,bits,N,nq,k,kw", CONFIGS, ids=[f"s{c[0]}" for c in CONFIGS])
def test_exact_ties_matches_oracle(vaqlib, oracle, seed, D, bits, N, nq, k, kw):
    c = make_case(seed, D, bits, N, nq, **kw)
    if c["eig"] is None:
        c["eig"] = np.eye(D, dtype=np.float32)  # (small integers stay integers: sums are exact, ties everywhere)
    c["X"][1] = np.nan  # FLT_MAX > NaN is false for every row: all slots stay -1 / FLT_MAX
    Xp = oracle.project(c["X"], c["eig"])
    o_lab, o_dis = oracle.search(Xp, c["cents"], c["codes"], k, max_bits=max(bits), projected=True, nthreads=8)
    ties = sum(int(np.any(np.diff(o_dis[q][o_lab[q] >= 0]) == 0)) for q in range(nq))
    assert ties >= 1 or k == 1 or N <= k, "the case is meant to have equal distances"
    v = make_index(c)
    for qb, ea, bf in [(0, 3, 1), (1, 1, 0), (2, 2, 1), (4, 0, 1)]:
        v.set_option("queries_per_pass", qb)
        v.set_option("early_abandon", ea)
        v.set_option("best_first", bf)
        same(v.search(c["X"], k), k, o_lab, o_dis, f"qb={qb} ea={ea} bf={bf}")
    # the option off: the documented (distance, label) order again
    v.set_option("exact_ties", 0)
    a = v.search(c["X"], k)
    assert np.array_equal(a.distances.reshape(nq, k).view(np.uint32), o_dis.view(np.uint32))
    v.close()


def test_exact_ties_at_c2_size(vaqlib, oracle):
    """SIFT-1M shape (1M x 8 B, k = 100) with duplicates planted at each query's k-th distance and
    inside its top k; 64 further queries without planted ties (most end up copied, not replayed)."""
    k, nq, N = 100, 72, 1_000_000
    c = make_case(7321, 128, [8] * 8, N, nq)
    Xp = oracle.project(c["X"], c["eig"])
    rng = np.random.default_rng(1)
    for q in range(8):
        d = oracle.all_dists(oracle.create_lut(Xp[q], c["cents"], 8), c["codes"])
        order = np.argpartition(d, k)[: k + 1]
        kth = order[np.argsort(d[order])[k - 1]]
        inner = order[np.argsort(d[order])[k // 2]]
        for dst in rng.integers(0, N, size=4):
            c["codes"][dst] = c["codes"][kth]
        for dst in rng.integers(0, N, size=2):
            c["codes"][dst] = c["codes"][inner]
    o_lab, o_dis = oracle.search(Xp, c["cents"], c["codes"], k, max_bits=8, projected=True, nthreads=8)
    v = make_index(c)
    for bf in (1, 0):
        v.set_option("best_first", bf)
        same(v.search(c["X"], k), k, o_lab, o_dis, f"planted ties bf={bf}")
    v.close()


def test_exact_ties_with_bucket_major_and_appends(vaqlib, oracle):
    """The scan behind the option may be any form (here the bucket-major rounds, forced), and rows
    appended later are replayed in their place: original order = label order."""
    import vaq_amd
    bits = [8] * 8
    N, nq, k = 320_000, 40, 20
    c = make_case(411, 64, bits, N, nq, dup_frac=0.4)
    Xp = oracle.project(c["X"], c["eig"])
    o_lab, o_dis = oracle.search(Xp, c["cents"], c["codes"], k, max_bits=8, projected=True, nthreads=8)
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = list(bits)
    v.mCentroidsPerSubs = c["cents"]
    v.mEigenVectors = c["eig"]
    v.mCodebook = c["codes"][:250_000]
    v._ensure_codes()
    v.set_option("exact_ties", 1)
    half_lab, half_dis = oracle.search(Xp, c["cents"], c["codes"][:250_000], k, max_bits=8, projected=True, nthreads=8)
    same(v.search(c["X"], k), k, half_lab, half_dis, "before the append")
    v.add_codes(c["codes"][250_000:])
    v.set_option("timing", 1)
    for bm in (2, 0):
        v.set_option("bucket_major", bm)
        same(v.search(c["X"], k), k, o_lab, o_dis, f"after the append, bucket_major={bm}")
        assert v.last_timing()["bucket_major"] == (1 if bm else 0)
    v.close()
