#!/usr/bin/env python3
"""Randomised parity soak on a GPU box: random index shapes, bit allocations, sizes,
k, and scan options, each checked against the CPU oracle under the tie contract.

    python tools/fuzz_parity.py [seconds] [seed]

Every third index is also regrouped by random triangle-inequality clusters and
searched with TI|EA / TI at random visit fractions.  A progress line is printed
every ~20 s (a silent GPU job is taken for hung).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import vaq_amd
from helpers import assert_topk_matches, make_case
from oracle import pyoracle as po

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
only_bm = len(sys.argv) > 3 and sys.argv[3] == "bm"  # every index a byte-coded one large enough for the bucket-major rounds
rng = np.random.default_rng(seed)
po.build(ref=False)
t0 = time.time(); n_cases = 0; n_searches = 0; ties = 0; n_ti = 0; n_bf = 0; n_append = 0; n_multi = 0; t_print = t0
n_bm = 0; n_exact = 0


def visited_dists(c, ti, T, seg, visit, k, Xp):
    out = np.full((Xp.shape[0], c["codes"].shape[0]), np.inf, dtype=np.float32)
    sizes = np.diff(ti["start"])
    for q in range(Xp.shape[0]):
        qcc, order = po.ti_query_order(Xp[q, :seg * c["L"]], ti["clusters"])
        nv = int(np.float32(T) * np.float32(visit)) if visit < 1 else T
        cum = np.cumsum(sizes[order])
        enough = int(np.searchsorted(cum, k) + 1) if cum[-1] >= k else T
        nv = min(T, max(nv, enough))
        rows = np.concatenate([ti["member"][ti["start"][t]:ti["start"][t + 1]] for t in order[:nv]])
        lut = po.create_lut(Xp[q], c["cents"], max(c["bits"]))
        out[q, rows] = np.sqrt(po.all_dists(lut, c["codes"][rows]))
    return out


while time.time() - t0 < budget:
    if time.time() - t_print > 20:
        t_print = time.time()
        print(f"[{t_print - t0:.0f}s] {n_cases} indexes, {n_searches} searches ({n_ti} TI, {n_bf} best-first, {n_bm} bucket-major, "
              f"{n_exact} exact-ties)", flush=True)
    M = int(rng.choice([4, 8, 8, 8, 12, 16, 16, 20, 32, 64]))
    L = int(rng.choice([1, 2, 4, 8, 16]))
    D = M * L
    kind = rng.integers(0, 5)  # (two of five: every code 8 bits -- the byte layout and its best-first form)
    kind = 0 if kind == 4 else kind
    if only_bm:
        kind, M = 0, int(rng.choice([8, 16, 32]))
        L = int(rng.choice([1, 4, 8]))
        D = M * L
    if kind == 0:
        bits = [8] * M
    elif kind == 1:
        bits = [int(rng.integers(1, 9)) for _ in range(M)]
    elif kind == 2:
        bits = sorted([int(rng.integers(1, 13)) for _ in range(M)], reverse=True)
    else:
        bits = [int(rng.integers(3, 6))] * M
    while sum(bits) > 256:
        bits[int(np.argmax(bits))] -= 1
    N = int(rng.choice([1, 63, 64, 65, 500, 4097, 20000, 150000]))
    if kind == 0 and M in (8, 16, 32) and (only_bm or rng.integers(0, 2)):
        N = int(rng.choice([260_000, 300_001, 520_000]))  # (enough rows for whole-first-code bucket keys: the bucket-major rounds)
    nq = int(rng.choice([1, 2, 3, 7, 33]))
    if only_bm:
        nq = int(rng.choice([5, 33, 130, 700]))
    if N <= 20000 and rng.integers(0, 12) == 0:
        nq = 1100  # (from 1024 queries on, one-workgroup-per-query launches rank the queries by cost)
    k = int(rng.choice([1, 5, 64, 100, 128, 129, 500]))
    if only_bm:
        k = int(rng.choice([1, 5, 64, 100, 128, 129, 256]))
    c = make_case(int(rng.integers(1 << 30)), D, bits, N, nq, dup_frac=float(rng.choice([0, 0.05, 0.5])),
                  integer=bool(rng.integers(0, 4) == 0), rotate=bool(rng.integers(0, 2)))
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = bits; v.mCentroidsPerSubs = c["cents"]; v.mEigenVectors = c["eig"]; v.mCodebook = c["codes"]
    if rng.integers(0, 2):  # bucket key width (incl. keys that continue into the second code)
        v._ensure_index()
        v.set_option("bucket_bits", int(rng.choice([8, 9, 10, 12])) if only_bm else int(rng.integers(1, 13)))
    if rng.integers(0, 4) == 0:
        v._ensure_index()
        v.set_option("sub_order", 0)  # (rows of a bucket left in label order: no runs to skip)
    if N >= 500 and rng.integers(0, 4) == 0:  # the same rows arriving in pieces (vaqhip_index_add_codes_u16)
        cuts = sorted(set(int(x) for x in rng.integers(1, N, size=int(rng.integers(1, 4))))) + [N]
        v.mCodebook = c["codes"][:cuts[0]]
        v._ensure_codes()
        for a_, b_ in zip(cuts[:-1], cuts[1:]):
            v.add_codes(c["codes"][a_:b_])
        n_append += 1
    Xp = po.project(c["X"], c["eig"]) if c["eig"] is not None else c["X"]
    o_lab, o_dis = po.search(Xp, c["cents"], c["codes"], k, max_bits=max(bits), projected=True, nthreads=8)
    ad = np.stack([po.all_dists(po.create_lut(Xp[q], c["cents"], max(bits)), c["codes"]) for q in range(nq)])
    for _ in range(4):
        opts = dict(queries_per_pass=int(rng.choice([0, 1, 2, 4])), early_abandon=int(rng.integers(0, 4)),
                    slices=int(rng.choice([0, 0, 1, 2, 5, 300])), hot_buckets=int(rng.choice([0, 3, 16, 32])),
                    waves_per_workgroup=int(rng.choice([0, 4, 8, 16])), seed_thresholds=int(rng.integers(0, 2)),
                    ordered_slices=int(rng.integers(0, 2)), best_first=int(rng.integers(0, 3) > 0),
                    group_queries=int(rng.choice([0, 1, 2, 2])), cost_order=int(rng.integers(0, 2)),
                    defer_units=int(rng.choice([0, 0, 0, 1, 4])),
                    bucket_major=int(rng.choice([0, 1, 2, 2])), bm_boot=int(rng.integers(0, 3)), bm_round=int(rng.choice([0, 1, 6])),
                    bm_candidates=int(rng.choice([0, 0, 1, 16, 300])), bm_units=int(rng.choice([0, 1, 3])),
                    bm_queries_per_group=int(rng.choice([0, 2, 4])), bm_waves=int(rng.choice([0, 4, 8, 16])),
                    bm_runs=int(rng.integers(0, 2)))
        if only_bm:
            opts.update(ordered_slices=0, slices=int(rng.choice([0, 1])), best_first=1, bucket_major=2, queries_per_pass=0)
        for key, val in opts.items():
            v.set_option(key, val)
        v.set_option("timing", 1)
        try:
            a = v.search(c["X"], k)
            n_bf += v.last_timing()["best_first"]
            n_bm += v.last_timing()["bucket_major"]
        except vaq_amd.VaqHipError as e:
            if e.code == -2:  # outside this build's limits (reported, not a parity failure)
                n_unsupported = globals().get("n_unsupported", 0) + 1
                globals()["n_unsupported"] = n_unsupported
                break
            raise
        try:
            ties += assert_topk_matches(a.labels.reshape(nq, k), a.distances.reshape(nq, k), o_lab, o_dis, ad,
                                        what=f"M={M} L={L} bits={bits} N={N} nq={nq} k={k} {opts}")
        except AssertionError as e:
            print("MISMATCH", e); print("case seed info:", M, L, bits, N, nq, k, opts)
            print("timing/plan of the failing search:", v.last_timing(), v.info())
            # which option matters: flip each one back to its default in turn
            for key2, dflt in (("best_first", 0), ("best_first", 1), ("ordered_slices", 0), ("slices", 0), ("seed_thresholds", 0),
                               ("hot_buckets", 16), ("waves_per_workgroup", 0), ("queries_per_pass", 1), ("group_queries", 0),
                               ("cost_order", 0), ("defer_units", 0), ("bucket_major", 0), ("bm_runs", 0), ("bm_boot", 0)):
                v.set_option(key2, dflt)
                b2 = v.search(c["X"], k)
                same = np.array_equal(b2.distances.reshape(nq, k), o_dis)
                print(f"  with {key2}={dflt}: distances {'match' if same else 'DIFFER'}  plan {v.last_timing()}")
                v.set_option(key2, opts[key2])
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_fail_case.npz"), X=c["X"], codes=c["codes"],
                     eig=c["eig"] if c["eig"] is not None else np.zeros(0), bits=np.array(bits),
                     **{f"cent{s_}": c["cents"][s_] for s_ in range(M)}, o_lab=o_lab, o_dis=o_dis,
                     g_lab=a.labels.reshape(nq, k), g_dis=a.distances.reshape(nq, k))
            sys.exit(1)
        n_searches += 1
        if k < 1024 and rng.integers(0, 3) == 0:
            # option exact_ties behind the same scan form: labels and distances equal the oracle's slot for slot
            v.set_option("exact_ties", 1)
            x = v.search(c["X"], k)
            v.set_option("exact_ties", 0)
            if not (np.array_equal(x.labels.reshape(nq, k), o_lab) and
                    np.array_equal(x.distances.reshape(nq, k).view(np.uint32), o_dis.view(np.uint32))):
                print("MISMATCH (exact_ties)", M, L, bits, N, nq, k, opts)
                bad = np.nonzero((x.labels.reshape(nq, k) != o_lab).any(axis=1))[0]
                print("queries", bad[:5], x.labels.reshape(nq, k)[bad[0]], o_lab[bad[0]])
                sys.exit(1)
            n_searches += 1; n_exact += 1
    if n_cases % 5 == 1 and N >= 64:
        # the multi-device index with logical shards on this GPU (exchange by copies) == the same answers
        from vaq_amd.index import VaqHipMulti
        g = int(rng.choice([1, 2, 3, 5, 8]))
        try:
            m = VaqHipMulti([0] * g, bits, c["cents"], c["eig"])
            m.set_codes(c["codes"])
            m.set_option("queries_per_pass", int(rng.choice([0, 1, 2, 4])))
            a = m.search(c["X"], k)
            ties += assert_topk_matches(a.labels.reshape(nq, k), a.distances.reshape(nq, k), o_lab, o_dis, ad,
                                        what=f"multi g={g} M={M} L={L} bits={bits} N={N} nq={nq} k={k}")
            m.close()
            n_searches += 1; n_multi += 1
        except vaq_amd.VaqHipError as e:
            if e.code != -2:
                raise
        except AssertionError as e:
            print("MISMATCH (multi)", e); sys.exit(1)
    if n_cases % 3 == 0 and N >= 64 and M % 4 == 0 and sum(1 << b for b in bits) * 4 < 120000:
        # triangle-inequality form on the same rows
        T = int(rng.choice([1, 2, 7, 40, 300]))
        seg = int(rng.integers(1, M + 1))
        if seg * L <= 1024:
            pick = rng.integers(0, N, size=T)
            cl = np.concatenate([c["cents"][s][c["codes"][pick, s].astype(np.int64)] for s in range(seg)], axis=1)
            if rng.integers(0, 3) == 0:
                cl = cl + rng.normal(size=cl.shape).astype(np.float32)
            cl = np.ascontiguousarray(cl, dtype=np.float32)
            ti = po.cluster_ti(c["codes"], c["cents"], cl, seg, nthreads=8)
            v.mTIClusters = cl; v.mTISegmentNum = seg
            for _ in range(3):
                visit = float(rng.choice([1.0, 0.5, 0.1, 0.013]))
                ea = bool(rng.integers(0, 4) != 0)
                v.mMethods = vaq_amd.NNMethod.TI | (vaq_amd.NNMethod.EA if ea else 0)
                v.mVisit = visit
                v.set_option("slices", int(rng.choice([0, 0, 1, 3, 40])))
                v.set_option("waves_per_workgroup", int(rng.choice([0, 4, 8, 16])))
                a = v.search(c["X"], k)
                ol, od, _ = po.search_ti(Xp, c["cents"], ti, k, visit=visit, max_bits=max(bits), ea=ea,
                                         nthreads=8, projected=True)
                try:
                    alld = visited_dists(c, ti, T, seg, visit, k, Xp) if ea else None
                    if ea:
                        ties += assert_topk_matches(a.labels.reshape(nq, k), a.distances.reshape(nq, k), ol, od,
                                                    alld, what=f"TI M={M} L={L} bits={bits} N={N} T={T} seg={seg}")
                    else:  # first k rows of the visiting order: the same rows, whatever their distances tie like
                        assert np.array_equal(a.distances.reshape(nq, k), od)
                        assert np.array_equal(np.sort(a.labels.reshape(nq, k), 1), np.sort(ol, 1))
                except AssertionError as e:
                    print("MISMATCH (TI)", e)
                    print("case:", M, L, bits, N, nq, k, "T", T, "seg", seg, "visit", visit, "ea", ea); sys.exit(1)
                n_searches += 1; n_ti += 1
    v.close(); n_cases += 1
print("unsupported (EUNSUPPORTED) cases:", globals().get("n_unsupported", 0))
print(f"fuzz ok: {n_cases} indexes ({n_append} built by appends), {n_searches} searches ({n_bf} in the best-first form, {n_bm} with "
      f"bucket-major rounds, {n_exact} with exact_ties, "
      f"{n_multi} on the multi-device index), {ties} boundary-tie queries, "
      f"{time.time()-t0:.0f}s, seed {seed}")
