"""GPU parity at BASELINE.json's large configurations, on one device:

  C4  100M x 128, 8 x 256 (8 B/row, 800 MB: the whole database of the 4-GPU config)
  C5  1B x 128, 16 x 256 (16 B/row, 16 GB: the whole database of the 8-GPU config), and a
      64M-row cut of it (1 GB: beyond the 256 MB Infinity Cache, second-code bucket keys,
      seeding pre-pass and the in-place scan form all chosen by the library itself)

Each case checks a few queries against the CPU oracle at FULL size (the oracle scans 100M rows
in about half a second per query and thread) and then the properties that need no oracle:
every scan form / queries-per-pass / slicing returns bit-identical results, and the row-sharded
layout of the config (4 or 8 contiguous shards with global labels + the merge kernel that
follows the RCCL all-gather) equals the single index."""
import numpy as np
import pytest

from helpers import assert_topk_matches

pytestmark = pytest.mark.gpu


def big_case(seed, bits, N, nq, D=128, dup=1000):
    """Gaussian codebooks and queries (helpers.make_case's recipe), uniform-random codes drawn
    on the GPU; `dup` rows are copies of other rows so that exactly equal distances occur."""
    import torch
    rng = np.random.default_rng(seed)
    M = len(bits)
    L = D // M
    cents = [(rng.normal(size=(1 << b, L)) * 30.0).astype(np.float32) for b in bits]
    X = (rng.normal(size=(nq, D)) * 30.0).astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(D, D)))
    eig = q.astype(np.float32)
    g = torch.Generator(device="cuda").manual_seed(seed)
    codes = torch.empty((N, M), dtype=torch.int16, device="cuda")
    step = 1 << 26
    for r in range(0, N, step):
        m = min(step, N - r)
        for s, b in enumerate(bits):
            codes[r:r + m, s] = torch.randint(0, 1 << b, (m,), generator=g, device="cuda", dtype=torch.int16)
    if dup:
        src = torch.randint(0, N, (dup,), generator=g, device="cuda")
        dst = torch.randint(0, N, (dup,), generator=g, device="cuda")
        codes[dst] = codes[src]
    return dict(D=D, M=M, bits=list(bits), cents=cents, X=X, eig=eig, codes=codes)


def index_of(c, codes, id_base=0):
    import vaq_amd
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = list(c["bits"])
    v.mCentroidsPerSubs = c["cents"]
    v.mEigenVectors = c["eig"]
    v.mCodebook = codes
    v.id_base = id_base
    v._ensure_codes()
    v.mCodebook = None  # the packed copy lives in the index
    return v


def search_np(v, Xd, k):
    import torch
    l, d = v.search_device(Xd, k)
    torch.cuda.synchronize()
    return l.cpu().numpy(), d.cpu().numpy()


def check_oracle(oracle, c, host_codes, lab, dis, nq_chk, k, what):
    Xp = oracle.project(c["X"][:nq_chk], c["eig"])
    o_lab, o_dis = oracle.search(Xp, c["cents"], host_codes, k, max_bits=max(c["bits"]), projected=True,
                                 nthreads=nq_chk)
    ties = 0
    for q in range(nq_chk):
        try:  # every distance of the database is only needed to adjudicate a boundary tie
            assert_topk_matches(lab[q:q + 1], dis[q:q + 1], o_lab[q:q + 1], o_dis[q:q + 1], None, what=what)
        except AssertionError:
            ad = oracle.all_dists(oracle.create_lut(Xp[q], c["cents"], max(c["bits"])), host_codes)[None]
            ties += assert_topk_matches(lab[q:q + 1], dis[q:q + 1], o_lab[q:q + 1], o_dis[q:q + 1], ad, what=what)
            del ad
    return ties


def check_forms(v, c, Xd, k, base, forms):
    """every (Qb, early-abandon form, slices, seeding, hot buckets) returns the same bits"""
    seen = {}
    for qb, ea, sl, seed, hot in forms:
        v.set_option("queries_per_pass", qb)
        v.set_option("early_abandon", ea)
        v.set_option("slices", sl)
        v.set_option("seed_thresholds", seed)
        v.set_option("hot_buckets", hot)
        v.set_option("timing", 1)
        l, d = search_np(v, Xd, k)
        t = v.last_timing()
        v.set_option("timing", 0)
        seen[(qb, ea, sl, seed, hot)] = t
        assert np.array_equal(l, base[0]) and np.array_equal(d.view(np.uint32), base[1].view(np.uint32)), \
            (qb, ea, sl, seed, hot)
    for key, val in (("queries_per_pass", 0), ("early_abandon", 3), ("slices", 0), ("seed_thresholds", 1),
                     ("hot_buckets", 16)):
        v.set_option(key, val)
    return seen


def check_sorted_unique(lab, dis, N, k):
    assert np.all(np.diff(dis, axis=1) >= 0)
    assert lab.min() >= 0 and lab.max() < N
    assert all(len(set(r.tolist())) == k for r in lab)
    same = np.diff(dis, axis=1) == 0
    assert np.all(np.diff(lab, axis=1)[same] > 0)


def check_shards(c, codes, N, n_shards, Xd, k, base):
    """the config's row-sharded layout on one device: n_shards contiguous shards with global
    labels, their top-k lists in the packed all-gather layout, one merge == the single index"""
    import torch
    from vaq_amd.index import merge_topk_packed_device
    from vaq_amd.sharding import shard_bounds
    nq = Xd.shape[0]
    packed = torch.empty((n_shards, 2, nq, k), dtype=torch.int32, device="cuda")
    for r in range(n_shards):
        lo, hi = shard_bounds(N, n_shards, r)
        vs = index_of(c, codes[lo:hi], id_base=lo)
        vs.search_device(Xd, k, out=(packed[r, 0], packed[r, 1].view(torch.float32)))
        torch.cuda.synchronize()
        vs.close()
    ml, md = merge_topk_packed_device(packed, n_shards, nq, k)
    torch.cuda.synchronize()
    assert np.array_equal(ml.cpu().numpy(), base[0])
    assert np.array_equal(md.cpu().numpy().view(np.uint32), base[1].view(np.uint32))


def test_c4_100m_rows(vaqlib, oracle):
    """BASELINE configs[3]: 100M x 8 B.  4 queries against the oracle at full size, 32 queries
    through every scan form, 4 row shards + merge == the single index."""
    import torch
    N, k, nq = 100_000_000, 100, 32
    c = big_case(4004, [8] * 8, N, nq)
    v = index_of(c, c["codes"])
    Xd = torch.from_numpy(c["X"]).cuda()
    v.set_option("timing", 1)
    base = search_np(v, Xd, k)
    t = v.last_timing()
    v.set_option("timing", 0)
    assert t["bucket_major"] == 1, t  # 32 queries on 800 MB of codes: the bucket-major rounds
    v.set_option("bucket_major", 0)
    v.set_option("timing", 1)
    old = search_np(v, Xd, k)
    t = v.last_timing()
    v.set_option("timing", 0)
    assert t["bucket_major"] == 0 and t["slices"] > 1, t  # ... and without them the plan cuts 100M rows into slices
    assert np.array_equal(old[0], base[0]) and np.array_equal(old[1].view(np.uint32), base[1].view(np.uint32))
    v.set_option("bucket_major", 1)
    check_sorted_unique(base[0], base[1], N, k)
    host = c["codes"].cpu().numpy().view(np.uint16)
    check_oracle(oracle, c, host, base[0], base[1], 4, k, "c4 100M")
    del host
    check_forms(v, c, Xd, k, base, [(1, 1, 0, 1, 16), (2, 1, 0, 1, 16), (4, 1, 0, 1, 16), (1, 2, 0, 1, 16),
                                    (2, 2, 0, 0, 16), (4, 2, 300, 1, 0), (2, 0, 0, 1, 16), (2, 1, 1, 1, 16),
                                    (4, 1, 2048, 0, 16)])
    # fewer queries than a pass holds, and a ragged batch
    for n in (1, 3, 5):
        l, d = search_np(v, Xd[:n].contiguous(), k)
        assert np.array_equal(l, base[0][:n]) and np.array_equal(d, base[1][:n])
    v.close()
    check_shards(c, c["codes"], N, 4, Xd, k, base)


def test_c5_cut_64m_rows(vaqlib, oracle):
    """64M x 16 B = 1 GB: every regime the 1B scan uses is the library's own choice here --
    bucket key continued into the second code (>= 16M rows), seeding pre-pass (>= 256 slices),
    in-place form for few queries, Qb = 4 for >= 32."""
    import torch
    N, k, nq = 64_000_000, 100, 32
    c = big_case(5005, [8] * 16, N, nq)
    v = index_of(c, c["codes"])
    Xd = torch.from_numpy(c["X"]).cuda()
    v.set_option("timing", 1)
    auto = search_np(v, Xd, k)
    assert v.last_timing()["bucket_major"] == 1  # (from 8 queries on: the bucket-major rounds)
    v.set_option("bucket_major", 0)
    base = search_np(v, Xd, k)
    t32 = v.last_timing()
    assert np.array_equal(auto[0], base[0]) and np.array_equal(auto[1].view(np.uint32), base[1].view(np.uint32))
    l2, d2 = search_np(v, Xd[:2].contiguous(), k)
    t2 = v.last_timing()
    v.set_option("bucket_major", 1)
    v.set_option("timing", 0)
    assert t32["queries_per_pass"] == 4 and t32["seed_slices"] > 0 and t32["early_abandon"] == 1, t32
    assert t2["queries_per_pass"] == 2 and t2["early_abandon"] == 2 and t2["seed_slices"] > 0, t2
    assert np.array_equal(l2, base[0][:2]) and np.array_equal(d2, base[1][:2])
    check_sorted_unique(base[0], base[1], N, k)
    host = c["codes"].cpu().numpy().view(np.uint16)
    check_oracle(oracle, c, host, base[0], base[1], 4, k, "c5 cut 64M")
    del host
    check_forms(v, c, Xd, k, base, [(1, 1, 0, 1, 16), (2, 1, 0, 1, 16), (2, 2, 0, 1, 16), (4, 2, 0, 0, 16),
                                    (1, 2, 0, 1, 0), (4, 1, 0, 0, 16), (2, 0, 0, 1, 16), (4, 1, 1000, 1, 16)])
    # many queries: the library switches to the bucket-major rounds (vaq_scan_bm.hip: every bucket
    # streamed once for all the queries that reach it); same answers as the shared-stream forms, as
    # the best-first form (one query per workgroup, several slices per query, thresholds exchanged
    # between them) and as the rounds in every shape
    big = np.concatenate([c["X"]] * 8)[:240] + np.float32(0)
    bigd = torch.from_numpy(np.ascontiguousarray(big)).cuda()
    v.set_option("timing", 1)
    lb_, db_ = search_np(v, bigd, k)
    tb = v.last_timing()
    assert tb["bucket_major"] == 1 and tb["queries_per_pass"] == 1 and tb["slices"] == 1, tb
    for j in range(240):
        assert np.array_equal(lb_[j], base[0][j % 32]) and np.array_equal(db_[j], base[1][j % 32]), j
    v.set_option("bucket_major", 0)
    l1_, d1_ = search_np(v, bigd, k)
    t1 = v.last_timing()
    assert t1["bucket_major"] == 0 and t1["best_first"] == 1 and t1["queries_per_pass"] == 1 and t1["slices"] > 1, t1
    assert np.array_equal(l1_, lb_) and np.array_equal(d1_, db_)
    v.set_option("best_first", 0)
    l0_, d0_ = search_np(v, bigd, k)
    assert np.array_equal(l0_, lb_) and np.array_equal(d0_, db_)
    v.set_option("best_first", 1)
    v.set_option("bucket_major", 1)
    for opts in (dict(bm_boot=0), dict(bm_boot=2, bm_round=0), dict(bm_boot=2, bm_round=2, bm_candidates=64),
                 dict(bm_boot=0, bm_units=4, bm_runs=0), dict(bm_queries_per_group=2, bm_waves=8)):
        for key, val in opts.items():
            v.set_option(key, val)
        l2_, d2_ = search_np(v, bigd, k)
        assert v.last_timing()["bucket_major"] == 1, opts
        assert np.array_equal(l2_, lb_) and np.array_equal(d2_, db_), opts
        for key, val in (("bm_boot", 1), ("bm_round", 6), ("bm_candidates", 0), ("bm_units", 0), ("bm_runs", 1),
                         ("bm_queries_per_group", 0), ("bm_waves", 0)):
            v.set_option(key, val)
    v.set_option("timing", 0)
    # the streaming measurement form (bucket_skip = 0) returns the same results
    v.set_option("bucket_skip", 0)
    for n in (2, 32):
        l, d = search_np(v, Xd[:n].contiguous(), k)
        assert np.array_equal(l, base[0][:n]) and np.array_equal(d, base[1][:n])
    v.set_option("bucket_skip", 1)
    v.close()
    check_shards(c, c["codes"], N, 8, Xd, k, base)


def test_c5_1b_rows(vaqlib, oracle):
    """BASELINE configs[4] at full size on one device: 1B x 16 B.  2 queries against the oracle
    over all 1e9 rows; 2 / 32 queries through the forms the bench uses (streaming pass, default
    mode, Qb 2 / 4); 8 row shards of 125M + merge == the single index."""
    import torch
    N, k, nq = 1_000_000_000, 100, 32
    c = big_case(6006, [8] * 16, N, nq)
    v = index_of(c, c["codes"])
    Xd = torch.from_numpy(c["X"]).cuda()
    base = search_np(v, Xd, k)
    check_sorted_unique(base[0], base[1], N, k)
    X2 = Xd[:2].contiguous()
    v.set_option("timing", 1)
    l2, d2 = search_np(v, X2, k)
    t2 = v.last_timing()
    v.set_option("timing", 0)
    assert t2["early_abandon"] == 2 and t2["seed_slices"] > 0, t2
    assert np.array_equal(l2, base[0][:2]) and np.array_equal(d2, base[1][:2])
    # the bench's roofline launch: one pass, every bucket visited, no pre-pass
    v.set_option("bucket_skip", 0)
    v.set_option("seed_thresholds", 0)
    v.set_option("queries_per_pass", 2)
    l, d = search_np(v, X2, k)
    assert np.array_equal(l, l2) and np.array_equal(d, d2)
    v.set_option("bucket_skip", 1)
    v.set_option("seed_thresholds", 1)
    l, d = search_np(v, Xd, k)  # 32 queries at Qb = 2
    assert np.array_equal(l, base[0]) and np.array_equal(d, base[1])
    v.set_option("queries_per_pass", 0)
    v.close()
    check_shards(c, c["codes"], N, 8, Xd, k, base)
    host = c["codes"].cpu().numpy().view(np.uint16)
    del c["codes"]
    torch.cuda.empty_cache()
    check_oracle(oracle, c, host, base[0], base[1], 8, k, "c5 1B")


def test_c5_encoded_rows(vaqlib, oracle):
    """C5's shape on ENCODED rows (clustered SIFT-shaped vectors through the product's encoder, bench.py's
    recipe): 200M x 16 B, where a query reaches a few per cent of the buckets -- unlike uniform-random codes,
    where nothing can be pruned.  512 queries through the bucket-major rounds (the library's choice), 8 of
    them against the oracle over all rows, and the rounds against the one-workgroup-per-query form."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from vaq_amd import harness
    N, k, nq = 200_000_000, 100, 512
    dev = torch.device("cuda", 0)
    v, host, cents, _ = bench.build_index([8] * 16, N, 0, N, dev, 0, 1, 0, iters=8, keep_host_rows=N)
    queries = harness.sift_like(nq, 128, stream=7, device=dev)
    v.set_option("timing", 1)
    base = search_np(v, queries, k)
    t = v.last_timing()
    assert t["bucket_major"] == 1, t
    check_sorted_unique(base[0], base[1], N, k)
    v.set_option("bucket_major", 0)
    l0, d0 = search_np(v, queries, k)
    assert v.last_timing()["bucket_major"] == 0
    assert np.array_equal(l0, base[0]) and np.array_equal(d0.view(np.uint32), base[1].view(np.uint32))
    v.set_option("bucket_major", 1)
    v.set_option("timing", 0)
    c = dict(X=queries[:8].cpu().numpy(), eig=v.mEigenVectors, cents=cents, bits=[8] * 16)
    v.close()
    check_oracle(oracle, c, host, base[0], base[1], 8, k, "c5 encoded 200M")
