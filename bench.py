#!/usr/bin/env python3
"""bench.py -- queries/sec of the VAQ ADC search path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|c2|c3|c4|c5]

One "step" = one pass of the hot path (project -> LUT build -> code scan with
top-k -> merge [-> RCCL all-gather + merge across GPUs]) over one batch of nq
synthetic queries that are already resident in HBM, results left in HBM.

Workloads (BASELINE.json configs):
  c2  SIFT-1M-shaped, d=128, 8 subspaces x 256 centroids, 10k queries, k=100
      (the configuration the metric is quoted on; the N = 1 default)
  c3  same data, non-uniform bits {12,10,9,8,8,7,6,4}
  c4  100M x 128, 8 x 256, 10k queries
  c5  1B x 128, 16 x 256, 10k queries (the N > 1 default; --rows scales it)

N = 1 (default invocation): `value` is the c2 rate.  c2's code array (8 MB) lives in
cache and most of it is pruned unread, so its scan kernel has no HBM roofline; the
`roofline` object is therefore measured, in the same process, on the one launch of this
path that IS an HBM stream: a single pass over 1B x 16 B of encoded codes with every
bucket visited (`roofline.workload` names it; --no-c5-leg skips it and leaves
`roofline` to the counter-based description of the c2 kernel alone).  The same leg
times the N > 1 default workload on one GPU (`scale_base`), so the per-N lines of a
scaling run have their one-GPU reference.

N > 1 (launched by torch.distributed.run, one rank per GPU): the north-star path --
strong scaling, the code rows sharded contiguously across the ranks, every
rank answers all queries on its shard, one RCCL all-gather of the per-shard top-k and a
merge kernel finish the step.  `--shard queries` replicates the codes and splits the
queries instead; `--scaling replicas` runs N independent replicas (no collective,
labelled as such).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
GEN = 1 << 20           # rows per generated chunk; chunk c is the same on every rank layout
D = 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3", "c4", "c5"])
    ap.add_argument("--rows", type=int, default=0, help="override database rows (total)")
    ap.add_argument("--nq", type=int, default=0, help="override queries per step")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--qb", type=int, default=0, help="queries per pass (0 = library default)")
    ap.add_argument("--slices", type=int, default=0)
    ap.add_argument("--ea", type=int, default=3, help="early abandon: 0 off, 1 queue, 2 in place, 3 auto")
    ap.add_argument("--nwaves", type=int, default=0, help="wavefronts per scan workgroup (0 = auto)")
    ap.add_argument("--seed", type=int, default=1, help="threshold-seeding pre-pass on/off")
    ap.add_argument("--order", type=int, default=0, help="best-first slice order on/off (experimental)")
    ap.add_argument("--seed-frac", type=int, default=0, help="pre-pass scans N / this many rows (0 = default 64)")
    ap.add_argument("--hot", type=int, default=-1, help="best-first buckets per workgroup (0..32, -1 = default)")
    ap.add_argument("--bf", type=int, default=-1, help="best-first scan form on/off (-1 = library default)")
    ap.add_argument("--cost-order", type=int, default=-1, help="expensive queries first (one best-first workgroup per "
                                                               "query): 0 / 1, -1 = library default")
    ap.add_argument("--defer", type=int, default=-1, help="work units of a query's first round before the rest goes to a "
                                                          "second launch (0 = off, -1 = library default)")
    ap.add_argument("--ti", default="", help="T[,seg]: triangle-inequality form with T clusters over the first "
                                             "seg subspaces (default all), method EA_TI (not the headline metric)")
    ap.add_argument("--visit", type=float, default=1.0, help="--visit-cluster of demo_vaq (with --ti)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="any vaqhip_set_option key (experiments), e.g. --set fused_front=0")
    ap.add_argument("--no-skip", action="store_true", help="visit every bucket (streaming-rate measurement)")
    ap.add_argument("--bucket-bits", type=int, default=0, help="bits of the first code that key the row buckets (0 = auto)")
    ap.add_argument("--random-codes", action="store_true", help="uniform-random codes instead of encoded vectors")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--parity-queries", type=int, default=256,
                    help="queries of the timed result checked against the CPU oracle (N = 1, c2/c3)")
    ap.add_argument("--no-c5-leg", action="store_true",
                    help="N = 1 default run: skip the in-process 1B x 16 B roofline / scale_base leg")
    ap.add_argument("--c5-rows", type=int, default=1_000_000_000, help="rows of that leg (rehearsals)")
    ap.add_argument("--c5-launches", type=int, default=12)
    ap.add_argument("--c5-parity-queries", type=int, default=16,
                    help="queries of that leg's 10k-query result checked against the CPU oracle over ALL its rows "
                         "(0 = skip; keeps a uint16 host copy of the codes, 32 GB at 1B rows)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "strong", "replicas", "weak"],
                    help="N > 1: strong (default) = the same queries, rows (or queries) split over the GPUs; "
                         "replicas (alias weak) = every GPU its own nq queries on a replicated index, no collective")
    ap.add_argument("--shard", default="auto", choices=["auto", "rows", "queries"],
                    help="multi-GPU: shard code rows (all-gather + merge; default), or replicate codes and shard queries")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--exchange-thresholds", type=int, default=0,
                    help="rows mode, N > 1: 1 = all-reduce(MIN) the per-query thresholds between the first rounds and the rest "
                         "of the search (the staged search of the C ABI).  Off by default: measured on 8 shards of a 250M-row "
                         "cut it saves 1.6 %% of the shards' scan time (tools/exp_threshold_exchange.py) -- the first rounds "
                         "already leave every shard close to its final thresholds -- against one more collective per step")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate GPUs)")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def base_chunk(c, N, dev):
    from vaq_amd import harness
    m = min(GEN, N - c * GEN)
    return harness.sift_like(m, D, stream=1000 + c, device=dev)


def build_index(bits, N, lo, hi, dev, device_index, world, rank, iters, random_codes=False, ti=None,
                bucket_bits=0, keep_host_rows=0, gt_queries=None, gt_k=100):
    """Train (harness: PCA + k-means on the first chunk; rank 0's state is broadcast), encode
    rows [lo, hi) with the product's own encoder and hand them to a VaqHip index.
    Returns (index, host uint16 copy of the first keep_host_rows rows or None, cents, ti info).
    gt_queries (raw vectors): exact L2 top-gt_k of these queries over rows [lo, hi) is accumulated
    chunk by chunk WHILE the base vectors are generated (a 1B-row base is never generated twice);
    left in build_index.ground_truth = (squared distances, global row ids) on this rank."""
    import torch.distributed as dist
    import vaq_amd
    from vaq_amd import harness
    M = len(bits)
    n_local = hi - lo
    train = base_chunk(0, N, dev)[: min(N, 262144)]
    eig = harness.pca_eigenvectors(train).to(dev)
    if world > 1:
        dist.broadcast(eig, 0)
    tp = train @ eig
    cents = harness.train_codebooks(tp, bits, iters=iters)
    if world > 1:
        for s in range(M):
            t = torch.from_numpy(cents[s]).to(dev)
            dist.broadcast(t, 0)
            cents[s] = t.cpu().numpy()
    del tp, train

    v = vaq_amd.VaqHip(device=device_index)
    v.parseMethodString("VAQ%dm%dmin%dmax%dvar1,HEAP" % (sum(bits), M, min(bits), max(bits)))
    v.mBitsAlloc = list(bits)
    v.mCentroidsPerSubs = cents
    v.mEigenVectors = eig.cpu().numpy()
    v.id_base = lo

    codes = torch.empty((n_local, M), dtype=torch.int16, device=dev)
    build_index.ground_truth = None
    gt_d = gt_i = qq = None
    if gt_queries is not None and not random_codes:
        nqg = gt_queries.shape[0]
        gt_d = torch.full((nqg, gt_k), float("inf"), device=dev)
        gt_i = torch.full((nqg, gt_k), -1, dtype=torch.long, device=dev)
        qq = (gt_queries * gt_queries).sum(1, keepdim=True)
    if not random_codes:
        c0, c1 = lo // GEN, ((hi + GEN - 1) // GEN if hi > lo else lo // GEN)
        for c in range(c0, c1):
            X = base_chunk(c, N, dev)
            a, b = max(lo, c * GEN), min(hi, (c + 1) * GEN)
            xs = X[a - c * GEN: b - c * GEN]
            # the product's own encoder (VAQ::encode on the GPU; projects with eig first)
            codes[a - lo: b - lo] = v.encode_device(xs.contiguous(), projected=False)
            if gt_d is not None and xs.shape[0] > 0:
                d = qq - 2.0 * gt_queries @ xs.T + (xs * xs).sum(1).unsqueeze(0)
                dv, di = torch.topk(d, min(gt_k, xs.shape[0]), dim=1, largest=False)
                cat_d = torch.cat([gt_d, dv], 1)
                cat_i = torch.cat([gt_i, di + a], 1)
                sel = torch.topk(cat_d, gt_k, dim=1, largest=False)
                gt_d, gt_i = sel.values, torch.gather(cat_i, 1, sel.indices)
                del d, dv, di, cat_d, cat_i
            del X, xs
        if gt_d is not None:
            build_index.ground_truth = (gt_d, gt_i)
    else:
        g = torch.Generator(device=dev).manual_seed(harness.SEED + rank)
        step_rows = 1 << 24
        for r in range(0, n_local, step_rows):
            m = min(step_rows, n_local - r)
            codes[r: r + m] = torch.randint(0, 256, (m, M), generator=g, device=dev, dtype=torch.int16)
    ti_info = (0, 0)
    if ti:
        # VAQ::clusterTI: centres = k-means over decoded code rows (at most 256 per centre, as
        # KMeans::staticFitCodebook samples, KMeans.hpp:618-650); training, so it runs in the harness
        ti_T, ti_seg, visit = ti
        ti_seg = ti_seg or M
        g = torch.Generator(device="cpu").manual_seed(harness.SEED)
        pick = torch.randperm(n_local, generator=g)[: min(n_local, 256 * ti_T)].to(dev)
        samp = codes[pick].to(torch.int64) & 0xffff
        dec = torch.cat([torch.from_numpy(cents[s]).to(dev)[samp[:, s]] for s in range(ti_seg)], dim=1)
        cl = harness.kmeans(dec, ti_T, iters=25, seed=harness.SEED)
        if world > 1:
            dist.broadcast(cl, 0)
        v.mTIClusters = cl.cpu().numpy()
        v.mTISegmentNum = ti_seg
        v.mTIClusterNum = ti_T
        v.mMethods = vaq_amd.NNMethod.TI | vaq_amd.NNMethod.EA
        v.mVisit = visit
        ti_info = (ti_T, ti_seg)
        del dec, samp, pick
    if bucket_bits:
        v._ensure_index()
        v.set_option("bucket_bits", bucket_bits)
    v.mCodebook = codes
    v._ensure_codes()
    host_codes = None
    if keep_host_rows > 0:
        host_codes = codes[: min(n_local, keep_host_rows)].cpu().numpy().view(np.uint16)
    del codes
    v.mCodebook = None  # the packed copy lives in the index
    torch.cuda.empty_cache()
    return v, host_codes, cents, ti_info


def settle(run_step, seconds=0.3, world=1, device=None):
    """Part of setup, untimed: the index build frees several GB of staging memory and the
    driver reclaims it asynchronously -- a one-off 40-60 ms stall of the GPU queue lands some
    milliseconds later (tools/step_times.py; INTEGRATION.md "first search after set_codes").
    Run the step until `seconds` have passed so that it is not mistaken for a step time.
    With several ranks the step contains a collective, so every rank must run it the SAME number
    of times: the count is agreed on (max over the ranks) after one measured step -- a loop that
    each rank ends by its own clock leaves one rank waiting in an all-gather nobody else enters."""
    if world > 1:
        import torch.distributed as dist
        sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)
        t = time.perf_counter()
        run_step()
        sync()
        dt = max(time.perf_counter() - t, 1e-4)
        n = torch.tensor([int(min(500, max(1, seconds / dt)))], dtype=torch.int64, device=device)
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        for _ in range(int(n.item())):
            run_step()
            sync()
        return
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        run_step()
        torch.cuda.synchronize()


def scan_kernel_name(info, tm, ti=False, no_skip=False):
    """Name rocprofv3 lists the scan kernel of this plan under (vaq_scan_bytes.hip / _bits.hip)."""
    if ti:
        return "scan_%s_ti_kernel" % ("bytes" if info["layout"] == 0 else "bits")
    if info["layout"] == 0:
        if tm.get("bucket_major"):
            return "scan_bm_kernel<%d, 4, true>" % info["M"]
        if tm["early_abandon"] == 2:
            return "scan_bytes_inplace_kernel<%d, %d, %s>" % (info["M"], tm["queries_per_pass"],
                                                              "true" if no_skip else "false")
        if tm.get("best_first"):
            return "scan_bytes_bf_kernel<%d, %s>" % (info["M"], "true" if info.get("bucket_shift", 0) == 0 else "false")
        return "scan_bytes_kernel<%d, %d, %d>" % (info["M"], tm["queries_per_pass"], tm["early_abandon"])
    if tm.get("best_first"):
        return "scan_bits_bf_kernel (W=%d)" % ((info["total_bits"] + 31) // 32)
    return "scan_bits_kernel (W=%d, Qb=%d, ea=%d)" % ((info["total_bits"] + 31) // 32, tm["queries_per_pass"],
                                                       tm["early_abandon"])


PROFILE_ROUND = "r03"


def load_profile_json(name):
    """Counter figures collected by tools/profile_default.sh + profile_collect.py -- attached only when
    they were measured on THIS build of the library (the files carry the sources' hash); a kernel that
    keeps its name across a tuning change must not be described by the old build's counters."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return {}
    from vaq_amd import build
    if d.get("lib_source_hash") != build.source_hash():
        return {}
    return d


def c5_leg(args, dev, device_index, k):
    """N = 1 default run: the HBM roofline of the path, measured in this process.
    (1) one pass over c5_rows x 16 B of encoded codes, every bucket visited (bucket_skip = 0,
        Qb = 2, 2 queries, no pre-pass): the launch whose algorithmic bytes all cross HBM;
    (2) the same 2 queries in the default mode (bucket skipping + threshold pre-pass);
    (3) the N > 1 default workload (c5, all its queries) on this one GPU: `scale_base`."""
    from vaq_amd import harness
    bits = [8] * 16
    N = args.c5_rows
    nq_full = 10_000
    queries = harness.sift_like(nq_full, D, stream=7, device=dev)
    t0 = time.time()
    n_par = 0 if args.no_cpu else max(0, args.c5_parity_queries)
    v, host_codes, cents, _ = build_index(bits, N, 0, N, dev, device_index, 1, 0, iters=8,
                                          keep_host_rows=N if n_par else 0,
                                          gt_queries=None if args.no_recall else queries[:100].contiguous(), gt_k=k)
    gt = build_index.ground_truth
    info = v.info()
    build_s = time.time() - t0
    log(f"[c5 leg] index of {N} rows built in {build_s:.1f}s")
    q2 = queries[:2].contiguous()
    out2 = (torch.empty((2, k), dtype=torch.int32, device=dev), torch.empty((2, k), dtype=torch.float32, device=dev))

    def timed(fn, launches, warm):
        settle(fn)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        v.set_option("timing", 1)
        v.last_timing()
        t = time.perf_counter()
        for _ in range(launches):
            fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / launches * 1e3
        tm = v.last_timing()
        v.set_option("timing", 0)
        return tm, wall

    # (1) streaming pass
    v.set_option("queries_per_pass", 2)
    v.set_option("seed_thresholds", 0)
    v.set_option("bucket_skip", 0)
    tm, wall = timed(lambda: v.search_device(q2, k, out=out2), args.c5_launches, 3)
    stream_labels = out2[0].clone()
    stream_dists = out2[1].clone()
    algo = float(N) * info["algo_code_bytes"] * tm["passes"]
    achieved = algo / (tm["scan_ms"] * 1e-3) / 1e9
    kname = scan_kernel_name(info, tm, no_skip=True)
    roof = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
        "workload": "synthetic %dx128 m16x256 (encoded), ONE pass: 2 queries, Qb=2, every bucket visited "
                    "(bucket_skip=0), no pre-pass" % N,
        "kernel": kname, "kernel_ms": round(tm["scan_ms"], 4), "launches_timed": tm["n_searches"],
        "wall_ms_per_launch": round(wall, 4), "queries_per_pass": tm["queries_per_pass"], "passes": tm["passes"],
        "rows": N, "code_bytes": info["algo_code_bytes"], "algorithmic_bytes_per_launch": algo,
        "slices": tm["slices"], "workgroups": tm["workgroups"], "lds_bytes": tm["lds_bytes"],
        "other_kernels_ms": {"project": round(tm["project_ms"], 4), "lut_build": round(tm["lut_ms"], 4),
                             "merge": round(tm["merge_ms"], 4)},
        "index_build_s": round(build_s, 1),
    }
    tr = load_profile_json(PROFILE_ROUND + "_traffic.json").get("c5_stream")
    if tr and tr.get("rows") == N and tr.get("kernel") == kname:
        roof["traffic"] = tr["hbm_bytes_per_launch"]
        roof["traffic_source"] = tr["source"]

    # (2) the same two queries, default mode
    v.set_option("queries_per_pass", 0)
    v.set_option("seed_thresholds", 1)
    v.set_option("bucket_skip", 1)
    tm2, wall2 = timed(lambda: v.search_device(q2, k, out=out2), args.c5_launches, 3)
    same = bool(torch.equal(out2[0], stream_labels) and torch.equal(out2[1], stream_dists))
    roof["default_mode"] = {
        "what": "same 2 queries with bucket skipping and the threshold pre-pass on (the library's defaults)",
        "kernel": scan_kernel_name(info, tm2), "kernel_ms": round(tm2["scan_ms"], 4),
        "pre_pass_ms": round(tm2["seed_ms"], 4), "merge_ms": round(tm2["merge_ms"], 4),
        "wall_ms_per_launch": round(wall2, 4), "queries_per_s": round(2 / (wall2 * 1e-3), 1),
        "results_identical_to_streaming_pass": same,
    }
    assert same, "c5 leg: the streaming pass and the default mode disagree"

    # (3) the N > 1 default workload on one GPU
    outf = (torch.empty((nq_full, k), dtype=torch.int32, device=dev),
            torch.empty((nq_full, k), dtype=torch.float32, device=dev))
    tm3, wall3 = timed(lambda: v.search_device(queries, k, out=outf), 2, 1)
    scale_base = {
        "workload": "synthetic %dx128 m16x256 nq10k k100 (the N > 1 default), 1 GPU" % N,
        "value": round(nq_full / (wall3 * 1e-3), 2), "unit": "queries/s", "ms_per_step": round(wall3, 3),
        "steps": 2, "kernel": scan_kernel_name(info, tm3), "kernel_ms": round(tm3["scan_ms"], 3),
        "pre_pass_ms": round(tm3["seed_ms"], 3), "merge_ms": round(tm3["merge_ms"], 3),
        "queries_per_pass": tm3["queries_per_pass"], "passes": tm3["passes"],
    }
    pm = load_profile_json(PROFILE_ROUND + "_bound.json").get("c5_bm")
    if pm and pm.get("rows") == N and pm.get("queries") == nq_full and tm3.get("bucket_major"):
        scale_base["counters"] = pm
    scale_base["form"] = ("bucket-major rounds (vaq_scan_bm.hip): every bucket streamed once for all the queries that "
                          "reach it" if tm3.get("bucket_major") else "one best-first workgroup per (query, slice)")
    assert bool(torch.equal(outf[0][:2], stream_labels) and torch.equal(outf[1][:2], stream_dists)), \
        "c5 leg: the 10k-query batch and the streaming pass disagree on the first two queries"
    if n_par and host_codes is not None:
        # the timed 10k-query result against the CPU oracle over ALL rows (a <= 32-query
        # subset at full N): distances bit for bit, labels under the tie contract
        from oracle import pyoracle as po
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import assert_topk_matches
        po.build(ref=False)
        threads = max(1, min(po.max_threads(), os.cpu_count() or 1, n_par))
        qh = queries[:n_par].cpu().numpy()
        t1 = time.perf_counter()
        ol, od = po.search(qh, cents, host_codes, k, eig=v.mEigenVectors, nthreads=threads)
        dt = time.perf_counter() - t1
        gl, gd = outf[0][:n_par].cpu().numpy(), outf[1][:n_par].cpu().numpy()
        Xp = po.project(qh, v.mEigenVectors)
        boundary = 0
        for q in range(n_par):
            try:
                assert_topk_matches(gl[q:q + 1], gd[q:q + 1], ol[q:q + 1], od[q:q + 1], None, what="c5 leg parity")
            except AssertionError:  # (every distance of the database is only needed to adjudicate a boundary tie)
                ad = po.all_dists(po.create_lut(Xp[q], cents, 8), host_codes)[None]
                boundary += assert_topk_matches(gl[q:q + 1], gd[q:q + 1], ol[q:q + 1], od[q:q + 1], ad, what="c5 leg parity")
                del ad
        scale_base["parity"] = {"checked_queries": n_par, "rows": N, "distances_bit_exact": True,
                                "boundary_tie_queries": boundary, "oracle_seconds": round(dt, 1), "oracle_threads": threads,
                                "what": "queries 0..%d of the timed 10k-query result against oracle/vaq_oracle.c over "
                                        "all %d encoded rows" % (n_par - 1, N)}
        log(f"[c5 leg] {n_par} queries of the 10k-query result match the oracle over {N} rows ({dt:.1f}s, {threads} threads)")
    del host_codes
    if gt is not None:
        lab = outf[0][:100].cpu().numpy()
        scale_base["recall"] = {
            "recall_at_100": round(harness.avg_recall(lab, gt[1].cpu().numpy()), 4),
            "recall_1nn_in_100": round(harness.recall_at_r(lab, gt[1].cpu().numpy()), 4), "queries": 100,
            "ground_truth": "exact L2 brute force over all %d rows, accumulated while the base was generated" % N}
    v.close()
    del v
    torch.cuda.empty_cache()
    return roof, scale_base


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import vaq_amd  # noqa: F401
    from vaq_amd import build, harness, sharding
    from vaq_amd.index import merge_topk_packed_device
    build.build_lib()

    k = args.k
    plan = sharding.bench_plan(world, args.workload, args.scaling, args.shard, rows=args.rows, nq=args.nq)
    wl, N, nq, bits, name = plan["workload"], plan["rows"], plan["nq"], plan["bits"], plan["name"]
    replicas, mode = plan["replicas"], plan["mode"]
    M = len(bits)

    # ---------------------------------------------------------------- setup --
    t_setup = time.time()
    if mode == "rows":
        lo, hi = sharding.shard_bounds(N, world, rank)
    else:
        lo, hi = 0, N
    n_local = hi - lo
    ti = None
    if args.ti:
        parts = [int(x) for x in args.ti.split(",")]
        ti = (parts[0], parts[1] if len(parts) > 1 else 0, args.visit)
    small = wl in ("c2", "c3")
    want_cpu = rank == 0 and not args.no_cpu and world == 1
    real_codes = not args.random_codes
    # replicas: every rank draws its own query batch (disjoint generator streams)
    queries = harness.sift_like(nq, D, stream=7 + (100 * rank if replicas else 0), device=dev)
    # big bases: the exact ground truth of the first queries is accumulated while the base is generated
    RECALL_Q = min(nq, 100)
    big_gt = (not args.no_recall and real_codes and N > 4_000_000 and not replicas and mode == "rows")
    v, host_codes, cents, (ti_T, ti_seg) = build_index(
        bits, N, lo, hi, dev, local_rank, world, rank, iters=15 if small else 8,
        random_codes=args.random_codes, ti=ti, bucket_bits=args.bucket_bits,
        keep_host_rows=(1_000_000 if small else 4_000_000) if want_cpu else 0,
        gt_queries=queries[:RECALL_Q].contiguous() if big_gt else None, gt_k=k)
    shard_gt = build_index.ground_truth
    q_lo, q_hi = (0, nq) if (mode == "rows" or replicas) else sharding.shard_bounds(nq, world, rank)
    my_queries = queries[q_lo:q_hi].contiguous()
    nq_local = q_hi - q_lo
    if args.qb:
        v.set_option("queries_per_pass", args.qb)
    if args.slices:
        v.set_option("slices", args.slices)
    v.set_option("early_abandon", args.ea)
    if args.nwaves:
        v.set_option("waves_per_workgroup", args.nwaves)
    v.set_option("seed_thresholds", args.seed)
    v.set_option("ordered_slices", args.order)
    if args.seed_frac:
        v.set_option("seed_fraction", args.seed_frac)
    if args.hot >= 0:
        v.set_option("hot_buckets", args.hot)
    if args.no_skip:
        v.set_option("bucket_skip", 0)
    if args.bf >= 0:
        v.set_option("best_first", args.bf)
    if args.defer >= 0:
        v.set_option("defer_units", args.defer)
    if args.cost_order >= 0:
        v.set_option("cost_order", args.cost_order)
    for kv in args.set:
        key, _, val = kv.partition("=")
        v.set_option(key, int(val))
    info = v.info()

    # every buffer of the steady-state loop is allocated once, here; with several ranks the
    # labels and distances of a rank travel in ONE all-gather (packed [2, n, k] int32 buffer)
    collective = world > 1 and not replicas
    per_q = (nq + world - 1) // world
    n_pack = nq if (mode == "rows" or replicas) else per_q
    pack_local, lab_view, dis_view = sharding.make_packed(n_pack, k, dev)
    out_local = (lab_view[:nq_local], dis_view[:nq_local])
    if nq_local < n_pack:  # short last query slice: the padding rows stay empty
        lab_view[nq_local:].fill_(-1)
        dis_view[nq_local:].fill_(3.4028234663852886e38)
    gathered = torch.empty((world, 2, n_pack, k), dtype=torch.int32, device=dev) if collective else None
    out_final = (torch.empty((nq, k), dtype=torch.int32, device=dev),
                 torch.empty((nq, k), dtype=torch.float32, device=dev)) if collective else None
    # device time of the exchange step, per step: events on the stream everything is enqueued on
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)] if collective else None

    # rows mode: every shard finds ITS k best -- one all-reduce(MIN) of the nq thresholds between the
    # first rounds and the rest lets each shard scan only what the GLOBAL k-th distance allows
    # (vaq_amd/sharding.py "Threshold exchange"); all ranks must agree on taking that path
    staged = bool(collective and mode == "rows" and args.exchange_thresholds and not args.ti and
                  sharding.staged_agreed(v, nq_local, k))
    thr_buf = torch.empty((nq_local,), dtype=torch.int32, device=dev) if staged else None

    def run_step(i=-1):
        if staged:
            sharding.search_staged(v, my_queries, k, out_local, thr_buf, host_bounce=args.backend != "nccl")
            l, d = out_local
        else:
            l, d = v.search_device(my_queries, k, out=out_local)
        if collective:
            if i >= 0:
                ev[i][0].record()
            sharding.all_gather_packed(pack_local, gathered)
            if i >= 0:
                ev[i][1].record()
            if mode == "rows":
                l, d = merge_topk_packed_device(gathered, world, nq, k, out=out_final)
            else:  # disjoint query slices: the gathered planes ARE the result (strided views)
                l = gathered[:, 0].reshape(world * per_q, k)[:nq]
                d = gathered[:, 1].reshape(world * per_q, k)[:nq].view(torch.float32)
            if i >= 0:
                ev[i][2].record()
        return l, d

    log(f"[rank {rank}] setup {time.time() - t_setup:.1f}s rows_local={n_local} nq={nq} mode={mode} "
        f"replicas={replicas} info={info}")

    # ---------------------------------------------------------------- timed --
    settle(run_step, world=world if collective else 1, device=dev)
    for _ in range(args.warmup):
        run_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        labels, dists_ = run_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # Per-kernel device times: a SEPARATE pass of the same steps with the library's event recording on
    # (HIP events on the launch stream around each kernel).  Not inside the timed region: the six event
    # records per search cost 34 us of a 0.48 ms C2 step (measured: 0.514 ms with them, 0.480 without) --
    # the timed steps above are the plain product path, `kernel_ms` below describes the same kernels.
    v.set_option("timing", 1)
    v.last_timing()  # reset the event ring
    for _ in range(min(args.steps, 20)):
        run_step()
    torch.cuda.synchronize()
    tm = v.last_timing()
    v.set_option("timing", 0)
    exchange = None
    if collective:
        coll = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        mrg = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        stats = torch.tensor([elapsed, coll, mrg, tm["scan_ms"], float(n_local)], dtype=torch.float64, device=dev)
        allst = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allst, stats)
        allst = torch.stack(allst).cpu().numpy()
        elapsed = float(allst[:, 0].max())
        exchange = {
            "world": world, "collective": "all_gather_into_tensor of one packed [2][nq][k] int32 buffer per rank "
                                          "(%s)" % ("RCCL" if args.backend == "nccl" else args.backend),
            "bytes_per_rank": int(2 * n_pack * k * 4),
            "threshold_exchange": ("all-reduce(MIN) of %d int32 thresholds between the first rounds and the rest of every "
                                   "shard's search" % nq_local) if staged else "none",
            "collective_ms": round(float(allst[:, 1].max()), 4), "merge_ms": round(float(allst[:, 2].max()), 4),
            "scan_ms_per_rank": [round(float(x), 4) for x in allst[:, 3]],
            "rows_local": [int(x) for x in allst[:, 4]],
        }
    elif world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    nq_total = nq * world if replicas else nq
    qps = nq_total * args.steps / elapsed

    # ------------------------------------------------------------- roofline --
    # unit = one database row scanned in one pass; bytes = ceil(sum bits / 8).  That
    # product is HBM traffic only when every row is read and nothing is shared through cache: a
    # single pass over a database far beyond the 256 MB Infinity Cache with bucket skipping off.
    # Whatever else this launch was, its kernel is described by `headline_kernel`; an HBM fraction
    # is printed for it only in that streaming case.
    kname = scan_kernel_name(info, tm, ti=bool(args.ti), no_skip=args.no_skip)
    passes = tm["passes"]
    eff = float(n_local) * info["algo_code_bytes"] * nq_local / (tm["scan_ms"] * 1e-3) / 1e9 if tm["scan_ms"] > 0 else 0.0
    streaming = (args.no_skip and passes == 1 and tm["seed_slices"] == 0 and
                 float(n_local) * info["algo_code_bytes"] > 1e9)
    headline_kernel = {
        "kernel": kname, "kernel_ms": round(tm["scan_ms"], 4), "launches_timed": tm["n_searches"],
        "queries_per_pass": tm["queries_per_pass"], "passes": passes,
        "other_kernels_ms": {"project": round(tm["project_ms"], 4), "lut_build": round(tm["lut_ms"], 4),
                             "threshold_seed": round(tm["seed_ms"], 4), "merge": round(tm["merge_ms"], 4)},
        "pre_pass_note": "threshold_seed = whatever runs between the table build and the scan: the threshold pre-pass "
                         "of streamed databases, or the ranking of the queries by cost (query_cost_kernel + "
                         "cost_sort_kernel) where one best-first workgroup serves each query",
        "slices": tm["slices"], "workgroups": tm["workgroups"], "lds_bytes": tm["lds_bytes"],
        "effective_per_query_GBps": round(eff, 1),
        "effective_note": "rows x code bytes x queries / kernel time: counts rows pruned unread and rows "
                          "served from L2 / Infinity Cache, so it is NOT an HBM rate and may exceed the peak",
    }
    roofline = None
    if streaming:
        algo = float(n_local) * info["algo_code_bytes"] * passes
        ach = algo / (tm["scan_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "workload": name + " (this run)",
                    "kernel": kname, "kernel_ms": round(tm["scan_ms"], 4), "launches_timed": tm["n_searches"],
                    "algorithmic_bytes_per_launch": algo}
        tr = load_profile_json(PROFILE_ROUND + "_traffic.json").get("c5_stream")
        if tr and tr.get("rows") == n_local and tr.get("kernel") == kname:
            roofline["traffic"] = tr["hbm_bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
    else:
        # counter-based description of a cache-resident, pruned scan (separate rocprofv3 --pmc passes
        # of this same command: tools/profile_gpu.sh; the summary is committed under profiles/)
        pm = load_profile_json(PROFILE_ROUND + "_bound.json").get(wl)
        if pm and world == 1 and not args.rows and not args.nq and not args.ti and pm.get("kernel") == kname:
            headline_kernel["bound"] = pm

    # --------------------------------------------------------------- recall --
    recall = None
    if not args.no_recall and real_codes and N <= 4_000_000 and rank == 0:
        nq_r = min(nq, 1000)
        Xq = queries[:nq_r]
        nchunks = (N + GEN - 1) // GEN
        gt = harness.brute_force_topk(Xq, ((c * GEN, base_chunk(c, N, dev)) for c in range(nchunks)), k)
        recall = {
            "recall_at_100": round(harness.avg_recall(labels[:nq_r].cpu().numpy(), gt.cpu().numpy()), 4),
            "recall_1nn_in_100": round(harness.recall_at_r(labels[:nq_r].cpu().numpy(), gt.cpu().numpy()), 4),
            "queries": nq_r, "ground_truth": "exact L2 brute force (torch) on the same synthetic base",
        }

    if big_gt and shard_gt is not None:
        gd, gi = shard_gt
        if world > 1:  # the shards' exact top-k -> the global one
            all_d = [torch.empty_like(gd) for _ in range(world)]
            all_i = [torch.empty_like(gi) for _ in range(world)]
            dist.all_gather(all_d, gd)
            dist.all_gather(all_i, gi)
            cat_d, cat_i = torch.cat(all_d, 1), torch.cat(all_i, 1)
            sel = torch.topk(cat_d, k, dim=1, largest=False)
            gi = torch.gather(cat_i, 1, sel.indices)
        if rank == 0:
            lab = labels[:RECALL_Q].cpu().numpy()
            recall = {
                "recall_at_100": round(harness.avg_recall(lab, gi.cpu().numpy()), 4),
                "recall_1nn_in_100": round(harness.recall_at_r(lab, gi.cpu().numpy()), 4),
                "queries": RECALL_Q,
                "ground_truth": "exact L2 brute force (torch) over all %d rows, accumulated chunk by chunk "
                                "while the base was generated" % N,
            }

    # --------------------------------------------------------- cpu baseline --
    cpu = None
    parity = None
    # (rank 0 at N = 1 only: with several ranks on the box the host cores are shared with their
    #  runtime threads and the other ranks would sit in a barrier for the duration)
    if want_cpu:
        from oracle import pyoracle as po
        po.build(ref=False)
        threads = max(1, min(po.max_threads(), os.cpu_count() or 1))
        qh = queries.cpu().numpy()
        eig_h = v.mEigenVectors
        n_rows_cpu = host_codes.shape[0]
        if args.ti:
            # the same method on the CPU: VAQ::clusterTI + searchTriangleInequality restated
            t1 = time.perf_counter()
            ti_h = po.cluster_ti(host_codes, cents, v.mTIClusters, ti_seg, nthreads=threads)
            log(f"[cpu] clusterTI restatement {time.perf_counter() - t1:.1f}s")

            def cpu_search(X, nthreads):
                l_, d_, _ = po.search_ti(X, cents, ti_h, k, visit=args.visit, eig=eig_h, nthreads=nthreads)
                return l_, d_
        else:
            def cpu_search(X, nthreads):
                return po.search(X, cents, host_codes, k, eig=eig_h, nthreads=nthreads)
        # calibrate on `threads` queries, then size the sample for ~cpu_seconds
        t1 = time.perf_counter()
        cpu_search(qh[:threads], threads)
        per_round = max(1e-3, time.perf_counter() - t1)
        n_cpu = int(min(nq, max(threads, threads * (args.cpu_seconds / per_round))))
        t1 = time.perf_counter()
        cl, cd = cpu_search(qh[:n_cpu], threads)
        dt = time.perf_counter() - t1
        scale = n_rows_cpu / float(N)  # rows scanned per query relative to the full job
        cpu_qps = n_cpu / dt * scale
        # single thread = the reference's execution model (VAQ.cpp:786)
        n1 = max(1, min(n_cpu, int(3.0 / (per_round)) + 1))
        t1 = time.perf_counter()
        cpu_search(qh[:n1], 1)
        dt1 = time.perf_counter() - t1
        cpu = {
            "value": round(cpu_qps, 2), "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": f"{n_cpu} of {nq} queries x {n_rows_cpu} rows (uint16 row-major codes), "
                      f"oracle/vaq_oracle.c, OpenMP over queries, {dt:.1f}s"
                      + ("" if scale == 1.0 else f", scaled x{scale:.4g} to {N} rows"),
            "single_thread_qps": round(n1 / dt1 * scale, 2),
        }
        if n_rows_cpu == n_local and args.ti:
            # distances must agree exactly; labels may differ only inside runs of equal distance
            chk = min(n_cpu, 32)
            gd = dists_[:chk].cpu().numpy()
            assert np.array_equal(gd, cd[:chk]), "bench parity (TI): distances differ"
            gl = labels[:chk].cpu().numpy()
            for q in range(chk):
                for dv in np.unique(gd[q][:-1][gd[q][:-1] != gd[q][-1]]):
                    assert set(gl[q][gd[q] == dv]) == set(cl[q][cd[q] == dv]), "bench parity (TI): labels differ"
            parity = {"checked_queries": chk, "contract": "TI: distances bit-exact, labels equal per run of equal distance"}
        elif n_rows_cpu == n_local:
            # same inputs: the CPU port is also the parity checker for the bench's own (timed) result
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import assert_topk_matches
            chk = min(n_cpu, args.parity_queries)
            gl, gd = labels[:chk].cpu().numpy(), dists_[:chk].cpu().numpy()
            Xp = po.project(qh[:chk], eig_h)
            boundary = 0
            interior = 0
            for q in range(chk):
                ad = po.all_dists(po.create_lut(Xp[q], cents, max(bits)), host_codes)[None]
                boundary += assert_topk_matches(gl[q:q + 1], gd[q:q + 1], cl[q:q + 1], cd[q:q + 1], ad,
                                                what="bench parity")
                interior += int(np.any(np.diff(gd[q]) == 0))
            parity = {
                "checked_queries": chk, "distances_bit_exact": True,
                "boundary_tie_queries": boundary,
                "queries_with_equal_distances_inside_topk": interior,
                "contract": "distances bit-exact rank for rank; labels equal after sorting each run of equal "
                            "distances; a boundary tie (k-th == (k+1)-th distance) keeps the smallest labels "
                            "where the reference's choice depends on its heap (DESIGN.md 'Ties')",
            }
            # option "exact_ties": the same batch with the reference's own choice among equal distances
            # (its heap replayed for the queries that have ties): labels identical slot for slot
            v.set_option("exact_ties", 1)
            xl, xd = v.search_device(my_queries, k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                v.search_device(my_queries, k, out=(xl, xd))
            torch.cuda.synchronize()
            x_ms = (time.perf_counter() - t1) / 3 * 1e3
            v.set_option("exact_ties", 0)
            xl_h, xd_h = xl[:chk].cpu().numpy(), xd[:chk].cpu().numpy()
            assert np.array_equal(xd_h.view(np.uint32), cd[:chk].view(np.uint32)), "bench parity (exact_ties): distances differ"
            differ = int((xl_h != cl[:chk]).any(axis=1).sum())
            assert differ == 0, "bench parity (exact_ties): labels differ for %d queries" % differ
            parity["exact_ties"] = {
                "checked_queries": chk, "labels_identical_slot_for_slot": True, "boundary_tie_queries": 0,
                "ms_per_step": round(x_ms, 4), "ms_per_step_default": round(ms_per_step, 4),
                "note": "option exact_ties = 1: the scan runs with k + 1, queries whose k + 1 smallest distances are "
                        "distinct are copied, the others replayed through the reference's heap in original row order "
                        "(vaq_exact.hip); plain equality of labels against oracle/vaq_oracle.c",
            }
        cpu["parity_checked_queries"] = parity["checked_queries"] if parity else 0
    v.close()
    del v
    torch.cuda.empty_cache()

    # ----------------------------------------- N = 1 default: the HBM leg --
    scale_base = None
    if world == 1 and args.workload == "auto" and not args.no_c5_leg and not args.ti:
        roofline, scale_base = c5_leg(args, dev, local_rank, k)
    if roofline is None and tm.get("bucket_major"):
        roofline = {"bound": "latency / memory side (bucket-major rounds: a bucket's rows are streamed once per group of four "
                             "queries and meet the other groups in the XCD's L2 or the Infinity Cache; no single-pass HBM "
                             "roofline applies -- at 1B rows the step moves 442 GB at the fabric's ~6.6 TB/s, on smaller "
                             "shards the kernel waits at 4 waves per SIMD; counters under profiles/)",
                    "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": kname, "kernel_ms": round(tm["scan_ms"], 4)}
        pm = load_profile_json(PROFILE_ROUND + "_bound.json").get("c5_bm")
        if pm and pm.get("rows") == n_local and pm.get("queries") == nq_local:
            roofline["counters"] = pm
            roofline["traffic"] = pm.get("memory_side_bytes_per_step")
    if roofline is None:
        roofline = {"bound": "lds+issue (cache-resident, bucket-pruned scan: no HBM roofline applies)",
                    "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None,
                    "kernel": kname, "kernel_ms": round(tm["scan_ms"], 4)}

    if rank == 0:
        out = {
            "metric": "queries/sec (ADC search, recall@100 reported alongside)",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if replicas else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": name, "rows": N, "rows_per_gpu": n_local if mode != "rows" else (N + world - 1) // world,
                       "queries_per_step": nq_total,
                       "queries_per_gpu": nq if (replicas or world == 1 or mode == "rows") else per_q, "k": k,
                       "bits": bits, "code_bytes": info["algo_code_bytes"],
                       "codes": "encoded" if real_codes else "uniform-random",
                       "method": ("EA_TI%dm%d visit=%g (k-means centres over decoded codes)" % (ti_T, ti_seg, args.visit))
                       if args.ti else "HEAP/EA (exhaustive)",
                       "sharding": "none" if world == 1 else (
                           "replicas: index replicated on every GPU, each GPU answers its own batch of queries, "
                           "results stay with their owner (no data-path collective)" if replicas else
                           "rows: contiguous shards, RCCL all-gather of per-shard top-k + merge" if mode == "rows"
                           else "queries: codes replicated (fit HBM), disjoint query slices, RCCL all-gather of results")},
            "roofline": roofline, "headline_kernel": headline_kernel, "cpu_baseline": cpu, "recall": recall,
            "parity": parity,
        }
        if exchange:
            out["exchange"] = exchange
        if scale_base:
            out["scale_base"] = scale_base
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
