#!/usr/bin/env python3
"""bench.py -- queries/sec of the VAQ ADC search path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5]

One "step" = one pass of the hot path (project -> LUT build -> code scan with
top-k -> merge [-> all-gather + merge across GPUs]) over one batch of nq
synthetic queries that are already resident in HBM, results left in HBM.

Workloads (BASELINE.json configs):
  c2  SIFT-1M-shaped, d=128, 8 subspaces x 256 centroids, 10k queries, k=100
      (the configuration the metric is quoted on; default)
  c3  same data, non-uniform bits {12,10,9,8,8,7,6,4}
  c5  1B x 128, 16 x 256 (uniform-random codes unless --encode; --rows scales it)
For N > 1 (launched by torch.distributed.run, one rank per GPU):
  --scaling weak (default for c2/c3, whose codes fit every GPU): the index is replicated,
      each GPU answers its OWN batch of nq queries, results stay on the GPU that owns the
      queries -- independent units, no data-path collective; value = N * nq * K / time.
  --scaling strong (default for c5): total work fixed.  --shard rows: the code rows are
      sharded contiguously across ranks, every rank answers all queries on its
      shard, one RCCL all-gather of the per-shard top-k plus a merge kernel finishes the step;
      --shard queries: codes replicated, the nq queries split across ranks, one all-gather.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5"])
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
    ap.add_argument("--ti", default="", help="T[,seg]: triangle-inequality form with T clusters over the first "
                                             "seg subspaces (default all), method EA_TI (not the headline metric)")
    ap.add_argument("--visit", type=float, default=1.0, help="--visit-cluster of demo_vaq (with --ti)")
    ap.add_argument("--no-skip", action="store_true", help="visit every bucket (streaming-rate measurement)")
    ap.add_argument("--bucket-bits", type=int, default=0, help="bits of the first code that key the row buckets (0 = auto)")
    ap.add_argument("--encode", action="store_true", help="c5: encode real vectors instead of random codes")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="N > 1: weak = every GPU its own nq queries on a replicated index (default c2/c3); "
                         "strong = the same nq queries and rows split over the GPUs (default c5)")
    ap.add_argument("--shard", default="auto", choices=["auto", "rows", "queries"],
                    help="multi-GPU: shard code rows (all-gather + merge), or replicate codes and shard queries")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (needs --backend gloo; RCCL refuses duplicate GPUs)")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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

    import vaq_amd
    from vaq_amd import build, harness, sharding
    from vaq_amd.index import merge_topk_device
    build.build_lib()

    D, k = 128, args.k
    if args.workload == "c2":
        bits, N, nq, name = [8] * 8, 1_000_000, 10_000, "sift1m-shaped d128 m8x256 nq10k k100"
    elif args.workload == "c3":
        bits, N, nq, name = list(harness.C3_BITS), 1_000_000, 10_000, "sift1m-shaped d128 bits{12,10,9,8,8,7,6,4} nq10k k100"
    else:
        bits, N, nq, name = [8] * 16, 1_000_000_000, 256, "synthetic 1Bx128 m16x256 k100"
    if args.rows:
        N = args.rows
    if args.nq:
        nq = args.nq
    M = len(bits)

    # ---------------------------------------------------------------- setup --
    t_setup = time.time()
    code_bytes_est = (sum(bits) + 7) // 8
    weak = world > 1 and (args.scaling == "weak" or (args.scaling == "auto" and args.workload != "c5"))
    mode = "queries" if weak else sharding.choose_mode(N, code_bytes_est, nq, world, args.shard)
    if mode == "rows":
        lo, hi = sharding.shard_bounds(N, world, rank)
    else:
        lo, hi = 0, N
    shard = hi - lo if mode == "queries" else (N + world - 1) // world
    n_local = hi - lo
    GEN = 1 << 20  # rows per generated chunk; chunk c is the same on every rank layout

    def base_chunk(c):
        m = min(GEN, N - c * GEN)
        return harness.sift_like(m, D, stream=1000 + c, device=dev)

    real_codes = args.workload != "c5" or args.encode
    # train on the first chunk (every rank derives identical state; rank 0's is broadcast)
    train = base_chunk(0)[: min(N, 262144)]
    eig = harness.pca_eigenvectors(train).to(dev)
    if world > 1:
        dist.broadcast(eig, 0)
    tp = train @ eig
    cents = harness.train_codebooks(tp, bits, iters=15 if args.workload != "c5" else 8)
    if world > 1:
        for s in range(M):
            t = torch.from_numpy(cents[s]).to(dev)
            dist.broadcast(t, 0)
            cents[s] = t.cpu().numpy()
    del tp

    v = vaq_amd.VaqHip(device=local_rank)
    v.parseMethodString("VAQ%dm%dmin%dmax%dvar1,HEAP" % (sum(bits), M, min(bits), max(bits)))
    v.mBitsAlloc = bits
    v.mCentroidsPerSubs = cents
    v.mEigenVectors = eig.cpu().numpy()
    v.id_base = lo

    codes = torch.empty((n_local, M), dtype=torch.int16, device=dev)
    if real_codes:
        c0, c1 = lo // GEN, (hi + GEN - 1) // GEN if hi > lo else lo // GEN
        for c in range(c0, c1):
            X = base_chunk(c)
            a, b = max(lo, c * GEN), min(hi, (c + 1) * GEN)
            xs = X[a - c * GEN: b - c * GEN]
            # the product's own encoder (VAQ::encode on the GPU; projects with eig first)
            codes[a - lo: b - lo] = v.encode_device(xs.contiguous(), projected=False)
            del X, xs
    else:
        g = torch.Generator(device=dev).manual_seed(harness.SEED + rank)
        step_rows = 1 << 24
        for r in range(0, n_local, step_rows):
            m = min(step_rows, n_local - r)
            codes[r: r + m] = torch.randint(0, 256, (m, M), generator=g, device=dev, dtype=torch.int16)
    ti_T = ti_seg = 0
    if args.ti:
        # VAQ::clusterTI: centres = k-means over decoded code rows (at most 256 per centre, as
        # KMeans::staticFitCodebook samples, KMeans.hpp:618-650); training, so it runs in the harness
        parts = [int(x) for x in args.ti.split(",")]
        ti_T, ti_seg = parts[0], (parts[1] if len(parts) > 1 else M)
        g = torch.Generator(device="cpu").manual_seed(harness.SEED)
        pick = torch.randperm(n_local, generator=g)[: min(n_local, 256 * ti_T)].to(dev)
        samp = codes[pick].to(torch.int64) & 0xffff
        L = D // M
        dec = torch.cat([torch.from_numpy(cents[s]).to(dev)[samp[:, s]] for s in range(ti_seg)], dim=1)
        cl = harness.kmeans(dec, ti_T, iters=25, seed=harness.SEED)
        if world > 1:
            dist.broadcast(cl, 0)
        v.mTIClusters = cl.cpu().numpy()
        v.mTISegmentNum = ti_seg
        v.mTIClusterNum = ti_T
        v.mMethods = vaq_amd.NNMethod.TI | vaq_amd.NNMethod.EA
        v.mVisit = args.visit
        del dec, samp, pick
    if args.bucket_bits:
        v._ensure_index()
        v.set_option("bucket_bits", args.bucket_bits)
    v.mCodebook = codes
    v._ensure_codes()
    host_codes = None
    if rank == 0 and not args.no_cpu and world == 1:
        n_cpu_rows = min(n_local, 1_000_000 if args.workload != "c5" else 4_000_000)
        host_codes = codes[:n_cpu_rows].cpu().numpy().view(np.uint16)
    del codes
    v.mCodebook = None  # packed copy lives in the index; keep _codes_sig
    torch.cuda.empty_cache()

    # weak scaling: every rank draws its own query batch (disjoint generator streams)
    queries = harness.sift_like(nq, D, stream=7 + (100 * rank if weak else 0), device=dev)
    q_lo, q_hi = (0, nq) if (mode == "rows" or weak) else sharding.shard_bounds(nq, world, rank)
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
    info = v.info()

    # every buffer of the steady-state loop is allocated once, here; with several ranks the
    # labels and distances of a rank travel in ONE all-gather (packed [2, n, k] int32 buffer)
    per_q = (nq + world - 1) // world
    n_pack = nq if (mode == "rows" or weak) else per_q
    pack_local, lab_view, dis_view = sharding.make_packed(n_pack, k, dev)
    out_local = (lab_view[:nq_local], dis_view[:nq_local])
    if nq_local < n_pack:  # short last query slice: the padding rows stay empty
        lab_view[nq_local:].fill_(-1)
        dis_view[nq_local:].fill_(3.4028234663852886e38)
    gathered = torch.empty((world, 2, n_pack, k), dtype=torch.int32, device=dev) if (world > 1 and not weak) else None
    out_final = (torch.empty((nq, k), dtype=torch.int32, device=dev),
                 torch.empty((nq, k), dtype=torch.float32, device=dev)) if (world > 1 and not weak) else None
    from vaq_amd.index import merge_topk_packed_device

    def run_step():
        l, d = v.search_device(my_queries, k, out=out_local)
        if world > 1 and not weak:
            sharding.all_gather_packed(pack_local, gathered)
            if mode == "rows":
                l, d = merge_topk_packed_device(gathered, world, nq, k, out=out_final)
            else:  # disjoint query slices: the gathered planes ARE the result (strided views)
                l = gathered[:, 0].reshape(world * per_q, k)[:nq]
                d = gathered[:, 1].reshape(world * per_q, k)[:nq].view(torch.float32)
        return l, d

    log(f"[rank {rank}] setup {time.time() - t_setup:.1f}s rows_local={n_local} nq={nq} info={info}")

    # ---------------------------------------------------------------- timed --
    # Settle (part of setup, untimed): the index build above frees several GB of
    # staging memory, and the driver reclaims it asynchronously -- a one-off
    # 40-60 ms stall of the GPU queue lands some milliseconds later
    # (tools/step_times.py shows it in hipDeviceSynchronize, with normal kernel
    # durations).  Run the step until 0.3 s have passed so it is not mistaken for
    # a step time; the W warmup steps and K timed steps follow as the contract says.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < 0.3:
        run_step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        run_step()
    torch.cuda.synchronize()
    v.set_option("timing", 1)
    v.last_timing()  # reset the event ring
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        labels, dists_ = run_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tm = v.last_timing()  # mean per-kernel device time over exactly these K steps (HIP events
    #                       recorded on the launch stream by the library)
    v.set_option("timing", 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    nq_total = nq * world if weak else nq
    qps = nq_total * args.steps / elapsed

    # ------------------------------------------------------------- roofline --
    # unit = one database row scanned in one pass; bytes = ceil(sum bits / 8);
    # one launch scans n_local rows in `passes` = ceil(nq / Qb) passes.
    # (per rank: this rank's rows x the passes its own queries need)
    algo_bytes = float(n_local) * info["algo_code_bytes"] * tm["passes"]
    achieved = algo_bytes / (tm["scan_ms"] * 1e-3) / 1e9 if tm["scan_ms"] > 0 else 0.0
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
        "kernel": "scan_%s%s_kernel" % ("bytes" if info["layout"] == 0 else "bits", "_ti" if args.ti else ""),
        "kernel_ms": round(tm["scan_ms"], 4), "launches_timed": tm["n_searches"],
        "queries_per_pass": tm["queries_per_pass"], "passes": tm["passes"],
        "algorithmic_bytes_per_launch": algo_bytes,
        "effective_per_query_GBps": round(float(n_local) * info["algo_code_bytes"] * nq_local /
                                          (tm["scan_ms"] * 1e-3) / 1e9, 1) if tm["scan_ms"] > 0 else 0.0,
        "other_kernels_ms": {"project": round(tm["project_ms"], 4), "lut_build": round(tm["lut_ms"], 4),
                             "threshold_seed": round(tm["seed_ms"], 4), "merge": round(tm["merge_ms"], 4)},
        "slices": tm["slices"], "workgroups": tm["workgroups"], "lds_bytes": tm["lds_bytes"],
    }

    # HBM-side traffic of the scan kernel comes from separate rocprofv3 --pmc passes of this
    # same command (tools/profile_gpu.sh); the corrected per-launch figure is committed under
    # profiles/ and attached here when it matches the workload and plan.
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json"))).get(args.workload)
        if tr and world == 1 and not args.rows and not args.nq and not args.ti \
                and tm["queries_per_pass"] == tr["queries_per_pass"]:
            roofline["traffic"] = tr["hbm_bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
    except (OSError, ValueError):
        pass

    # --------------------------------------------------------------- recall --
    recall = None
    if not args.no_recall and real_codes and N <= 4_000_000 and rank == 0:
        nq_r = min(nq, 1000)
        Xq = queries[:nq_r]
        nchunks = (N + GEN - 1) // GEN
        gt = harness.brute_force_topk(Xq, ((c * GEN, base_chunk(c)) for c in range(nchunks)), k)
        recall = {
            "recall_at_100": round(harness.avg_recall(labels[:nq_r].cpu().numpy(), gt.cpu().numpy()), 4),
            "recall_1nn_in_100": round(harness.recall_at_r(labels[:nq_r].cpu().numpy(), gt.cpu().numpy()), 4),
            "queries": nq_r, "ground_truth": "exact L2 brute force (torch) on the same synthetic base",
        }

    # --------------------------------------------------------- cpu baseline --
    cpu = None
    # (rank 0 at N = 1 only: with several ranks on the box the host cores are shared with their
    #  runtime threads and the other ranks would sit in a barrier for the duration)
    if rank == 0 and not args.no_cpu and world == 1:
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
        if world == 1 and n_rows_cpu == n_local and args.ti:
            # distances must agree exactly; labels may differ only inside runs of equal distance
            chk = min(n_cpu, 32)
            gd = dists_[:chk].cpu().numpy()
            assert np.array_equal(gd, cd[:chk]), "bench parity (TI): distances differ"
            gl = labels[:chk].cpu().numpy()
            for q in range(chk):
                for dv in np.unique(gd[q][:-1][gd[q][:-1] != gd[q][-1]]):
                    assert set(gl[q][gd[q] == dv]) == set(cl[q][cd[q] == dv]), "bench parity (TI): labels differ"
            cpu["parity_checked_queries"] = chk
        elif world == 1 and n_rows_cpu == n_local:
            # same inputs: the CPU port is also the parity checker for the bench's own result
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import assert_topk_matches
            chk = min(n_cpu, 32)
            Xp = po.project(qh[:chk], eig_h)
            ad = np.stack([po.all_dists(po.create_lut(Xp[q], cents, max(bits)), host_codes) for q in range(chk)])
            assert_topk_matches(labels[:chk].cpu().numpy(), dists_[:chk].cpu().numpy(), cl[:chk], cd[:chk], ad,
                                what="bench parity")
            cpu["parity_checked_queries"] = chk

    if rank == 0:
        out = {
            "metric": "queries/sec (ADC search, recall@100 reported alongside)",
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": name, "rows": N, "rows_per_gpu": shard, "queries_per_step": nq_total,
                       "queries_per_gpu": nq if (weak or world == 1) else (nq if mode == "rows" else per_q), "k": k,
                       "bits": bits, "code_bytes": info["algo_code_bytes"],
                       "codes": "encoded" if real_codes else "uniform-random",
                       "method": ("EA_TI%dm%d visit=%g (k-means centres over decoded codes)" % (ti_T, ti_seg, args.visit))
                       if args.ti else "HEAP/EA (exhaustive)",
                       "sharding": "none" if world == 1 else (
                           "weak: index replicated on every GPU, each GPU answers its own batch of queries, "
                           "results stay with their owner (no data-path collective)" if weak else
                           "rows: contiguous shards, RCCL all-gather of per-shard top-k + merge" if mode == "rows"
                           else "queries: codes replicated (fit HBM), disjoint query slices, RCCL all-gather of results")},
            "roofline": roofline, "cpu_baseline": cpu, "recall": recall,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
