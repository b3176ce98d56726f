#!/usr/bin/env python3
"""Development bench of the bucket-major second pass (vaq_scan_bm.hip) on an ENCODED C5 / C4 cut:
builds the index once with bench.py's own recipe, then times the query batch with the pass off
and on (and over a few launch shapes), asserting identical results.

    python tools/bm_bench.py --rows 125000000 --nq 10000 [--m 16] [--sweep]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=125_000_000)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--random-codes", action="store_true")
    ap.add_argument("--skip-base", action="store_true", help="do not time the one-workgroup-per-query form")
    ap.add_argument("--units", default="", help="comma-separated bm_units values to time (instead of the default shape)")
    ap.add_argument("--rounds", default="", help="comma-separated bm_boot:bm_round pairs to time")
    ap.add_argument("--bucket-bits", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0, help="bm_waves of the default shape (0 = library default)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from vaq_amd import build, harness
    build.build_lib()
    bits = [8] * a.m
    t0 = time.time()
    v, _, _, _ = bench.build_index(bits, a.rows, 0, a.rows, dev, 0, 1, 0, iters=8, random_codes=a.random_codes,
                                   bucket_bits=a.bucket_bits)
    print(f"index of {a.rows} rows built in {time.time() - t0:.1f}s: {v.info()}", flush=True)
    queries = harness.sift_like(a.nq, 128, stream=7, device=dev)
    k = a.k
    out = (torch.empty((a.nq, k), dtype=torch.int32, device=dev), torch.empty((a.nq, k), dtype=torch.float32, device=dev))

    def timed(label, **opts):
        for key, val in opts.items():
            v.set_option(key, val)
        v.search_device(queries, k, out=out)
        torch.cuda.synchronize()
        v.set_option("timing", 1)
        v.last_timing()
        t = time.perf_counter()
        for _ in range(a.steps):
            v.search_device(queries, k, out=out)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / a.steps * 1e3
        tm = v.last_timing()
        v.set_option("timing", 0)
        rec = dict(label=label, opts=opts, wall_ms=round(wall, 3), scan_ms=round(tm["scan_ms"], 3), seed_ms=round(tm["seed_ms"], 3),
                   merge_ms=round(tm["merge_ms"], 3), bucket_major=tm["bucket_major"], slices=tm["slices"],
                   qps=round(a.nq / wall * 1e3, 1))
        print(json.dumps(rec), flush=True)
        return rec, out[0].clone(), out[1].clone()

    recs = []
    ref = None
    if not a.skip_base:
        r, l0, d0 = timed("one workgroup per query (bucket_major=0)", bucket_major=0)
        recs.append(r)
        ref = (l0, d0)
    shapes = [dict(bucket_major=2, bm_waves=a.waves)]
    if a.units:
        shapes = [dict(bucket_major=2, bm_boot=0, bm_units=int(u)) for u in a.units.split(",")]
    if a.rounds:  # "boot:round" pairs, e.g. 1:6,1:0,0:6
        shapes = [dict(bucket_major=2, bm_boot=2 * int(x.split(":")[0]), bm_round=int(x.split(":")[1])) for x in a.rounds.split(",")]
    if a.sweep:
        shapes += [dict(bucket_major=2, bm_units=u) for u in (8, 32, 128, 512)]
        shapes += [dict(bucket_major=2, bm_units=0, bm_queries_per_group=2), dict(bucket_major=2, bm_queries_per_group=4, bm_waves=8)]
    for s in shapes:
        r, l, d = timed("bucket-major", **s)
        recs.append(r)
        if ref is None:
            ref = (l, d)
        same = bool(torch.equal(l, ref[0]) and torch.equal(d, ref[1]))
        print("   identical to the reference form:", same, flush=True)
        assert same
    if a.out:
        json.dump(recs, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
