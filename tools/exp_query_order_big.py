#!/usr/bin/env python3
"""Experiment: on a big streamed database (Qb = 4 queries per pass) does batching queries with
the same nearest first-subspace centroid shrink the union of buckets a pass must visit?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from vaq_amd import harness

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bits = [8] * 16
v, _, cents, _ = bench.build_index(bits, N, 0, N, dev, 0, 1, 0, iters=8)
q = harness.sift_like(nq, 128, stream=7, device=dev)
eig = torch.from_numpy(v.mEigenVectors).to(dev)
qp = q @ eig
L = 8
c0 = torch.from_numpy(cents[0]).to(dev)
near = ((qp[:, None, :L] - c0[None]) ** 2).sum(-1).argmin(1)
c1 = torch.from_numpy(cents[1]).to(dev)
near1 = ((qp[:, None, L:2 * L] - c1[None]) ** 2).sum(-1).argmin(1)
orders = {"as given": torch.arange(nq, device=dev), "by nearest first code": torch.argsort(near, stable=True),
          "by (first, second) nearest codes": torch.argsort(near * 256 + near1, stable=True)}
for name, o in orders.items():
    qq = q[o].contiguous()
    for qb in (2, 4):
        v.set_option("queries_per_pass", qb)
        v.set_option("timing", 0)
        v.search_device(qq, 100)
        torch.cuda.synchronize()
        v.set_option("timing", 1); v.last_timing()
        t = time.perf_counter()
        for _ in range(3):
            v.search_device(qq, 100)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t) / 3 * 1e3
        tm = v.last_timing()
        print(f"{name:36s} Qb={qb} scan {tm['scan_ms']:.2f} ms seed {tm['seed_ms']:.2f} step {wall:.2f} ms  ({nq / wall * 1e3:.0f} q/s)", flush=True)
