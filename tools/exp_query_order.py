#!/usr/bin/env python3
"""Experiment: does ordering the queries by their nearest first-subspace centroid (so that
consecutive workgroups -- same XCD -- scan the same buckets) speed the C2 scan up?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from vaq_amd import harness

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
bits = [8] * 8 if wl == "c2" else list(harness.C3_BITS)
v, _, cents, _ = bench.build_index(bits, 1_000_000, 0, 1_000_000, dev, 0, 1, 0, iters=15)
q = harness.sift_like(10_000, 128, stream=7, device=dev)
eig = torch.from_numpy(v.mEigenVectors).to(dev)
qp = q @ eig
L = 128 // len(bits)
c0 = torch.from_numpy(cents[0]).to(dev)
d0 = ((qp[:, None, :L] - c0[None]) ** 2).sum(-1)
near = d0.argmin(1)
orders = {"as given": torch.arange(10_000, device=dev), "by nearest first code": torch.argsort(near),
          "random": torch.randperm(10_000, device=dev)}
# a smarter key: order the centroids of subspace 0 along their first coordinate, then by that rank
rank = torch.argsort(torch.argsort(c0[:, 0]))
orders["by rank of nearest centroid along dim 0"] = torch.argsort(rank[near])
orders["by first projected coordinate"] = torch.argsort(qp[:, 0])
for name, o in orders.items():
    qq = q[o].contiguous()
    v.set_option("timing", 0)
    for _ in range(5):
        v.search_device(qq, 100)
    torch.cuda.synchronize()
    v.set_option("timing", 1); v.last_timing()
    t = time.perf_counter()
    for _ in range(20):
        v.search_device(qq, 100)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 20 * 1e3
    tm = v.last_timing()
    print(f"{name:45s} scan {tm['scan_ms']:.4f} ms  step {wall:.4f} ms", flush=True)
