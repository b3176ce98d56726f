#!/usr/bin/env python3
"""What the threshold exchange between row shards buys: G shards of an encoded C5 cut as G indexes on ONE GPU,
searched one after the other -- every shard on its own, and staged (first rounds, MIN of the thresholds over
the shards, the rest).  Prints the summed device time of the shards' searches; results are asserted equal."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from vaq_amd import harness, sharding
from vaq_amd.index import merge_topk_packed_device

N = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nq, k = 10_000, 100
dev = torch.device("cuda", 0)
shards = []
for g in range(G):
    lo, hi = sharding.shard_bounds(N, G, g)
    v, _, _, _ = bench.build_index([8] * 16, N, lo, hi, dev, 0, 1, 0, iters=8)
    shards.append(v)
q = harness.sift_like(nq, 128, stream=7, device=dev)
packed = torch.empty((G, 2, nq, k), dtype=torch.int32, device=dev)
thr = torch.empty((G, nq), dtype=torch.int32, device=dev)

def plain():
    for g, v in enumerate(shards):
        v.search_device(q, k, out=(packed[g, 0], packed[g, 1].view(torch.float32)))

def staged(exchange=True):
    for g, v in enumerate(shards):
        v.search_begin_device(q, k, (packed[g, 0], packed[g, 1].view(torch.float32)), thr[g])
    t = thr.min(dim=0).values.contiguous() if exchange else None
    for g, v in enumerate(shards):
        v.search_finish_device(t)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ml, md = merge_topk_packed_device(packed, G, nq, k)
    torch.cuda.synchronize()
    return ms, ml.clone(), md.clone()

a = timed(plain)
b = timed(lambda: staged(False))
c = timed(lambda: staged(True))
assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[1], c[1]) and torch.equal(a[2], c[2])
print(f"{G} shards of {N // G} rows, {nq} queries: every shard on its own {a[0]:.2f} ms (sum over the shards), staged without "
      f"exchange {b[0]:.2f} ms, with the thresholds MIN-reduced over the shards {c[0]:.2f} ms")
