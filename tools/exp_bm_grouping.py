#!/usr/bin/env python3
"""How much of the bucket-major pass's row work is shared between the queries of a group?  With the
FINAL thresholds of a batch: run (c0, c1) is alive for query q iff l0[c0] + l1[c1] <= thr_q; a group of
four queries streams the UNION of its members' alive runs.  Compares orders of a bucket's query list."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from vaq_amd import harness

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
nq, k, bt = 10_000, 100, 2
dev = torch.device("cuda", 0)
v, host, cents, _ = bench.build_index([8] * 16, N, 0, N, dev, 0, 1, 0, iters=8, keep_host_rows=N)
codes = torch.from_numpy(host.astype(np.int32)[:, :2].copy()).to(dev)
runs = torch.bincount(codes[:, 0] * 256 + codes[:, 1], minlength=65536).float().view(256, 256)
q = harness.sift_like(nq, 128, stream=7, device=dev)
l, d = v.search_device(q, k)
torch.cuda.synchronize()
thr = d[:, -1]
lut = torch.from_numpy(v.build_lut(q.cpu().numpy())).to(dev).view(nq, 16, 256)
l0, l1 = lut[:, 0], lut[:, 1]
tot_individual = 0.0
res = {}
G = 1 << bt
R = 256 >> bt
orders = {}
# per (query, group): nearest and second nearest second code of the group
l1g = l1.view(nq, G, R)
near = l1g.argsort(dim=2)[:, :, :2]
key_near2 = near[:, :, 0] * 256 + near[:, :, 1]
key_near1 = near[:, :, 0]
rng = torch.Generator(device="cpu").manual_seed(1)
key_rand = torch.randint(0, 1 << 20, (nq, G), generator=rng).to(dev)
# coordinate-based keys: the query's own coordinates in the second subspace are not available here; use
# the distances to two fixed reference codes of the group as a 2-D embedding
key_emb = (l1g[:, :, 0] / l1g[:, :, 0].max() * 1023).long() * 1024 + (l1g[:, :, 1] / l1g[:, :, 1].max() * 1023).long()
Xp = torch.from_numpy(v.project(q.cpu().numpy())).to(dev)[:, 8:16]   # the queries' coordinates in the second subspace
def quant(x, bits):
    lo, hi = x.min(), x.max()
    return ((x - lo) / (hi - lo) * ((1 << bits) - 1e-3)).long()
def morton(cols, bits):
    code = torch.zeros(nq, dtype=torch.long, device=dev)
    qs = [quant(Xp[:, c], bits) for c in cols]
    for b in range(bits - 1, -1, -1):
        for x in qs:
            code = (code << 1) | ((x >> b) & 1)
    return code
keys = [("random", key_rand), ("nearest code", key_near1), ("nearest + second nearest", key_near2)]
keys.append(("coordinate 0", quant(Xp[:, 0], 16)[:, None].expand(nq, G)))
for cols, bits in (((0, 1), 8), ((0, 1, 2), 6), ((0, 1, 2, 3), 5), ((0, 1, 2, 3, 4, 5), 3)):
    keys.append(("morton %s x %d bits" % (cols, bits), morton(cols, bits)[:, None].expand(nq, G)))
keys.append(("nearest code, then coordinate 0", key_near1 * 65536 + quant(Xp[:, 0], 16)[:, None]))
for name, key in keys:
    work_union = 0.0
    work_each = 0.0
    for c0 in range(256):
        for g in range(G):
            # queries whose bucket (c0, g) is in reach
            p01 = l0[:, c0:c0 + 1] + l1g[:, g, :]              # nq x R
            alive = p01 <= thr[:, None]
            inb = alive.any(dim=1)
            idx = inb.nonzero()[:, 0]
            if idx.numel() == 0:
                continue
            a = alive[idx]
            w = runs[c0, g * R:(g + 1) * R]
            order = key[idx, g].argsort()
            a = a[order]
            n = a.shape[0]
            pad = (-n) % 4
            if pad:
                a = torch.cat([a, torch.zeros((pad, R), dtype=torch.bool, device=dev)])
            u = a.view(-1, 4, R).any(dim=1)
            work_union += float((u.float() @ w).sum())
            work_each += float((a.float() @ w).sum())
    res[name] = (work_union, work_each)
    print(f"{name:28s} rows streamed per group-slot {work_union:.4e}  sum of members' own rows {work_each:.4e}  "
          f"-> {4 * work_union / work_each:.2f} x the ideal (1.0 = every streamed row wanted by all four)", flush=True)
