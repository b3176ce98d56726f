#!/usr/bin/env python3
"""Experiment: how much of the C2 scan's fixed ~0.22 ms is stragglers (queries that cost several
times the mean, finishing last) and how much is pipeline fill?  Batches of IDENTICAL queries have
no stragglers: their time against the batch size gives the pure fill/drain intercept, and their
spread over different queries gives the per-query cost distribution."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from vaq_amd import harness

dev = torch.device("cuda", 0)
v, _, cents, _ = bench.build_index([8] * 8, 1_000_000, 0, 1_000_000, dev, 0, 1, 0, iters=15)
q = harness.sift_like(10_000, 128, stream=7, device=dev)

def scan_ms(qq, reps=10):
    v.set_option("timing", 0)
    for _ in range(3):
        v.search_device(qq, 100)
    torch.cuda.synchronize()
    v.set_option("timing", 1); v.last_timing()
    for _ in range(reps):
        v.search_device(qq, 100)
    torch.cuda.synchronize()
    return v.last_timing()["scan_ms"]

print("mixed batch: ", {n: round(scan_ms(q[:n].contiguous()), 4) for n in (1792, 2500, 5000, 10000)})
costs = []
for i in range(0, 10000, 250):
    one = q[i:i + 1].expand(10000, 128).contiguous()
    costs.append(scan_ms(one, reps=4))
costs = np.array(costs)
print("10k copies of one query, 40 different queries: min %.3f  median %.3f  mean %.3f  p90 %.3f  max %.3f ms" %
      (costs.min(), np.median(costs), costs.mean(), np.quantile(costs, 0.9), costs.max()))
i_med = int(np.argsort(costs)[len(costs) // 2]) * 250
one = q[i_med:i_med + 1]
print("identical (median-cost) query: ", {n: round(scan_ms(one.expand(n, 128).contiguous()), 4) for n in (1792, 2500, 5000, 10000, 16000)})

# can a cheap function of the lookup tables predict a query's cost (to schedule the long ones first)?
if len(sys.argv) > 1:
    eig = torch.from_numpy(v.mEigenVectors).to(dev)
    qp = q @ eig
    idx = list(range(0, 10000, 40))
    cost = np.array([scan_ms(q[i:i + 1].expand(4096, 128).contiguous(), reps=3) for i in idx])
    L = 16
    feats = {}
    luts = []
    for s in range(8):
        c = torch.from_numpy(cents[s]).to(dev)
        luts.append(((qp[idx][:, None, s * L:(s + 1) * L] - c[None]) ** 2).sum(-1))  # [n, 256]
    mins = torch.stack([l.min(1).values for l in luts], 1)
    feats["sum of per-table minima"] = mins.sum(1)
    feats["min of table 0"] = mins[:, 0]
    feats["sum of minima of tables 1.."] = mins[:, 1:].sum(1)
    feats["(sum of minima) / (mean of table 0)"] = mins.sum(1) / luts[0].mean(1)
    srt0 = luts[0].sort(1).values
    feats["table 0: 16th smallest - smallest"] = srt0[:, 15] - srt0[:, 0]
    feats["sum of minima / (16th smallest of table 0)"] = mins.sum(1) / srt0[:, 15]
    # rows whose bucket bound is below the sum of minima + a margin: a direct estimate of the work
    import scipy.stats as st
    for name, f in feats.items():
        f = f.cpu().numpy()
        print("%-48s spearman %.3f" % (name, st.spearmanr(f, cost).correlation))

    # LPT in practice: order ALL queries by a predicted cost, most expensive first, and time the batch
    luts_all = []
    for s in range(8):
        c = torch.from_numpy(cents[s]).to(dev)
        luts_all.append(((qp[:, None, s * L:(s + 1) * L] - c[None]) ** 2).sum(-1))
    mins_all = torch.stack([l.min(1).values for l in luts_all], 1)
    S = mins_all.sum(1)
    srt = luts_all[0].sort(1).values
    pred = {"sum of minima / 16th smallest of table 0": S / srt[:, 15],
            "-(16th smallest - smallest of table 0)": -(srt[:, 15] - srt[:, 0])}
    # rows in buckets that can hold a row within beta x (sum of minima): needs the bucket sizes
    codes0 = None
    print("as given:", round(scan_ms(q), 4))
    for name, f in pred.items():
        o = torch.argsort(f, descending=True)
        print("longest (predicted) first, by %-44s %.4f ms" % (name + ":", scan_ms(q[o].contiguous())))
        o = torch.argsort(f, descending=False)
        print("shortest first (control), by %-44s %.4f ms" % (name + ":", scan_ms(q[o].contiguous())))
    f = pred["sum of minima / 16th smallest of table 0"]
    srt_idx = torch.argsort(f, descending=True)
    for frac in (0.01, 0.03, 0.1, 0.25):
        n_top = int(10000 * frac)
        top = srt_idx[:n_top]
        mask = torch.ones(10000, dtype=torch.bool, device=dev); mask[top] = False
        rest = torch.arange(10000, device=dev)[mask]
        o = torch.cat([top[torch.randperm(n_top, device=dev)], rest])
        print("predicted top %4.0f%% first (shuffled), rest as given: %.4f ms" % (frac * 100, scan_ms(q[o].contiguous())))
        # and the opposite control: predicted top moved to the END
        o2 = torch.cat([rest, top])
        print("   control, predicted top %4.0f%% LAST:               %.4f ms" % (frac * 100, scan_ms(q[o2].contiguous())))
