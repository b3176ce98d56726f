#!/usr/bin/env python3
"""Experiment: what would longest-first dispatch buy the C2 scan if a query's cost were known
exactly?  A VAQ_WGTIME build (VAQ_VARIANT=wgtime VAQ_EXTRA_FLAGS=-DVAQ_WGTIME) returns every
workgroup's lifetime in place of the k-th distance; the batch is then re-run with the queries in
descending order of that time, and with only the top p % moved to the front."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VAQHIP_LIB"] = os.path.join(ROOT, "vaq_amd/lib/variants/wgtime/libvaqhip.so")
import numpy as np, torch
import bench
from vaq_amd import harness

dev = torch.device("cuda", 0)
v, _, cents, _ = bench.build_index([8] * 8, 1_000_000, 0, 1_000_000, dev, 0, 1, 0, iters=15)
v.set_option("group_queries", 0)
v.set_option("cost_order", 0)  # (the library's own ranking off: block b serves query b of the batch as given)
v.set_option("defer_units", 0)
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

lab, dist = v.search_device(q, 100)
torch.cuda.synchronize()
cyc = dist.reshape(10000, 100)[:, 99].float().cpu().numpy()
print("workgroup lifetime (counter units): mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" %
      (cyc.mean(), np.median(cyc), np.quantile(cyc, 0.9), np.quantile(cyc, 0.99), cyc.max()))
D = dist.reshape(10000, 100).float().cpu().numpy()
names = ["steps", "drains", "flushes", "pool compactions", "eligible buckets", "rounds"]
import scipy.stats as st
for j, nm in enumerate(names):
    x = D[:, 98 - j]
    top = np.argsort(-cyc)[:100]
    print("%-18s mean %.1f  top-1%% by lifetime mean %.1f  spearman with lifetime %.3f" % (nm, x.mean(), x[top].mean(), st.spearmanr(x, cyc).correlation))
print("as given: %.4f ms" % scan_ms(q))
order = np.argsort(-cyc)
print("longest first (exact): %.4f ms" % scan_ms(q[torch.from_numpy(order.copy()).to(dev)].contiguous()))
print("shortest first: %.4f ms" % scan_ms(q[torch.from_numpy(order[::-1].copy()).to(dev)].contiguous()))
for pct in (1, 3, 10, 30):
    n = 10000 * pct // 100
    top = order[:n]
    rest = np.setdiff1d(np.arange(10000), top, assume_unique=False)
    o = np.concatenate([top, rest])
    print("top %d %% first: %.4f ms" % (pct, scan_ms(q[torch.from_numpy(o).to(dev)].contiguous())))
# interleaved: long and short alternate (every 8th workgroup is one of the longest)
rng = np.random.default_rng(1)
o = order.copy()
blocks = o.reshape(8, 1250)  # rows: cost octiles, longest first
print("octiles interleaved (one of each per 8 workgroups): %.4f ms" % scan_ms(q[torch.from_numpy(blocks.T.reshape(-1).copy()).to(dev)].contiguous()))

# keep the mix, but let no expensive query START late: expensive ones in the last part of the order
# trade places with cheap ones from the first part
for tail_frac, hard_q in ((0.3, 0.85), (0.3, 0.7), (0.5, 0.8), (0.2, 0.9)):
    o = np.arange(10000)
    cut = int(10000 * (1 - tail_frac))
    thr_hard = np.quantile(cyc, hard_q)
    late_hard = [i for i in range(cut, 10000) if cyc[i] > thr_hard]
    early_easy = [i for i in range(0, cut) if cyc[i] < np.median(cyc)]
    rng.shuffle(early_easy)
    for a, b in zip(late_hard, early_easy):
        o[a], o[b] = o[b], o[a]
    print("expensive (> p%d) out of the last %d %%: %d swaps, %.4f ms" %
          (int(hard_q * 100), int(tail_frac * 100), min(len(late_hard), len(early_easy)), scan_ms(q[torch.from_numpy(o).to(dev)].contiguous())))

# The kernel deals virtual ids to XCDs in contiguous ranges (xcd_virtual_id: XCD x serves queries
# [x G/8, (x+1) G/8) in that order), so a sorted order gives ONE XCD all the expensive queries.
# Longest-first done properly: rank r goes to XCD r % 8, position r // 8.
def xcd_place(order):
    # (block b serves query b of a single-slice best-first launch since the ranked dispatch went in;
    #  before, XCD x served the contiguous range [x G/8, (x+1) G/8) and this function dealt a
    #  ranking over the XCDs)
    return order
print("longest first, dealt over the XCDs: %.4f ms" % scan_ms(q[torch.from_numpy(xcd_place(order)).to(dev)].contiguous()))
print("shortest first, dealt over the XCDs: %.4f ms" % scan_ms(q[torch.from_numpy(xcd_place(order[::-1].copy())).to(dev)].contiguous()))
for pct in (1, 3, 10, 30):
    n = 10000 * pct // 100
    top = order[:n]
    rest = np.setdiff1d(np.arange(10000), top)
    rng.shuffle(rest)
    o = np.concatenate([top, rest])
    print("top %d %% first, dealt over the XCDs, rest shuffled: %.4f ms" % (pct, scan_ms(q[torch.from_numpy(xcd_place(o)).to(dev)].contiguous())))
o = np.arange(10000); rng.shuffle(o)
print("random shuffle: %.4f ms" % scan_ms(q[torch.from_numpy(o).to(dev)].contiguous()))
