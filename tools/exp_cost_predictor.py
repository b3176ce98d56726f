#!/usr/bin/env python3
"""Experiment: which cheap function of a query's lookup tables predicts its scan cost well enough
to dispatch expensive queries first?  Exact costs come from a -DVAQ_WGTIME build
(VAQ_VARIANT=wgtime); candidates are rank-correlated with them and the batch is re-run in each
candidate's order (block b serves the b-th most expensive query: blocks are dealt round-robin over
the XCDs and dispatched in order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VAQHIP_LIB"] = os.path.join(ROOT, "vaq_amd/lib/variants/wgtime/libvaqhip.so")
import numpy as np, torch, scipy.stats as st
import bench
from vaq_amd import harness

dev = torch.device("cuda", 0)
v, _, cents, _ = bench.build_index([8] * 8, 1_000_000, 0, 1_000_000, dev, 0, 1, 0, iters=15)
v.set_option("group_queries", 0)
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

def xcd_place(order):
    G8 = len(order) // 8
    out = np.empty_like(order)
    r = np.arange(len(order))
    out[(r % 8) * G8 + r // 8] = order
    return out

lab, dist = v.search_device(q, 100)
torch.cuda.synchronize()
cyc = dist.reshape(10000, 100)[:, 99].float().cpu().numpy()
steps = dist.reshape(10000, 100)[:, 98].float().cpu().numpy()
print("as given %.4f ms; exact longest first %.4f ms" % (scan_ms(q), scan_ms(q[torch.from_numpy(xcd_place(np.argsort(-cyc))).to(dev)].contiguous())))
eig = torch.from_numpy(v.mEigenVectors).to(dev)
qp = q @ eig
L = 16
luts = []
for s in range(8):
    c = torch.from_numpy(cents[s]).to(dev)
    luts.append(((qp[:, None, s * L:(s + 1) * L] - c[None]) ** 2).sum(-1))  # [nq, 256]
srt = [l.sort(1).values for l in luts]
mins = torch.stack([x[:, 0] for x in srt], 1)
S = mins.sum(1)
feats = {"sum of minima / 16th smallest of table 0": S / srt[0][:, 15]}
for j in (8, 16, 32, 64):
    sp = srt[0][:, j - 1] - srt[0][:, 0]
    feats["-(table 0: %d-th smallest - smallest)" % j] = -sp
    feats["-(table 0: %d-th smallest - smallest) / sum of minima" % j] = -sp / S
    feats["-(table 0: %d-th smallest - smallest) / sqrt(sum of minima)" % j] = -sp / S.sqrt()
# second table too: the bucket key continues into it
sp01 = (srt[0][:, 31] - srt[0][:, 0]) + (srt[1][:, 31] - srt[1][:, 0])
feats["-(32nd - smallest of table 0 + the same of table 1)"] = -sp01
blk = cyc.reshape(8, 1250).mean(1)
print("mean lifetime of the 8 contiguous query ranges (one per XCD as given):", np.round(blk / 1000).astype(int), " corr(lifetime, index) %.3f" % np.corrcoef(cyc, np.arange(10000))[0, 1])
for name, f in feats.items():
    f = f.cpu().numpy()
    o = np.argsort(-f)
    t = scan_ms(q[torch.from_numpy(xcd_place(o)).to(dev)].contiguous())
    print("%-62s spearman: lifetime %.3f steps %.3f   batch in that order %.4f ms" % (name, st.spearmanr(f, cyc).correlation, st.spearmanr(f, steps).correlation, t))
