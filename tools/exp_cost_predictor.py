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
v.set_option("cost_order", 0)  # (the library's own ranking off: block b serves query b of the batch as given)
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
    # (block b serves query b of a single-slice best-first launch since the ranked dispatch went in;
    #  before, XCD x served the contiguous range [x G/8, (x+1) G/8) and this function dealt a
    #  ranking over the XCDs)
    return order

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
feats = {}
sp = {j: (srt[0][:, j - 1] - srt[0][:, 0]) for j in (4, 8, 16, 32, 64)}
feats["-(16th - smallest of table 0)  [the library's key]"] = -sp[16]
feats["-(8th + 16th + 32nd - 3 smallest)"] = -(sp[8] + sp[16] + sp[32])
feats["-(4th + 8th + 16th + 32nd + 64th - 5 smallest)"] = -(sp[4] + sp[8] + sp[16] + sp[32] + sp[64])
for m in (16, 32, 64, 128):
    feats["-(sum of the %d smallest of table 0 - %d x smallest)" % (m, m)] = -(srt[0][:, :m].sum(1) - m * srt[0][:, 0])
# entries of table 0 within d of the minimum, d = a fraction of the 16th-smallest spread of the OTHER tables summed
other = sum((srt[s_][:, 15] - srt[s_][:, 0]) for s_ in range(1, 8))
for frac in (0.4,):
    feats["table-0 entries within %.2f x (sum of the other tables' 16th-smallest spreads)" % frac] = ((luts[0] - srt[0][:, :1]) <= frac * other[:, None]).sum(1).float()
# least squares on log features against log lifetime (fit on even queries, ordered by the prediction on all)
import numpy as _np
F = torch.stack([sp[j].clamp_min(1e-3).log() for j in (4, 8, 16, 32, 64)] + [S.log(), other.log()], 1).cpu().numpy()
Fa = _np.concatenate([F, _np.ones((F.shape[0], 1))], 1)
w, *_ = _np.linalg.lstsq(Fa[::2], _np.log(cyc[::2]), rcond=None)
feats["least squares on log spreads (4..64), log sum of minima, log other spreads"] = torch.from_numpy(Fa @ w).float()
print("least-squares weights:", _np.round(w, 3))
for name, f in feats.items():
    f = f.cpu().numpy()
    o = np.argsort(-f)
    t = scan_ms(q[torch.from_numpy(xcd_place(o)).to(dev)].contiguous())
    print("%-62s spearman: lifetime %.3f steps %.3f   batch in that order %.4f ms" % (name, st.spearmanr(f, cyc).correlation, st.spearmanr(f, steps).correlation, t))
