#!/usr/bin/env python3
"""Where the time of bench.build_index goes (per 1M-row chunk): generate, ground truth, project+encode,
then set_codes (sort + pack)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from vaq_amd import harness
import vaq_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
dev = torch.device("cuda", 0)
bits = [8] * 16
train = bench.base_chunk(0, N, dev)[:262144]
eig = harness.pca_eigenvectors(train).to(dev)
cents = harness.train_codebooks(train @ eig, bits, iters=8)
v = vaq_amd.VaqHip()
v.mBitsAlloc = bits
v.mCentroidsPerSubs = cents
v.mEigenVectors = eig.cpu().numpy()
codes = torch.empty((N, 16), dtype=torch.int16, device=dev)
q = harness.sift_like(100, 128, stream=7, device=dev)
qq = (q * q).sum(1, keepdim=True)
t = {"gen": 0.0, "gt": 0.0, "encode": 0.0}
def sync():
    torch.cuda.synchronize()
    return time.perf_counter()
for c in range(N // bench.GEN):
    t0 = sync()
    X = bench.base_chunk(c, N, dev)
    t1 = sync()
    d = qq - 2.0 * q @ X.T + (X * X).sum(1).unsqueeze(0)
    dv, di = torch.topk(d, 100, dim=1, largest=False)
    t2 = sync()
    codes[c * bench.GEN:(c + 1) * bench.GEN] = v.encode_device(X.contiguous(), projected=False)
    t3 = sync()
    t["gen"] += t1 - t0; t["gt"] += t2 - t1; t["encode"] += t3 - t2
t0 = sync()
v.mCodebook = codes
v._ensure_codes()
t1 = sync()
t["set_codes"] = t1 - t0
print({k: round(x, 2) for k, x in t.items()}, "rows", N)
