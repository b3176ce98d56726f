#!/usr/bin/env python3
"""Diagnostic: per-step wall time of the default workload (finds host-side stalls)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vaq_amd
from vaq_amd import harness
dev = torch.device("cuda", 0)
X = harness.sift_like(1_000_000, 128, stream=1000, device=dev)
eig = harness.pca_eigenvectors(X[:262144]).to(dev)
cents = harness.train_codebooks(X[:262144] @ eig, [8] * 8, iters=5)
v = vaq_amd.VaqHip()
v.mBitsAlloc = [8] * 8; v.mCentroidsPerSubs = cents; v.mEigenVectors = eig.cpu().numpy()
v.mCodebook = v.encode_device(X, projected=False); v._ensure_codes()
q = harness.sift_like(10000, 128, stream=7, device=dev)
timing = len(sys.argv) > 1
if timing:
    v.set_option("timing", 1)
out = (torch.empty((10000, 100), dtype=torch.int32, device=dev), torch.empty((10000, 100), dtype=torch.float32, device=dev))
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v.search_device(q, 100, out=out)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"step {i}: {1e3*(t1-t0):.3f} ms")
