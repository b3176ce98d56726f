#!/usr/bin/env python3
"""Re-run a case saved by tools/fuzz_parity.py (gpurun_out/fuzz_fail_case.npz) under option sweeps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vaq_amd
z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "fuzz_fail_case.npz"))
bits = z["bits"].tolist(); M = len(bits)
cents = [z[f"cent{s}"] for s in range(M)]
eig = z["eig"] if z["eig"].size else None
o_dis = z["o_dis"]; nq, k = o_dis.shape
for bb in (0, 8, 9, 10):
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = bits; v.mCentroidsPerSubs = cents; v.mEigenVectors = eig; v.mCodebook = z["codes"]
    v._ensure_index()
    if bb: v.set_option("bucket_bits", bb)
    for sl in (0, 2, 5, 300):
        for bf in (0, 1):
            v.set_option("slices", sl); v.set_option("best_first", bf); v.set_option("early_abandon", 1)
            v.set_option("waves_per_workgroup", 4); v.set_option("timing", 1)
            bad = 0
            for rep in range(20):
                a = v.search(z["X"], k)
                bad += int(not np.array_equal(a.distances.reshape(nq, k), o_dis))
            t = v.last_timing()
            print(f"bucket_bits={bb} slices={sl}->{t['slices']} bf={bf}->{t['best_first']}: {bad}/20 runs differ", flush=True)
    v.close()
# which rows are lost, and where do they sit (sorted position / slice)?
if len(sys.argv) > 2:
    v = vaq_amd.VaqHip()
    v.mBitsAlloc = bits; v.mCentroidsPerSubs = cents; v.mEigenVectors = eig; v.mCodebook = z["codes"]
    v._ensure_index(); v.set_option("bucket_bits", 10); v.set_option("slices", 5); v.set_option("best_first", 1)
    v.set_option("early_abandon", 1); v.set_option("waves_per_workgroup", 4)
    o_lab = z["o_lab"]
    codes = z["codes"].astype(np.int64)
    key = (codes[:, 0] << 2) | (codes[:, 1] >> 6)
    order = np.argsort(key, kind="stable")          # the index's bucketed row order
    pos_of = np.empty_like(order); pos_of[order] = np.arange(order.size)
    for rep in range(3):
        a = v.search(z["X"], k)
        gl = a.labels.reshape(nq, k)
        for q in range(nq):
            lost = sorted(set(o_lab[q].tolist()) - set(gl[q].tolist()))
            if lost:
                print("rep", rep, "q", q, "lost labels", lost, "sorted pos", [int(pos_of[l]) for l in lost],
                      "slice(1024 rows)", [int(pos_of[l]) // 1024 for l in lost], "bucket", [int(key[l]) for l in lost],
                      "bucket rows", [int((key == key[l]).sum()) for l in lost])
