import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from vaq_amd import harness
dev = torch.device("cuda", 0)
v, _, cents, _ = bench.build_index([8] * 8, 1_000_000, 0, 1_000_000, dev, 0, 1, 0, iters=15)
q = harness.sift_like(10_000, 128, stream=7, device=dev)
def scan_ms(qq, reps=10):
    v.set_option("timing", 0)
    for _ in range(3): v.search_device(qq, 100)
    torch.cuda.synchronize()
    v.set_option("timing", 1); v.last_timing()
    for _ in range(reps): v.search_device(qq, 100)
    torch.cuda.synchronize()
    return v.last_timing()["scan_ms"]
def xcd_place(order):
    G8 = len(order) // 8
    out = np.empty_like(order); r = np.arange(len(order))
    out[(r % 8) * G8 + r // 8] = order
    return out
v.set_option("cost_order", 0)
print("no ranking, as given (XCD x serves queries [1250 x, 1250 x + 1250)): %.4f" % scan_ms(q))
print("no ranking, block b serves query b: %.4f" % scan_ms(q[torch.from_numpy(xcd_place(np.arange(10000))).to(dev)].contiguous()))
rng = np.random.default_rng(3); o = np.arange(10000); rng.shuffle(o)
print("no ranking, random order: %.4f" % scan_ms(q[torch.from_numpy(o).to(dev)].contiguous()))
v.set_option("cost_order", 1)
print("ranking: %.4f" % scan_ms(q))
