#!/bin/bash
# tools/big_sweep.sh <rows> <nq> "<args>" ... -- one encoded C5-shaped index, several option sets
# (bench.py rebuilds the index per invocation; this keeps it: options only change the plan)
cd "$GRAFT_REPO_ROOT"
rows=$1; nq=$2; shift 2
python3 - "$rows" "$nq" "$@" <<'PY'
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
from vaq_amd import harness
rows, nq = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
t = time.time()
v, _, _, _ = bench.build_index([8] * 16, rows, 0, rows, dev, 0, 1, 0, iters=8)
print("built %d rows in %.0f s" % (rows, time.time() - t), flush=True)
q = harness.sift_like(nq, 128, stream=7, device=dev)
ref = None
for spec in sys.argv[3:]:
    opts = dict(kv.split("=") for kv in spec.split()) if spec.strip() else {}
    for key in ("queries_per_pass", "slices", "best_first", "group_queries", "waves_per_workgroup"):
        v.set_option(key, int(opts.get(key, {"best_first": 1, "group_queries": 1}.get(key, 0))))
    v.set_option("timing", 0)
    l, d = v.search_device(q, 100); torch.cuda.synchronize()
    if ref is None: ref = (l.clone(), d.clone())
    assert torch.equal(l, ref[0]) and torch.equal(d, ref[1]), spec
    v.set_option("timing", 1); v.last_timing()
    t = time.perf_counter()
    for _ in range(2): v.search_device(q, 100)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 2 * 1e3
    tm = v.last_timing()
    print("%-44s step %9.2f ms  scan %9.2f  seed %6.2f merge %5.2f  qb %d slices %d bf %d  (%.0f q/s)" %
          (spec or "(defaults)", wall, tm["scan_ms"], tm["seed_ms"], tm["merge_ms"], tm["queries_per_pass"], tm["slices"],
           tm["best_first"], nq / wall * 1e3), flush=True)
PY
