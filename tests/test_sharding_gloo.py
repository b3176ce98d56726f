"""The N > 1 path on CPU: two gloo ranks, each answering all queries on its
contiguous row shard (with the CPU oracle standing in for the GPU scan), one
all-gather of per-shard top-k, merge by (distance, label) -- must equal the
single-index result.  Exercises vaq_amd.sharding exactly as bench.py uses it."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def cpu_merge(gd, gl, k):
    """CPU stand-in for vaqhip_merge_topk_device: k smallest by (distance, label);
    empty slots are -1 / FLT_MAX and sort last."""
    gd = gd.numpy()
    gl = gl.numpy()
    world, nq, _ = gd.shape
    out_l = np.full((nq, k), -1, np.int32)
    out_d = np.full((nq, k), np.finfo(np.float32).max, np.float32)
    for q in range(nq):
        d = gd[:, q, :].ravel()
        l = gl[:, q, :].ravel()
        ok = l >= 0
        d, l = d[ok], l[ok]
        order = np.lexsort((l, d))[:k]
        out_l[q, : len(order)] = l[order]
        out_d[q, : len(order)] = d[order]
    return torch.from_numpy(out_l), torch.from_numpy(out_d)


def _worker(rank, world, port, N, k, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_case
    from oracle import pyoracle as po
    from vaq_amd import sharding
    c = make_case(77, 32, [8] * 8, N, 5, dup_frac=0.05)
    lo, hi = sharding.shard_bounds(N, world, rank)
    Xp = po.project(c["X"], c["eig"])
    lab, dis = po.search(Xp, c["cents"], c["codes"][lo:hi], k, projected=True)
    lab = np.where(lab >= 0, lab + lo, -1).astype(np.int32)  # global labels, as id_base does
    ml, md = sharding.gather_and_merge(torch.from_numpy(lab), torch.from_numpy(dis), k, cpu_merge)
    if rank == 0:
        full_l, full_d = po.search(Xp, c["cents"], c["codes"], k, projected=True)
        ad = np.stack([po.all_dists(po.create_lut(Xp[q], c["cents"], 8), c["codes"]) for q in range(5)])
        from helpers import assert_topk_matches
        assert_topk_matches(ml.numpy(), md.numpy(), full_l, full_d, ad, what="gloo shards")
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N,k", [(3000, 50), (150, 100), (1, 10)])
def test_two_rank_shard_and_merge(oracle, N, k):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, k, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def _worker_queries(rank, world, port, nq, k, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_case
    from oracle import pyoracle as po
    from vaq_amd import sharding
    c = make_case(78, 32, [8] * 8, 2000, nq)
    lo, hi = sharding.shard_bounds(nq, world, rank)
    if hi > lo:
        lab, dis = po.search(c["X"][lo:hi], c["cents"], c["codes"], k, eig=c["eig"])
    else:
        lab, dis = np.empty((0, k), np.int32), np.empty((0, k), np.float32)
    gl, gd = sharding.gather_query_slices(torch.from_numpy(lab), torch.from_numpy(dis), nq)
    full_l, full_d = po.search(c["X"], c["cents"], c["codes"], k, eig=c["eig"])
    assert np.array_equal(gl.numpy(), full_l) and np.array_equal(gd.numpy(), full_d)
    if rank == 0:
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq", [7, 2, 1])
def test_two_rank_query_sharding(oracle, nq):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_queries, args=(r, 2, port, nq, 20, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def _worker_packed(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vaq_amd import sharding
    n, k = 5, 3
    buf, lab, dis = sharding.make_packed(n, k, "cpu")
    lab.copy_(torch.arange(n * k, dtype=torch.int32).view(n, k) + 100 * rank)
    dis.copy_(torch.arange(n * k, dtype=torch.float32).view(n, k) * 0.5 + rank)
    g = torch.empty((world, 2, n, k), dtype=torch.int32)
    sharding.all_gather_packed(buf, g)
    for r in range(world):
        assert torch.equal(g[r, 0], torch.arange(n * k, dtype=torch.int32).view(n, k) + 100 * r)
        assert torch.equal(g[r, 1].view(torch.float32), torch.arange(n * k, dtype=torch.float32).view(n, k) * 0.5 + r)
    if rank == 0:
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_packed_all_gather():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def test_choose_mode():
    from vaq_amd.sharding import choose_mode
    assert choose_mode(1_000_000, 8, 10_000, 8) == "queries"        # C2: 8 MB of codes, plenty of queries
    assert choose_mode(1_000_000, 8, 100, 8) == "rows"              # too few queries to split
    assert choose_mode(10**9, 16, 10_000, 8) == "queries"           # 16 GB still replicates on 288 GB parts
    assert choose_mode(4 * 10**9, 16, 10_000, 8) == "rows"          # 64 GB: shard the rows
    assert choose_mode(10**9, 16, 10_000, 8, "rows") == "rows"
    assert choose_mode(10**6, 8, 10_000, 1) == "rows"


def test_shard_bounds():
    from vaq_amd.sharding import shard_bounds
    for N in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(N, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi >= lo for lo, hi in spans)


def _settle_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import bench
    calls = [0]

    def step():  # a step with a collective in it, much slower on one rank than on the other
        time.sleep(0.002 if rank == 0 else 0.011)
        t = torch.ones(1)
        dist.all_reduce(t)
        calls[0] += 1

    bench.settle(step, seconds=0.1, world=world)
    ret[rank] = calls[0]
    dist.barrier()
    dist.destroy_process_group()


def test_bench_settle_runs_the_same_steps_on_every_rank():
    """bench.py's settle loop contains the all-gather of the N > 1 path: ranks that ended it by their
    own clocks ran it different numbers of times and one of them hung in the collective (seen in a
    2-rank rehearsal).  The count is now agreed on; both ranks must report the same number."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_settle_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0] == ret[1] and ret[0] >= 2, dict(ret)
