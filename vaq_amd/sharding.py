"""Row sharding of the code database across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).  Rows are split into contiguous ranges,
every rank answers all queries on its shard with GLOBAL labels
(id_base = first row of the shard), and one all-gather of the per-shard
top-k followed by a k-min merge by (distance, label) yields, on every rank,
the same result a single index over all rows returns -- shards are
contiguous in label order and the single-GPU rule is also (distance, label).
The only collective on the path is that all-gather: nq*k*(4+4) bytes per rank.
"""
from __future__ import annotations

from typing import Callable, Tuple

# Row-shard when one GPU should not (or cannot) hold the whole code array, or
# when there are too few queries to give every GPU its own batch; otherwise
# replicate the (small) code array and split the QUERIES: an MI355X has 288 GB
# of HBM, so an 8 MB .. 16 GB code array is cheap to replicate, and query
# sharding needs no merge at all (one all-gather of disjoint result rows).
REPLICATE_MAX_BYTES = 32 << 30
MIN_QUERIES_PER_RANK = 64


def choose_mode(n_rows: int, code_bytes: int, nq: int, world: int, requested: str = "auto") -> str:
    """'rows' (contiguous row shards + all-gather + merge) or
    'queries' (replicated codes, disjoint query slices)."""
    if requested in ("rows", "queries"):
        return requested
    if world == 1:
        return "rows"
    fits = n_rows * code_bytes <= REPLICATE_MAX_BYTES
    enough = nq >= MIN_QUERIES_PER_RANK * world
    return "queries" if (fits and enough) else "rows"


# BASELINE.json configs the bench knows (bits per subspace, rows, queries per step)
BENCH_WORKLOADS = {
    "c2": dict(bits=[8] * 8, rows=1_000_000, nq=10_000, name="sift1m-shaped d128 m8x256 nq10k k100"),
    "c3": dict(bits=[12, 10, 9, 8, 8, 7, 6, 4], rows=1_000_000, nq=10_000,
               name="sift1m-shaped d128 bits{12,10,9,8,8,7,6,4} nq10k k100"),
    "c4": dict(bits=[8] * 8, rows=100_000_000, nq=10_000, name="synthetic 100Mx128 m8x256 nq10k k100"),
    "c5": dict(bits=[8] * 16, rows=1_000_000_000, nq=10_000, name="synthetic 1Bx128 m16x256 nq10k k100"),
}


def bench_plan(world: int, workload: str = "auto", scaling: str = "auto", shard: str = "auto",
               rows: int = 0, nq: int = 0) -> dict:
    """What `bench.py --gpus <world>` runs.  Defaults: one GPU -> c2 (the configuration the metric
    is quoted on); several GPUs -> the north-star path: c5, strong scaling, contiguous row
    shards + one RCCL all-gather + merge.  Replicas (every GPU its own queries on a replicated
    index, no collective) only when asked for by name."""
    if workload == "auto":
        workload = "c2" if world == 1 else "c5"
    w = dict(BENCH_WORKLOADS[workload])
    if rows:
        w["rows"] = rows
    if nq:
        w["nq"] = nq
    replicas = world > 1 and scaling in ("replicas", "weak")
    if replicas:
        mode = "queries"  # the whole index on every rank
    elif shard in ("rows", "queries"):
        mode = shard
    else:
        mode = "rows"
    return dict(workload=workload, rows=w["rows"], nq=w["nq"], bits=w["bits"], name=w["name"],
                replicas=replicas, mode=mode, scaling="weak" if replicas else "strong")


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of `rank`: ceil(N/world) rows per shard,
    the last shards may be short or empty."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    hi = min(n_rows, (rank + 1) * per)
    return lo, hi


def gather_and_merge(labels, dists, k: int, merge_fn: Callable, group=None, bufs=None):
    """labels/dists: this rank's [nq, k] result (global labels, empty slots
    -1 / FLT_MAX).  Returns the merged [nq, k] result (identical on all ranks).
    merge_fn(dist_lists[world, nq, k], label_lists[world, nq, k], k) -> (labels, dists)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    nq = labels.shape[0]
    # concatenation along dim 0 ([world*nq, k]) is the layout every backend accepts
    if bufs is not None:
        gl, gd = bufs  # caller-owned [world*nq, k] buffers: no allocation in the loop
    else:
        gl = torch.empty((world * nq, k), dtype=labels.dtype, device=labels.device)
        gd = torch.empty((world * nq, k), dtype=dists.dtype, device=dists.device)
    dist.all_gather_into_tensor(gl, labels.contiguous(), group=group)
    dist.all_gather_into_tensor(gd, dists.contiguous(), group=group)
    return merge_fn(gd.view(world, nq, k), gl.view(world, nq, k), k)


def gather_query_slices(labels, dists, nq_total: int, group=None, bufs=None):
    """Query sharding: this rank answered queries [lo, hi) = shard_bounds(nq_total,
    world, rank) against ALL rows.  All-gathers the disjoint slices into the full
    [nq_total, k] result on every rank (slices are padded to ceil(nq/world) rows)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = (nq_total + world - 1) // world
    k = labels.shape[1]
    if labels.shape[0] == per:
        pl, pd = labels.contiguous(), dists.contiguous()
    else:  # short last slice: pad to the common length
        pl = torch.full((per, k), -1, dtype=labels.dtype, device=labels.device)
        pd = torch.zeros((per, k), dtype=dists.dtype, device=dists.device)
        pl[: labels.shape[0]] = labels
        pd[: dists.shape[0]] = dists
    if bufs is not None:
        gl, gd = bufs  # caller-owned [world*per, k]
    else:
        gl = torch.empty((world * per, k), dtype=labels.dtype, device=labels.device)
        gd = torch.empty((world * per, k), dtype=dists.dtype, device=dists.device)
    dist.all_gather_into_tensor(gl, pl, group=group)
    dist.all_gather_into_tensor(gd, pd, group=group)
    return gl[:nq_total], gd[:nq_total]


# ---------------------------------------------------------------------------
# One-collective forms: labels and distances travel in ONE all-gather of a packed
# int32 buffer [2, n, k] per rank (plane 0 labels, plane 1 float32 bits).  At
# millisecond step times the second collective's launch latency is visible.
# ---------------------------------------------------------------------------
def make_packed(n: int, k: int, device):
    """Per-rank result buffer; search writes into .labels / .dists views of it."""
    import torch
    buf = torch.empty((2, n, k), dtype=torch.int32, device=device)
    return buf, buf[0], buf[1].view(torch.float32)


def all_gather_packed(packed_local, gathered, group=None):
    """gathered: int32 [world, 2, n, k] (caller-owned).  One collective."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(gathered.view(-1), packed_local.view(-1), group=group)
    return gathered


# ---------------------------------------------------------------------------
# Threshold exchange (round 3).  Every shard finds ITS k best, so its admission thresholds are
# looser than the global k-th distance allows.  With the staged search of the C ABI
# (vaqhip_search_begin_device / _finish_device) each rank runs the first rounds -- every query's
# nearest buckets of its shard --, ONE all-reduce(MIN) of nq int32 words (distance bit patterns:
# ordered like the distances) gives every rank the tightest bound any shard has found, and the rest
# of each shard is scanned under it.  A shard may then return fewer than k rows for a query; the
# merge of the gathered lists is unchanged and equals the single-index result bit for bit.
# ---------------------------------------------------------------------------
def staged_agreed(index, nq: int, k: int, group=None) -> bool:
    """All ranks must take the same path (there is a collective inside): MIN over the ranks of
    "my index can split this search"."""
    import torch
    import torch.distributed as dist
    ok = 1 if index.staged_supported(nq, k) else 0
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(ok)
    dev = torch.device("cuda", index.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def search_staged(index, d_queries, k: int, out, thr, group=None, host_bounce: bool = False):
    """begin -> all-reduce(MIN) of the thresholds -> finish.  thr: int32 [nq] CUDA scratch.
    host_bounce: the backend cannot reduce CUDA tensors (gloo rehearsals): go through the host."""
    import torch.distributed as dist
    index.search_begin_device(d_queries, k, out, thr)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if host_bounce:
            h = thr.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
            thr.copy_(h)
        else:
            dist.all_reduce(thr, op=dist.ReduceOp.MIN, group=group)
    index.search_finish_device(thr)
    return out
