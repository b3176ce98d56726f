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


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of `rank`: ceil(N/world) rows per shard,
    the last shards may be short or empty."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    hi = min(n_rows, (rank + 1) * per)
    return lo, hi


def gather_and_merge(labels, dists, k: int, merge_fn: Callable, group=None):
    """labels/dists: this rank's [nq, k] result (global labels, empty slots
    -1 / FLT_MAX).  Returns the merged [nq, k] result (identical on all ranks).
    merge_fn(dist_lists[world, nq, k], label_lists[world, nq, k], k) -> (labels, dists)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    nq = labels.shape[0]
    # concatenation along dim 0 ([world*nq, k]) is the layout every backend accepts
    gl = torch.empty((world * nq, k), dtype=labels.dtype, device=labels.device)
    gd = torch.empty((world * nq, k), dtype=dists.dtype, device=dists.device)
    dist.all_gather_into_tensor(gl, labels.contiguous(), group=group)
    dist.all_gather_into_tensor(gd, dists.contiguous(), group=group)
    return merge_fn(gd.view(world, nq, k), gl.view(world, nq, k), k)
