"""Shared test helpers: the tie contract and small synthetic indexes."""
from __future__ import annotations

import numpy as np


def make_case(seed, D, bits, N, nq, dup_frac=0.0, scale=30.0, rotate=True, integer=False):
    """Random index: gaussian centroids, queries, uint16 codes, optional rotation.
    dup_frac > 0 duplicates rows so exactly equal distances occur."""
    rng = np.random.default_rng(seed)
    M = len(bits)
    L = D // M
    if integer:  # small integers: all sums exact, massive ties
        cents = [rng.integers(-3, 4, size=(1 << b, L)).astype(np.float32) for b in bits]
        X = rng.integers(-3, 4, size=(nq, D)).astype(np.float32)
    else:
        cents = [(rng.normal(size=(1 << b, L)) * scale).astype(np.float32) for b in bits]
        X = (rng.normal(size=(nq, D)) * scale).astype(np.float32)
    codes = np.stack([rng.integers(0, 1 << b, size=N, dtype=np.int64) for b in bits], 1).astype(np.uint16)
    if dup_frac > 0 and N > 1:
        nd = int(N * dup_frac)
        src = rng.integers(0, N, nd)
        dst = rng.integers(0, N, nd)
        codes[dst] = codes[src]
    eig = None
    if rotate:
        q, _ = np.linalg.qr(rng.normal(size=(D, D)))
        eig = q.astype(np.float32)
    return dict(D=D, M=M, L=L, bits=list(bits), cents=cents, X=X, codes=codes, eig=eig)


def runs(d):
    """[(start, end)) runs of bit-equal values in a 1-D float array."""
    out = []
    s = 0
    for i in range(1, len(d) + 1):
        if i == len(d) or d[i] != d[s]:
            out.append((s, i))
            s = i
    return out


def assert_topk_matches(labels, dists, o_labels, o_dists, all_dists=None, id_base=0, what=""):
    """The parity contract (DESIGN.md "Ties"):
      * distances bit-exact, rank for rank (the k smallest distances are unique
        as a multiset whatever the tie order);
      * labels equal after sorting each run of bit-equal distances by label,
        except in a boundary tie (k-th and (k+1)-th distance equal), where the
        GPU must return the smallest labels carrying that distance;
      * unfilled slots are -1 / FLT_MAX on both sides.
    Returns the number of boundary-tie queries seen."""
    labels = np.asarray(labels)
    dists = np.asarray(dists, dtype=np.float32)
    o_labels = np.asarray(o_labels)
    o_dists = np.asarray(o_dists, dtype=np.float32)
    assert labels.shape == o_labels.shape and dists.shape == o_dists.shape, what
    assert np.array_equal(dists.view(np.uint32), o_dists.view(np.uint32)), \
        f"{what}: distances differ in {(dists.view(np.uint32) != o_dists.view(np.uint32)).sum()} slots"
    boundary = 0
    nq, k = labels.shape
    for q in range(nq):
        rs = runs(dists[q])
        for (a, b) in rs:
            g = labels[q, a:b]
            o = o_labels[q, a:b]
            assert np.all(np.diff(g) > 0) or b - a == 1 or np.all(g == -1), \
                f"{what}: q{q} labels not ascending inside tie run {a}:{b}: {g}"
            last = b == k
            if np.array_equal(np.sort(g), np.sort(o)):
                continue
            # only legal difference: boundary tie on the last run
            assert last and all_dists is not None, f"{what}: q{q} run {a}:{b} ids differ {g} vs {o}"
            cand = np.nonzero(all_dists[q] == dists[q, a])[0] + id_base
            assert len(cand) > (b - a), f"{what}: q{q} ids differ without a boundary tie"
            assert np.array_equal(g, np.sort(cand)[: b - a]), \
                f"{what}: q{q} boundary tie must keep the smallest labels"
            assert set(o.tolist()) <= set(cand.tolist())
            boundary += 1
    return boundary
