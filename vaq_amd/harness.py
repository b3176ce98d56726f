"""Harness utilities (NOT the hot path): synthetic SIFT-shaped data, the
PCA + per-subspace k-means that stands in for VAQ::train (which needs glpk /
armadillo and is out of scope), exact ground truth, recall.

Everything here runs on whatever torch device it is given (CPU in the build
container, the MI355X on the GPU box).  None of it is used by
vaqhip_search itself.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

SEED = 13517106  # the reference's fixed seed (utils/Random.hpp:18-28)

C3_BITS = [12, 10, 9, 8, 8, 7, 6, 4]  # "4-12 bits/subspace", 64-bit budget


def _rotation(d: int, seed: int, device) -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randn(d, d, generator=g, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)
    return q.to(torch.float32).to(device)


_SIFT_CONSTANTS = {}


def _sift_like_constants(d: int, seed: int, device: str, clusters: int):
    """(lam, R, C) of sift_like: the same for every chunk of a base, so they are made once -- the QR
    of the rotation and the centres are drawn on the CPU, and a 1B-row base is generated in a thousand
    chunks."""
    key = (d, seed, device, clusters)
    if key not in _SIFT_CONSTANTS:
        dev = torch.device(device)
        lam = torch.arange(1, d + 1, dtype=torch.float64) ** -1.2
        lam = (lam * d / lam.sum()).to(torch.float32).to(dev)
        R = _rotation(d, seed, dev)
        gc = torch.Generator(device="cpu").manual_seed(seed * 7 + 1)
        C = (torch.randn(clusters, d, generator=gc, dtype=torch.float32)).to(dev) * lam.sqrt()
        _SIFT_CONSTANTS[key] = (lam, R, C)
    return _SIFT_CONSTANTS[key]


def sift_like(n: int, d: int = 128, seed: int = SEED, stream: int = 0,
              device="cpu", chunk: int = 1 << 20, clusters: int = 1024,
              noise: float = 0.5) -> torch.Tensor:
    """'SIFT-shaped' vectors: d-dim, non-negative, integer-valued fp32 in
    [0, 255] (about a quarter of the entries are 0, as in SIFT), anisotropic
    and clustered: z = c_j + noise * e, with c_j one of `clusters` centres and
    e ~ N(0, diag(lam)), c_j ~ N(0, diag(lam)), lam_i ~ (i+1)^-1.2;
    x = clip(round(26 + 40 * R z), 0, 255) with a fixed random rotation R.
    The mixture gives nearest neighbours real structure (8 x 8-bit PQ reaches
    recall@100 of about 0.3 on 1M rows, like SIFT1M), which a single Gaussian
    does not."""
    device = torch.device(device)
    lam, R, C = _sift_like_constants(d, seed, str(device), clusters)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    g = torch.Generator(device=device).manual_seed(seed * 1000003 + stream)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        idx = torch.randint(0, clusters, (m,), generator=g, device=device)
        z = C[idx] + noise * torch.randn(m, d, generator=g, device=device, dtype=torch.float32) * lam.sqrt()
        x = 26.0 + 40.0 * (z @ R.T)
        out[s:s + m] = x.round_().clamp_(0, 255)
    return out


def pca_eigenvectors(X: torch.Tensor, max_rows: int = 131072) -> torch.Tensor:
    """Eigenvectors of X^T X on <= max_rows sampled rows, descending eigenvalue
    (the shape of VAQ.cpp:14-100).  Returns D x D float32 (columns = vectors)."""
    if X.shape[0] > max_rows:
        idx = torch.linspace(0, X.shape[0] - 1, max_rows, device=X.device).long()
        X = X[idx]
    Xd = X.to(torch.float64)
    cov = Xd.T @ Xd
    w, v = torch.linalg.eigh(cov.cpu())
    order = torch.argsort(w, descending=True)
    return v[:, order].to(torch.float32).contiguous()


def kmeans(X: torch.Tensor, K: int, iters: int = 25, seed: int = SEED) -> torch.Tensor:
    """Plain Lloyd k-means (stand-in for arma::kmeans static_subset, 25 iters,
    VAQ.cpp:528-631).  X: n x L on any device.  Returns K x L float32."""
    n = X.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed)
    if n >= K:
        init = torch.randperm(n, generator=g)[:K].to(X.device)
        C = X[init].clone()
    else:
        reps = (K + n - 1) // n
        C = X.repeat(reps, 1)[:K].clone()
        C += 1e-3 * torch.randn(C.shape, generator=g).to(X.device)
    xx = (X * X).sum(1, keepdim=True)
    for _ in range(iters):
        assign = torch.empty(n, dtype=torch.long, device=X.device)
        for s in range(0, n, 1 << 18):
            xs = X[s:s + (1 << 18)]
            d = xx[s:s + (1 << 18)] - 2.0 * xs @ C.T + (C * C).sum(1).unsqueeze(0)
            assign[s:s + (1 << 18)] = d.argmin(1)
        sums = torch.zeros_like(C).index_add_(0, assign, X)
        cnt = torch.zeros(K, device=X.device, dtype=X.dtype).index_add_(
            0, assign, torch.ones(n, device=X.device, dtype=X.dtype))
        nz = cnt > 0
        C[nz] = sums[nz] / cnt[nz].unsqueeze(1)
        if (~nz).any():  # re-seed empty clusters from random points
            ne = int((~nz).sum())
            C[~nz] = X[torch.randint(0, n, (ne,), generator=g).to(X.device)]
    return C.contiguous()


def train_codebooks(Xproj_sample: torch.Tensor, bits: Sequence[int], iters: int = 25,
                    seed: int = SEED, sample_per_centroid: int = 256) -> List[np.ndarray]:
    """Per-subspace k-means on projected training rows.  Returns the
    mCentroidsPerSubs list (K_s x L float32 numpy arrays)."""
    M = len(bits)
    D = Xproj_sample.shape[1]
    L = D // M
    cents = []
    for s, b in enumerate(bits):
        K = 1 << b
        n_use = min(Xproj_sample.shape[0], max(K * sample_per_centroid, 4096))
        xs = Xproj_sample[:n_use, s * L:(s + 1) * L].contiguous()
        cents.append(kmeans(xs, K, iters=iters, seed=seed + s).cpu().numpy().astype(np.float32))
    return cents


def encode_torch(Xproj: torch.Tensor, cents: Sequence[np.ndarray], chunk: int = 1 << 18) -> torch.Tensor:
    """Harness encoder (argmin of squared L2 per subspace; first minimum wins
    as in VAQ.cpp:728-748 up to float rounding of the distance form used here).
    Returns N x M int16 tensor holding uint16 codes."""
    n, D = Xproj.shape
    M = len(cents)
    L = D // M
    codes = torch.empty((n, M), dtype=torch.int16, device=Xproj.device)
    for s, c in enumerate(cents):
        C = torch.from_numpy(c).to(Xproj.device)
        cc = (C * C).sum(1).unsqueeze(0)
        for r in range(0, n, chunk):
            xs = Xproj[r:r + chunk, s * L:(s + 1) * L]
            d = cc - 2.0 * xs @ C.T
            codes[r:r + chunk, s] = d.argmin(1).to(torch.int16)
    return codes


def brute_force_topk(Xq: torch.Tensor, Xbase_chunks, k: int) -> torch.Tensor:
    """Exact L2 top-k ids.  Xbase_chunks yields (row_offset, tensor) pairs so
    large bases can be regenerated chunk-wise."""
    nq = Xq.shape[0]
    best_d = torch.full((nq, k), float("inf"), device=Xq.device)
    best_i = torch.full((nq, k), -1, dtype=torch.long, device=Xq.device)
    qq = (Xq * Xq).sum(1, keepdim=True)
    for off, B in Xbase_chunks:
        d = qq - 2.0 * Xq @ B.T + (B * B).sum(1).unsqueeze(0)
        kk = min(k, B.shape[0])
        dv, di = torch.topk(d, kk, dim=1, largest=False)
        cat_d = torch.cat([best_d, dv], 1)
        cat_i = torch.cat([best_i, di + off], 1)
        sel = torch.topk(cat_d, k, dim=1, largest=False)
        best_d = sel.values
        best_i = torch.gather(cat_i, 1, sel.indices)
    return best_i


def avg_recall(labels: np.ndarray, topnn: np.ndarray) -> float:
    """getAvgRecall (utils/Experiment.hpp:252-271): mean |returned ∩ true top-K| / K."""
    labels = np.asarray(labels)
    topnn = np.asarray(topnn)
    nq, K = labels.shape
    hit = 0
    for q in range(nq):
        hit += len(set(labels[q].tolist()) & set(topnn[q, :K].tolist()))
    return hit / (nq * K)


def recall_at_r(labels: np.ndarray, topnn: np.ndarray) -> float:
    """getRecallAtR (utils/Experiment.hpp:288-303): true 1-NN within the returned K."""
    labels = np.asarray(labels)
    topnn = np.asarray(topnn)
    return float(np.mean([topnn[q, 0] in set(labels[q].tolist()) for q in range(labels.shape[0])]))
