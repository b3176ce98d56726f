"""The reference's on-disk formats in NumPy (harness side; the C++ versions
are in include/vaqhip_io.hpp).

  saveCentroids / loadCentroids   utils/IO.hpp:736-754 / 522-549
  saveCodebook / loadCodebook     utils/IO.hpp:756-772 / 551-571
  fvecs / ivecs / bvecs           utils/IO.hpp:91-233, 334-361
`size_t` is 8 bytes little-endian (x86-64, the only platform the reference builds on).
"""
from __future__ import annotations

from typing import List

import numpy as np


def save_centroids(centroids: List[np.ndarray], path: str) -> None:
    with open(path, "wb") as f:
        f.write(np.uint64(len(centroids)).tobytes())
        for c in centroids:
            c = np.ascontiguousarray(c, dtype=np.float32)
            f.write(np.array(c.shape, dtype=np.uint64).tobytes())
            f.write(c.tobytes())


def load_centroids(path: str) -> List[np.ndarray]:
    buf = open(path, "rb").read()
    n = int(np.frombuffer(buf, np.uint64, 1, 0)[0])
    off = 8
    out = []
    for _ in range(n):
        r, c = (int(x) for x in np.frombuffer(buf, np.uint64, 2, off))
        off += 16
        out.append(np.frombuffer(buf, np.float32, r * c, off).reshape(r, c).copy())
        off += 4 * r * c
    return out


def save_codebook(codes: np.ndarray, path: str) -> None:
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    with open(path, "wb") as f:
        f.write(np.array(codes.shape, dtype=np.uint64).tobytes())
        f.write(codes.tobytes())


def load_codebook(path: str) -> np.ndarray:
    buf = open(path, "rb").read()
    r, c = (int(x) for x in np.frombuffer(buf, np.uint64, 2, 0))
    return np.frombuffer(buf, np.uint16, r * c, 16).reshape(r, c).copy()


def _read_vecs(path: str, dtype, max_rows: int = -1) -> np.ndarray:
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size == 0:
        return np.empty((0, 0), dtype)
    dim = int(raw[:4].view(np.int32)[0])
    rec = 4 + dim * np.dtype(dtype).itemsize
    n = raw.size // rec
    if max_rows >= 0:
        n = min(n, max_rows)
    rows = raw[: n * rec].reshape(n, rec)
    if not np.all(rows[:, :4].view(np.int32) == dim):
        raise ValueError("N and actual dimension mismatch")
    return rows[:, 4:].copy().view(dtype).reshape(n, dim)


def read_fvecs(path, max_rows=-1):
    return _read_vecs(path, np.float32, max_rows)


def read_ivecs(path, max_rows=-1):
    return _read_vecs(path, np.int32, max_rows)


def read_bvecs(path, max_rows=-1):
    return _read_vecs(path, np.uint8, max_rows).astype(np.float32)


def write_vecs(path: str, a: np.ndarray) -> None:
    a = np.ascontiguousarray(a)
    n, d = a.shape
    with open(path, "wb") as f:
        for i in range(n):
            f.write(np.int32(d).tobytes())
            f.write(a[i].tobytes())
