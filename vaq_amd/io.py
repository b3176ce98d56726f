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


def pack_rows_msb(codes: np.ndarray, bits) -> np.ndarray:
    """CodebookType (N x M uint16) -> the reference's BitVector-packed rows (N x W uint64): field s
    at bits [P_s, P_s + b_s) from the MSB of word 0, straddling fields split high part first
    (BitVecEngine.hpp:564-588; include/vaqhip_io.hpp:packRowsMSB is the C++ twin)."""
    codes = np.asarray(codes, dtype=np.uint64)
    n = codes.shape[0]
    W = (int(sum(bits)) + 63) // 64
    out = np.zeros((n, W), dtype=np.uint64)
    pos = 0
    for s, b in enumerate(bits):
        v = codes[:, s] & np.uint64((1 << b) - 1)
        w = pos // 64
        if w != (pos + b - 1) // 64:  # sliced
            right = b - ((w + 1) * 64 - pos)
            out[:, w] |= v >> np.uint64(right)
            out[:, w + 1] |= (v & np.uint64((1 << right) - 1)) << np.uint64(64 - right)
        else:
            out[:, w] |= v << np.uint64(64 - (pos % 64) - b)
        pos += b
    return out


def unpack_rows_msb(packed: np.ndarray, bits) -> np.ndarray:
    """Inverse of pack_rows_msb: N x W uint64 -> N x M uint16."""
    packed = np.asarray(packed, dtype=np.uint64)
    n = packed.shape[0]
    out = np.zeros((n, len(bits)), dtype=np.uint16)
    pos = 0
    for s, b in enumerate(bits):
        w = pos // 64
        if w != (pos + b - 1) // 64:
            right = b - ((w + 1) * 64 - pos)
            left = b - right
            v = ((packed[:, w] & np.uint64((1 << left) - 1)) << np.uint64(right)) | (packed[:, w + 1] >> np.uint64(64 - right))
        else:
            v = (packed[:, w] >> np.uint64(64 - (pos % 64) - b)) & np.uint64((1 << b) - 1)
        out[:, s] = v.astype(np.uint16)
        pos += b
    return out
