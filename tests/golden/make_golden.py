"""Generate the committed golden vectors (run HERE, where /root/reference exists).

    python tests/golden/make_golden.py

Each case stores inputs (centroids, rotation, queries, uint16 codes) and
expected outputs:
  lut        CreateLUT restatement (oracle/vaq_oracle.c), cross-checked at
             generation time against the reference's own fma()
             (oracle/_ref: ref_lut_column_fma) and fvec_L2sqr_ny
  labels,    searchHeap restatement, cross-checked at generation time against
  dists      the REAL reference heap (utils/Heap.cpp HeapArray::addn/reorder,
             compiled in oracle/_ref) fed with the restated distance array
The script refuses to write a fixture whose cross-checks fail.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po  # noqa: E402
from helpers import make_case  # noqa: E402

CASES = [
    # name, seed, D, bits, N, nq, ks, kwargs
    ("d32_m8_b8", 1, 32, [8] * 8, 2000, 8, [1, 10, 100], {}),
    ("d128_m8_b8", 2, 128, [8] * 8, 5000, 8, [100], {"dup_frac": 0.02}),
    ("d128_m16_b8", 3, 128, [8] * 16, 4000, 8, [100], {}),
    ("d128_m8_nonuniform", 4, 128, [12, 10, 9, 8, 8, 7, 6, 4], 4000, 6, [100], {}),
    ("d16_m4_b3_ties", 5, 16, [3] * 4, 5000, 16, [100], {"integer": True}),
    ("d20_m4_smallk", 6, 20, [2, 1, 2, 3], 300, 4, [10], {}),
    ("d48_m4_smallk_l12", 7, 48, [2, 2, 1, 2], 200, 4, [5], {}),
    ("d128_m32_b78", 8, 128, [8, 7] * 16, 3000, 4, [100], {}),
    ("n_lt_k", 9, 32, [8] * 8, 37, 3, [100], {}),
]


def main():
    po.build(ref=True)
    assert po.have_ref(), "oracle/_ref is needed to cross-check the fixtures"
    manifest = {}
    for name, seed, D, bits, N, nq, ks, kw in CASES:
        c = make_case(seed, D, bits, N, nq, **kw)
        M, L = c["M"], c["L"]
        max_bits = max(bits)
        Xp = po.project(c["X"], c["eig"])
        luts = np.stack([po.create_lut(Xp[q], c["cents"], max_bits) for q in range(nq)])
        # cross-check LUT columns against the reference primitives
        for q in range(nq):
            for s in range(M):
                K = 1 << bits[s]
                qs = Xp[q, s * L:(s + 1) * L]
                if K >= 8:
                    r = po.ref_lut_column_fma(qs, c["cents"][s])
                else:
                    r = po.ref_l2sqr_ny(qs, c["cents"][s])
                assert np.array_equal(luts[q, s, :K].view(np.uint32), r.view(np.uint32)), (name, q, s)
                assert not luts[q, s, K:].any()
        out = {"X": c["X"], "eig": c["eig"], "Xproj": Xp, "codes": c["codes"], "lut": luts,
               "bits": np.array(bits, np.int32)}
        for s in range(M):
            out[f"cent{s}"] = c["cents"][s]
        boundary = 0
        for k in ks:
            labels, dists = po.search(Xp, c["cents"], c["codes"], k, max_bits=max_bits, projected=True)
            l_ea, d_ea = po.search(Xp, c["cents"], c["codes"], k, max_bits=max_bits, projected=True,
                                   method=po.METHOD_EA)
            assert np.array_equal(labels, l_ea) and np.array_equal(d_ea.view(np.uint32), dists.view(np.uint32))
            for q in range(nq):
                ad = po.all_dists(luts[q], c["codes"])
                rl, rd = po.ref_topk_from_dists(ad, k)
                assert np.array_equal(rl, labels[q]) and np.array_equal(rd.view(np.uint32), dists[q].view(np.uint32)), (name, k, q)
                if k < N:
                    srt = np.sort(ad)
                    boundary += int(srt[k - 1] == srt[k])
            out[f"labels_k{k}"] = labels
            out[f"dists_k{k}"] = dists
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        manifest[name] = {"seed": seed, "D": D, "bits": bits, "N": N, "nq": nq, "ks": ks,
                          "boundary_tie_queries": boundary, "bytes": os.path.getsize(path), **kw}
        print(name, manifest[name])
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
