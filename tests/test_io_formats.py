"""On-disk formats against the reference's own IO code
compiled in oracle/_ref, and the C++ adapter headers compile with plain g++."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from vaq_amd import io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    cents = [rng.normal(size=(1 << b, 4)).astype(np.float32) for b in (8, 3, 12, 1)]
    codes = rng.integers(0, 65535, size=(1000, 4)).astype(np.uint16)
    io.save_centroids(cents, str(tmp_path / "c.bin"))
    io.save_codebook(codes, str(tmp_path / "cb.bin"))
    back = io.load_centroids(str(tmp_path / "c.bin"))
    assert all(np.array_equal(a, b) for a, b in zip(cents, back))
    assert np.array_equal(io.load_codebook(str(tmp_path / "cb.bin")), codes)
    v = rng.normal(size=(7, 5)).astype(np.float32)
    io.write_vecs(str(tmp_path / "v.fvecs"), v)
    assert np.array_equal(io.read_fvecs(str(tmp_path / "v.fvecs")), v)
    assert io.read_fvecs(str(tmp_path / "v.fvecs"), 3).shape == (3, 5)


def test_against_reference_io(oracle, tmp_path):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    r = oracle.ref()
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 4096, size=(321, 8)).astype(np.uint16)
    # reference writes, we read
    p = str(tmp_path / "ref_cb.bin").encode()
    r.ref_save_codebook(codes.ctypes.data_as(C.c_void_p), C.c_size_t(321), C.c_size_t(8), p)
    assert np.array_equal(io.load_codebook(p.decode()), codes)
    # we write, reference reads
    p2 = str(tmp_path / "our_cb.bin")
    io.save_codebook(codes, p2)
    out = np.zeros_like(codes)
    rows, cols = C.c_size_t(), C.c_size_t()
    assert r.ref_load_codebook(p2.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size),
                               C.byref(rows), C.byref(cols)) == 0
    assert (rows.value, cols.value) == (321, 8) and np.array_equal(out, codes)
    # centroids both ways
    cents = [rng.normal(size=(1 << b, 16)).astype(np.float32) for b in (8, 10, 4, 6)]
    arr = (C.POINTER(C.c_float) * 4)(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in cents])
    rws = (C.c_size_t * 4)(*[c.shape[0] for c in cents])
    cls = (C.c_size_t * 4)(*[c.shape[1] for c in cents])
    p3 = str(tmp_path / "ref_c.bin")
    r.ref_save_centroids(arr, rws, cls, C.c_size_t(4), p3.encode())
    back = io.load_centroids(p3)
    assert all(np.array_equal(a, b) for a, b in zip(cents, back))
    p4 = str(tmp_path / "our_c.bin")
    io.save_centroids(cents, p4)
    flat = np.zeros(sum(c.size for c in cents), np.float32)
    orow, ocol, nsub = (C.c_size_t * 8)(), (C.c_size_t * 8)(), C.c_size_t()
    assert r.ref_load_centroids(p4.encode(), flat.ctypes.data_as(C.c_void_p), C.c_size_t(flat.size),
                                orow, ocol, C.c_size_t(8), C.byref(nsub)) == 0
    assert nsub.value == 4 and np.array_equal(flat, np.concatenate([c.ravel() for c in cents]))


def test_reference_siftsmall_shape():
    """The query / ground-truth files the reference ships parse to the shapes
    its demo expects (scripts/run_demos.sh: 100 queries, d=128, k=100).  Read
    from /root/reference only when present (never on the GPU box)."""
    q = "/root/reference/data/siftsmall/siftsmall_query.fvecs"
    if not os.path.exists(q):
        pytest.skip("reference data not present")
    a = io.read_fvecs(q)
    g = io.read_ivecs("/root/reference/data/siftsmall/siftsmall_groundtruth.ivecs")
    assert a.shape == (100, 128) and g.shape == (100, 100)


def test_cpp_adapter_compiles(vaqlib, tmp_path):
    """include/vaqhip.hpp + vaqhip_io.hpp + examples/demo_vaqhip.cpp build with
    plain g++ and link against the C ABI only (no HIP, no Eigen, no torch)."""
    from vaq_amd import build
    exe = build.build_demo()
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "missing --centroids" in r.stderr


def test_bitvector_packed_rows_msb(tmp_path):
    """The reference's packed convention (packer BitVecEngine.hpp:564-588): MSB-first
    fields in 64-bit words, straddling fields split high part first.  Known-answer vectors worked
    out from that code by hand, the Python and C++ twins against each other, and the round trip."""
    # one field of 4 bits = 0xA at the top of word 0; then 12 bits = 0x123
    assert io.pack_rows_msb(np.array([[0xA, 0x123]]), [4, 12])[0, 0] == np.uint64(0xA123 << 48)
    # a field straddling words 0 and 1: 60 filler bits (4 x 15), then 8 bits = 0xAB -> 0xA | 0xB << 60
    p = io.pack_rows_msb(np.array([[0, 0, 0, 0, 0xAB]]), [15, 15, 15, 15, 8])
    assert p.shape == (1, 2) and p[0, 0] == np.uint64(0xA) and p[0, 1] == np.uint64(0xB << 60)
    rng = np.random.default_rng(3)
    for bits in ([8] * 8, [12, 10, 9, 8, 8, 7, 6, 4], [15, 1, 8, 8, 13, 13, 12, 2], [8] * 16, [13] * 9 + [1] * 3):
        codes = np.stack([rng.integers(0, 1 << b, size=257) for b in bits], 1).astype(np.uint16)
        packed = io.pack_rows_msb(codes, bits)
        assert packed.shape == (257, (sum(bits) + 63) // 64)
        assert np.array_equal(io.unpack_rows_msb(packed, bits), codes)
    # the C++ twin (include/vaqhip_io.hpp) produces the same words
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include <cstdio>
#include "vaqhip_io.hpp"
using namespace vaqhip;
int main(int argc, char **argv) {
  std::vector<int> bits = {15, 1, 8, 8, 13, 13, 12, 2};
  CodebookType cb = loadCodebook(argv[1]);
  std::vector<uint64_t> p = packRowsMSB(cb, bits);
  CodebookType back = unpackRowsMSB(p.data(), cb.rows(), bits);
  for (size_t i = 0; i < cb.rows(); i++)
    for (size_t j = 0; j < cb.cols(); j++)
      if (back(i, j) != cb(i, j)) return 3;
  FILE *f = std::fopen(argv[2], "wb");
  std::fwrite(p.data(), 8, p.size(), f);
  std::fclose(f);
  return 0;
}
""")
    bits = [15, 1, 8, 8, 13, 13, 12, 2]
    codes = np.stack([rng.integers(0, 1 << b, size=100) for b in bits], 1).astype(np.uint16)
    io.save_codebook(codes, str(tmp_path / "cb.bin"))
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    subprocess.check_call([exe, str(tmp_path / "cb.bin"), str(tmp_path / "p.bin")])
    cpp = np.fromfile(str(tmp_path / "p.bin"), dtype=np.uint64).reshape(100, 2)
    assert np.array_equal(cpp, io.pack_rows_msb(codes, bits))
