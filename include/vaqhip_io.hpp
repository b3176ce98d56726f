// vaqhip_io.hpp -- readers/writers for the reference's on-disk formats, so an
// index written by `demo_vaq --save/--save-enc` loads here and vice versa.
//   saveCentroids / loadCentroids   utils/IO.hpp:736-754 / 522-549
//       size_t nsub; { size_t rows, cols; float[rows*cols] row-major } x nsub
//   saveCodebook / loadCodebook     utils/IO.hpp:756-772 / 551-571
//       size_t rows, cols; uint16_t[rows*cols] row-major
//   fvecs / ivecs / bvecs           utils/IO.hpp:91-233, 334-361
//       per vector: int32 dim; dim x {float | int32 | uint8}
//   writeKNNResults                 utils/IO.hpp:720-734 (labels as CSV, one query per line)
// Errors throw (the reference prints and carries on).
#ifndef VAQHIP_IO_HPP_
#define VAQHIP_IO_HPP_

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "vaqhip.hpp"

namespace vaqhip {

namespace detail {
struct File {
  FILE *f;
  File(const std::string &p, const char *mode) : f(std::fopen(p.c_str(), mode)) {
    if (!f) throw std::runtime_error("vaqhip_io: cannot open " + p);
  }
  ~File() { if (f) std::fclose(f); }
  void read(void *dst, size_t sz, size_t n) {
    if (n && std::fread(dst, sz, n, f) != n) throw std::runtime_error("vaqhip_io: short read");
  }
  void write(const void *src, size_t sz, size_t n) {
    if (n && std::fwrite(src, sz, n, f) != n) throw std::runtime_error("vaqhip_io: short write");
  }
};
} // namespace detail

inline void saveCentroids(const std::vector<RowMatrixF> &centroids, const std::string &path) {
  detail::File f(path, "wb");
  size_t dim = centroids.size();
  f.write(&dim, sizeof(size_t), 1);
  for (const RowMatrixF &c : centroids) {
    size_t row = c.rows(), col = c.cols();
    f.write(&row, sizeof(size_t), 1);
    f.write(&col, sizeof(size_t), 1);
    f.write(c.data(), sizeof(float), row * col);
  }
}

inline std::vector<RowMatrixF> loadCentroids(const std::string &path) {
  detail::File f(path, "rb");
  size_t dim = 0;
  f.read(&dim, sizeof(size_t), 1);
  if (dim > 4096) throw std::runtime_error("vaqhip_io: implausible subspace count");
  std::vector<RowMatrixF> out(dim);
  for (size_t i = 0; i < dim; i++) {
    size_t row = 0, col = 0;
    f.read(&row, sizeof(size_t), 1);
    f.read(&col, sizeof(size_t), 1);
    out[i] = RowMatrixF(row, col);
    f.read(out[i].data(), sizeof(float), row * col);
  }
  return out;
}

inline void saveCodebook(const CodebookType &cb, const std::string &path) {
  detail::File f(path, "wb");
  size_t row = cb.rows(), col = cb.cols();
  f.write(&row, sizeof(size_t), 1);
  f.write(&col, sizeof(size_t), 1);
  f.write(cb.data(), sizeof(uint16_t), row * col);
}

inline CodebookType loadCodebook(const std::string &path) {
  detail::File f(path, "rb");
  size_t row = 0, col = 0;
  f.read(&row, sizeof(size_t), 1);
  f.read(&col, sizeof(size_t), 1);
  CodebookType cb(row, col);
  f.read(cb.data(), sizeof(uint16_t), row * col);
  return cb;
}

// fvecs / ivecs / bvecs into a row-major matrix of `N + pad` columns (the
// reference zero-pads the dimension up to a multiple of the subspace count,
// demo_vaq.cpp:66-72)
template <typename Src, typename Dst>
inline RowMatrix<Dst> readVecs(const std::string &path, int N, int maxRow = -1, int padCols = 0) {
  detail::File f(path, "rb");
  std::vector<Dst> data;
  std::vector<Src> row((size_t)N);
  size_t rows = 0;
  for (;;) {
    int dim = 0;
    if (std::fread(&dim, sizeof(int), 1, f.f) != 1) break;
    if (dim != N) throw std::runtime_error("vaqhip_io: N and actual dimension mismatch");
    f.read(row.data(), sizeof(Src), (size_t)N);
    for (int j = 0; j < N; j++) data.push_back((Dst)row[j]);
    for (int j = 0; j < padCols; j++) data.push_back((Dst)0);
    rows++;
    if (maxRow != -1 && (int)rows >= maxRow) break;
  }
  RowMatrix<Dst> m;
  m.v.swap(data);
  m.r = rows;
  m.c = (size_t)(N + padCols);
  return m;
}
inline RowMatrixF readFVecs(const std::string &p, int N, int maxRow = -1, int pad = 0) {
  return readVecs<float, float>(p, N, maxRow, pad);
}
inline RowMatrixF readBVecs(const std::string &p, int N, int maxRow = -1, int pad = 0) {
  return readVecs<uint8_t, float>(p, N, maxRow, pad);
}
inline RowMatrix<int> readIVecs(const std::string &p, int N) { return readVecs<int, int>(p, N); }


// ---------------------------------------------------------------------------
// BitVector-packed rows in the reference's convention: `bitvectors` =
// rows of 64-bit words (BitVector.hpp:13-22), field s occupying bits [P_s, P_s + b_s) counted
// from the MOST significant bit of word 0, P_s = sum of the bits before it; in-word placement
// `code << (64 - (P_s % 64) - b_s)`, a field that straddles two words split high part first
// (the packer of BitVecEngine.hpp:564-588).  The index itself packs rows LSB-first in 32-bit
// words (a fixed permutation chosen for the GPU's field extraction) and takes
// `CodebookType`; these two functions convert between that matrix and the reference's packed
// form for callers that keep codes packed.  Words per row: actualBitVLen = (sum bits + 63) / 64.
// ---------------------------------------------------------------------------
inline size_t packedWordsPerRow(const std::vector<int> &bits) {
  size_t t = 0;
  for (int b : bits) t += (size_t)b;
  return (t + 63) / 64;
}

inline std::vector<uint64_t> packRowsMSB(const CodebookType &codes, const std::vector<int> &bits) {
  const size_t W = packedWordsPerRow(bits);
  std::vector<uint64_t> out(codes.rows() * W, 0ull);
  for (size_t i = 0; i < codes.rows(); i++) {
    size_t pos = 0;
    for (size_t s = 0; s < bits.size(); s++) {
      const int b = bits[s];
      const uint64_t bucket = (uint64_t)codes(i, s) & ((b >= 64) ? ~0ull : ((1ull << b) - 1ull));
      const size_t w = pos / 64;
      if (b > 0 && w != (pos + b - 1) / 64) {  // sliced
        const int right = b - (int)((w + 1) * 64 - pos);
        out[i * W + w] |= bucket >> right;
        out[i * W + w + 1] |= (bucket & ((1ull << right) - 1ull)) << (64 - right);
      } else if (b > 0) {
        out[i * W + w] |= bucket << (64 - (pos % 64) - b);
      }
      pos += (size_t)b;
    }
  }
  return out;
}

inline CodebookType unpackRowsMSB(const uint64_t *packed, size_t N, const std::vector<int> &bits) {
  const size_t W = packedWordsPerRow(bits);
  CodebookType codes(N, bits.size());
  for (size_t i = 0; i < N; i++) {
    size_t pos = 0;
    for (size_t s = 0; s < bits.size(); s++) {
      const int b = bits[s];
      const size_t w = pos / 64;
      uint64_t v = 0;
      if (b > 0 && w != (pos + b - 1) / 64) {
        const int right = b - (int)((w + 1) * 64 - pos);
        const int left = b - right;
        v = ((packed[i * W + w] & ((1ull << left) - 1ull)) << right) | (packed[i * W + w + 1] >> (64 - right));
      } else if (b > 0) {
        v = (packed[i * W + w] >> (64 - (pos % 64) - b)) & ((1ull << b) - 1ull);
      }
      codes(i, s) = (uint16_t)v;
      pos += (size_t)b;
    }
  }
  return codes;
}

inline void writeKNNResults(const std::string &path, const LabelDistVecF &results, size_t nrows) {
  const size_t k = nrows ? results.labels.size() / nrows : 0;
  std::ofstream out(path);
  for (size_t i = 0; i < nrows; i++) {
    for (size_t j = 0; j < k; j++) {
      out << results.labels[i * k + j];
      if (j != k - 1) out << ',';
    }
    out << std::endl;
  }
}

// getAvgRecall / getRecallAtR, utils/Experiment.hpp:252-271 / 288-303
inline double getAvgRecall(const std::vector<int> &labels, const RowMatrix<int> &topnn, int K) {
  const int nq = (int)(labels.size() / K);
  double ans = 0;
  for (int q = 0; q < nq; q++) {
    int ct = 0;
    for (int ki = 0; ki < K; ki++)
      for (int j = 0; j < K; j++)
        if (labels[(size_t)q * K + ki] == topnn(q, j)) { ct++; break; }
    ans += (double)ct / K;
  }
  return nq ? ans / nq : 0.0;
}
inline double getRecallAtR(const std::vector<int> &labels, const RowMatrix<int> &topnn, int K) {
  const int nq = (int)(labels.size() / K);
  double ans = 0;
  for (int q = 0; q < nq; q++)
    for (int ki = 0; ki < K; ki++)
      if (topnn(q, 0) == labels[(size_t)q * K + ki]) { ans += 1; break; }
  return nq ? ans / nq : 0.0;
}

} // namespace vaqhip
#endif
