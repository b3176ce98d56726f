// ref_harness.cpp -- thin extern "C" entry points over the REAL reference
// sources, compiled where they lie under /root/reference (never copied).
// TEST INFRASTRUCTURE ONLY; output goes to oracle/_ref/ (git-ignored).
//
// What the reference lets us build in this image (no glpk / armadillo / BLAS):
//   bitvecengine/utils/Heap.hpp + Heap.cpp   top-k heap (f::heap_*, HeapArray)
//   bitvecengine/utils/Math.hpp              fvec_L2sqr_ny (LUT fallback, K_s < 8)
//   bitvecengine/utils/AVXUtils.hpp          fma() used by CreateLUT
//   bitvecengine/utils/IO.hpp                save/load Centroids + Codebook, fvecs/ivecs
//   bitvecengine/utils/Experiment.hpp        getAvgRecall / getRecallAtR
//   external/eigen + utils/Types.hpp         the matrix types and the Eigen expressions of
//                                            VAQ::encodeImpl / ProjectOnEigenVectors (what Eigen's
//                                            reductions and GEMM actually sum)
// What it does not: bitvecengine/VAQ.{hpp,cpp} and BitVecEngine.hpp include
// glpk.h / armadillo, so VAQ::CreateLUT, VAQ::searchHeap and VAQ::encode are
// unbuildable here ("Oracle").
#include <cstdint>
#include <complex>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "BitVector.hpp"   // IO.hpp uses bitv/bitvectors without including it
#include "utils/Types.hpp"
#include "utils/AVXUtils.hpp"
#include "utils/Math.hpp"
#include "utils/IO.hpp"
#include "utils/Experiment.hpp"

extern "C" {

// HeapArray<CMax<float,int>>::heapify / addn / reorder (utils/Heap.cpp:6-41):
// addn's body is the insert rule of VAQ::searchHeap (VAQ.cpp:1750-1753).
void ref_topk_from_dists(const float *dist, int64_t n, int k, int *out_ids,
                         float *out_val) {
  f::float_maxheap_t res = {size_t(1), size_t(k), out_ids, out_val};
  res.heapify();
  res.addn(size_t(n), dist, 0);
  res.reorder();
}

// same with caller-supplied ids (Heap.cpp:43-70)
void ref_topk_from_dists_ids(const float *dist, const int *ids, int64_t n,
                             int k, int *out_ids, float *out_val) {
  f::float_maxheap_t res = {size_t(1), size_t(k), out_ids, out_val};
  res.heapify();
  res.addn_with_ids(size_t(n), dist, ids, 0);
  res.reorder();
}

// raw heap primitives, for step-by-step comparison
void ref_heap_heapify(size_t k, float *val, int *ids) {
  f::heap_heapify<f::CMax<float, int>>(k, val, ids);
}
void ref_heap_pop(size_t k, float *val, int *ids) {
  f::heap_pop<f::CMax<float, int>>(k, val, ids);
}
void ref_heap_push(size_t k, float *val, int *ids, float v, int id) {
  f::heap_push<f::CMax<float, int>>(k, val, ids, v, id);
}
size_t ref_heap_reorder(size_t k, float *val, int *ids) {
  return f::heap_reorder<f::CMax<float, int>>(k, val, ids);
}

// utils/Math.hpp:147-171
void ref_l2sqr_ny(float *dis, const float *x, const float *y, size_t d,
                  size_t ny) {
  fvec_L2sqr_ny(dis, x, y, d, ny);
}

// One LUT column the way VAQ.hpp:135-159 spells it, using the reference's own
// fma() (utils/AVXUtils.hpp:11-15) on 8-wide stripes of the column-major
// centroid matrix.  The loop is the harness's; the primitive is the
// reference's.  cent_cmajor: K x L column-major, 32-byte aligned, K % 8 == 0.
void ref_lut_column_fma(const float *qsub, const float *cent_cmajor, int K,
                        int L, float *out) {
  const int nstripes = K / 8;
  __m256 acc[(1 << 15) / 8]; // as VAQ.hpp:136 `accumulators[(1 << maxbit)/8]`, maxbit <= 15
  for (int i = 0; i < nstripes; i++) acc[i] = _mm256_setzero_ps();
  for (int j = 0; j < L; j++) {
    const float *cp = cent_cmajor + size_t(K) * j;
    __m256 qb = _mm256_set1_ps(qsub[j]);
    for (int i = 0; i < nstripes; i++) {
      __m256 col = _mm256_loadu_ps(cp);
      cp += 8;
      __m256 diff = _mm256_sub_ps(qb, col);
      acc[i] = fma(diff, diff, acc[i]);
    }
  }
  for (int i = 0; i < nstripes; i++) _mm256_storeu_ps(out + 8 * i, acc[i]);
}

// VAQ::encodeImpl's inner statements (VAQ.cpp:736-745) on the reference's own matrix type
// (RowMatrixXf, utils/Types.hpp:16) and its vendored Eigen: the block-difference-squaredNorm
// expression, the strict `<`, the uint16 code are spelled as there; the harness supplies the
// loops' bounds.  Pins what Eigen's (vectorised) reduction actually sums for a sub-vector of
// length L -- the source leaves that order to the library.  Column s of the codes for n rows.
void ref_encode_column(const float *Xproj, int64_t n, int D, int s, int L, const float *cent, int K,
                       uint16_t *codes_col) {
  Eigen::Map<const RowMatrixXf> XTrain(Xproj, n, D);
  Eigen::Map<const RowMatrixXf> centroids(cent, K, L);
  for (int64_t rowIdx = 0; rowIdx < n; rowIdx++) {
    uint16_t bestCode = 0;
    float bsf = std::numeric_limits<float>::max();
    for (uint16_t code = 0; code < static_cast<uint16_t>(K); code++) {
      float dist = (XTrain.block(rowIdx, s * L, 1, L) - centroids.block(code, 0, 1, L)).squaredNorm();
      if (dist < bsf) {
        bestCode = code;
        bsf = dist;
      }
    }
    codes_col[rowIdx] = bestCode;
  }
}
// the same expression's value for one (row, centroid) pair
float ref_encode_dist(const float *x, const float *c, int L) {
  Eigen::Map<const RowMatrixXf> X(x, 1, L);
  Eigen::Map<const RowMatrixXf> Cm(c, 1, L);
  return (X.block(0, 0, 1, L) - Cm.block(0, 0, 1, L)).squaredNorm();
}

// VAQ::ProjectOnEigenVectors (VAQ.hpp:198-201): (X * mEigenVectors).real() with the reference's
// complex eigenvector matrix (imaginary parts zero after train(), VAQ.cpp:14-100) -- Eigen's GEMM
// order, the one thing on the path the source does not define.
void ref_project(const float *X, int64_t n, int D, const float *eig_real, float *out) {
  Eigen::Map<const RowMatrixXf> Xm(X, n, D);
  Eigen::Matrix<std::complex<float>, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> E(D, D);
  for (int i = 0; i < D; i++)
    for (int j = 0; j < D; j++) E(i, j) = std::complex<float>(eig_real[(size_t)i * D + j], 0.0f);
  RowMatrixXf P = (Xm * E).real();
  std::memcpy(out, P.data(), sizeof(float) * (size_t)n * D);
}

// utils/IO.hpp:736-772 / 522-571
void ref_save_codebook(const uint16_t *codes, size_t rows, size_t cols,
                       const char *path) {
  CodebookType cb(rows, cols);
  std::memcpy(cb.data(), codes, rows * cols * sizeof(uint16_t));
  saveCodebook<CodebookType>(cb, path);
}
int ref_load_codebook(const char *path, uint16_t *out, size_t cap,
                      size_t *rows, size_t *cols) {
  CodebookType cb = loadCodebook<CodebookType>(path);
  *rows = cb.rows();
  *cols = cb.cols();
  if (size_t(cb.size()) > cap) return -1;
  std::memcpy(out, cb.data(), cb.size() * sizeof(uint16_t));
  return 0;
}
void ref_save_centroids(const float *const *cent, const size_t *rows,
                        const size_t *cols, size_t nsub, const char *path) {
  CentroidsPerSubsType c(nsub);
  for (size_t i = 0; i < nsub; i++) {
    c[i].resize(rows[i], cols[i]);
    std::memcpy(c[i].data(), cent[i], rows[i] * cols[i] * sizeof(float));
  }
  saveCentroids(c, path);
}
// flat output: for each subspace rows*cols floats, back to back
int ref_load_centroids(const char *path, float *out, size_t cap, size_t *rows,
                       size_t *cols, size_t max_sub, size_t *nsub) {
  CentroidsPerSubsType c = loadCentroids(path);
  *nsub = c.size();
  if (c.size() > max_sub) return -1;
  size_t off = 0;
  for (size_t i = 0; i < c.size(); i++) {
    rows[i] = c[i].rows();
    cols[i] = c[i].cols();
    if (off + size_t(c[i].size()) > cap) return -1;
    std::memcpy(out + off, c[i].data(), c[i].size() * sizeof(float));
    off += c[i].size();
  }
  return 0;
}

// utils/Experiment.hpp:252-271 and 288-303 (labels overloads)
double ref_avg_recall(const int *labels, int nq, int K, const int *topnn,
                      int stride) {
  std::vector<int> lab(labels, labels + size_t(nq) * K);
  std::vector<std::vector<int>> gt(nq);
  for (int q = 0; q < nq; q++)
    gt[q].assign(topnn + size_t(q) * stride, topnn + size_t(q) * stride + stride);
  return getAvgRecall<0>(lab, gt, K);
}
double ref_recall_at_r(const int *labels, int nq, int K, const int *topnn,
                       int stride) {
  std::vector<int> lab(labels, labels + size_t(nq) * K);
  std::vector<std::vector<int>> gt(nq);
  for (int q = 0; q < nq; q++)
    gt[q].assign(topnn + size_t(q) * stride, topnn + size_t(q) * stride + stride);
  return getRecallAtR<0>(lab, gt, K);
}

} // extern "C"
