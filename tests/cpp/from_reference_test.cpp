// VaqHip::fromReference against a stand-in for the reference's `class VAQ` AFTER clusterTI():
// clusterTI leaves mCodebook regrouped by cluster (VAQ.cpp:984-996) while search() keeps
// returning ORIGINAL row numbers through mTIClustersMember (VAQ.cpp:1575-1590).  The adapter
// must therefore put the rows back in original order and copy the TI state, or its labels
// would be positions in the grouped matrix.  (The stand-in only has the members fromReference
// reads; it is not the reference's class.)
#include <algorithm>
#include <complex>
#include <cstdio>
#include <random>

#include "vaqhip.hpp"

using namespace vaqhip;

struct ComplexMat {
  size_t r = 0, c = 0;
  std::vector<std::complex<float>> d;
  size_t rows() const { return r; }
  size_t cols() const { return c; }
  const std::complex<float> &operator()(size_t i, size_t j) const { return d[i * c + j]; }
};

struct MockVAQ {
  int mBitBudget = 64, mSubspaceNum = 8, mMinBitsPerSubs = 8, mMaxBitsPerSubs = 8, mHighestSubs = 8;
  uint32_t mMethods = 0;
  std::vector<int> mBitsAlloc;
  std::vector<RowMatrixF> mCentroidsPerSubs;
  ComplexMat mEigenVectors;
  CodebookType mCodebook;
  int mTIClusterNum = 0, mTISegmentNum = -1;
  float mVisit = 1.0f;
  RowMatrixF mTIClusters;
  std::vector<std::vector<int>> mTIClustersMember;
};

int main() {
  const int M = 8, L = 4, D = M * L, N = 6000, nq = 16, k = 20, T = 12, seg = 4;
  std::mt19937 rng(77);
  std::normal_distribution<float> nd(0.f, 30.f);
  VaqHip a;
  a.mBitsAlloc.assign(M, 8);
  for (int s = 0; s < M; s++) {
    RowMatrixF c(256, L);
    for (size_t i = 0; i < 256; i++)
      for (int j = 0; j < L; j++) c(i, j) = nd(rng);
    a.mCentroidsPerSubs.push_back(c);
  }
  a.mCodebook = CodebookType(N, M);
  for (int i = 0; i < N; i++)
    for (int s = 0; s < M; s++) a.mCodebook(i, s) = (uint16_t)(rng() % 256);
  a.mTIClusterNum = T;
  a.mTISegmentNum = seg;
  a.mTIClusters = RowMatrixF(T, (size_t)seg * L);
  for (int t = 0; t < T; t++)
    for (int j = 0; j < seg * L; j++) a.mTIClusters(t, j) = nd(rng);
  a.mMethods = VaqHip::NNMethod::TI | VaqHip::NNMethod::EA;
  a.mVisit = 1.0f;
  RowMatrixF q(nq, D);
  for (int i = 0; i < nq; i++)
    for (int j = 0; j < D; j++) q(i, j) = nd(rng);
  LabelDistVecF ra = a.search(q, k);

  // the reference object after clusterTI: an arbitrary partition into T member lists in an
  // arbitrary order, and the codebook regrouped accordingly
  MockVAQ v;
  v.mMethods = a.mMethods;
  v.mBitsAlloc = a.mBitsAlloc;
  v.mCentroidsPerSubs = a.mCentroidsPerSubs;
  v.mEigenVectors.r = v.mEigenVectors.c = D;
  v.mEigenVectors.d.assign((size_t)D * D, {0.f, 0.f});
  for (int i = 0; i < D; i++) v.mEigenVectors.d[(size_t)i * D + i] = {1.f, 0.5f};  // (imaginary part must be ignored)
  v.mTIClusterNum = T;
  v.mTISegmentNum = seg;
  v.mVisit = 1.0f;
  v.mTIClusters = a.mTIClusters;
  v.mTIClustersMember.resize(T);
  std::vector<int> perm(N);
  for (int i = 0; i < N; i++) perm[i] = i;
  std::shuffle(perm.begin(), perm.end(), rng);
  for (int i = 0; i < N; i++) v.mTIClustersMember[rng() % T].push_back(perm[i]);
  v.mCodebook = CodebookType(N, M);
  size_t r = 0;
  for (const auto &cm : v.mTIClustersMember)
    for (const int idx : cm) {
      for (int s = 0; s < M; s++) v.mCodebook(r, s) = a.mCodebook((size_t)idx, s);
      r++;
    }
  VaqHip b;
  b.fromReference(v);
  if (b.mTIClusterNum != T || b.mTISegmentNum != seg || b.mTIClusters.rows() != (size_t)T) {
    std::printf("FAIL: TI state not copied\n");
    return 1;
  }
  LabelDistVecF rb = b.search(q, k);
  for (size_t i = 0; i < ra.labels.size(); i++)
    if (ra.labels[i] != rb.labels[i] || ra.distances[i] != rb.distances[i]) {
      std::printf("FAIL: slot %zu: %d %g vs %d %g\n", i, ra.labels[i], ra.distances[i], rb.labels[i], rb.distances[i]);
      return 1;
    }
  // TI selected but no clusters yet: must fail loudly, not search the wrong thing
  MockVAQ w = v;
  w.mTIClusters = RowMatrixF();
  w.mTIClustersMember.clear();
  bool threw = false;
  try {
    VaqHip c;
    c.fromReference(w);
  } catch (const Error &) {
    threw = true;
  }
  if (!threw) {
    std::printf("FAIL: TI without clusters accepted\n");
    return 1;
  }
  std::printf("from_reference ok: %d queries, labels are original rows\n", nq);
  return 0;
}
