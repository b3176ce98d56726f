// vaqhip.hpp -- header-only C++ adapter over the C ABI (vaqhip.h) with the
// reference's own names, so driver code written against `class VAQ`
// (bitvecengine/VAQ.hpp:36-113) swaps the type and keeps its calls:
//
//     VaqHip vaq;                                   // was: VAQ vaq;
//     vaq.parseMethodString(args["method"]);        // demo_vaq.cpp:59
//     vaq.mCentroidsPerSubs = ...; vaq.mBitsAlloc = ...; vaq.mCodebook = ...;
//     LabelDistVecF answers = vaq.search(queries, k, true);   // demo_vaq.cpp:339
//
// No Eigen dependency: matrices are passed as any type with data()/rows()/
// cols() in row-major float (Eigen's RowMatrixXf qualifies), or as the plain
// RowMatrixF below.  VaqHip::fromReference() copies the state out of a
// reference VAQ object by duck typing (compiled only where that class exists).
#ifndef VAQHIP_HPP_
#define VAQHIP_HPP_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "vaqhip.h"

namespace vaqhip {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
  if (rc < 0) throw Error(rc, std::string("vaqhip: ") + vaqhip_last_error());
}

// utils/Types.hpp:98-104
template <typename T = float> struct LabelDistVec {
  std::vector<int> labels;
  std::vector<T> distances;
};
using LabelDistVecF = LabelDistVec<>;

// minimal row-major matrix (stands in for RowMatrixXf / CodebookType)
template <typename T> struct RowMatrix {
  std::vector<T> v;
  size_t r = 0, c = 0;
  RowMatrix() = default;
  RowMatrix(size_t rows_, size_t cols_) : v(rows_ * cols_), r(rows_), c(cols_) {}
  T *data() { return v.data(); }
  const T *data() const { return v.data(); }
  size_t rows() const { return r; }
  size_t cols() const { return c; }
  T &operator()(size_t i, size_t j) { return v[i * c + j]; }
  const T &operator()(size_t i, size_t j) const { return v[i * c + j]; }
};
using RowMatrixF = RowMatrix<float>;
using CodebookType = RowMatrix<uint16_t>;  // utils/Types.hpp:31

class VaqHip {
public:
  struct NNMethod {  // VAQ.hpp:38-49
    enum { Sort = 0x01u, EA = 0x02u, TI = 0x04u, Fast = 0x08u, Fast2 = 0x10u, Fast3 = 0x20u,
           Fast4 = 0x40u, Heap = 0x80u };
  };

  // VAQ.hpp:51-75 (the members search() reads)
  int mBitBudget = 0, mSubspaceNum = 0;
  float mPercentVarExplained = 1.0f;
  int mMinBitsPerSubs = 0, mMaxBitsPerSubs = 0;
  uint32_t mMethods = NNMethod::Heap;
  RowMatrixF mEigenVectors;                  // real part of the reference's complex matrix; empty = identity
  std::vector<RowMatrixF> mCentroidsPerSubs;  // K_s x L each
  std::vector<int> mBitsAlloc;
  CodebookType mCodebook;                     // N x M uint16
  // VAQ.hpp:77-84: triangle-inequality clusters.  mTIClusters is T x (mTISegmentNum * mSubsLen);
  // the reference fills it in clusterTI() with a k-means over decoded codes -- here the caller
  // provides it (like the codebooks), or clusterTI(false) draws random decoded rows.
  int mTIClusterNum = 0, mTISegmentNum = -1;
  float mTIVariance = 1.0f;
  float mVisit = 1.0f;
  RowMatrixF mTIClusters;
  int64_t mIdBase = 0;                        // shard offset (not in the reference: single node)
  int mDevice = 0;
  // Several GPUs of the node (not in the reference, which is one host thread on one CPU): with
  // setDevices({0, 1, ..}) the code rows are sharded contiguously over them and search() ends
  // with the RCCL all-gather + merge of vaqhip_multi_search; results are unchanged.
  std::vector<int> mDevices;
  void setDevices(const std::vector<int> &devices) {
    invalidate();
    mDevices = devices;
    if (!devices.empty()) mDevice = devices[0];
  }

  VaqHip() = default;
  VaqHip(const VaqHip &) = delete;
  VaqHip &operator=(const VaqHip &) = delete;
  ~VaqHip() {
    vaqhip_index_destroy(h_);
    vaqhip_multi_destroy(mh_);
  }

  int mHighestSubs() const { return (int)mBitsAlloc.size(); }
  int mSubsLen() const { return mCentroidsPerSubs.empty() ? 0 : (int)mCentroidsPerSubs[0].cols(); }
  int mTotalDim() const { return mHighestSubs() * mSubsLen(); }
  uint32_t searchMethod() const { return mMethods; }  // VAQ.hpp:106-108

  // VAQ::parseMethodString, VAQ.cpp:1189-1267.  HEAP, EA and TI<T>[m<seg>] are this
  // path; other search tokens throw (the reference would run a different algorithm).
  void parseMethodString(const std::string &methodString) {
    std::stringstream ss(methodString);
    std::string token;
    while (std::getline(ss, token, ',')) {
      if (token.rfind("VAQ", 0) == 0) {
        int tb, sv, mn, mx;
        float var;
        if (std::sscanf(token.c_str(), "VAQ%dm%dmin%dmax%dvar%f", &tb, &sv, &mn, &mx, &var) == 5) {
          mBitBudget = tb; mSubspaceNum = sv; mMinBitsPerSubs = mn; mMaxBitsPerSubs = mx;
          mPercentVarExplained = var;
        }
      } else if (token.find("SORT") != std::string::npos || token.find("HEAP") != std::string::npos ||
                 token.find("EA") != std::string::npos || token.find("TI") != std::string::npos ||
                 token.find("FAST") != std::string::npos) {
        uint32_t m = 0;
        std::stringstream sm(token);
        std::string t;
        while (std::getline(sm, t, '_')) {
          if (t.find("SORT") != std::string::npos) m |= NNMethod::Sort;
          else if (t.find("HEAP") != std::string::npos) m |= NNMethod::Heap;
          else if (t.find("EA") != std::string::npos) m |= NNMethod::EA;
          else if (t.find("TI") != std::string::npos) {  // VAQ.cpp:1236-1251
            unsigned long cluster = 0, segment = 0;
            float minvar = 1.0f;
            if (std::sscanf(t.c_str(), "TI%luvar%f", &cluster, &minvar) == 2) {
              m |= NNMethod::TI; mTIClusterNum = (int)cluster; mTIVariance = minvar;
            } else if (std::sscanf(t.c_str(), "TI%lum%lu", &cluster, &segment) == 2) {
              m |= NNMethod::TI; mTIClusterNum = (int)cluster; mTISegmentNum = (int)segment;
            } else if (std::sscanf(t.c_str(), "TI%lu", &cluster) == 1) {
              m |= NNMethod::TI; mTIClusterNum = (int)cluster;
            }
          }
          else if (t.find("FAST3") != std::string::npos) m |= NNMethod::Fast3;
          else if (t.find("FAST2") != std::string::npos) m |= NNMethod::Fast2;
          else if (t.find("FAST") != std::string::npos) m |= NNMethod::Fast;
        }
        if (m & ~(uint32_t)(NNMethod::Heap | NNMethod::EA | NNMethod::TI))
          throw Error(VAQHIP_EUNSUPPORTED, "vaqhip: method '" + token + "' is outside the HEAP/EA/TI path");
        mMethods = m;
      }
    }
  }

  // Push the public members to the GPU (called lazily by search()).
  void sync() {
    const int M = mHighestSubs();
    if (M == 0 || (int)mCentroidsPerSubs.size() != M) throw Error(VAQHIP_EINVAL, "vaqhip: state not set");
    if (!mDevices.empty()) {
      syncMulti();
      return;
    }
    if (!h_) {
      std::vector<const float *> cp(M);
      for (int s = 0; s < M; s++) {
        if ((int)mCentroidsPerSubs[s].rows() != (1 << mBitsAlloc[s]))
          throw Error(VAQHIP_EINVAL, "vaqhip: centroid rows != 1 << bits");
        cp[s] = mCentroidsPerSubs[s].data();
      }
      check(vaqhip_index_create(&h_, mTotalDim(), M, mBitsAlloc.data(), cp.data(),
                                mEigenVectors.rows() ? mEigenVectors.data() : nullptr, mDevice));
      codes_set_ = false;
      ti_set_ = false;
    }
    if ((mMethods & NNMethod::TI) && !ti_set_) {  // before the codes: they are then grouped once
      const int seg = mTISegmentNum == -1 ? M : mTISegmentNum;  // VAQ.cpp:890-892
      if (mTIClusters.rows() == 0 || (int)mTIClusters.cols() != seg * mSubsLen())
        throw Error(VAQHIP_ESTATE, "vaqhip: method TI needs mTIClusters (T x seg*L); see clusterTI()");
      check(vaqhip_index_set_ti_clusters(h_, mTIClusters.data(), (int)mTIClusters.rows(), seg));
      ti_set_ = true;
    } else if (!(mMethods & NNMethod::TI) && ti_set_) {
      check(vaqhip_index_set_ti_clusters(h_, nullptr, 0, 0));
      ti_set_ = false;
    }
    check(vaqhip_index_set_method(h_, mMethods, mVisit));
    if (!codes_set_) {
      if (mCodebook.cols() != (size_t)M && mCodebook.rows() != 0)
        throw Error(VAQHIP_EINVAL, "vaqhip: mCodebook is not N x M");
      check(vaqhip_index_set_codes_u16(h_, mCodebook.data(), (int64_t)mCodebook.rows(), mIdBase));
      codes_set_ = true;
    }
  }
  // the same for the multi-device index (mDevices set)
  void syncMulti() {
    const int M = mHighestSubs();
    if (!mh_) {
      std::vector<const float *> cp(M);
      for (int s = 0; s < M; s++) {
        if ((int)mCentroidsPerSubs[s].rows() != (1 << mBitsAlloc[s]))
          throw Error(VAQHIP_EINVAL, "vaqhip: centroid rows != 1 << bits");
        cp[s] = mCentroidsPerSubs[s].data();
      }
      checkMulti(vaqhip_multi_create(&mh_, mTotalDim(), M, mBitsAlloc.data(), cp.data(),
                                     mEigenVectors.rows() ? mEigenVectors.data() : nullptr, (int)mDevices.size(),
                                     mDevices.data(), 0u));
      codes_set_ = false;
      ti_set_ = false;
    }
    if ((mMethods & NNMethod::TI) && !ti_set_) {
      const int seg = mTISegmentNum == -1 ? M : mTISegmentNum;
      if (mTIClusters.rows() == 0 || (int)mTIClusters.cols() != seg * mSubsLen())
        throw Error(VAQHIP_ESTATE, "vaqhip: method TI needs mTIClusters (T x seg*L); see clusterTI()");
      checkMulti(vaqhip_multi_set_ti_clusters(mh_, mTIClusters.data(), (int)mTIClusters.rows(), seg));
      ti_set_ = true;
    } else if (!(mMethods & NNMethod::TI) && ti_set_) {
      checkMulti(vaqhip_multi_set_ti_clusters(mh_, nullptr, 0, 0));
      ti_set_ = false;
    }
    checkMulti(vaqhip_multi_set_method(mh_, mMethods, mVisit));
    if (!codes_set_) {
      if (mCodebook.cols() != (size_t)M && mCodebook.rows() != 0)
        throw Error(VAQHIP_EINVAL, "vaqhip: mCodebook is not N x M");
      checkMulti(vaqhip_multi_set_codes_u16(mh_, mCodebook.data(), (int64_t)mCodebook.rows(), mIdBase));
      codes_set_ = true;
    }
  }
  // call after changing any public member
  void invalidate() {
    vaqhip_multi_destroy(mh_);
    mh_ = nullptr;
    vaqhip_index_destroy(h_);
    h_ = nullptr;
    codes_set_ = false;
    ti_set_ = false;
  }

  // VAQ::clusterTI, VAQ.hpp:106 / VAQ.cpp:878-999.  useKMeans = false is the reference's
  // other branch (:901-911): mTIClusterNum random code rows, decoded over the first
  // mTISegmentNum subspaces.  The reference's k-means branch (:897-900) is training and
  // is not provided: fill mTIClusters yourself for that.  The grouping itself happens
  // on the GPU at the next search().
  void clusterTI(bool useKMeans = false, bool verbose = false) {
    (void)verbose;
    if (mTIVariance < 1.0f)
      throw Error(VAQHIP_EUNSUPPORTED, "vaqhip: TI<T>var<v> needs train()'s variance profile; use TI<T>m<seg>");
    if (mTISegmentNum == -1) mTISegmentNum = mHighestSubs();
    if (mTIClusters.rows() == 0) {
      if (useKMeans)
        throw Error(VAQHIP_EUNSUPPORTED, "vaqhip: the k-means of clusterTI is training; set mTIClusters");
      if (mCodebook.rows() == 0) throw Error(VAQHIP_ESTATE, "vaqhip: clusterTI needs mCodebook");
      const int L = mSubsLen();
      mTIClusters = RowMatrixF((size_t)mTIClusterNum, (size_t)mTISegmentNum * L);
      for (int i = 0; i < mTIClusterNum; i++) {
        const size_t r = (size_t)std::rand() % mCodebook.rows();  // VAQ.cpp:903
        for (int s = 0; s < mTISegmentNum; s++)
          for (int j = 0; j < L; j++) mTIClusters(i, (size_t)s * L + j) = mCentroidsPerSubs[s](mCodebook(r, s), j);
      }
    }
    mMethods |= NNMethod::TI;
    ti_set_ = false;
  }

  // VAQ::search, VAQ.hpp:102 / VAQ.cpp:776-847
  template <class Mat> LabelDistVecF search(const Mat &XTest, const int k, bool verbose = false) {
    (void)verbose;
    if (!(mMethods & (NNMethod::Heap | NNMethod::EA | NNMethod::TI)))
      throw Error(VAQHIP_EUNSUPPORTED, "vaqhip: only HEAP / EA / TI are implemented on this path");
    sync();
    LabelDistVecF ret;
    const size_t nq = (size_t)XTest.rows();
    ret.labels.resize(k * nq);
    ret.distances.resize(k * nq);
    if ((int)XTest.cols() != mTotalDim()) throw Error(VAQHIP_EINVAL, "vaqhip: XTest has the wrong width");
    if (mh_)
      checkMulti(vaqhip_multi_search(mh_, XTest.data(), (int)nq, k, 0, ret.labels.data(), ret.distances.data()));
    else
      check(vaqhip_search(h_, XTest.data(), (int)nq, k, ret.labels.data(), ret.distances.data()));
    return ret;
  }

  // VAQ::encode, VAQ.cpp:663-748: fills mCodebook from rows already in PCA
  // space (as the reference expects after train()); projected = false applies
  // mEigenVectors first.
  template <class Mat> void encode(const Mat &XTrain, bool projected = true) {
    const bool had = codes_set_;
    codes_set_ = true;  // sync() must not try to upload the (empty) codebook first
    try { sync(); } catch (...) { codes_set_ = had; throw; }
    codes_set_ = false;
    mCodebook = CodebookType((size_t)XTrain.rows(), (size_t)mHighestSubs());
    vaqhip_index *enc = mh_ ? vaqhip_multi_shard(mh_, 0) : h_;  // (every shard holds the codebooks)
    check(vaqhip_encode(enc, XTrain.data(), (int64_t)XTrain.rows(), projected ? 1 : 0, mCodebook.data()));
  }

  // VAQ::refine, VAQ.hpp:104 / VAQ.cpp:849-876
  template <class MatQ, class MatT>
  LabelDistVecF refine(const MatQ &XTest, const LabelDistVecF &answersIn, const MatT &XTrain, const int k) {
    const int nq = (int)XTest.rows();
    const int R = nq ? (int)(answersIn.labels.size() / nq) : 0;
    LabelDistVecF ret;
    ret.labels.resize((size_t)k * nq);
    ret.distances.resize((size_t)k * nq);
    check(vaqhip_refine(mDevice, XTest.data(), nq, (int)XTest.cols(), XTrain.data(), (int64_t)XTrain.rows(),
                        answersIn.labels.data(), R, k, ret.labels.data(), ret.distances.data()));
    return ret;
  }

  // Copy the search state out of a reference `VAQ` object (duck-typed: the
  // members of VAQ.hpp:51-75, Eigen matrices).  Instantiate only in a
  // translation unit that includes the reference's VAQ.hpp.
  template <class RefVAQ> void fromReference(const RefVAQ &v) {
    invalidate();
    mBitBudget = v.mBitBudget; mSubspaceNum = v.mSubspaceNum;
    mMinBitsPerSubs = v.mMinBitsPerSubs; mMaxBitsPerSubs = v.mMaxBitsPerSubs;
    mMethods = v.mMethods;
    mBitsAlloc.assign(v.mBitsAlloc.begin(), v.mBitsAlloc.begin() + v.mHighestSubs);
    mCentroidsPerSubs.clear();
    for (int s = 0; s < v.mHighestSubs; s++) {
      const auto &c = v.mCentroidsPerSubs[s];  // RowMatrix<float>
      RowMatrixF m(c.rows(), c.cols());
      for (size_t i = 0; i < m.rows(); i++)
        for (size_t j = 0; j < m.cols(); j++) m(i, j) = c(i, j);
      mCentroidsPerSubs.push_back(m);
    }
    const size_t D = v.mEigenVectors.rows();
    mEigenVectors = RowMatrixF(D, v.mEigenVectors.cols());
    for (size_t i = 0; i < D; i++)
      for (size_t j = 0; j < mEigenVectors.cols(); j++) mEigenVectors(i, j) = v.mEigenVectors(i, j).real();
    mCodebook = CodebookType(v.mCodebook.rows(), v.mCodebook.cols());
    // VAQ::clusterTI ends by REGROUPING mCodebook (VAQ.cpp:984-996: row r of the grouped matrix
    // is original row mTIClustersMember[c][r - start_c]) while search() keeps returning ORIGINAL
    // row numbers (VAQ.cpp:1575-1590).  This index regroups its own copy and labels by position
    // in the matrix it is given, so a codebook that was already regrouped is put back into
    // original order first; before clusterTI (no members yet) it is copied as it is.
    size_t members = 0;
    for (const auto &cm : v.mTIClustersMember) members += cm.size();
    if (members == 0) {
      for (size_t i = 0; i < mCodebook.rows(); i++)
        for (size_t j = 0; j < mCodebook.cols(); j++) mCodebook(i, j) = v.mCodebook(i, j);
    } else {
      if (members != mCodebook.rows())
        throw Error(VAQHIP_ESTATE, "vaqhip: mTIClustersMember does not cover mCodebook");
      size_t r = 0;
      for (const auto &cm : v.mTIClustersMember)
        for (const int idx : cm) {
          for (size_t j = 0; j < mCodebook.cols(); j++) mCodebook((size_t)idx, j) = v.mCodebook(r, j);
          r++;
        }
    }
    // the TI state search() reads (VAQ.hpp:77-84)
    mTIClusterNum = v.mTIClusterNum;
    mTISegmentNum = v.mTISegmentNum;
    mVisit = v.mVisit;
    mTIClusters = RowMatrixF((size_t)v.mTIClusters.rows(), (size_t)v.mTIClusters.cols());
    for (size_t i = 0; i < mTIClusters.rows(); i++)
      for (size_t j = 0; j < mTIClusters.cols(); j++) mTIClusters(i, j) = v.mTIClusters(i, j);
    if ((mMethods & NNMethod::TI) && mTIClusters.rows() == 0)
      throw Error(VAQHIP_ESTATE, "vaqhip: the reference object selects TI but has no mTIClusters yet "
                                 "(call its clusterTI() first, or clusterTI() here)");
  }

  vaqhip_index *handle() { return h_; }
  vaqhip_multi *multiHandle() { return mh_; }

private:
  static void checkMulti(int rc) {
    if (rc < 0) throw Error(rc, std::string("vaqhip: ") + vaqhip_multi_last_error());
  }
  vaqhip_index *h_ = nullptr;
  vaqhip_multi *mh_ = nullptr;
  bool codes_set_ = false;
  bool ti_set_ = false;
};

// ---------------------------------------------------------------------------
// BitVecEngine::queryLUT (BitVecEngine.hpp:1222-1343), the reference's other
// entry on this path: one scalar quantiser per PCA dimension, columns summed
// one by one.  The engine keeps its state private (centroidsMat, solutionX,
// eigenVectors, nonZeroAllocCount: BitVecEngine.hpp:33-39); here it is public.
// ---------------------------------------------------------------------------
struct IdxDistPairFloat {  // utils/Types.hpp:41-57
  int idx;
  float dist;
};

class BitVecEngineHip {
public:
  std::vector<float> centroidsMat;  // column-major, `centroidRows` (256) rows x nonZeroAllocCount columns
  int centroidRows = 256;
  std::vector<int> solutionX;       // bits per dimension
  RowMatrixF eigenVectors;          // real part, D x D, D = nonZeroAllocCount (empty = identity)
  int mDevice = 0;

  BitVecEngineHip() = default;
  BitVecEngineHip(const BitVecEngineHip &) = delete;
  BitVecEngineHip &operator=(const BitVecEngineHip &) = delete;
  ~BitVecEngineHip() { vaqhip_index_destroy(h_); }
  void invalidate() { vaqhip_index_destroy(h_); h_ = nullptr; }

  // queries: any row-major float matrix with data()/rows()/cols() (the reference
  // takes a column-major Eigen::MatrixXf: pass its transpose's storage or a RowMatrixF)
  template <class Mat, class Codebook>
  std::vector<std::vector<IdxDistPairFloat>> queryLUT(const Mat &queries, const int k, const Codebook &codebook) {
    const int nd = (int)solutionX.size();
    if (!h_) {
      std::vector<std::vector<float>> cols(nd);
      std::vector<const float *> cp(nd);
      for (int d = 0; d < nd; d++) {
        cols[d].assign(centroidsMat.begin() + (size_t)d * centroidRows,
                       centroidsMat.begin() + (size_t)d * centroidRows + ((size_t)1 << solutionX[d]));
        cp[d] = cols[d].data();
      }
      check(vaqhip_index_create_ex(&h_, nd, nd, solutionX.data(), cp.data(),
                                   eigenVectors.rows() ? eigenVectors.data() : nullptr, mDevice,
                                   VAQHIP_SUM_SEQUENTIAL));
    }
    check(vaqhip_index_set_codes_u16(h_, codebook.data(), (int64_t)codebook.rows(), 0));
    const int nq = (int)queries.rows();
    std::vector<int> lab((size_t)nq * k);
    std::vector<float> dis((size_t)nq * k);
    check(vaqhip_search(h_, queries.data(), nq, k, lab.data(), dis.data()));
    std::vector<std::vector<IdxDistPairFloat>> answers(nq);
    for (int q = 0; q < nq; q++)
      for (int i = 0; i < k && lab[(size_t)q * k + i] >= 0; i++)
        answers[q].push_back({lab[(size_t)q * k + i], dis[(size_t)q * k + i]});
    return answers;
  }

private:
  vaqhip_index *h_ = nullptr;
};

} // namespace vaqhip
#endif
