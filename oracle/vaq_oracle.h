/*
 * vaq_oracle.h -- CPU restatement of the reference's ADC search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / reported CPU baseline.
 * The product path (vaq_amd/, include/vaqhip.h) never links or calls it.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * arithmetic it restates.  Pinning status (see DESIGN.md "Oracle"):
 *   - heap top-k semantics : pinned against the reference's own
 *                            utils/Heap.hpp compiled in oracle/_ref and against
 *                            tests/golden/heap_*.npz produced by it.
 *   - K_s<8 LUT fallback   : pinned against the reference's utils/Math.hpp
 *                            fvec_L2sqr_ny compiled in oracle/_ref.
 *   - on-disk formats, recall metrics : pinned against utils/IO.hpp and
 *                            utils/Experiment.hpp compiled in oracle/_ref.
 *   - CreateLUT AVX2 loop and searchHeap sum order : PARITY UNPINNED.
 *     bitvecengine/VAQ.hpp / VAQ.cpp need glpk + armadillo, which this image
 *     lacks, so they are unbuildable here; the loops are restated from the
 *     source text (VAQ.hpp:128-167, VAQ.cpp:1729-1758) and only the `fma`
 *     primitive (utils/AVXUtils.hpp:11-15) is exercised from the reference.
 */
#ifndef VAQ_ORACLE_H_
#define VAQ_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- top-k heap: utils/Heap.hpp:73-88 (CMax), 115-144 (pop), 151-169 (push),
 *      211-235 (heapify, k0 = 0), 322-349 (reorder) ---------------------- */
void   vo_heap_heapify(size_t k, float *val, int *ids);
void   vo_heap_pop(size_t k, float *val, int *ids);
void   vo_heap_push(size_t k, float *val, int *ids, float v, int id);
size_t vo_heap_reorder(size_t k, float *val, int *ids);

/* Feed n distances (ids 0..n-1, or ids[] when given) through the exact
 * insert rule of VAQ.cpp:1750-1753 and finish with heap_reorder. */
void vo_topk_from_dists(const float *dist, const int *ids_or_null, int64_t n,
                        int k, int *out_ids, float *out_val);

/* ---- VAQ::ProjectOnEigenVectors, VAQ.hpp:198-201 ------------------------
 * out[n x D] = X[n x D] * E[D x D] (real part; E row-major).  The reference
 * evaluates this with Eigen's blocked GEMM whose summation order is not
 * defined by the source; the restatement fixes it to an fmaf chain over the
 * inner index ascending (tolerance, not bit-exactness, is claimed vs Eigen). */
void vo_project(const float *X, int64_t n, int D, const float *E, float *out);

/* ---- VAQ::CreateLUT<maxbit>, VAQ.hpp:128-167 ----------------------------
 * lut is the reference LUTType: column-major ksub x M, i.e. lut[s*ksub + c].
 * ncent[s] >= 8 : AVX2 branch, fmaf chain over j ascending from 0.0f.
 * ncent[s] <  8 : fvec_L2sqr_ny (utils/Math.hpp:147-171) with its SSE
 *                 reduction orders for L in {1,2,4,8,12}, else sequential.
 * cent[s] is the row-major ncent[s] x L matrix (mCentroidsPerSubs[s]). */
void vo_create_lut(const float *qproj, int M, int L, const int *ncent,
                   const float *const *cent, int ksub, float *lut);

/* ---- VAQ::searchHeap, VAQ.cpp:1729-1758 --------------------------------- */
void vo_search_heap(const float *lut, int ksub, const uint16_t *codes,
                    int64_t N, int M, int k, int *ids, float *dis);
/* ---- VAQ::searchEarlyAbandon, VAQ.cpp:1694-1727 -------------------------- */
void vo_search_ea(const float *lut, int ksub, const uint16_t *codes,
                  int64_t N, int M, int k, int *ids, float *dis);
/* all N distances in the reference's summation order (test helper) */
void vo_all_dists(const float *lut, int ksub, const uint16_t *codes,
                  int64_t N, int M, float *out);

/* ---- VAQ::search, VAQ.cpp:776-847 (HEAP = 0x80, EA = 0x02 only) ---------- */
typedef struct {
  int D, M, L;             /* mTotalDim, mHighestSubs, mSubsLen           */
  int max_bits;            /* mMaxBitsPerSubs -> ksub = 1 << max_bits     */
  const int *ncent;        /* mCentroidsNum[M]                            */
  const float *const *cent;/* mCentroidsPerSubs[M], row-major ncent x L   */
  const float *eig;        /* real(mEigenVectors) D x D row-major, or NULL = identity */
  const uint16_t *codes;   /* mCodebook, N x M row-major                  */
  int64_t N;
} vo_index;

#define VO_METHOD_EA   0x02u
#define VO_METHOD_HEAP 0x80u

/* nthreads = 1 is the reference's execution model (VAQ.cpp:786 is a plain
 * sequential loop); nthreads > 1 runs the unmodified per-query algorithm
 * under OpenMP over queries (second timing).
 * projected != 0: X is already in PCA space (skip vo_project). */
int vo_search(const vo_index *ix, const float *X, int nq, int k,
              unsigned method, int nthreads, int projected,
              int *labels, float *distances);

/* ---- VAQ::encodeImpl, VAQ.cpp:728-748 (input already projected) --------- */
void vo_encode(const float *Xproj, int64_t n, int M, int L, const int *ncent,
               const float *const *cent, int nthreads, uint16_t *codes);

/* ---- VAQ::refine, VAQ.cpp:849-876 ---------------------------------------- */
void vo_refine(const float *Xq, int nq, int D, const float *Xtrain,
               const int *labels_in, int refine_num, int k,
               int *labels, float *distances);

/* ---- getAvgRecall / getRecallAtR, utils/Experiment.hpp:252-271, 288-303 -- */
double vo_avg_recall(const int *labels, int nq, int K, const int *topnn,
                     int topnn_stride);
double vo_recall_at_r(const int *labels, int nq, int K, const int *topnn,
                      int topnn_stride);

/* ---- BitVecEngine::queryLUT, BitVecEngine.hpp:1222-1343 ------------------
 * 1-D subspaces, LUT stride fixed 256, sequential column sum with early
 * abandon, libstdc++-style heap is NOT restated: results are returned in
 * (dist asc) order of the k best under the same insert rule.  cent is the
 * column-major centroidsMat (rows = 256, one column per dimension). */
void vo_query_lut_1d(const float *qproj, int ndim, const int *bits,
                     const float *cent_colmajor, int cent_rows,
                     const uint16_t *codes, int64_t N, int code_cols, int k,
                     int *ids, float *dis);

/* ---- triangle-inequality cluster pruning ---------------------------------
 * VAQ::clusterTI (VAQ.cpp:878-999) after mTIClusters exists; the TI branch of
 * VAQ::search (:799-826); VAQ::searchTriangleInequality (:1540-1692).
 * PARITY UNPINNED (VAQ.cpp is unbuildable here); fvec_L2sqr_ny and the heap
 * underneath are the pinned pieces.  std::sort ties are restated as stable
 * (ascending row / cluster index). */
#define VO_METHOD_TI 0x04u
void vo_cluster_ti(const uint16_t *codes, int64_t N, int M, int L,
                   const float *const *cent, const float *clusters, int T,
                   int seg, int nthreads, int *member /*[N]*/,
                   int *start /*[T+1]*/, float *code2cc /*[N], by original row*/,
                   uint16_t *grouped /*[N x M] or NULL*/);
void vo_ti_query_order(const float *qproj, const float *clusters, int T, int d,
                       float *qcc /*[T]*/, int *order /*[T]*/);
void vo_search_ti(const float *lut, int ksub, const uint16_t *grouped, int M,
                  const int *member, const int *start, const float *code2cc,
                  int T, const float *qcc, const int *order, float visit,
                  int use_ea, int k, int *ids, float *dis, long *pruned_out);
typedef struct {
  const float *clusters;   /* mTIClusters, T x (seg*L) row-major            */
  int T, seg;              /* mTIClusterNum, mTISegmentNum                  */
  float visit;             /* mVisit                                        */
  const int *member;       /* mTIClustersMember flattened (original rows)   */
  const int *start;        /* mClusterMembersStartIdx + total, [T+1]        */
  const float *code2cc;    /* mCodeToCCDist, by original row                */
  const uint16_t *grouped; /* mCodebook after regrouping                    */
} vo_ti;
int vo_search_ti_all(const vo_index *ix, const vo_ti *ti, const float *X, int nq,
                     int k, unsigned method, int nthreads, int projected,
                     int *labels, float *distances, long *total_pruned);

int vo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
