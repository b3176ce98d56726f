/*
 * vaq_oracle.c -- CPU restatement of the reference's ADC search path.
 * TEST INFRASTRUCTURE ONLY (see vaq_oracle.h for the rules and the pinning
 * status of each function).  Build: oracle/Makefile, which compiles this
 * file with -ffp-contract=off so that every rounding below is the one the
 * source spells out; fused multiply-adds appear only as explicit fmaf().
 */
#include "vaq_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * Heap.  CMax<float,int>::cmp(a,b) = a > b (utils/Heap.hpp:78-80); neutral =
 * FLT_MAX (:81-83).
 * ---------------------------------------------------------------------- */
static inline int cmax(float a, float b) { return a > b; }

/* utils/Heap.hpp:115-144.  1-based sift-down of the last element from the
 * root; on equal children the comparison at :127 is false, so the RIGHT
 * child is taken. */
void vo_heap_pop(size_t k, float *val, int *ids) {
  val--; ids--;
  float v = val[k];
  size_t i = 1, i1, i2;
  for (;;) {
    i1 = i << 1;
    i2 = i1 + 1;
    if (i1 > k) break;
    if (i2 == k + 1 || cmax(val[i1], val[i2])) {
      if (cmax(v, val[i1])) break;
      val[i] = val[i1]; ids[i] = ids[i1]; i = i1;
    } else {
      if (cmax(v, val[i2])) break;
      val[i] = val[i2]; ids[i] = ids[i2]; i = i2;
    }
  }
  val[i] = val[k];
  ids[i] = ids[k];
}

/* utils/Heap.hpp:151-169.  Sift-up from slot k. */
void vo_heap_push(size_t k, float *val, int *ids, float v, int id) {
  val--; ids--;
  size_t i = k, f;
  while (i > 1) {
    f = i >> 1;
    if (!cmax(v, val[f])) break;
    val[i] = val[f]; ids[i] = ids[f]; i = f;
  }
  val[i] = v;
  ids[i] = id;
}

/* utils/Heap.hpp:211-235 with k0 = 0: fill with neutral / -1. */
void vo_heap_heapify(size_t k, float *val, int *ids) {
  for (size_t i = 0; i < k; i++) { val[i] = FLT_MAX; ids[i] = -1; }
}

/* utils/Heap.hpp:322-349. */
size_t vo_heap_reorder(size_t k, float *val, int *ids) {
  size_t i, ii;
  for (i = 0, ii = 0; i < k; i++) {
    float v = val[0];
    int id = ids[0];
    vo_heap_pop(k - i, val, ids);
    val[k - ii - 1] = v;
    ids[k - ii - 1] = id;
    if (id != -1) ii++;
  }
  size_t nel = ii;
  memmove(val, val + k - ii, ii * sizeof(*val));
  memmove(ids, ids + k - ii, ii * sizeof(*ids));
  for (; ii < k; ii++) { val[ii] = FLT_MAX; ids[ii] = -1; }
  return nel;
}

/* The insert rule of VAQ.cpp:1750-1753 applied to a precomputed distance
 * array. */
void vo_topk_from_dists(const float *dist, const int *ids_or_null, int64_t n,
                        int k, int *out_ids, float *out_val) {
  vo_heap_heapify((size_t)k, out_val, out_ids);
  for (int64_t i = 0; i < n; i++) {
    float d = dist[i];
    if (cmax(out_val[0], d)) {
      vo_heap_pop((size_t)k, out_val, out_ids);
      vo_heap_push((size_t)k, out_val, out_ids, d,
                   ids_or_null ? ids_or_null[i] : (int)i);
    }
  }
  vo_heap_reorder((size_t)k, out_val, out_ids);
}

/* ------------------------------------------------------------------------
 * Projection, VAQ.hpp:198-201: (X * mEigenVectors).real().  With X real the
 * real part of each product is x*re(e) exactly; only the summation order is
 * Eigen's.  Fixed here to an fmaf chain over the inner index.
 * ---------------------------------------------------------------------- */
void vo_project(const float *X, int64_t n, int D, const float *E, float *out) {
  for (int64_t r = 0; r < n; r++) {
    const float *x = X + r * (int64_t)D;
    float *o = out + r * (int64_t)D;
    for (int c = 0; c < D; c++) {
      float acc = 0.0f;
      for (int j = 0; j < D; j++) acc = fmaf(x[j], E[(size_t)j * D + c], acc);
      o[c] = acc;
    }
  }
}

/* ------------------------------------------------------------------------
 * fvec_L2sqr_ny, utils/Math.hpp:147-171 and its SSE kernels :38-128.  The
 * lanes of an __m128 are reduced by two _mm_hadd_ps: (a0+a1)+(a2+a3).
 * ElementOpL2::op is tmp*tmp of the difference (:131-143); `accu += op(..)`
 * is a separate multiply and add here (no contraction).
 * ---------------------------------------------------------------------- */
static inline float sq(float x, float y) { float t = x - y; return t * t; }

static void l2sqr_ny(float *dis, const float *x, const float *y, size_t d,
                     size_t ny) {
  size_t i;
  switch (d) {
  case 1: /* :38-58 */
    for (i = 0; i < ny; i++) dis[i] = sq(x[0], y[i]);
    return;
  case 2: /* :60-78 : hadd(accu,accu)[0] = a0+a1, [1]... lane 3 = a2+a3 */
    for (i = 0; i < ny; i++) dis[i] = sq(x[0], y[2 * i]) + sq(x[1], y[2 * i + 1]);
    return;
  case 4: /* :82-95 */
    for (i = 0; i < ny; i++, y += 4)
      dis[i] = (sq(x[0], y[0]) + sq(x[1], y[1])) + (sq(x[2], y[2]) + sq(x[3], y[3]));
    return;
  case 8: /* :97-111 */
    for (i = 0; i < ny; i++, y += 8) {
      float a0 = sq(x[0], y[0]) + sq(x[4], y[4]);
      float a1 = sq(x[1], y[1]) + sq(x[5], y[5]);
      float a2 = sq(x[2], y[2]) + sq(x[6], y[6]);
      float a3 = sq(x[3], y[3]) + sq(x[7], y[7]);
      dis[i] = (a0 + a1) + (a2 + a3);
    }
    return;
  case 12: /* :113-128 */
    for (i = 0; i < ny; i++, y += 12) {
      float a0 = (sq(x[0], y[0]) + sq(x[4], y[4])) + sq(x[8], y[8]);
      float a1 = (sq(x[1], y[1]) + sq(x[5], y[5])) + sq(x[9], y[9]);
      float a2 = (sq(x[2], y[2]) + sq(x[6], y[6])) + sq(x[10], y[10]);
      float a3 = (sq(x[3], y[3]) + sq(x[7], y[7])) + sq(x[11], y[11]);
      dis[i] = (a0 + a1) + (a2 + a3);
    }
    return;
  default: /* :8-34 sequential */
    for (i = 0; i < ny; i++, y += d) {
      float res = 0;
      for (size_t j = 0; j < d; j++) { float t = x[j] - y[j]; res += t * t; }
      dis[i] = res;
    }
  }
}

/* ------------------------------------------------------------------------
 * VAQ::CreateLUT<maxbit>, VAQ.hpp:128-167.
 *   :130      lut.setZero()
 *   :134      AVX2 branch when mCentroidsNum[s] >= 8
 *   :144-154  for j: diff = q[s*L+j] - C_s[c][j]; acc = fma(diff,diff,acc)
 *             (utils/AVXUtils.hpp:11-15 is a true vfmadd231ps), acc from 0
 *   :161-165  otherwise fvec_L2sqr_ny on the row-major centroids
 * The 8-wide striping of :135-153 does not change per-centroid arithmetic.
 * ---------------------------------------------------------------------- */
void vo_create_lut(const float *qproj, int M, int L, const int *ncent,
                   const float *const *cent, int ksub, float *lut) {
  memset(lut, 0, sizeof(float) * (size_t)ksub * (size_t)M);
  for (int s = 0; s < M; s++) {
    float *col = lut + (size_t)s * ksub;
    const float *q = qproj + (size_t)s * L;
    if (ncent[s] >= 8) {
      for (int c = 0; c < ncent[s]; c++) {
        const float *cr = cent[s] + (size_t)c * L;
        float acc = 0.0f;
        for (int j = 0; j < L; j++) {
          float diff = q[j] - cr[j];
          acc = fmaf(diff, diff, acc);
        }
        col[c] = acc;
      }
    } else {
      l2sqr_ny(col, q, cent[s], (size_t)L, (size_t)ncent[s]);
    }
  }
}

/* ------------------------------------------------------------------------
 * Row distance of VAQ.cpp:1737-1748: groups of four subspaces,
 *   dism = luts0[c0]; dism += luts1[c1]; dism += luts2[c2]; dism += luts3[c3];
 *   dist += dism;            (dist starts at 0)
 * ---------------------------------------------------------------------- */
static inline float row_dist(const float *lut, int ksub, const uint16_t *c,
                             int M) {
  float dist = 0;
  const float *l = lut;
  for (int col = 0; col < M; col += 4) {
    float dism;
    dism  = l[c[col]];     l += ksub;
    dism += l[c[col + 1]]; l += ksub;
    dism += l[c[col + 2]]; l += ksub;
    dism += l[c[col + 3]]; l += ksub;
    dist += dism;
  }
  return dist;
}

void vo_all_dists(const float *lut, int ksub, const uint16_t *codes, int64_t N,
                  int M, float *out) {
  for (int64_t i = 0; i < N; i++) out[i] = row_dist(lut, ksub, codes + i * M, M);
}

/* VAQ::searchHeap, VAQ.cpp:1729-1758. */
void vo_search_heap(const float *lut, int ksub, const uint16_t *codes,
                    int64_t N, int M, int k, int *ids, float *dis) {
  vo_heap_heapify((size_t)k, dis, ids);
  for (int64_t i = 0; i < N; i++) {
    float dist = row_dist(lut, ksub, codes + i * M, M);
    if (cmax(dis[0], dist)) {
      vo_heap_pop((size_t)k, dis, ids);
      vo_heap_push((size_t)k, dis, ids, dist, (int)i);
    }
  }
  vo_heap_reorder((size_t)k, dis, ids);
}

/* VAQ::searchEarlyAbandon, VAQ.cpp:1694-1727: the group loop stops once
 * dist >= bsfK (:1708); bsfK = heap top after each insert (:1721). */
void vo_search_ea(const float *lut, int ksub, const uint16_t *codes, int64_t N,
                  int M, int k, int *ids, float *dis) {
  vo_heap_heapify((size_t)k, dis, ids);
  float bsfK = FLT_MAX;
  for (int64_t i = 0; i < N; i++) {
    const uint16_t *c = codes + i * M;
    const float *l = lut;
    float dist = 0;
    for (int col = 0; col < M && dist < bsfK; col += 4) {
      float dism;
      dism  = l[c[col]];     l += ksub;
      dism += l[c[col + 1]]; l += ksub;
      dism += l[c[col + 2]]; l += ksub;
      dism += l[c[col + 3]]; l += ksub;
      dist += dism;
    }
    if (cmax(dis[0], dist)) {
      vo_heap_pop((size_t)k, dis, ids);
      vo_heap_push((size_t)k, dis, ids, dist, (int)i);
      bsfK = dis[0];
    }
  }
  vo_heap_reorder((size_t)k, dis, ids);
}

/* ------------------------------------------------------------------------
 * VAQ::search, VAQ.cpp:776-847: project (:777), one LUT of 1<<maxbits rows
 * reused per query (:780), per query CreateLUT then searchEarlyAbandon /
 * searchHeap (:828-831).  Only those two method bits are restated.
 * ---------------------------------------------------------------------- */
int vo_search(const vo_index *ix, const float *X, int nq, int k,
              unsigned method, int nthreads, int projected, int *labels,
              float *distances) {
  if (ix->M % 4 != 0) return -1; /* VAQ.cpp:1741-1746 reads 4 codes per step */
  if (!(method & (VO_METHOD_EA | VO_METHOD_HEAP))) return -2;
  const int D = ix->D, M = ix->M, L = ix->L;
  const int ksub = 1 << ix->max_bits;
  float *xp = NULL;
  const float *Q = X;
  if (!projected && ix->eig) {
    xp = (float *)malloc(sizeof(float) * (size_t)nq * D);
    vo_project(X, nq, D, ix->eig, xp);
    Q = xp;
  }
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    float *lut = (float *)malloc(sizeof(float) * (size_t)ksub * M);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int q = 0; q < nq; q++) {
      vo_create_lut(Q + (size_t)q * D, M, L, ix->ncent, ix->cent, ksub, lut);
      int *ids = labels + (size_t)q * k;
      float *dis = distances + (size_t)q * k;
      if (method & VO_METHOD_EA)
        vo_search_ea(lut, ksub, ix->codes, ix->N, M, k, ids, dis);
      else
        vo_search_heap(lut, ksub, ix->codes, ix->N, M, k, ids, dis);
    }
    free(lut);
  }
  free(xp);
  return 0;
}

/* ------------------------------------------------------------------------
 * VAQ::encodeImpl, VAQ.cpp:728-748: per subspace, per row, argmin over codes
 * of (x_block - c_row).squaredNorm() with strict `<` (first minimum wins).
 * Eigen's squaredNorm reduction order is not defined by the source; it is
 * restated as a sequential sum of squares (agreed with
 * the reference's encode on 160000/160000 codes in the survey probe).
 * ---------------------------------------------------------------------- */
void vo_encode(const float *Xproj, int64_t n, int M, int L, const int *ncent,
               const float *const *cent, int nthreads, uint16_t *codes) {
  const int D = M * L;
  if (nthreads < 1) nthreads = 1;
  for (int s = 0; s < M; s++) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads)
#endif
    for (int64_t r = 0; r < n; r++) {
      const float *x = Xproj + r * D + (size_t)s * L;
      uint16_t best = 0;
      float bsf = FLT_MAX;
      for (int c = 0; c < ncent[s]; c++) {
        const float *cr = cent[s] + (size_t)c * L;
        float dist = 0;
        for (int j = 0; j < L; j++) { float t = x[j] - cr[j]; dist += t * t; }
        if (dist < bsf) { best = (uint16_t)c; bsf = dist; }
      }
      codes[r * M + s] = best;
    }
  }
}

/* ------------------------------------------------------------------------
 * VAQ::refine, VAQ.cpp:849-876: exact squared L2 between the raw query and
 * the raw dataset row of each candidate, same heap.  squaredNorm order
 * restated as sequential (as in vo_encode).
 * ---------------------------------------------------------------------- */
void vo_refine(const float *Xq, int nq, int D, const float *Xtrain,
               const int *labels_in, int refine_num, int k, int *labels,
               float *distances) {
  for (int q = 0; q < nq; q++) {
    int *ids = labels + (size_t)q * k;
    float *dis = distances + (size_t)q * k;
    vo_heap_heapify((size_t)k, dis, ids);
    const float *x = Xq + (size_t)q * D;
    for (int i = 0; i < refine_num; i++) {
      int lab = labels_in[(size_t)q * refine_num + i];
      const float *y = Xtrain + (size_t)lab * D;
      float dist = 0;
      for (int j = 0; j < D; j++) { float t = x[j] - y[j]; dist += t * t; }
      if (cmax(dis[0], dist)) {
        vo_heap_pop((size_t)k, dis, ids);
        vo_heap_push((size_t)k, dis, ids, dist, lab);
      }
    }
    vo_heap_reorder((size_t)k, dis, ids);
  }
}

/* utils/Experiment.hpp:252-271 (labels overload, IdxOffset = 0). */
double vo_avg_recall(const int *labels, int nq, int K, const int *topnn,
                     int topnn_stride) {
  double ans = 0.0;
  for (int q = 0; q < nq; q++) {
    int ct = 0;
    for (int ki = 0; ki < K; ki++)
      for (int j = 0; j < K; j++)
        if (labels[(size_t)q * K + ki] == topnn[(size_t)q * topnn_stride + j]) { ct++; break; }
    ans += ((double)ct) / K;
  }
  return ans / nq;
}

/* utils/Experiment.hpp:288-303 (labels overload). */
double vo_recall_at_r(const int *labels, int nq, int K, const int *topnn,
                      int topnn_stride) {
  double ans = 0.0;
  for (int q = 0; q < nq; q++) {
    int truenn = topnn[(size_t)q * topnn_stride];
    for (int ki = 0; ki < K; ki++)
      if (truenn == labels[(size_t)q * K + ki]) { ans += 1; break; }
  }
  return ans / nq;
}

/* ------------------------------------------------------------------------
 * BitVecEngine::queryLUT, BitVecEngine.hpp:1222-1343 (projected query in).
 *   :1230      LUT is 256 x ndim column-major (stride fixed 256)
 *   :1236-1262 bits >= 3: one fma(diff,diff,0) per centroid
 *   :1263-1267 bits <  3: diff*diff
 *   :1296-1300 sequential column sum with early abandon (dist < bsfK)
 *   :1301-1311 insert iff dist < bsfK (= current k-th best once k rows seen)
 * ---------------------------------------------------------------------- */
void vo_query_lut_1d(const float *qproj, int ndim, const int *bits,
                     const float *cent_colmajor, int cent_rows,
                     const uint16_t *codes, int64_t N, int code_cols, int k,
                     int *ids, float *dis) {
  float *lut = (float *)calloc((size_t)256 * ndim, sizeof(float));
  for (int d = 0; d < ndim; d++) {
    int nc = 1 << bits[d];
    for (int c = 0; c < nc; c++) {
      float diff = qproj[d] - cent_colmajor[(size_t)d * cent_rows + c];
      lut[(size_t)d * 256 + c] = (bits[d] >= 3) ? fmaf(diff, diff, 0.0f) : diff * diff;
    }
  }
  vo_heap_heapify((size_t)k, dis, ids);
  float bsfK = FLT_MAX;
  int64_t filled = 0;
  for (int64_t i = 0; i < N; i++) {
    const uint16_t *c = codes + i * code_cols;
    float dist = 0;
    for (int col = 0; col < ndim && dist < bsfK; col++)
      dist += lut[(size_t)col * 256 + c[col]];
    if (dist < bsfK) {
      vo_heap_pop((size_t)k, dis, ids);
      vo_heap_push((size_t)k, dis, ids, dist, (int)i);
      filled++;
      if (i >= k) bsfK = dis[0];
    }
  }
  (void)filled;
  vo_heap_reorder((size_t)k, dis, ids);
  free(lut);
}

/* ------------------------------------------------------------------------
 * Triangle-inequality cluster pruning.
 *
 * VAQ::clusterTI, VAQ.cpp:878-999, from the point where mTIClusters exists
 * (its k-means, KMeans::staticFitCodebook with srand(time), is training and is
 * injected like the codebooks):
 *   :926-950  per code row: x = concatenation of the centroids of its first
 *             `seg` codes; dists = fvec_L2sqr_ny(x, mTIClusters); the row joins
 *             the cluster with the smallest sqrt(dist) (strict `<`, first wins);
 *             mCodeToCCDist[row] = that sqrt
 *   :972-979  members of a cluster sorted by mCodeToCCDist descending
 *             (std::sort, order among equal keys is implementation defined;
 *             restated as: ties by ascending row)
 *   :984-996  mCodebook regrouped cluster by cluster, mClusterMembersStartIdx
 * ---------------------------------------------------------------------- */
typedef struct { float d; int i; } fi_pair;
static int fi_desc(const void *a, const void *b) {
  const fi_pair *x = (const fi_pair *)a, *y = (const fi_pair *)b;
  if (x->d > y->d) return -1;
  if (x->d < y->d) return 1;
  return (x->i > y->i) - (x->i < y->i);
}
static int fi_asc(const void *a, const void *b) {
  const fi_pair *x = (const fi_pair *)a, *y = (const fi_pair *)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->i > y->i) - (x->i < y->i);
}

void vo_cluster_ti(const uint16_t *codes, int64_t N, int M, int L,
                   const float *const *cent, const float *clusters, int T,
                   int seg, int nthreads, int *member, int *start,
                   float *code2cc, uint16_t *grouped) {
  const int d = seg * L;
  int *assign = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
  if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    float *x = (float *)malloc(sizeof(float) * (size_t)d);
    float *dists = (float *)malloc(sizeof(float) * (size_t)T);
#ifdef _OPENMP
#pragma omp for
#endif
    for (int64_t i = 0; i < N; i++) {
      for (int s = 0; s < seg; s++)
        memcpy(x + (size_t)s * L, cent[s] + (size_t)codes[i * M + s] * L, sizeof(float) * (size_t)L);
      l2sqr_ny(dists, x, clusters, (size_t)d, (size_t)T);
      float closest = FLT_MAX;
      int idx = -1;
      for (int c = 0; c < T; c++) {
        float dist = sqrtf(dists[c]);
        if (dist < closest) { idx = c; closest = dist; }
      }
      code2cc[i] = closest;
      assign[i] = idx;  /* -1 only if every distance is NaN/inf (the reference then indexes [-1]) */
    }
    free(x);
    free(dists);
  }
  int *cnt = (int *)calloc((size_t)T + 1, sizeof(int));
  for (int64_t i = 0; i < N; i++) cnt[(assign[i] < 0 ? 0 : assign[i]) + 1]++;
  start[0] = 0;
  for (int c = 0; c < T; c++) start[c + 1] = start[c] + cnt[c + 1];
  fi_pair *tmp = (fi_pair *)malloc(sizeof(fi_pair) * (size_t)(N > 0 ? N : 1));
  int *fill = (int *)malloc(sizeof(int) * (size_t)T);
  for (int c = 0; c < T; c++) fill[c] = start[c];
  for (int64_t i = 0; i < N; i++) {
    const int c = assign[i] < 0 ? 0 : assign[i];
    tmp[fill[c]].d = code2cc[i];
    tmp[fill[c]].i = (int)i;
    fill[c]++;
  }
  for (int c = 0; c < T; c++)
    qsort(tmp + start[c], (size_t)(start[c + 1] - start[c]), sizeof(fi_pair), fi_desc);
  for (int64_t r = 0; r < N; r++) {
    member[r] = tmp[r].i;
    if (grouped) memcpy(grouped + r * M, codes + (int64_t)tmp[r].i * M, sizeof(uint16_t) * (size_t)M);
  }
  free(fill);
  free(tmp);
  free(cnt);
  free(assign);
}

/* VAQ::search, TI branch, VAQ.cpp:799-826: qToCCDist = sqrt(fvec_L2sqr_ny(
 * first seg*L projected dims, mTIClusters)); clusters visited in ascending
 * qToCCDist (std::sort; ties restated as ascending cluster index). */
void vo_ti_query_order(const float *qproj, const float *clusters, int T, int d,
                       float *qcc, int *order) {
  l2sqr_ny(qcc, qproj, clusters, (size_t)d, (size_t)T);
  fi_pair *tmp = (fi_pair *)malloc(sizeof(fi_pair) * (size_t)T);
  for (int c = 0; c < T; c++) {
    qcc[c] = sqrtf(qcc[c]);
    tmp[c].d = qcc[c];
    tmp[c].i = c;
  }
  qsort(tmp, (size_t)T, sizeof(fi_pair), fi_asc);
  for (int c = 0; c < T; c++) order[c] = tmp[c].i;
  free(tmp);
}

/* VAQ::searchTriangleInequality, VAQ.cpp:1540-1692, both branches.
 *   :1548-1551 maxClusterVisit = int(float(T) * mVisit) when mVisit < 1
 *   :1555      clusters in qToCCIdx order while idx < maxClusterVisit, or
 *              until k rows have been seen (retrievedEnough, :1611)
 *   :1564-1568 once k rows are in: the rest of a cluster is pruned when
 *              bsfK <= qToCCDist[cluster] - mCodeToCCDist[row]
 *   EA (:1553-1616): the group loop runs while dist < bsfK^2; a row enters when
 *              dist < bsfK^2; stored distance is sqrt(dist); bsfK = heap top
 *   no EA (:1617-1686): bsfKSquared stays 0 (it is never updated in the fill
 *              phase, :1672-1674), so after the first k rows nothing enters:
 *              the result is the first k rows of the visiting order.  Restated
 *              as written.
 * `grouped` is the regrouped mCodebook, `member` = mTIClustersMember
 * flattened (labels are original rows), code2cc indexed by original row. */
void vo_search_ti(const float *lut, int ksub, const uint16_t *grouped, int M,
                  const int *member, const int *start, const float *code2cc,
                  int T, const float *qcc, const int *order, float visit,
                  int use_ea, int k, int *ids, float *dis, long *pruned_out) {
  vo_heap_heapify((size_t)k, dis, ids);
  float bsfK = 0, bsfKSquared = 0;
  int counter = 0;
  long pruned = 0;
  int maxVisit = T;
  if (visit < 1) maxVisit = (int)((float)T * visit);
  int retrievedEnough = 0;
  for (int ci = 0; (ci < maxVisit) || (!retrievedEnough && ci < T); ci++) {
    const int c = order[ci];
    const int s0 = start[c], s1 = start[c + 1];
    if (s1 == s0) continue;
    const uint16_t *codes = grouped + (size_t)M * s0;
    int inter = 0;
    for (int r = s0; r < s1; r++) {
      const int dataIndex = member[r];
      if (counter >= k) {
        if (bsfK <= (qcc[c] - code2cc[dataIndex])) {
          pruned += (s1 - s0) - inter;
          break;
        }
        float dist = 0;
        const float *l = lut;
        int col;
        for (col = 0; col < M && (!use_ea || dist < bsfKSquared); col += 4) {
          float dism;
          dism  = l[*codes++]; l += ksub;
          dism += l[*codes++]; l += ksub;
          dism += l[*codes++]; l += ksub;
          dism += l[*codes++]; l += ksub;
          dist += dism;
        }
        codes += (M - col);
        if (dist < bsfKSquared) {
          dist = sqrtf(dist);
          vo_heap_pop((size_t)k, dis, ids);
          vo_heap_push((size_t)k, dis, ids, dist, dataIndex);
          bsfK = dis[0];
          bsfKSquared = bsfK * bsfK;
        }
      } else {
        float dist = 0;
        const float *l = lut;
        for (int col = 0; col < M; col += 4) {
          float dism;
          dism  = l[*codes++]; l += ksub;
          dism += l[*codes++]; l += ksub;
          dism += l[*codes++]; l += ksub;
          dism += l[*codes++]; l += ksub;
          dist += dism;
        }
        dist = sqrtf(dist);
        vo_heap_pop((size_t)k, dis, ids);
        vo_heap_push((size_t)k, dis, ids, dist, dataIndex);
        if (dist > bsfK) {
          bsfK = dist;
          if (use_ea) bsfKSquared = bsfK * bsfK;  /* :1602 vs :1672-1674 */
        }
        counter++;
      }
      inter++;
    }
    if (counter >= k) retrievedEnough = 1;
  }
  for (int i = maxVisit; i < T; i++) pruned += start[order[i] + 1] - start[order[i]];
  if (pruned_out) *pruned_out = pruned;
  vo_heap_reorder((size_t)k, dis, ids);
}

/* VAQ::search with NNMethod::TI set, VAQ.cpp:776-847. */
int vo_search_ti_all(const vo_index *ix, const vo_ti *ti, const float *X, int nq,
                     int k, unsigned method, int nthreads, int projected,
                     int *labels, float *distances, long *total_pruned) {
  if (ix->M % 4 != 0) return -1;
  const int D = ix->D, M = ix->M, L = ix->L;
  const int ksub = 1 << ix->max_bits;
  float *xp = NULL;
  const float *Q = X;
  if (!projected && ix->eig) {
    xp = (float *)malloc(sizeof(float) * (size_t)nq * D);
    vo_project(X, nq, D, ix->eig, xp);
    Q = xp;
  }
  if (nthreads < 1) nthreads = 1;
  long pruned_sum = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads) reduction(+ : pruned_sum)
#endif
  {
    float *lut = (float *)malloc(sizeof(float) * (size_t)ksub * M);
    float *qcc = (float *)malloc(sizeof(float) * (size_t)ti->T);
    int *order = (int *)malloc(sizeof(int) * (size_t)ti->T);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int q = 0; q < nq; q++) {
      vo_create_lut(Q + (size_t)q * D, M, L, ix->ncent, ix->cent, ksub, lut);
      vo_ti_query_order(Q + (size_t)q * D, ti->clusters, ti->T, ti->seg * L, qcc, order);
      long pr = 0;
      vo_search_ti(lut, ksub, ti->grouped, M, ti->member, ti->start, ti->code2cc, ti->T, qcc,
                   order, ti->visit, (method & VO_METHOD_EA) != 0, k, labels + (size_t)q * k,
                   distances + (size_t)q * k, &pr);
      pruned_sum += pr;
    }
    free(order);
    free(qcc);
    free(lut);
  }
  free(xp);
  if (total_pruned) *total_pruned = pruned_sum;
  return 0;
}

int vo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
