// vaq_ti.hip -- gfx950 kernels around the triangle-inequality (TI) cluster
// pruning of the reference: the index-side regrouping of VAQ::clusterTI
// (VAQ.cpp:878-999) and the per-query cluster order of the TI branch of
// VAQ::search (VAQ.cpp:799-826).  The pruned scan itself is the TI form of the
// scan kernels in vaq_kernels.hip.
//
// Compiled with -ffp-contract=off like every file of the library: distances
// are the reference's  res += tmp * tmp  (utils/Math.hpp:8-19), multiply and
// add unfused, or its SSE orders for d in {1,2,4,8,12} (:38-128).
#include "vaq_kernels.h"

#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <float.h>
#include <limits.h>
#include <math.h>

namespace vaq {

// ---------------------------------------------------------------------------
// packed rows (index order) -> CodebookType rows (uint16 N x M) in ORIGINAL row
// order: the inverse of pack_codes_kernel.  Lets the index be regrouped after
// the codes were handed over (the reference calls clusterTI after encode).
// ---------------------------------------------------------------------------
__global__ void unpack_codes_kernel(const uint32_t *__restrict__ packed, int64_t n, int M, int layout,
                                    int W, const SubDesc *__restrict__ sub,
                                    const uint32_t *__restrict__ perm, uint16_t *__restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  uint16_t *o = out + (perm ? (int64_t)perm[r] : r) * M;
  if (layout == LAYOUT_BYTES) {
    const uint32_t *rp = packed + r * (M / 4);
    for (int w = 0; w < M / 4; w++) {
      const uint32_t v = rp[w];
      o[w * 4 + 0] = (uint16_t)(v & 0xffu);
      o[w * 4 + 1] = (uint16_t)((v >> 8) & 0xffu);
      o[w * 4 + 2] = (uint16_t)((v >> 16) & 0xffu);
      o[w * 4 + 3] = (uint16_t)(v >> 24);
    }
  } else {
    const uint32_t *rp = packed + (r / TILE_ROWS) * (int64_t)(TILE_ROWS * W) + (r % TILE_ROWS);
    for (int s = 0; s < M; s++) {
      const SubDesc sd = sub[s];
      const uint32_t lo = rp[sd.word * TILE_ROWS];
      const uint32_t hi = (sd.word + 1 < W) ? rp[(sd.word + 1) * TILE_ROWS] : 0u;
      const uint64_t both = ((uint64_t)hi << 32) | lo;
      o[s] = (uint16_t)((both >> sd.shift) & (uint64_t)(sd.ncent - 1));
    }
  }
}

hipError_t launch_unpack_codes(const uint32_t *packed, int64_t n, int M, int layout, int W,
                               const SubDesc *sub, const uint32_t *perm, uint16_t *out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_codes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, packed, n, M,
                     layout, W, sub, perm, out);
  return hipGetLastError();
}

// one term of fvec_L2sqr_ny: ElementOpL2::op, utils/Math.hpp:131-136
__device__ __forceinline__ float sq_term(float x, float y) {
  const float t = x - y;
  return t * t;
}

// fvec_L2sqr_ny's value for one y (utils/Math.hpp:147-171): x[j] is read as
// xs[j * xstride]; the SSE kernels for d in {1,2,4,8,12} reduce their four
// lanes as (a0 + a1) + (a2 + a3) (two _mm_hadd_ps), everything else is the
// sequential loop of fvec_L2sqr_ref.
__device__ __forceinline__ float l2sqr_ref_order(const float *xs, int xstride, const float *__restrict__ yp,
                                                 int d, int ystride = 1) {
#define X(j) xs[(j) * xstride]
  struct YS {
    const float *p;
    int st;
    __device__ __forceinline__ float operator[](int j) const { return p[(size_t)j * st]; }
  };
  const YS y{yp, ystride};
  switch (d) {
  case 1: return sq_term(X(0), y[0]);
  case 2: return sq_term(X(0), y[0]) + sq_term(X(1), y[1]);
  case 4: return (sq_term(X(0), y[0]) + sq_term(X(1), y[1])) + (sq_term(X(2), y[2]) + sq_term(X(3), y[3]));
  case 8: {
    const float a0 = sq_term(X(0), y[0]) + sq_term(X(4), y[4]);
    const float a1 = sq_term(X(1), y[1]) + sq_term(X(5), y[5]);
    const float a2 = sq_term(X(2), y[2]) + sq_term(X(6), y[6]);
    const float a3 = sq_term(X(3), y[3]) + sq_term(X(7), y[7]);
    return (a0 + a1) + (a2 + a3);
  }
  case 12: {
    const float a0 = (sq_term(X(0), y[0]) + sq_term(X(4), y[4])) + sq_term(X(8), y[8]);
    const float a1 = (sq_term(X(1), y[1]) + sq_term(X(5), y[5])) + sq_term(X(9), y[9]);
    const float a2 = (sq_term(X(2), y[2]) + sq_term(X(6), y[6])) + sq_term(X(10), y[10]);
    const float a3 = (sq_term(X(3), y[3]) + sq_term(X(7), y[7])) + sq_term(X(11), y[11]);
    return (a0 + a1) + (a2 + a3);
  }
  default: {
    float res = 0.0f;
    for (int j = 0; j < d; j++) res += sq_term(X(j), y[j]);
    return res;
  }
  }
#undef X
}

// ---------------------------------------------------------------------------
// VAQ::clusterTI, VAQ.cpp:926-950: per code row, x = the centroids of its
// first `seg` codes side by side; the row joins the cluster with the smallest
// sqrt(fvec_L2sqr_ny(x, cluster)) (strict `<`: the first minimum wins, and
// sqrt can merge nearby values, so the comparison is made on the sqrt as the
// reference does); xcc = that distance (mCodeToCCDist).
// One row per thread; the workgroup's decoded rows sit in LDS as [dim][row]
// (conflict-free), the cluster centres are read with wave-uniform addresses.
// ---------------------------------------------------------------------------
__global__ void ti_assign_kernel(const uint16_t *__restrict__ codes, int64_t n, int M, int L, int seg,
                                 const SubDesc *__restrict__ sub, const float *__restrict__ cent,
                                 const float *__restrict__ clusters, int T, float *__restrict__ scratch,
                                 int *__restrict__ assign, float *__restrict__ xcc) {
  extern __shared__ float xs_lds[];  // [d][R]
  const int R = blockDim.x;
  const int d = seg * L;
  // centres of more dims than LDS holds for 64 rows: the decoded rows live in a global
  // scratch tile per workgroup instead (same [dim][row] shape, each thread its own column)
  float *xs = scratch ? scratch + (size_t)blockIdx.x * d * R : xs_lds;
  for (int64_t blk = blockIdx.x; blk * R < n; blk += gridDim.x) {
    const int64_t r = blk * R + threadIdx.x;
    if (r >= n) continue;
    for (int s = 0; s < seg; s++) {
      const SubDesc sd = sub[s];
      const float *c = cent + sd.cent_off + (size_t)(codes[r * M + s] & (sd.ncent - 1)) * L;
      for (int j = 0; j < L; j++) xs[(s * L + j) * R + threadIdx.x] = c[j];
    }
    // each thread reads back only its own column: no barrier needed
    float closest = FLT_MAX;
    int idx = 0;  // (the reference would index [-1] if no distance were < FLT_MAX)
    bool found = false;
    for (int c = 0; c < T; c++) {
      const float dist = sqrtf(l2sqr_ref_order(xs + threadIdx.x, R, clusters + (size_t)c * d, d));
      if (dist < closest) {
        closest = dist;
        idx = c;
        found = true;
      }
    }
    assign[r] = idx;
    xcc[r] = found ? closest : FLT_MAX;
  }
}

// key = cluster << 32 | ~bits(xcc): ascending key order = cluster ascending, xcc
// DEscending (VAQ.cpp:972-979 sorts members farthest first; xcc >= 0 so its bit
// pattern is monotone).  The radix sort is stable, so equal keys keep ascending
// original rows (the reference's std::sort leaves that order unspecified).
__global__ void ti_keys_kernel(const int *__restrict__ assign, const float *__restrict__ xcc, int64_t n,
                               uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = ((uint64_t)(uint32_t)assign[i] << 32) | (uint64_t)(~__builtin_bit_cast(unsigned, xcc[i]));
  idx[i] = (uint32_t)i;
}

__global__ void ti_bounds_kernel(const uint64_t *__restrict__ keys, int64_t n, int *__restrict__ start,
                                 float *__restrict__ xcc_sorted) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(keys[i] >> 32);
  if (i == 0 || (int)(keys[i - 1] >> 32) != c) start[c] = (int)i;
  xcc_sorted[i] = __builtin_bit_cast(float, ~(unsigned)(keys[i] & 0xffffffffu));
}

// rows per workgroup and its LDS bytes; 0 bytes = the tile does not fit LDS (global scratch)
static size_t ti_assign_lds(int d, int *rows_out) {
  int R = 256;
  while (R > 64 && (size_t)R * d * 4 > 48 * 1024) R >>= 1;
  if (rows_out) *rows_out = R;
  const size_t bytes = (size_t)R * d * 4;
  return bytes <= 128 * 1024 ? bytes : 0;
}

// Regroup: d_perm[n] (index row -> original row), d_start[T+1] (first index row
// of each cluster that occurs, -1 otherwise; the caller back-fills),
// d_xcc_sorted[n].  Synchronises the stream.
hipError_t ti_group_rows(const uint16_t *d_codes, int64_t n, int M, int L, int seg, const SubDesc *sub,
                         const float *cent, const float *d_clusters, int T, uint32_t *d_perm,
                         int *d_start, float *d_xcc_sorted, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d_start, 0xff, (size_t)(T + 1) * sizeof(int), st);
  if (e != hipSuccess || n == 0) return e;
  int *assign = nullptr;
  float *xcc = nullptr, *scratch = nullptr;
  uint64_t *keys_in = nullptr, *keys_out = nullptr;
  uint32_t *idx_in = nullptr;
  void *temp = nullptr;
  size_t temp_bytes = 0;
  auto cleanup = [&]() {
    (void)hipFree(assign); (void)hipFree(xcc); (void)hipFree(scratch); (void)hipFree(keys_in); (void)hipFree(keys_out);
    (void)hipFree(idx_in); (void)hipFree(temp);
  };
  if ((e = hipMalloc(&assign, (size_t)n * 4)) != hipSuccess || (e = hipMalloc(&xcc, (size_t)n * 4)) != hipSuccess ||
      (e = hipMalloc(&keys_in, (size_t)n * 8)) != hipSuccess ||
      (e = hipMalloc(&keys_out, (size_t)n * 8)) != hipSuccess ||
      (e = hipMalloc(&idx_in, (size_t)n * 4)) != hipSuccess) {
    cleanup();
    return e;
  }
  int R = 0;
  const size_t lds = ti_assign_lds(seg * L, &R);
  int64_t grid = (n + R - 1) / R;
  if (lds == 0) {
    grid = std::min<int64_t>(grid, 2048);
    e = hipMalloc(&scratch, (size_t)grid * R * seg * L * sizeof(float));
  } else {
    grid = std::min<int64_t>(grid, (int64_t)1 << 30);
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(ti_assign_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ti_assign_kernel, dim3((unsigned)grid), dim3(R), lds, st, d_codes, n, M, L, seg, sub,
                       cent, d_clusters, T, scratch, assign, xcc);
    e = hipGetLastError();
  }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ti_keys_kernel, dim3(blocks), dim3(256), 0, st, assign, xcc, n, keys_in, idx_in);
    e = hipGetLastError();
  }
  unsigned cbits = 1;
  while ((1u << cbits) < (unsigned)T) cbits++;
  if (e == hipSuccess)
    e = rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, idx_in, d_perm, (size_t)n, 0u,
                                  32u + cbits, st);
  if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
  if (e == hipSuccess)
    e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, idx_in, d_perm, (size_t)n, 0u,
                                  32u + cbits, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(ti_bounds_kernel, dim3(blocks), dim3(256), 0, st, keys_out, n, d_start, d_xcc_sorted);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  return e;
}

// ---------------------------------------------------------------------------
// TI branch of VAQ::search, VAQ.cpp:799-826, one workgroup per query:
//   qToCCDist[c] = sqrt(fvec_L2sqr_ny(first d projected dims, mTIClusters[c]))
//   clusters sorted by qToCCDist ascending (std::sort; ties here: ascending
//   cluster index)
// and the head of searchTriangleInequality (:1548-1555, :1611): the clusters
// visited are the first  max(maxVisit, shortest prefix holding >= k rows)  of
// that order.
// Outputs, per query: order[T] (cluster ids), qcc[T] (their distances, same
// order), nvisit.
// ---------------------------------------------------------------------------
constexpr int TI_PLAN_THREADS = 256;

__global__ __launch_bounds__(TI_PLAN_THREADS) void ti_plan_kernel(
    const float *__restrict__ qproj, int D, int d, const float *__restrict__ clusters_t, int T, int Tp,
    const int *__restrict__ start, int max_visit, int k, int *__restrict__ order,
    float *__restrict__ qcc, int *__restrict__ nvisit) {
  extern __shared__ unsigned char smem[];
  // one 64-bit key per cluster: distance bits (>= 0, so they order like the values; NaN sorts
  // last) above the cluster index -- ascending keys = ascending (distance, cluster)
  unsigned long long *key = reinterpret_cast<unsigned long long *>(smem);  // [Tp]
  float *qs = reinterpret_cast<float *>(key + Tp);                         // [d]
  const int q = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < d; j += TI_PLAN_THREADS) qs[j] = qproj[(size_t)q * D + j];
  __syncthreads();
  for (int c = tid; c < Tp; c += TI_PLAN_THREADS) {
    unsigned long long k = ~0ull;
    if (c < T) {
      // clusters_t is dimension-major (T floats per dimension): lanes read consecutive floats
      const float v = sqrtf(l2sqr_ref_order(qs, 1, clusters_t + c, d, T));
      k = ((unsigned long long)__builtin_bit_cast(unsigned, v) << 32) | (unsigned)c;
    }
    key[c] = k;
  }
  __syncthreads();
  for (int size = 2; size <= Tp; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (Tp >> 1); t += TI_PLAN_THREADS) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const unsigned long long a = key[i], b = key[j];
        if ((a > b) == ((i & size) == 0)) {
          key[i] = b;
          key[j] = a;
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < T; i += TI_PLAN_THREADS) {
    const unsigned long long k = key[i];
    order[(size_t)q * T + i] = (int)(unsigned)k;
    qcc[(size_t)q * T + i] = __builtin_bit_cast(float, (unsigned)(k >> 32));
  }
  if (tid == 0) {
    int p = 0;
    int64_t cum = 0;
    while (p < T && cum < k) {
      const int c = (int)(unsigned)key[p];
      cum += start[c + 1] - start[c];
      p++;
    }
    nvisit[q] = p > max_visit ? p : max_visit;
  }
}

hipError_t launch_ti_plan(const float *qproj, int nq, int D, int d, const float *clusters_t, int T,
                          const int *start, int max_visit, int k, int *order, float *qcc, int *nvisit,
                          hipStream_t st) {
  if (nq == 0) return hipSuccess;
  int Tp = 2;
  while (Tp < T) Tp <<= 1;
  const size_t lds = (size_t)Tp * 8 + (size_t)d * 4;  // keys + the query's first d dims
  hipLaunchKernelGGL(ti_plan_kernel, dim3(nq), dim3(TI_PLAN_THREADS), lds, st, qproj, D, d, clusters_t, T, Tp,
                     start, max_visit, k, order, qcc, nvisit);
  return hipGetLastError();
}

} // namespace vaq
