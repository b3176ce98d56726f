// vaq_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the VAQ ADC search path.
//
// Compiled with -ffp-contract=off: every float add below is a plain IEEE fp32
// add in the order the reference's source spells out, and fused multiply-adds
// appear only as explicit __builtin_fmaf (utils/AVXUtils.hpp:11-15 is a real
// vfmadd231ps).  Wavefront = 64 lanes everywhere.
//
// Kernels
//   project_kernel      VAQ::ProjectOnEigenVectors          VAQ.hpp:198-201
//   lut_build_kernel    VAQ::CreateLUT<maxbit>              VAQ.hpp:128-167
//   pack_codes_kernel   CodebookType (uint16 N x M) -> packed device rows
//   scan_*_kernel       VAQ::searchHeap / searchEarlyAbandon: vaq_scan.h, instantiated in
//                       vaq_scan_bytes.hip and vaq_scan_bits.hip
//   merge_kernel        heap_reorder's sorted output         utils/Heap.hpp:322-349
#include "vaq_scan.h"
#include "vaq_scan_bf.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <float.h>
#include <limits.h>

namespace vaq {
// ---------------------------------------------------------------------------
// VAQ::ProjectOnEigenVectors, VAQ.hpp:198-201: out = (X * mEigenVectors).real()
// One workgroup per row; thread c owns output column c; fmaf chain over the
// inner index ascending (the reference leaves the order to Eigen's GEMM).
// ---------------------------------------------------------------------------
// checked: BitVecEngine::ProjectOnEigenVectors(Z, withChecking = true) (BitVecEngine.hpp:53-71, called
// by queryLUT at :1226): a coordinate that comes out NaN or infinite is replaced by 0.  E == nullptr
// stands for the identity matrix: the product z * I is then NaN in EVERY column as soon as one
// component of z is not finite (NaN * 0 and inf * 0 are NaN), i.e. the whole row becomes 0.
__global__ void project_kernel(const float *__restrict__ X, int D,
                               const float *__restrict__ E, float *__restrict__ out, int checked) {
  extern __shared__ float xs[];
  __shared__ int bad;
  const int64_t r = blockIdx.x;
  if (threadIdx.x == 0) bad = 0;
  for (int j = threadIdx.x; j < D; j += blockDim.x) xs[j] = X[r * D + j];
  __syncthreads();
  if (!E) {
    for (int j = threadIdx.x; j < D; j += blockDim.x)
      if (!(fabsf(xs[j]) <= FLT_MAX)) bad = 1;
    __syncthreads();
    const bool zero = checked && bad;
    for (int c = threadIdx.x; c < D; c += blockDim.x) out[r * D + c] = zero ? 0.0f : xs[c];
    return;
  }
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float acc = 0.0f;
    for (int j = 0; j < D; j++) acc = __builtin_fmaf(xs[j], E[(size_t)j * D + c], acc);
    if (checked && !(fabsf(acc) <= FLT_MAX)) acc = 0.0f;
    out[r * D + c] = acc;
  }
}

// Many rows (the encoder's input: VAQ::encode projects the whole dataset, VAQ.cpp:294): a workgroup
// takes RPT * (256 / D) rows, thread (h, c) keeps RPT outputs of column c in registers, the rows sit in
// LDS and are read four inner indices at a time (one ds_read_b128, the same address for the whole
// wave: a broadcast).  Every output is still ONE fmaf chain over the inner index ascending from 0,
// so the result is bit for bit the one-row-per-workgroup kernel's; per four inner steps a thread
// issues 4 loads of E, RPT LDS reads and 4 RPT FMAs.
template <int D_, int RPT>
__global__ __launch_bounds__(256) void project_tile_kernel(const float *__restrict__ X, int64_t n,
                                                           const float *__restrict__ E, float *__restrict__ out,
                                                           int checked) {
  constexpr int G = 256 / D_, R = G * RPT;
  __shared__ __attribute__((aligned(16))) float xs[R * D_];
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * R;
  for (int i = tid; i < R * D_ / 4; i += 256) {
    const int64_t row = row0 + (i * 4) / D_;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (row < n) v = reinterpret_cast<const float4 *>(X + row0 * D_)[i];
    reinterpret_cast<float4 *>(xs)[i] = v;
  }
  __syncthreads();
  const int c = tid % D_, h = tid / D_;
  float acc[RPT];
#pragma unroll
  for (int r = 0; r < RPT; r++) acc[r] = 0.0f;
  for (int j = 0; j < D_; j += 4) {
    const float e0 = E[(size_t)(j + 0) * D_ + c], e1 = E[(size_t)(j + 1) * D_ + c];
    const float e2 = E[(size_t)(j + 2) * D_ + c], e3 = E[(size_t)(j + 3) * D_ + c];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
      const float4 xv = *reinterpret_cast<const float4 *>(&xs[(h * RPT + r) * D_ + j]);
      acc[r] = __builtin_fmaf(xv.x, e0, acc[r]);
      acc[r] = __builtin_fmaf(xv.y, e1, acc[r]);
      acc[r] = __builtin_fmaf(xv.z, e2, acc[r]);
      acc[r] = __builtin_fmaf(xv.w, e3, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < RPT; r++) {
    const int64_t row = row0 + h * RPT + r;
    float a = acc[r];
    if (checked && !(fabsf(a) <= FLT_MAX)) a = 0.0f;
    if (row < n) out[row * D_ + c] = a;
  }
}

// The tiled kernel from this many rows on: a query batch too (10 k rows: 0.025 -> 0.013 ms with 8 rows per
// thread, C2 step 0.491 -> 0.480 ms); the encoder's millions of rows take 16 per thread.
#ifndef VAQ_PROJECT_TILE_MIN
#define VAQ_PROJECT_TILE_MIN 2048
#endif
#ifndef VAQ_PROJECT_TILE_RPT_SMALL
#define VAQ_PROJECT_TILE_RPT_SMALL 8
#endif
constexpr int64_t PROJECT_TILE_MIN_ROWS = VAQ_PROJECT_TILE_MIN;
constexpr int64_t PROJECT_TILE_BIG_ROWS = 65536;

template <int RPT>
static hipError_t launch_project_tile(const float *X, int64_t n, int D, const float *E, float *out, hipStream_t st, int checked) {
  const int R = (256 / D) * RPT;
  const dim3 grid((unsigned)((n + R - 1) / R));
  if (D == 64) hipLaunchKernelGGL((project_tile_kernel<64, RPT>), grid, dim3(256), 0, st, X, n, E, out, checked);
  else if (D == 128) hipLaunchKernelGGL((project_tile_kernel<128, RPT>), grid, dim3(256), 0, st, X, n, E, out, checked);
  else hipLaunchKernelGGL((project_tile_kernel<256, RPT>), grid, dim3(256), 0, st, X, n, E, out, checked);
  return hipGetLastError();
}

hipError_t launch_project(const float *X, int64_t n, int D, const float *E, float *out,
                          hipStream_t st, int checked) {
  if (n == 0) return hipSuccess;
  if (E && n >= PROJECT_TILE_MIN_ROWS && (D == 64 || D == 128 || D == 256)) {
    if (n >= PROJECT_TILE_BIG_ROWS) return launch_project_tile<16>(X, n, D, E, out, st, checked);
    return launch_project_tile<VAQ_PROJECT_TILE_RPT_SMALL>(X, n, D, E, out, st, checked);
  }
  // (one workgroup per row: batching 8 rows per workgroup, as the LUT build does with queries,
  //  was measured slower here -- 0.044 vs 0.030 ms for 10 k rows: too few workgroups to fill the chip)
  int threads = D < 256 ? ((D + 63) / 64) * 64 : 256;
  hipLaunchKernelGGL(project_kernel, dim3((unsigned)n), dim3(threads), D * sizeof(float), st, X, D,
                     E, out, checked);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// VAQ::CreateLUT<maxbit>, VAQ.hpp:128-167.  grid = (query, subspace); one
// thread per centroid.
//   ncent >= 8 (:134-159): acc = fma(diff, diff, acc) for j ascending, from 0.
//   ncent <  8 (:161-165): fvec_L2sqr_ny, utils/Math.hpp:147-171, with the
//       SSE reduction orders of :38-128 for L in {1,2,4,8,12}; sequential
//       multiply-then-add otherwise (:8-19).
// Output is the packed LUT (no zero tails): lut[q][lut_off[s] + c].
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sqdiff(float x, float y) {
  float t = x - y;
  return t * t;
}

__global__ void lut_build_kernel(const float *__restrict__ qproj, int D, int L,
                                 const SubDesc *__restrict__ sub,
                                 const float *__restrict__ cent_t, int lut_floats,
                                 float *__restrict__ lut, int only_small) {
  extern __shared__ float qs[];
  const int q = blockIdx.x, s = blockIdx.y;
  const SubDesc sd = sub[s];
  if ((int)(blockIdx.z * blockDim.x) >= sd.ncent) return;  // chunk beyond this subspace's codebook
  if (only_small && sd.ncent >= 8) return;                 // (the batched kernel did those)
  for (int j = threadIdx.x; j < L; j += blockDim.x) qs[j] = qproj[(size_t)q * D + (size_t)s * L + j];
  __syncthreads();
  float *out = lut + (size_t)q * lut_floats + sd.lut_off;
  // dimension-major codebook: y(j) of centroid c at cs[j * K + c], so a wave reads 64
  // consecutive floats per dimension
  const float *cs = cent_t + sd.cent_off;
  const int K = sd.ncent;
#define Y(j) cs[(size_t)(j) * K + c]
  for (int c = blockIdx.z * blockDim.x + threadIdx.x; c < K; c += gridDim.z * blockDim.x) {
    float r;
    if (K >= 8) {
      float acc = 0.0f;
      for (int j = 0; j < L; j++) {
        float diff = qs[j] - Y(j);
        acc = __builtin_fmaf(diff, diff, acc);
      }
      r = acc;
    } else if (L == 1) {
      r = sqdiff(qs[0], Y(0));
    } else if (L == 2) {
      r = sqdiff(qs[0], Y(0)) + sqdiff(qs[1], Y(1));
    } else if (L == 4) {
      r = (sqdiff(qs[0], Y(0)) + sqdiff(qs[1], Y(1))) + (sqdiff(qs[2], Y(2)) + sqdiff(qs[3], Y(3)));
    } else if (L == 8) {
      float a0 = sqdiff(qs[0], Y(0)) + sqdiff(qs[4], Y(4));
      float a1 = sqdiff(qs[1], Y(1)) + sqdiff(qs[5], Y(5));
      float a2 = sqdiff(qs[2], Y(2)) + sqdiff(qs[6], Y(6));
      float a3 = sqdiff(qs[3], Y(3)) + sqdiff(qs[7], Y(7));
      r = (a0 + a1) + (a2 + a3);
    } else if (L == 12) {
      float a0 = (sqdiff(qs[0], Y(0)) + sqdiff(qs[4], Y(4))) + sqdiff(qs[8], Y(8));
      float a1 = (sqdiff(qs[1], Y(1)) + sqdiff(qs[5], Y(5))) + sqdiff(qs[9], Y(9));
      float a2 = (sqdiff(qs[2], Y(2)) + sqdiff(qs[6], Y(6))) + sqdiff(qs[10], Y(10));
      float a3 = (sqdiff(qs[3], Y(3)) + sqdiff(qs[7], Y(7))) + sqdiff(qs[11], Y(11));
      r = (a0 + a1) + (a2 + a3);
    } else {
      float res = 0.0f;
      for (int j = 0; j < L; j++) {
        float t = qs[j] - Y(j);
        res += t * t;
      }
      r = res;
    }
    out[c] = r;
  }
#undef Y
}

// The same for codebooks of >= 8 centroids (VAQ.hpp:134-159), LUT_QT queries per workgroup: a
// thread owns one centroid and reads each of its coordinates ONCE for the whole batch of queries
// (the per-query kernel above spends its time launching 80 000 workgroups of 16 FMAs a thread
// at C2).  Per (query, centroid) the chain is unchanged: acc = fma(diff, diff, acc), j ascending.
constexpr int LUT_QT = 8;
__global__ __launch_bounds__(256) void lut_build_batch_kernel(const float *__restrict__ qproj, int nq, int D, int L,
                                                              const SubDesc *__restrict__ sub,
                                                              const float *__restrict__ cent_t, int lut_floats,
                                                              float *__restrict__ lut) {
  extern __shared__ float qs[];  // [LUT_QT][L]
  const int q0 = blockIdx.x * LUT_QT, s = blockIdx.y;
  const SubDesc sd = sub[s];
  const int K = sd.ncent;
  if (K < 8) return;  // (those subspaces are the per-query kernel's)
  for (int i = threadIdx.x; i < LUT_QT * L; i += blockDim.x) {
    const int qt = i / L, j = i - qt * L;
    qs[i] = q0 + qt < nq ? qproj[(size_t)(q0 + qt) * D + (size_t)s * L + j] : 0.0f;
  }
  __syncthreads();
  const float *cs = cent_t + sd.cent_off;
  for (int c = threadIdx.x; c < K; c += blockDim.x) {
    float acc[LUT_QT];
#pragma unroll
    for (int qt = 0; qt < LUT_QT; qt++) acc[qt] = 0.0f;
    for (int j = 0; j < L; j++) {
      const float y = cs[(size_t)j * K + c];
#pragma unroll
      for (int qt = 0; qt < LUT_QT; qt++) {
        const float diff = qs[qt * L + j] - y;
        acc[qt] = __builtin_fmaf(diff, diff, acc[qt]);
      }
    }
#pragma unroll
    for (int qt = 0; qt < LUT_QT; qt++)
      if (q0 + qt < nq) lut[(size_t)(q0 + qt) * lut_floats + sd.lut_off + c] = acc[qt];
  }
}

hipError_t launch_lut_build(const float *qproj, int nq, int D, int M, int L, const SubDesc *sub,
                            const float *cent_t, int lut_floats, int max_ncent, float *lut,
                            hipStream_t st, int min_ncent) {
  if (nq == 0) return hipSuccess;
  (void)max_ncent;
  if (nq >= 2 * LUT_QT && max_ncent >= 8) {
    hipLaunchKernelGGL(lut_build_batch_kernel, dim3((nq + LUT_QT - 1) / LUT_QT, M), dim3(256),
                       LUT_QT * L * sizeof(float), st, qproj, nq, D, L, sub, cent_t, lut_floats, lut);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || min_ncent >= 8) return e;
    // codebooks of fewer than 8 centroids (the reference's scalar branch) go through the per-query kernel
    hipLaunchKernelGGL(lut_build_kernel, dim3(nq, M, 1), dim3(64), L * sizeof(float), st, qproj, D, L, sub, cent_t,
                       lut_floats, lut, 1);
    return hipGetLastError();
  }
  // one workgroup per (query, subspace): splitting big codebooks over blockIdx.z was measured
  // slower (the extra, mostly empty workgroups cost more than the 16-iteration loop they save)
  hipLaunchKernelGGL(lut_build_kernel, dim3(nq, M, 1), dim3(256), L * sizeof(float), st, qproj, D, L,
                     sub, cent_t, lut_floats, lut, 0);
  return hipGetLastError();
}

// packed LUT -> the reference's LUTType (ksub x M column-major, zero tails; VAQ.cpp:780,
// VAQ.hpp:130).  Test hook only.
__global__ void lut_expand_kernel(const float *__restrict__ lp, int M,
                                  const SubDesc *__restrict__ sub, int lut_floats, int ksub,
                                  float *__restrict__ lr) {
  const int q = blockIdx.x, s = blockIdx.y;
  const SubDesc sd = sub[s];
  for (int c = threadIdx.x; c < ksub; c += blockDim.x)
    lr[((size_t)q * M + s) * ksub + c] =
        c < sd.ncent ? lp[(size_t)q * lut_floats + sd.lut_off + c] : 0.0f;
}

hipError_t launch_lut_expand(const float *lut_packed, int nq, int M, const SubDesc *sub,
                             int lut_floats, int ksub, float *lut_ref, hipStream_t st) {
  if (nq == 0) return hipSuccess;
  hipLaunchKernelGGL(lut_expand_kernel, dim3(nq, M), dim3(256), 0, st, lut_packed, M, sub,
                     lut_floats, ksub, lut_ref);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// CodebookType (uint16 N x M row-major, utils/Types.hpp:31) -> device layout.
//  LAYOUT_BYTES (every subspace 8 bits): row-major N x M bytes; byte s = code s.
//  LAYOUT_BITS : field s occupies bits [bit_off, bit_off+bits) of a W-dword
//      little-endian row, LSB-first; rows are stored in planar tiles of 64:
//      word w of row r at (r/64)*64*W + w*64 + (r%64), so a wavefront's load
//      of word w is one coalesced 256-byte access.
// One thread per output dword in [t_begin, t_end); `in` holds rows
// [row_begin, row_end) of the caller's matrix; rows outside it pack to zero
// (the padding past N).
// ---------------------------------------------------------------------------
__global__ void pack_codes_kernel(const uint16_t *__restrict__ in, int64_t row_begin,
                                  int64_t row_end, int M, int layout, int W,
                                  const SubDesc *__restrict__ sub, const uint32_t *__restrict__ perm,
                                  uint32_t *__restrict__ out, int64_t t_begin, int64_t t_end) {
  const int64_t t = t_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= t_end) return;
  uint32_t v = 0;
  if (layout == LAYOUT_BYTES) {
    const int wpr = M / 4;
    const int64_t r = t / wpr;
    const int w = (int)(t - r * wpr);
    if (r >= row_begin && r < row_end) {
      const uint16_t *c = in + ((perm ? (int64_t)perm[r] : r) - row_begin) * M + w * 4;
      v = (uint32_t)(c[0] & 0xff) | ((uint32_t)(c[1] & 0xff) << 8) |
          ((uint32_t)(c[2] & 0xff) << 16) | ((uint32_t)(c[3] & 0xff) << 24);
    }
  } else {
    const int64_t tile = t / ((int64_t)TILE_ROWS * W);
    const int rem = (int)(t - tile * TILE_ROWS * W);
    const int w = rem / TILE_ROWS;
    const int64_t r = tile * TILE_ROWS + (rem % TILE_ROWS);
    if (r >= row_begin && r < row_end) {
      const int64_t src = (perm ? (int64_t)perm[r] : r) - row_begin;
      for (int s = 0; s < M; s++) {
        const SubDesc sd = sub[s];
        const uint32_t code = (uint32_t)in[src * M + s] & (uint32_t)(sd.ncent - 1);
        if (sd.word == w) v |= code << sd.shift;
        else if (sd.word + 1 == w && sd.shift + sd.bits > 32) v |= code >> (32 - sd.shift);
      }
    }
  }
  out[t] = v;
}

int64_t packed_words(int64_t rows, int M, int layout, int W) {
  if (layout == LAYOUT_BYTES) return rows * (M / 4);
  return ((rows + TILE_ROWS - 1) / TILE_ROWS) * TILE_ROWS * W;
}

hipError_t launch_pack_codes(const uint16_t *codes_u16, int64_t row_begin, int64_t row_end,
                             int64_t out_row_end, int M, int layout, int W, const SubDesc *sub,
                             const uint32_t *perm, uint32_t *out, hipStream_t st) {
  const int64_t t_begin = packed_words(row_begin, M, layout, W);
  const int64_t t_end = packed_words(out_row_end, M, layout, W);
  if (t_end <= t_begin) return hipSuccess;
  const int64_t blocks = (t_end - t_begin + 255) / 256;
  hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, codes_u16,
                     row_begin, row_end, M, layout, W, sub, perm, out, t_begin, t_end);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------
// Appending rows to a bucketed index without rebuilding it: the new rows are sorted and packed
// on their own (sort_by_first_code + pack_codes, n_new rows), then merged bucket by bucket --
// bucket b of the result = bucket b of the old order followed by bucket b of the new rows (new
// labels are larger, so this IS the stable order a sort of all rows would give).  One thread per
// output row copies the row's packed words and its label; no unpacking, no global re-sort.
//   tot_start[b] = old_start[b] + new_start[b]   (K0 + 1 entries each)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int64_t packed_word_index(int64_t row, int w, int layout, int WPR) {
  if (layout == LAYOUT_BYTES) return row * WPR + w;
  return (row / TILE_ROWS) * (int64_t)(TILE_ROWS * WPR) + (int64_t)w * TILE_ROWS + (row % TILE_ROWS);
}

__global__ void merge_rows_kernel(const uint32_t *__restrict__ old_codes, const uint32_t *__restrict__ old_perm,
                                  const int *__restrict__ old_start, const uint32_t *__restrict__ new_codes,
                                  const uint32_t *__restrict__ new_perm, const int *__restrict__ new_start, int K0,
                                  int64_t n_old, int64_t n_total, int layout, int WPR, uint32_t *__restrict__ out_codes,
                                  uint32_t *__restrict__ out_perm) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_total) return;
  // bucket of output row r: largest b with old_start[b] + new_start[b] <= r
  int lo = 0, hi = K0;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)old_start[mid] + new_start[mid] <= r) lo = mid;
    else hi = mid;
  }
  const int b = lo;
  const int64_t off = r - ((int64_t)old_start[b] + new_start[b]);
  const int64_t old_cnt = (int64_t)old_start[b + 1] - old_start[b];
  const bool from_old = off < old_cnt;
  const int64_t src = from_old ? old_start[b] + off : new_start[b] + (off - old_cnt);
  const uint32_t *codes = from_old ? old_codes : new_codes;
  for (int w = 0; w < WPR; w++)
    out_codes[packed_word_index(r, w, layout, WPR)] = codes[packed_word_index(src, w, layout, WPR)];
  out_perm[r] = from_old ? old_perm[src] : (uint32_t)(n_old + new_perm[src]);
}

hipError_t launch_merge_rows(const uint32_t *old_codes, const uint32_t *old_perm, const int *old_start,
                             const uint32_t *new_codes, const uint32_t *new_perm, const int *new_start, int K0,
                             int64_t n_old, int64_t n_total, int M, int layout, int W, uint32_t *out_codes,
                             uint32_t *out_perm, hipStream_t st) {
  if (n_total <= 0) return hipSuccess;
  const int WPR = layout == LAYOUT_BYTES ? M / 4 : W;
  hipLaunchKernelGGL(merge_rows_kernel, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, st, old_codes, old_perm,
                     old_start, new_codes, new_perm, new_start, K0, n_old, n_total, layout, WPR, out_codes, out_perm);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Bucketed row order.  The index stores rows stably sorted by (the top bits
// of) the code of subspace 0 -- after PCA the highest-variance subspace.  A
// bucket's rows share the first LUT term (shift == 0) or at least a lower bound
// of it (shift > 0: the minimum over the bucket's 1 << shift codes), so the scan
// skips a whole bucket, without touching its codes, when that bound alone already
// exceeds the threshold of every query in the batch (the bucket-level form of
// VAQ::searchEarlyAbandon's test, VAQ.cpp:1708); with shift == 0 it also reads
// the term once per bucket (wave-uniform) instead of gathering it per row.
// The host picks shift so that buckets average >= ~2048 rows.
//   perm[r]          original row of sorted row r (labels are original rows)
//   bucket_start[b]  first sorted row whose code 0 is >= b  (b = 0 .. K0)
// ---------------------------------------------------------------------------
__global__ void first_code_keys_kernel(const uint16_t *__restrict__ codes, int64_t n, int M,
                                       unsigned mask, int shift, unsigned mask1, int shift1, int t,
                                       uint16_t *__restrict__ keys, uint32_t *__restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned key = (codes[i * M] & mask) >> shift;
  if (t > 0) key = (key << t) | ((codes[i * M + 1] & mask1) >> shift1);  // + top t bits of code 1 (t includes the fine bits)
  keys[i] = (uint16_t)key;
  idx[i] = (uint32_t)i;
}

// start[b] = first i with keys[i] >> fine == b, for the keys that occur (others stay -1);
// sub (optional): the same at the level of the whole key
__global__ void bucket_bounds_kernel(const uint16_t *__restrict__ keys, int64_t n, int fine, int *__restrict__ start,
                                     int *__restrict__ sub) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned k = keys[i], kp = i > 0 ? keys[i - 1] : 0u;
  if (i == 0 || (k >> fine) != (kp >> fine)) start[k >> fine] = (int)i;
  if (sub && (i == 0 || k != kp)) sub[k] = (int)i;
}

// fine > 0 (needs shift == 0): the rows of a bucket are ordered by the NEXT `fine` bits of the second
// code as well -- the sort key is bucket key << fine | those bits -- and d_sub_start[(K0 << fine) + 1]
// receives the first row of every (bucket, fine value) run, like d_bucket_start.  Rows of a run share
// their first two lookup-table terms whenever t + fine == bits1 (vaq_scan_bm.hip skips whole runs).
hipError_t sort_by_first_code(const uint16_t *d_codes, int64_t n, int M, int bits0, int shift, int bits1,
                              int t, uint32_t *d_perm, int *d_bucket_start, hipStream_t st, int fine,
                              int *d_sub_start) {
  const int kbits_b = bits0 - shift + t;  // bucket key: the first code's top bits0 - shift bits [+ t of code 1]
  const int kbits = kbits_b + fine;
  const int K0 = 1 << kbits_b;
  if (fine < 0 || kbits > 16 || (fine > 0 && (shift != 0 || t + fine > bits1 || !d_sub_start))) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(d_bucket_start, 0xff, (size_t)(K0 + 1) * sizeof(int), st);
  if (e == hipSuccess && fine > 0) e = hipMemsetAsync(d_sub_start, 0xff, ((size_t)(K0 << fine) + 1) * sizeof(int), st);
  if (e != hipSuccess || n == 0) return e;
  uint16_t *keys_in = nullptr, *keys_out = nullptr;
  uint32_t *idx_in = nullptr;
  void *temp = nullptr;
  size_t temp_bytes = 0;
  auto cleanup = [&]() {
    (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(idx_in); (void)hipFree(temp);
  };
  if ((e = hipMalloc(&keys_in, (size_t)n * 2)) != hipSuccess || (e = hipMalloc(&keys_out, (size_t)n * 2)) != hipSuccess ||
      (e = hipMalloc(&idx_in, (size_t)n * 4)) != hipSuccess) {
    cleanup();
    return e;
  }
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(first_code_keys_kernel, dim3(blocks), dim3(256), 0, st, d_codes, n, M,
                     (unsigned)((1 << bits0) - 1), shift, (unsigned)((1 << bits1) - 1), bits1 - (t + fine), t + fine, keys_in,
                     idx_in);
  // stable LSD radix sort on the b0 key bits: equal codes keep ascending original rows
  e = rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, idx_in, d_perm, (size_t)n, 0u,
                                (unsigned)kbits, st);
  if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 16);
  if (e == hipSuccess)
    e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, idx_in, d_perm, (size_t)n, 0u,
                                  (unsigned)kbits, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(bucket_bounds_kernel, dim3(blocks), dim3(256), 0, st, keys_out, n, fine, d_bucket_start,
                       fine > 0 ? d_sub_start : (int *)nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  return e;
}

// ---------------------------------------------------------------------------
// VAQ::encodeImpl, VAQ.cpp:728-748: per subspace, per row, argmin over codes of
// (x_block - c_row).squaredNorm(), strict `<` so the first minimum wins.  The
// reference leaves the summation order to Eigen's vectorised reduction; here
// (and in the oracle) it is the sequential  dist = 0; dist += t*t  over the
// subspace's dimensions, multiply and add unfused.
// grid = (row tile of 256, subspace); one row per thread, its L values in
// registers (LT = compile-time L) or read back from an LDS tile (LT = 0);
// centroids stream through LDS in chunks of 256 (broadcast reads).
// ---------------------------------------------------------------------------
constexpr int ENC_THREADS = 256;
constexpr int ENC_CHUNK = 256;

template <int LT>
__global__ __launch_bounds__(ENC_THREADS) void encode_kernel(
    const float *__restrict__ Xp, int64_t n, int D, int L, const SubDesc *__restrict__ sub,
    const float *__restrict__ cent, int M, uint16_t *__restrict__ codes) {
  extern __shared__ __attribute__((aligned(16))) float esm[];
  const int tid = threadIdx.x;
  const int s = blockIdx.y;
  const SubDesc sd = sub[s];
  const int64_t row = (int64_t)blockIdx.x * ENC_THREADS + tid;
  const bool valid = row < n;
  float *cs = esm;                                  // [ENC_CHUNK][L]
  float *xs = esm + (size_t)ENC_CHUNK * L;          // [L][ENC_THREADS] (LT == 0 only)
  float xr[LT > 0 ? LT : 1];
  const float *xp = Xp + (valid ? row : 0) * D + (int64_t)s * L;
  if (LT > 0) {
#pragma unroll
    for (int j = 0; j < LT; j++) xr[j] = xp[j];
  } else {
    for (int j = 0; j < L; j++) xs[j * ENC_THREADS + tid] = xp[j];
  }
  float bsf = FLT_MAX;
  int best = 0;
  const float *cg = cent + sd.cent_off;
  for (int c0 = 0; c0 < sd.ncent; c0 += ENC_CHUNK) {
    const int cn = sd.ncent - c0 < ENC_CHUNK ? sd.ncent - c0 : ENC_CHUNK;
    __syncthreads();
    for (int e = tid; e < cn * L; e += ENC_THREADS) cs[e] = cg[(size_t)c0 * L + e];
    __syncthreads();
    for (int c = 0; c < cn; c++) {
      const float *cr = cs + c * (LT > 0 ? LT : L);
      float dist = 0.0f;
      if (LT > 0) {
#pragma unroll
        for (int j = 0; j < LT; j++) {
          const float t = xr[j] - cr[j];
          dist += t * t;
        }
      } else {
        for (int j = 0; j < L; j++) {
          const float t = xs[j * ENC_THREADS + tid] - cr[j];
          dist += t * t;
        }
      }
      if (dist < bsf) {
        bsf = dist;
        best = c0 + c;
      }
    }
  }
  if (valid) codes[row * M + s] = (uint16_t)best;
}

hipError_t launch_encode(const float *Xp, int64_t n, int D, int M, int L, const SubDesc *sub,
                         const float *cent, uint16_t *codes, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const dim3 grid((unsigned)((n + ENC_THREADS - 1) / ENC_THREADS), M);
  const size_t lds_reg = (size_t)ENC_CHUNK * L * sizeof(float);
  switch (L) {
  case 4:  hipLaunchKernelGGL(encode_kernel<4>, grid, dim3(ENC_THREADS), lds_reg, st, Xp, n, D, L, sub, cent, M, codes); break;
  case 8:  hipLaunchKernelGGL(encode_kernel<8>, grid, dim3(ENC_THREADS), lds_reg, st, Xp, n, D, L, sub, cent, M, codes); break;
  case 16: hipLaunchKernelGGL(encode_kernel<16>, grid, dim3(ENC_THREADS), lds_reg, st, Xp, n, D, L, sub, cent, M, codes); break;
  case 32: hipLaunchKernelGGL(encode_kernel<32>, grid, dim3(ENC_THREADS), lds_reg, st, Xp, n, D, L, sub, cent, M, codes); break;
  default: {
    const size_t lds = lds_reg + (size_t)L * ENC_THREADS * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(encode_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(encode_kernel<0>, grid, dim3(ENC_THREADS), lds, st, Xp, n, D, L, sub, cent, M, codes);
  }
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Best-first order of the row slices of a multi-slice scan, one list per query
// batch: slices are ranked by the smallest first LUT term (or its per-bucket
// lower bound) that any query of the batch sees in any bucket the slice touches.
// Workgroups are dispatched in blockIdx order, so with this table the first
// ones scan the most promising rows, publish tight thresholds, and the rest
// skip their buckets unread -- it replaces the sampling pre-pass.
// grid = query batches; n_slices <= SLICE_ORDER_MAX (packed-key sort in LDS).
// ---------------------------------------------------------------------------
constexpr int SLICE_ORDER_MAX = 4096;

__global__ __launch_bounds__(256) void slice_order_kernel(
    const float *__restrict__ lut, int lut_floats, int nq, int qb, const int *__restrict__ bstart,
    int n_buckets, int bucket_shift, int64_t slice_rows, int n_slices, int64_t n_rows,
    int *__restrict__ order) {
  __shared__ unsigned key[SLICE_ORDER_MAX];
  __shared__ float lbk[HOT_MAX_BUCKETS];
  const int batch = blockIdx.x, tid = threadIdx.x;
  // per-bucket bound over the batch's queries
  for (int b = tid; b < n_buckets; b += 256) {
    float m = INFINITY;
    for (int q = 0; q < qb; q++) {
      int x = batch * qb + q;
      if (x >= nq) x = nq - 1;
      for (int c = b << bucket_shift; c < ((b + 1) << bucket_shift); c++) {
        const float v = lut[(size_t)x * lut_floats + c];
        m = v < m ? v : m;
      }
    }
    lbk[b] = m;
  }
  __syncthreads();
  int P = 2;
  while (P < n_slices) P <<= 1;
  const unsigned idx_mask = (unsigned)P - 1u;
  for (int s = tid; s < P; s += 256) {
    unsigned k = 0xffffffffu;
    if (s < n_slices) {
      const int64_t r0 = (int64_t)s * slice_rows;
      int64_t r1 = r0 + slice_rows;
      if (r1 > n_rows) r1 = n_rows;
      int lo = 0, hi = n_buckets;  // bucket of r0
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bstart[mid] <= r0) lo = mid; else hi = mid;
      }
      float m = INFINITY;
      for (int b = lo; b < n_buckets && bstart[b] < r1; b++)
        if (bstart[b + 1] > r0 && lbk[b] < m) m = lbk[b];
      k = m == m ? ((__builtin_bit_cast(unsigned, m) & ~idx_mask) | (unsigned)s) : (0xffffffffu & ~idx_mask) | (unsigned)s;
    }
    key[s] = k;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += 256) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const unsigned a = key[i], c = key[j];
        if ((a > c) == ((i & size) == 0)) { key[i] = c; key[j] = a; }
      }
      __syncthreads();
    }
  // padding keys (0xffffffff) sort last; real ones carry their slice index in the low bits
  for (int r = tid; r < n_slices; r += 256) order[(size_t)batch * n_slices + r] = (int)(key[r] & idx_mask);
}

hipError_t launch_slice_order(const float *lut, int lut_floats, int nq, int qb, const int *bstart,
                              int n_buckets, int bucket_shift, int64_t slice_rows, int n_slices,
                              int64_t n_rows, int *order, hipStream_t st) {
  if (n_slices > SLICE_ORDER_MAX || n_buckets > HOT_MAX_BUCKETS) return hipErrorInvalidValue;
  const int nqb = (nq + qb - 1) / qb;
  if (nqb == 0) return hipSuccess;
  hipLaunchKernelGGL(slice_order_kernel, dim3(nqb), dim3(256), 0, st, lut, lut_floats, nq, qb, bstart,
                     n_buckets, bucket_shift, slice_rows, n_slices, n_rows, order);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------
// Query grouping for multi-query passes over a streamed database.  A pass of Qb queries skips a
// bucket only when EVERY query of the pass can skip it, so a pass costs the UNION of its queries'
// buckets.  Queries whose nearest first and second codes agree have nearly the same buckets in
// reach: ordering the queries by (nearest first code, nearest second code) before cutting them
// into passes shrinks that union (125M x 16 B, 1024 queries, Qb = 4: 35.5 -> 28.0 ms).  Results
// are written to the queries' own slots, so the caller sees no difference.
// query_order_kernel: one workgroup; keys = (argmin of LUT table 0) << 16 | (argmin of table 1),
// then a bitonic sort of key << 32 | query in LDS (nq <= 16384: 128 KB); order[i] = i-th query.
// ---------------------------------------------------------------------------
constexpr int QORDER_THREADS = 1024;
constexpr int QORDER_MAX = 16384;
__global__ __launch_bounds__(QORDER_THREADS) void query_order_kernel(const float *__restrict__ lut, int lut_floats, int nq,
                                                                     int n0, int off1, int n1, int *__restrict__ order) {
  extern __shared__ unsigned long long qk[];  // [P], P = power of two >= nq
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = QORDER_THREADS / 64;
  int P = 2;
  while (P < nq) P <<= 1;
  for (int q = wave; q < P; q += nwaves) {
    unsigned long long key = ~0ull;
    if (q < nq) {
      const float *l = lut + (size_t)q * lut_floats;
      unsigned a[2];
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int off = t == 0 ? 0 : off1, n = t == 0 ? n0 : n1;
        // packed (value bits, index): entries are >= 0, so bit order == value order; NaN sorts last
        unsigned long long best = ~0ull;
        for (int e = lane; e < n; e += 64) {
          const float x = l[off + e];
          const unsigned long long c = ((unsigned long long)(x == x ? float_to_bits(x) : 0xffffffffu) << 32) | (unsigned)e;
          best = c < best ? c : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const unsigned long long x = __shfl_xor(best, o);
          best = x < best ? x : best;
        }
        a[t] = (unsigned)(best & 0xffffu);
      }
      key = ((unsigned long long)((a[0] << 16) | a[1]) << 32) | (unsigned)q;
    }
    if (lane == 0) qk[q] = key;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += QORDER_THREADS) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const unsigned long long x = qk[i], y = qk[j];
        if ((x > y) == ((i & size) == 0)) { qk[i] = y; qk[j] = x; }
      }
      __syncthreads();
    }
  for (int i = tid; i < nq; i += QORDER_THREADS) order[i] = (int)(qk[i] & 0xffffffffu);
}

hipError_t launch_query_order(const float *lut, int lut_floats, int nq, int n0, int off1, int n1, int *order,
                              hipStream_t st) {
  if (nq <= 0 || nq > QORDER_MAX) return hipErrorInvalidValue;
  int P = 2;
  while (P < nq) P <<= 1;
  const size_t lds = (size_t)P * sizeof(unsigned long long);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(query_order_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(query_order_kernel, dim3(1), dim3(QORDER_THREADS), lds, st, lut, lut_floats, nq, n0, off1, n1, order);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Dispatch order of the best-first form with one workgroup per query: expensive queries first.
// A query's cost spans 6x (C2: 318 wave steps on average, 1857 for the top 1 %) and a launch ends
// with its most expensive workgroups; started first they end inside the bulk (C2 scan 0.75 -> 0.55 ms
// with this predictor, 0.50 with the exact costs: tools/exp_cost_predictor.py).  What makes a
// query expensive is the number of buckets in its reach, i.e. how FLAT its first lookup table is
// near the minimum: cost key = (sum of the 16 smallest - 16 x the smallest) of the per-bucket minima of
// table 0 (of 256 of them, sampled, when there are more),
// ascending (Spearman 0.72 with a workgroup's lifetime, 0.83 with its steps).
//   query_cost_kernel   one wave per query -> key bits << 32 | query
//   cost_sort_kernel    one workgroup: counting sort by the key's top bits -> order[b] = query of block b
// Block b serves order[b]: blocks are dealt round-robin over the XCDs and dispatched in order, so
// every XCD gets every 8th query of the ranking.  Speed only: any order is correct.
// ---------------------------------------------------------------------------
template <int NREG>  // per-bucket minima a lane holds: buckets at the level of the first code <= 64 NREG
__global__ __launch_bounds__(256) void query_cost_kernel(const float *__restrict__ lut, int lut_floats, int nq, int n0,
                                                         int shift, unsigned long long *__restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const float *l = lut + (size_t)q * lut_floats;
  // (more than 256 buckets at the level of the first code: every stride-th one is sampled -- the
  //  statistic is a ranking aid, and reading a 4096-entry table per query costs more than it saves)
  const int nb_all = n0 >> shift;
  const int stride = nb_all > 64 * NREG ? nb_all / (64 * NREG) : 1;
  const int nb = nb_all / stride;
  unsigned v[NREG];
  unsigned vmin = 0xffffffffu, vmax = 0u;
#pragma unroll
  for (int i = 0; i < NREG; i++) {
    const int b = i * 64 + lane;
    unsigned m = 0xffffffffu;  // (absent or NaN: never counted)
    if (b < nb) {
      // (sampled in runs of one 64-byte line of the table, so that the unsampled lines are never read)
      const int g = (16 >> shift) > 0 ? (16 >> shift) : 1;
      const int bb = (b / g) * g * stride + (b % g);
      for (int c = bb << shift; c < ((bb + 1) << shift); c++) {
        const float x = l[c];
        const unsigned xb = x == x ? float_to_bits(x) : 0xffffffffu;  // entries are >= 0: bit order == value order
        m = xb < m ? xb : m;
      }
    }
    v[i] = m;
    vmin = m < vmin ? m : vmin;
    vmax = (m != 0xffffffffu && m > vmax) ? m : vmax;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned x = (unsigned)__shfl_xor((int)vmin, o), y = (unsigned)__shfl_xor((int)vmax, o);
    vmin = x < vmin ? x : vmin;
    vmax = y > vmax ? y : vmax;
  }
  int J = 16 / stride;
  J = J < 2 ? 2 : J;
  J = nb < J ? nb : J;
  unsigned lo = vmin, hi = vmax;  // smallest t with count(v <= t) >= J (vmax: all of them, unless NaNs leave fewer)
  if (lo > hi) lo = hi;
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    int c = 0;
#pragma unroll
    for (int i = 0; i < NREG; i++) c += __popcll(__ballot(v[i] <= mid));
    if (c >= J) hi = mid;
    else lo = mid + 1u;
  }
  // key = sum of the J smallest minima - J x the smallest: how flat the table is near its minimum
  // (tools/exp_cost_predictor.py: the batch in this order 0.532 ms, by the J-th smallest alone 0.556)
  float part = 0.0f;
  int below = 0;
#pragma unroll
  for (int i = 0; i < NREG; i++) {
    const bool lt = v[i] < lo;
    part += lt ? bits_to_float(v[i]) : 0.0f;
    below += __popcll(__ballot(lt));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  const float v0 = bits_to_float(vmin < 0x7f800000u ? vmin : 0x7f800000u);
  const float spread = (part + (float)(J - below) * bits_to_float(lo)) - (float)J * v0;
  const unsigned kb = (spread == spread && spread >= 0.0f) ? float_to_bits(spread) : (spread < 0.0f ? 0u : 0x7f800000u);  // (inf - inf: last)
  if (lane == 0) keys[q] = ((unsigned long long)kb << 32) | (unsigned)q;
}

// One workgroup: counting sort of the queries by the top 12 value bits of their cost key (sign 0,
// exponent, 4 mantissa bits: 6 % steps -- the ranking only has to put expensive queries ahead of cheap
// ones; a full bitonic sort of 16 k keys in one workgroup takes longer than the launch it speeds up).
constexpr int COST_CLASSES = 4096;
__global__ __launch_bounds__(QORDER_THREADS) void cost_sort_kernel(const unsigned long long *__restrict__ keys, int nq,
                                                                   int *__restrict__ order) {
  __shared__ unsigned hist[COST_CLASSES];
  __shared__ unsigned wave_tot[QORDER_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < COST_CLASSES; i += QORDER_THREADS) hist[i] = 0u;
  __syncthreads();
  for (int i = tid; i < nq; i += QORDER_THREADS) atomicAdd(&hist[(unsigned)(keys[i] >> 51) & (COST_CLASSES - 1)], 1u);
  __syncthreads();
  // exclusive prefix over the classes: 4 per thread, then a wave scan, then the waves' totals
  constexpr int PER = COST_CLASSES / QORDER_THREADS;
  unsigned c[PER], sum = 0u;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    c[j] = hist[tid * PER + j];
    sum += c[j];
  }
  unsigned inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned x = (unsigned)__shfl_up((int)inc, o);
    if (lane >= o) inc += x;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  unsigned base = inc - sum;
  for (int w = 0; w < wave; w++) base += wave_tot[w];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    hist[tid * PER + j] = base;
    base += c[j];
  }
  __syncthreads();
  for (int i = tid; i < nq; i += QORDER_THREADS) {
    const unsigned long long kq = keys[i];
    const unsigned pos = atomicAdd(&hist[(unsigned)(kq >> 51) & (COST_CLASSES - 1)], 1u);
    order[pos] = (int)(kq & 0xffffffffu);
  }
}

hipError_t launch_cost_order(const float *lut, int lut_floats, int nq, int n0, int shift, unsigned long long *keys,
                             int *order, hipStream_t st) {
  if (nq <= 0 || nq > QORDER_MAX || (n0 >> shift) > 1024 || (n0 >> shift) < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(query_cost_kernel<4>, dim3((nq + 3) / 4), dim3(256), 0, st, lut, lut_floats, nq, n0, shift, keys);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cost_sort_kernel, dim3(1), dim3(QORDER_THREADS), 0, st, keys, nq, order);
  return hipGetLastError();
}

// __global__ entry points: the SGPR-capped one for EA_NONE / EA_QUEUE, a plain one for EA_INPLACE
static int rows_per_item(int layout, int M) { return (layout == LAYOUT_BYTES && M < 16) ? 16 / M : 1; }

// rows the code buffer and every slice are padded to: one step of the largest workgroup
int scan_wg_step_rows(int layout, int M) { return SCAN_MAX_THREADS * rows_per_item(layout, M); }

// code dwords a survivor-queue entry carries besides the row id and the partial sums: the
// rest of the row for byte codes of up to 16 subspaces (phase B then needs no re-read)
static int queue_code_words(int layout, int M, int ea) {
  return (layout == LAYOUT_BYTES && M <= 16 && ea == EA_QUEUE) ? M / 4 - 1 : 0;
}

void scan_geometry(int layout, int M, int k, int ea, int *kp, int *ccap, int *qcap) {
  int p2 = 1;
  while (p2 < k) p2 <<= 1;
  *kp = p2;
  *ccap = VAQ_CCAP;  // a wave appends at most 64 rows per lock hold
  *qcap = ea == EA_QUEUE ? 128 : 0;  // at most 63 left over + 64 pushed before the next drain
}

// lut_floats: LUT entries staged in LDS (all of them, or the resident prefix of the bit-packed path)
size_t scan_lds_bytes(int layout, int M, int lut_floats, int qb, int k, int ea, int nwaves,
                      int n_buckets, int bucket_shift, int bucket_t) {
  int kp, ccap, qcap;
  scan_geometry(layout, M, k, ea, &kp, &ccap, &qcap);
  size_t lut = (size_t)(layout == LAYOUT_BYTES ? M * 256 : lut_floats) * 4 * qb;
  lut = (lut + 15) & ~(size_t)15;
  const size_t sb = ((size_t)SEL_HDR_WORDS * 4 + (size_t)(kp + ccap) * 8 + 15) & ~(size_t)15;
  const size_t lbb = (bucket_shift > 0 || bucket_t > 0) ? (((size_t)n_buckets * 4 * qb + 15) & ~(size_t)15) : 0;
  return lut + (size_t)qb * sb + lbb + hot_bytes(n_buckets) +
         (size_t)nwaves * qcap * 4 * (1 + qb + queue_code_words(layout, M, ea));
}

size_t scan_ti_lds_bytes(int n_clusters) { return ti_lds_bytes(n_clusters); }

hipError_t launch_scan(const ScanParams &p_in, int *grid_out, hipStream_t st) {
  ScanParams p = p_in;
  p.q_cw_words = queue_code_words(p.layout, p.M, p.ea);
  const int nqb = (p.nq + p.qb - 1) / p.qb;
  const int total = nqb * p.n_slices;
  const int grid = ((total + 7) / 8) * 8;
  if (grid_out) *grid_out = grid;
  if (total == 0) return hipSuccess;
  if (p.bf) return launch_scan_bf(p, grid, st);
  size_t lds = scan_lds_bytes(p.layout, p.M, p.lut_lds_entries, p.qb, p.k, p.ea, p.nwaves, p.n_buckets,
                              p.bucket_shift, p.bucket_t);
  if (p.bucket_t < 0 || p.bucket_t > GMIN_MAX_BITS || (p.bucket_t > 0 && p.bucket_shift != 0))
    return hipErrorInvalidValue;
  if (p.ti) {
    if (p.qb != 1 || p.ea != EA_QUEUE || p.bucket_shift != 0 || p.bucket_t != 0 || p.n_hot != 0 || !p.sqrt_out)
      return hipErrorInvalidValue;
    if (p.ti_cap < 1 || p.ti_cap > p.n_buckets) return hipErrorInvalidValue;
    lds += ti_lds_bytes(p.ti_cap);
  } else if (p.sqrt_out) {
    return hipErrorInvalidValue;  // only the TI kernels select on square roots
  }
  return p.layout == LAYOUT_BYTES ? launch_scan_bytes(p, lds, grid, st) : launch_scan_bits(p, lds, grid, st);
}

// ---------------------------------------------------------------------------
// Final k-min over the candidate lists of one query, output in the order
// heap_reorder produces (ascending; utils/Heap.hpp:322-349), empty slots
// -1 / FLT_MAX.  One workgroup per query; candidates are folded through a
// 2048-entry LDS buffer: [kept | new chunk] -> bitonic sort -> keep k.
// ---------------------------------------------------------------------------
constexpr int MERGE_THREADS = 256;
constexpr int MERGE_CAP = 2048;
constexpr int MERGE_FANIN = 16;  // lists folded by one workgroup (16 x k <= 2048 for k <= 128: one sort)

// grid = (query, group); group g folds lists [g*lists_per_group, ...).
//  final != 0 : write labels (+id_base) / distances with -1 / FLT_MAX in empty slots
//  final == 0 : write the group's k best as an intermediate list (raw ids, sentinels kept)
//  thr_out    : optional [nq] float bits; receives min(thr_out[q], k-th distance) when k rows exist
__global__ __launch_bounds__(MERGE_THREADS) void merge_kernel(
    const float *__restrict__ part_d, const int *__restrict__ part_id, int n_lists,
    int lists_per_group, int64_t list_stride, int64_t query_stride, int k, int64_t id_base,
    int in_final, int final, int32_t *__restrict__ out_id, float *__restrict__ out_d,
    unsigned *__restrict__ thr_out) {
  __shared__ float sd[MERGE_CAP];
  __shared__ int si[MERGE_CAP];
  const int q = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int l0 = g * lists_per_group;
  int l1 = l0 + lists_per_group;
  if (l1 > n_lists) l1 = n_lists;
  const int64_t total = (int64_t)(l1 > l0 ? l1 - l0 : 0) * k;
  int kept = 0;
  int64_t pos = 0;
  while (pos < total) {
    int take = MERGE_CAP - kept;
    if ((int64_t)take > total - pos) take = (int)(total - pos);
    for (int i = tid; i < take; i += MERGE_THREADS) {
      const int64_t c = pos + i;
      const int64_t l = c / k;
      const int j = (int)(c - l * k);
      const size_t a = (size_t)((l0 + l) * list_stride + (int64_t)q * query_stride + j);
      float d = part_d[a];
      int id = part_id[a];
      if (in_final && id < 0) { d = INFINITY; id = ID_SENTINEL; }
      sd[kept + i] = d;
      si[kept + i] = id;
    }
    const int n = kept + take;
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += MERGE_THREADS) { sd[i] = INFINITY; si[i] = ID_SENTINEL; }
    __syncthreads();
    bitonic_sort<true>(sd, si, P, tid, MERGE_THREADS);
    kept = n < k ? n : k;
    pos += take;
  }
  __syncthreads();
  const size_t o = ((size_t)q * gridDim.y + g) * k;
  for (int i = tid; i < k; i += MERGE_THREADS) {
    const bool ok = i < kept && si[i] != ID_SENTINEL;
    if (final) {
      out_id[o + i] = ok ? (int32_t)(si[i] + id_base) : -1;
      out_d[o + i] = ok ? sd[i] : FLT_MAX;
    } else {
      out_id[o + i] = ok ? si[i] : ID_SENTINEL;
      out_d[o + i] = ok ? sd[i] : INFINITY;
    }
  }
  if (thr_out && tid == 0 && kept >= k && si[k - 1] != ID_SENTINEL)
    atomicMin(&thr_out[q], __builtin_bit_cast(unsigned, sd[k - 1]));
}

// ---------------------------------------------------------------------------
// VAQ::refine, VAQ.cpp:849-876: exact squared L2 between the raw query and the
// raw dataset row of each of R candidates, then the same k-min.  The
// reference's squaredNorm order is Eigen's; here (and in the oracle) it is the
// sequential  dist = 0; dist += t*t.  One workgroup per query: thread t owns
// candidates t, t+256, ...; then a bitonic sort of the R (<= 2048) pairs.
// rows == nullptr: candidate vectors are read from `dataset` by label;
// otherwise `rows` holds them gathered as [nq][R][D].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(MERGE_THREADS) void refine_kernel(
    const float *__restrict__ Q, int D, const float *__restrict__ dataset,
    const float *__restrict__ rows, const int32_t *__restrict__ labels_in, int R, int k,
    int32_t *__restrict__ labels, float *__restrict__ dist) {
  __shared__ float sd[MERGE_CAP];
  __shared__ int si[MERGE_CAP];
  extern __shared__ float qs[];
  const int q = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < D; j += MERGE_THREADS) qs[j] = Q[(size_t)q * D + j];
  int P = 2;
  while (P < R) P <<= 1;
  __syncthreads();
  for (int i = tid; i < P; i += MERGE_THREADS) {
    float d = INFINITY;
    int id = ID_SENTINEL;
    if (i < R) {
      const int lab = labels_in[(size_t)q * R + i];
      if (lab >= 0) {
        const float *y = rows ? rows + ((size_t)q * R + i) * D : dataset + (size_t)lab * D;
        float acc = 0.0f;
        for (int j = 0; j < D; j++) {
          const float t = qs[j] - y[j];
          acc += t * t;
        }
        // heap admission rule (VAQ.cpp:867): heap_top > dist with the neutral FLT_MAX
        if (acc < FLT_MAX) { d = acc; id = lab; }
      }
    }
    sd[i] = d;
    si[i] = id;
  }
  __syncthreads();
  bitonic_sort<true>(sd, si, P, tid, MERGE_THREADS);
  for (int i = tid; i < k; i += MERGE_THREADS) {
    const bool ok = i < P && si[i] != ID_SENTINEL;
    labels[(size_t)q * k + i] = ok ? si[i] : -1;
    dist[(size_t)q * k + i] = ok ? sd[i] : FLT_MAX;
  }
}

hipError_t launch_refine(const float *Q, int nq, int D, const float *dataset, const float *rows,
                         const int32_t *labels_in, int R, int k, int32_t *labels, float *dist,
                         hipStream_t st) {
  if (nq == 0) return hipSuccess;
  if (R > MERGE_CAP || R < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(refine_kernel, dim3(nq), dim3(MERGE_THREADS), D * sizeof(float), st, Q, D, dataset,
                     rows, labels_in, R, k, labels, dist);
  return hipGetLastError();
}

// First merge level when the scan left MANY lists per query (one per row slice) that are
// mostly empty because the thresholds were tight: grid = (query, group of 256 lists), one
// thread per list; the per-list entry counts written by the scan let the workgroup gather
// only real candidates -- usually a single sort -- instead of reading 256 x k slots.
constexpr int MERGE_LPG = 256;

__global__ __launch_bounds__(MERGE_THREADS) void merge_compact_kernel(
    const float *__restrict__ part_d, const int *__restrict__ part_id,
    const int *__restrict__ part_cnt, int n_lists, int k, int32_t *__restrict__ out_id,
    float *__restrict__ out_d) {
  __shared__ float sd[MERGE_CAP];
  __shared__ int si[MERGE_CAP];
  __shared__ int pre[MERGE_LPG + 1];
  __shared__ int s_end;
  const int q = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int l0 = g * MERGE_LPG;
  const int nl = (n_lists - l0 < MERGE_LPG) ? n_lists - l0 : MERGE_LPG;
  const int my = (tid < nl) ? part_cnt[(size_t)q * n_lists + l0 + tid] : 0;
  // inclusive scan of the counts (Hillis-Steele over 256 entries)
  pre[tid + 1] = my;
  if (tid == 0) pre[0] = 0;
  __syncthreads();
  for (int off = 1; off < MERGE_LPG; off <<= 1) {
    const int v = (tid >= off) ? pre[tid + 1 - off] : 0;
    __syncthreads();
    pre[tid + 1] += v;
    __syncthreads();
  }
  int kept = 0, start = 0;
  while (start < nl) {
    const int room = MERGE_CAP - kept;
    if (tid == 0) s_end = start + 1;  // one list always fits: counts <= k <= 1024 <= room
    __syncthreads();
    if (tid >= start && tid < nl && pre[tid + 1] - pre[start] <= room) atomicMax(&s_end, tid + 1);
    __syncthreads();
    const int end = s_end;
    if (tid >= start && tid < end) {
      const size_t src = ((size_t)q * n_lists + l0 + tid) * k;
      const int dst = kept + pre[tid] - pre[start];
      for (int i = 0; i < my; i++) {
        sd[dst + i] = part_d[src + i];
        si[dst + i] = part_id[src + i];
      }
    }
    const int n = kept + pre[end] - pre[start];
    int P = 2;
    while (P < n) P <<= 1;
    __syncthreads();
    for (int i = n + tid; i < P; i += MERGE_THREADS) { sd[i] = INFINITY; si[i] = ID_SENTINEL; }
    __syncthreads();
    bitonic_sort<true>(sd, si, P, tid, MERGE_THREADS);
    kept = n < k ? n : k;
    start = end;
    __syncthreads();
  }
  const size_t o = ((size_t)q * gridDim.y + g) * k;
  for (int i = tid; i < k; i += MERGE_THREADS) {
    const bool ok = i < kept && si[i] != ID_SENTINEL;
    out_id[o + i] = ok ? si[i] : ID_SENTINEL;
    out_d[o + i] = ok ? sd[i] : INFINITY;
  }
}

size_t merge_scratch_elems(int n_lists, int nq, int k) {
  size_t lists = 1;  // room for one discarded result list (labels == nullptr)
  // worst case of either first level (compacting groups of 256, or plain groups of 16)
  for (size_t n = (size_t)n_lists; n > (size_t)MERGE_FANIN;) {
    n = (n + MERGE_FANIN - 1) / MERGE_FANIN;
    lists += n;
  }
  return lists * (size_t)nq * k;
}

// ---------------------------------------------------------------------------
// Queries the best-first form cut in two (ScanParams::defer_*).  The first launch wrote the k best
// of the buckets it scanned to labels/dist (the API's format: ascending, -1 / FLT_MAX in empty
// slots, labels with id_base added); the second launch's workgroups wrote n_lists partial lists
// (local labels) of the other buckets -- disjoint rows, so every (distance, label) pair is distinct.
// One workgroup per handed-over query: each thread counts the pairs below its own, compared as one
// 64-bit key, and writes its pair at that position if it is among the k best (VAQ::searchHeap's
// result is the k smallest pairs whatever the order they were met in, VAQ.cpp:1729-1758).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void defer_merge_kernel(const unsigned *__restrict__ defer_count, int defer_cap,
                                                          const DeferRec *__restrict__ defer_list, int n_lists, int k,
                                                          const float *__restrict__ part_d,
                                                          const int *__restrict__ part_id,
                                                          const int *__restrict__ part_cnt, int64_t id_base,
                                                          int32_t *labels, float *dist) {
  extern __shared__ unsigned long long dk[];  // [(1 + n_lists) * k] keys, ~0 = empty
  __shared__ unsigned n_valid;
  const unsigned asked = *defer_count;
  const int cnt = asked < (unsigned)defer_cap ? (int)asked : defer_cap;
  const int e = blockIdx.x;
  if (e >= cnt) return;
  const int q = defer_list[e].q;
  const int total = (1 + n_lists) * k;
  if (threadIdx.x == 0) n_valid = 0u;
  __syncthreads();
  unsigned mine = 0u;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int l = i / k, t = i - l * k;
    unsigned long long key = ~0ull;
    if (l == 0) {
      const int32_t lab = labels[(size_t)q * k + t];
      if (lab >= 0) key = ((unsigned long long)__float_as_uint(dist[(size_t)q * k + t]) << 32) | (unsigned)((int64_t)lab - id_base);
    } else {
      const size_t li = (size_t)e * n_lists + (l - 1);
      if (t < part_cnt[li])
        key = ((unsigned long long)__float_as_uint(part_d[li * k + t]) << 32) | (unsigned)part_id[li * k + t];
    }
    dk[i] = key;
    mine += key != ~0ull ? 1u : 0u;
  }
  if (mine) atomicAdd(&n_valid, mine);
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const unsigned long long ki = dk[i];
    if (ki == ~0ull) continue;
    int rank = 0;
    for (int j = 0; j < total; j++) rank += dk[j] < ki ? 1 : 0;
    if (rank < k) {
      labels[(size_t)q * k + rank] = (int32_t)((int64_t)(int)(unsigned)(ki & 0xffffffffull) + id_base);
      dist[(size_t)q * k + rank] = __uint_as_float((unsigned)(ki >> 32));
    }
  }
  for (int i = (int)n_valid + (int)threadIdx.x; i < k; i += blockDim.x) {
    labels[(size_t)q * k + i] = -1;
    dist[(size_t)q * k + i] = FLT_MAX;
  }
}

hipError_t launch_defer_merge(const unsigned *defer_count, int defer_cap, const DeferRec *defer_list, int n_lists, int k,
                              const float *part_d, const int *part_id, const int *part_cnt, int64_t id_base,
                              int32_t *labels, float *dist, hipStream_t st) {
  const size_t lds = (size_t)(1 + n_lists) * k * sizeof(unsigned long long);
  if (lds > 60 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(defer_merge_kernel, dim3(defer_cap), dim3(256), lds, st, defer_count, defer_cap, defer_list, n_lists,
                     k, part_d, part_id, part_cnt, id_base, labels, dist);
  return hipGetLastError();
}

hipError_t launch_merge(const float *part_d, const int *part_id, const int *part_cnt, int n_lists,
                        int64_t list_stride, int64_t query_stride, int nq, int k, int64_t id_base,
                        int in_final, int32_t *labels, float *dist, unsigned *thr_out,
                        float *scratch_d, int *scratch_id, hipStream_t st) {
  if (nq == 0 || k == 0) return hipSuccess;
  const float *cur_d = part_d;
  const int *cur_id = part_id;
  float *sd = scratch_d;
  int *si = scratch_id;
  if (!labels && (!sd || !si)) return hipErrorInvalidValue;
  if (part_cnt && n_lists > MERGE_FANIN && list_stride == k && query_stride == (int64_t)n_lists * k) {
    // scan partials with per-list counts: compact 256 lists per workgroup first
    const int groups = (n_lists + MERGE_LPG - 1) / MERGE_LPG;
    if (!sd || !si) return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_compact_kernel, dim3(nq, groups), dim3(MERGE_THREADS), 0, st, cur_d, cur_id,
                       part_cnt, n_lists, k, si, sd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    cur_d = sd;
    cur_id = si;
    sd += (size_t)nq * groups * k;
    si += (size_t)nq * groups * k;
    n_lists = groups;
    list_stride = k;
    query_stride = (int64_t)groups * k;
    in_final = 0;
  }
  // fold 64 lists at a time into intermediate lists until one workgroup can finish
  while (n_lists > MERGE_FANIN) {
    const int groups = (n_lists + MERGE_FANIN - 1) / MERGE_FANIN;
    if (!sd || !si) return hipErrorInvalidValue;
    hipLaunchKernelGGL(merge_kernel, dim3(nq, groups), dim3(MERGE_THREADS), 0, st, cur_d, cur_id,
                       n_lists, MERGE_FANIN, list_stride, query_stride, k, (int64_t)0, in_final, 0, si,
                       sd, (unsigned *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    cur_d = sd;
    cur_id = si;
    sd += (size_t)nq * groups * k;
    si += (size_t)nq * groups * k;
    n_lists = groups;
    list_stride = k;
    query_stride = (int64_t)groups * k;
    in_final = 0;
  }
  hipLaunchKernelGGL(merge_kernel, dim3(nq, 1), dim3(MERGE_THREADS), 0, st, cur_d, cur_id, n_lists,
                     n_lists > 0 ? n_lists : 1, list_stride, query_stride, k, id_base, in_final,
                     labels ? 1 : 0, labels ? labels : si, labels ? dist : sd, thr_out);
  return hipGetLastError();
}

} // namespace vaq
