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
//   scan_bytes_kernel   VAQ::searchHeap, 8-bit codes        VAQ.cpp:1729-1758
//   scan_bits_kernel    VAQ::searchHeap, 1..15-bit codes    VAQ.cpp:1729-1758
//   merge_kernel        heap_reorder's sorted output         utils/Heap.hpp:322-349
#include "vaq_kernels.h"

#include <float.h>
#include <limits.h>

namespace vaq {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int QB> struct LutVec;
template <> struct LutVec<1> { typedef float T; };
template <> struct LutVec<2> { typedef f32x2 T; };
template <> struct LutVec<4> { typedef f32x4 T; };

template <int QB> __device__ __forceinline__ float lv_get(const typename LutVec<QB>::T &v, int q);
template <> __device__ __forceinline__ float lv_get<1>(const float &v, int) { return v; }
template <> __device__ __forceinline__ float lv_get<2>(const f32x2 &v, int q) { return v[q]; }
template <> __device__ __forceinline__ float lv_get<4>(const f32x4 &v, int q) { return v[q]; }

template <int QB> __device__ __forceinline__ void lv_set(typename LutVec<QB>::T &v, int q, float x);
template <> __device__ __forceinline__ void lv_set<1>(float &v, int, float x) { v = x; }
template <> __device__ __forceinline__ void lv_set<2>(f32x2 &v, int q, float x) { v[q] = x; }
template <> __device__ __forceinline__ void lv_set<4>(f32x4 &v, int q, float x) { v[q] = x; }

// LDS operations of one wavefront execute in issue order; this only stops the
// compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// strict total order on (distance, id): the contract for ties
__device__ __forceinline__ bool pair_less(float da, int ia, float db, int ib) {
  return (da < db) || (da == db && ia < ib);
}

// ---------------------------------------------------------------------------
// VAQ::ProjectOnEigenVectors, VAQ.hpp:198-201: out = (X * mEigenVectors).real()
// One workgroup per row; thread c owns output column c; fmaf chain over the
// inner index ascending (the reference leaves the order to Eigen's GEMM).
// ---------------------------------------------------------------------------
__global__ void project_kernel(const float *__restrict__ X, int D,
                               const float *__restrict__ E, float *__restrict__ out) {
  extern __shared__ float xs[];
  const int64_t r = blockIdx.x;
  for (int j = threadIdx.x; j < D; j += blockDim.x) xs[j] = X[r * D + j];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float acc = 0.0f;
    for (int j = 0; j < D; j++) acc = __builtin_fmaf(xs[j], E[(size_t)j * D + c], acc);
    out[r * D + c] = acc;
  }
}

hipError_t launch_project(const float *X, int64_t n, int D, const float *E, float *out,
                          hipStream_t st) {
  if (n == 0) return hipSuccess;
  int threads = D < 256 ? ((D + 63) / 64) * 64 : 256;
  hipLaunchKernelGGL(project_kernel, dim3((unsigned)n), dim3(threads), D * sizeof(float), st, X, D,
                     E, out);
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
                                 const float *__restrict__ cent, int lut_floats,
                                 float *__restrict__ lut) {
  extern __shared__ float qs[];
  const int q = blockIdx.x, s = blockIdx.y;
  const SubDesc sd = sub[s];
  for (int j = threadIdx.x; j < L; j += blockDim.x) qs[j] = qproj[(size_t)q * D + (size_t)s * L + j];
  __syncthreads();
  float *out = lut + (size_t)q * lut_floats + sd.lut_off;
  const float *cs = cent + sd.cent_off;
  for (int c = threadIdx.x; c < sd.ncent; c += blockDim.x) {
    const float *y = cs + (size_t)c * L;
    float r;
    if (sd.ncent >= 8) {
      float acc = 0.0f;
      for (int j = 0; j < L; j++) {
        float diff = qs[j] - y[j];
        acc = __builtin_fmaf(diff, diff, acc);
      }
      r = acc;
    } else if (L == 1) {
      r = sqdiff(qs[0], y[0]);
    } else if (L == 2) {
      r = sqdiff(qs[0], y[0]) + sqdiff(qs[1], y[1]);
    } else if (L == 4) {
      r = (sqdiff(qs[0], y[0]) + sqdiff(qs[1], y[1])) + (sqdiff(qs[2], y[2]) + sqdiff(qs[3], y[3]));
    } else if (L == 8) {
      float a0 = sqdiff(qs[0], y[0]) + sqdiff(qs[4], y[4]);
      float a1 = sqdiff(qs[1], y[1]) + sqdiff(qs[5], y[5]);
      float a2 = sqdiff(qs[2], y[2]) + sqdiff(qs[6], y[6]);
      float a3 = sqdiff(qs[3], y[3]) + sqdiff(qs[7], y[7]);
      r = (a0 + a1) + (a2 + a3);
    } else if (L == 12) {
      float a0 = (sqdiff(qs[0], y[0]) + sqdiff(qs[4], y[4])) + sqdiff(qs[8], y[8]);
      float a1 = (sqdiff(qs[1], y[1]) + sqdiff(qs[5], y[5])) + sqdiff(qs[9], y[9]);
      float a2 = (sqdiff(qs[2], y[2]) + sqdiff(qs[6], y[6])) + sqdiff(qs[10], y[10]);
      float a3 = (sqdiff(qs[3], y[3]) + sqdiff(qs[7], y[7])) + sqdiff(qs[11], y[11]);
      r = (a0 + a1) + (a2 + a3);
    } else {
      float res = 0.0f;
      for (int j = 0; j < L; j++) {
        float t = qs[j] - y[j];
        res += t * t;
      }
      r = res;
    }
    out[c] = r;
  }
}

hipError_t launch_lut_build(const float *qproj, int nq, int D, int M, int L, const SubDesc *sub,
                            const float *cent, int lut_floats, float *lut, hipStream_t st) {
  if (nq == 0) return hipSuccess;
  hipLaunchKernelGGL(lut_build_kernel, dim3(nq, M), dim3(256), L * sizeof(float), st, qproj, D, L,
                     sub, cent, lut_floats, lut);
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
                                  const SubDesc *__restrict__ sub, uint32_t *__restrict__ out,
                                  int64_t t_begin, int64_t t_end) {
  const int64_t t = t_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= t_end) return;
  uint32_t v = 0;
  if (layout == LAYOUT_BYTES) {
    const int wpr = M / 4;
    const int64_t r = t / wpr;
    const int w = (int)(t - r * wpr);
    if (r >= row_begin && r < row_end) {
      const uint16_t *c = in + (r - row_begin) * M + w * 4;
      v = (uint32_t)(c[0] & 0xff) | ((uint32_t)(c[1] & 0xff) << 8) |
          ((uint32_t)(c[2] & 0xff) << 16) | ((uint32_t)(c[3] & 0xff) << 24);
    }
  } else {
    const int64_t tile = t / ((int64_t)TILE_ROWS * W);
    const int rem = (int)(t - tile * TILE_ROWS * W);
    const int w = rem / TILE_ROWS;
    const int64_t r = tile * TILE_ROWS + (rem % TILE_ROWS);
    if (r >= row_begin && r < row_end) {
      for (int s = 0; s < M; s++) {
        const SubDesc sd = sub[s];
        const uint32_t code = (uint32_t)in[(r - row_begin) * M + s] & (uint32_t)(sd.ncent - 1);
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
                             uint32_t *out, hipStream_t st) {
  const int64_t t_begin = packed_words(row_begin, M, layout, W);
  const int64_t t_end = packed_words(out_row_end, M, layout, W);
  if (t_end <= t_begin) return hipSuccess;
  const int64_t blocks = (t_end - t_begin + 255) / 256;
  hipLaunchKernelGGL(pack_codes_kernel, dim3((unsigned)blocks), dim3(256), 0, st, codes_u16,
                     row_begin, row_end, M, layout, W, sub, out, t_begin, t_end);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Bitonic sort of P (power of two) (distance, id) pairs in LDS, ascending by
// (distance, id).  WG = false: one wavefront, no barriers (LDS is in-order per
// wave); WG = true: the whole workgroup with __syncthreads.
// ---------------------------------------------------------------------------
template <bool WG>
__device__ __forceinline__ void bitonic_sort(float *d, int *id, int P, int tid, int nthreads) {
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int p = tid; p < (P >> 1); p += nthreads) {
        int i = 2 * p - (p & (stride - 1));
        int j = i + stride;
        bool asc = (i & size) == 0;
        float di = d[i], dj = d[j];
        int ii = id[i], ij = id[j];
        bool gt = pair_less(dj, ij, di, ii);
        if (gt == asc) {
          d[i] = dj; d[j] = di;
          id[i] = ij; id[j] = ii;
        }
      }
      if (WG) __syncthreads();
      else wave_lds_sync();
    }
  }
}

// ---------------------------------------------------------------------------
// Per-wavefront running k-min of VAQ::searchHeap (VAQ.cpp:1750-1753 with
// utils/Heap.hpp:115-169): a candidate buffer of kcap slots in LDS, private to
// the wave, plus the wave-uniform admission threshold (thr_d, thr_id) = the
// current k-th best.  A row is admitted iff it is strictly below the
// threshold in (distance, id) order -- the reference admits iff
// heap_top > dist, i.e. strictly better than its current k-th.  Initial
// threshold FLT_MAX reproduces heap_heapify's neutral (utils/Heap.hpp:211-235):
// a distance >= FLT_MAX is never admitted.
// Invariant: before a step that can admit up to A rows, cnt <= kcap - A.
// ---------------------------------------------------------------------------
struct WaveSel {
  float *d;
  int *id;
  int cnt;      // wave-uniform
  float thr_d;  // wave-uniform
  int thr_id;   // wave-uniform
};

__device__ __forceinline__ void wavesel_init(WaveSel &s, float *d, int *id) {
  s.d = d;
  s.id = id;
  s.cnt = 0;
  s.thr_d = FLT_MAX;
  s.thr_id = INT_MIN;
}

__device__ __forceinline__ void wavesel_admit(WaveSel &s, float dist, int rid, bool valid, int lane) {
  const bool pass = valid && pair_less(dist, rid, s.thr_d, s.thr_id);
  const unsigned long long m = __ballot(pass);
  if (m != 0ull) {
    const int pos = s.cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (pass) {
      s.d[pos] = dist;
      s.id[pos] = rid;
    }
    s.cnt += __popcll(m);
  }
  (void)lane;
}

// sort the buffer, keep the k best, refresh the threshold
__device__ __forceinline__ void wavesel_prune(WaveSel &s, int k, int lane) {
  int P = 2;
  while (P < s.cnt) P <<= 1;
  for (int i = s.cnt + lane; i < P; i += 64) {
    s.d[i] = INFINITY;
    s.id[i] = ID_SENTINEL;
  }
  wave_lds_sync();
  bitonic_sort<false>(s.d, s.id, P, lane, 64);
  if (s.cnt >= k) {
    s.cnt = k;
    s.thr_d = __builtin_bit_cast(
        float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s.d[k - 1])));
    s.thr_id = __builtin_amdgcn_readfirstlane(s.id[k - 1]);
  }
}

// final: sort what is left and write the wave's k best (sentinel-padded)
__device__ __forceinline__ void wavesel_flush(WaveSel &s, int k, int lane, float *out_d,
                                              int *out_id) {
  wavesel_prune(s, k, lane);
  const int c = s.cnt;
  for (int i = lane; i < k; i += 64) {
    const bool ok = i < c;
    out_d[i] = ok ? s.d[i] : INFINITY;
    out_id[i] = ok ? s.id[i] : ID_SENTINEL;
  }
}

// XCD-aware workgroup -> (slice, query batch) mapping.  Workgroups are dealt
// round-robin over the 8 XCDs, so b and b+8 share an L2; giving XCD x the
// contiguous range [x*G/8, (x+1)*G/8) of virtual ids makes the workgroups that
// stream the same code slice (consecutive query batches) share that L2.  Speed
// only: any placement is correct.
__device__ __forceinline__ int xcd_virtual_id(int b, int G) { return (b & 7) * (G >> 3) + (b >> 3); }

constexpr int SCAN_THREADS = SCAN_WAVES * 64;
constexpr int PREFETCH = 2;  // items loaded ahead of the one being processed

// ---- code-stream item: what one lane consumes per step --------------------
// LAYOUT_BYTES, M subspaces of 8 bits: an item is max(16, M) bytes = 16/M rows
// (M = 8: two rows) or one row (M = 16, 32), loaded as 16-byte dwordx4.
template <int M> struct BytesItem {
  static constexpr int BYTES = M < 16 ? 16 : M;
  static constexpr int ROWS = BYTES / M;
  static constexpr int LOADS = BYTES / 16;
  static constexpr int WPR = M / 4;  // dwords per row
  uint4 w[LOADS];
  __device__ __forceinline__ void load(const uint32_t *codes, int64_t item) {
    const uint4 *c = reinterpret_cast<const uint4 *>(codes) + item * LOADS;
#pragma unroll
    for (int i = 0; i < LOADS; i++) w[i] = c[i];
  }
  __device__ __forceinline__ uint32_t word(int row, int g) const {
    const int idx = row * WPR + g;
    const uint4 x = w[idx / 4];
    const int c = idx % 4;
    return c == 0 ? x.x : c == 1 ? x.y : c == 2 ? x.z : x.w;
  }
};

// ---------------------------------------------------------------------------
// VAQ::searchHeap for 8-bit codes (VAQ.cpp:1729-1758).  Per row:
//   dist = 0; for each group of 4 subspaces: dism = ((l0 + l1) + l2) + l3;
//   dist += dism   (:1737-1748), plain fp32 adds.
// A workgroup stages the LUTs of QB queries in LDS, interleaved per entry
// ([entry][query], so one ds_read_b32/b64/b128 serves all QB queries), and
// its four wavefronts stream the workgroup's row slice with coalesced 16-byte
// loads (wave w takes every 4th KiB), two items prefetched ahead.
// ---------------------------------------------------------------------------
template <int M, int QB>
__global__ __launch_bounds__(SCAN_THREADS) void scan_bytes_kernel(ScanParams p) {
  typedef typename LutVec<QB>::T LT;
  typedef BytesItem<M> Item;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nqb = (p.nq + QB - 1) / QB;
  const int total = nqb * p.n_slices;
  const int v = xcd_virtual_id(blockIdx.x, gridDim.x);
  if (v >= total) return;
  const int slice = v / nqb;
  const int qbatch = v - slice * nqb;

  LT *lut = reinterpret_cast<LT *>(smem);
  constexpr int LUT_ENTRIES = M * 256;
  int qi[QB];
#pragma unroll
  for (int q = 0; q < QB; q++) {
    int x = qbatch * QB + q;
    qi[q] = x < p.nq ? x : p.nq - 1;
  }
  for (int e = tid; e < LUT_ENTRIES; e += SCAN_THREADS) {
    LT val;
#pragma unroll
    for (int q = 0; q < QB; q++) lv_set<QB>(val, q, p.lut[(size_t)qi[q] * p.lut_floats + e]);
    lut[e] = val;
  }

  const int k = p.k, kcap = p.kcap;
  WaveSel sel[QB];
  {
    unsigned char *base = smem + (size_t)LUT_ENTRIES * sizeof(LT) + (size_t)wave * QB * kcap * 8;
#pragma unroll
    for (int q = 0; q < QB; q++) {
      float *d = reinterpret_cast<float *>(base + (size_t)q * kcap * 8);
      wavesel_init(sel[q], d, reinterpret_cast<int *>(d + kcap));
    }
  }
  __syncthreads();

  // slice = [r0, r0 + slice_rows); slice_rows is a multiple of the workgroup
  // step and the code buffer is padded to n_slices * slice_rows rows, so every
  // load is in bounds; rows >= n_rows are masked out.
  const int64_t r0 = (int64_t)slice * p.slice_rows;
  int64_t r1 = r0 + p.slice_rows;
  if (r1 > p.n_rows) r1 = p.n_rows;
  constexpr int STEP_ITEMS = SCAN_THREADS;  // items per workgroup step
  const int64_t item0 = r0 / Item::ROWS + (int64_t)wave * 64 + lane;
  const int64_t n_steps = (r1 > r0) ? ((r1 - r0) + (int64_t)STEP_ITEMS * Item::ROWS - 1) /
                                          ((int64_t)STEP_ITEMS * Item::ROWS)
                                    : 0;
  const int admit_limit = kcap - 64 * Item::ROWS;

  Item pf[PREFETCH];
#pragma unroll
  for (int i = 0; i < PREFETCH; i++)
    if (i < n_steps) pf[i].load(p.codes, item0 + (int64_t)i * STEP_ITEMS);

  for (int64_t st = 0; st < n_steps; st++) {
    const Item cur = pf[0];
#pragma unroll
    for (int i = 0; i + 1 < PREFETCH; i++) pf[i] = pf[i + 1];
    if (st + PREFETCH < n_steps)
      pf[PREFETCH - 1].load(p.codes, item0 + (st + PREFETCH) * STEP_ITEMS);

    const int64_t row0 = (item0 + st * STEP_ITEMS) * Item::ROWS;
#pragma unroll
    for (int r = 0; r < Item::ROWS; r++) {
      float acc[QB];
#pragma unroll
      for (int g = 0; g < Item::WPR; g++) {
        const uint32_t c4 = cur.word(r, g);
        const LT l0 = lut[(g * 4 + 0) * 256 + (c4 & 0xffu)];
        const LT l1 = lut[(g * 4 + 1) * 256 + ((c4 >> 8) & 0xffu)];
        const LT l2 = lut[(g * 4 + 2) * 256 + ((c4 >> 16) & 0xffu)];
        const LT l3 = lut[(g * 4 + 3) * 256 + (c4 >> 24)];
#pragma unroll
        for (int q = 0; q < QB; q++) {
          float dism = lv_get<QB>(l0, q);
          dism += lv_get<QB>(l1, q);
          dism += lv_get<QB>(l2, q);
          dism += lv_get<QB>(l3, q);
          acc[q] = (g == 0) ? dism : acc[q] + dism;  // dist = 0; dist += dism
        }
      }
      const int64_t row = row0 + r;
      const bool valid = row < r1;
#pragma unroll
      for (int q = 0; q < QB; q++) wavesel_admit(sel[q], acc[q], (int)row, valid, lane);
    }
#pragma unroll
    for (int q = 0; q < QB; q++)
      if (sel[q].cnt > admit_limit) wavesel_prune(sel[q], k, lane);
  }

  const int nslots = p.n_slices * SCAN_WAVES;
  const int slot = slice * SCAN_WAVES + wave;
#pragma unroll
  for (int q = 0; q < QB; q++) {
    const int x = qbatch * QB + q;
    if (x < p.nq) {
      const size_t o = ((size_t)x * nslots + slot) * k;
      wavesel_flush(sel[q], k, lane, p.part_d + o, p.part_id + o);
    }
  }
}

// ---------------------------------------------------------------------------
// VAQ::searchHeap for arbitrary 1..15-bit codes (the variance-aware
// non-uniform allocation).  Same arithmetic; codes are bit-packed
// (LAYOUT_BITS: planar 64-row tiles), LUT packed with per-subspace offsets.
// W = dwords per row; one row per lane per step.
// ---------------------------------------------------------------------------
template <int W> struct BitsItem {
  uint32_t w[W];
  __device__ __forceinline__ void load(const uint32_t *codes, int64_t tile, int lane) {
    const uint32_t *tp = codes + tile * (int64_t)(TILE_ROWS * W) + lane;
#pragma unroll
    for (int i = 0; i < W; i++) w[i] = tp[i * TILE_ROWS];
  }
};

template <int W, int QB>
__global__ __launch_bounds__(SCAN_THREADS) void scan_bits_kernel(ScanParams p) {
  typedef typename LutVec<QB>::T LT;
  typedef BitsItem<W> Item;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nqb = (p.nq + QB - 1) / QB;
  const int total = nqb * p.n_slices;
  const int v = xcd_virtual_id(blockIdx.x, gridDim.x);
  if (v >= total) return;
  const int slice = v / nqb;
  const int qbatch = v - slice * nqb;

  LT *lut = reinterpret_cast<LT *>(smem);
  const int lut_entries = p.lut_floats;
  int qi[QB];
#pragma unroll
  for (int q = 0; q < QB; q++) {
    int x = qbatch * QB + q;
    qi[q] = x < p.nq ? x : p.nq - 1;
  }
  for (int e = tid; e < lut_entries; e += SCAN_THREADS) {
    LT val;
#pragma unroll
    for (int q = 0; q < QB; q++) lv_set<QB>(val, q, p.lut[(size_t)qi[q] * p.lut_floats + e]);
    lut[e] = val;
  }
  const int k = p.k, kcap = p.kcap;
  WaveSel sel[QB];
  {
    const size_t lut_bytes = ((size_t)lut_entries * sizeof(LT) + 15) & ~(size_t)15;
    unsigned char *base = smem + lut_bytes + (size_t)wave * QB * kcap * 8;
#pragma unroll
    for (int q = 0; q < QB; q++) {
      float *d = reinterpret_cast<float *>(base + (size_t)q * kcap * 8);
      wavesel_init(sel[q], d, reinterpret_cast<int *>(d + kcap));
    }
  }
  __syncthreads();

  const int64_t r0 = (int64_t)slice * p.slice_rows;
  int64_t r1 = r0 + p.slice_rows;
  if (r1 > p.n_rows) r1 = p.n_rows;
  const int64_t tile0 = r0 / TILE_ROWS + wave;
  const int64_t n_steps =
      (r1 > r0) ? ((r1 - r0) + SCAN_THREADS - 1) / SCAN_THREADS : 0;
  const int admit_limit = kcap - 64;
  const SubDesc *__restrict__ sub = p.sub;
  const int *__restrict__ first_sub = p.first_sub;

  Item pf[PREFETCH];
#pragma unroll
  for (int i = 0; i < PREFETCH; i++)
    if (i < n_steps) pf[i].load(p.codes, tile0 + (int64_t)i * SCAN_WAVES, lane);

  for (int64_t st = 0; st < n_steps; st++) {
    const Item cur = pf[0];
#pragma unroll
    for (int i = 0; i + 1 < PREFETCH; i++) pf[i] = pf[i + 1];
    if (st + PREFETCH < n_steps)
      pf[PREFETCH - 1].load(p.codes, tile0 + (st + PREFETCH) * SCAN_WAVES, lane);

    float acc[QB], dism[QB];
#pragma unroll
    for (int q = 0; q < QB; q++) { acc[q] = 0.0f; dism[q] = 0.0f; }
    int s = 0;
#pragma unroll
    for (int wi = 0; wi < W; wi++) {
      const uint32_t lo = cur.w[wi];
      const uint32_t hi = (wi + 1 < W) ? cur.w[wi + 1 < W ? wi + 1 : wi] : 0u;
      const int s_end = first_sub[wi + 1];
      for (; s < s_end; s++) {
        const SubDesc sd = sub[s];
        const uint32_t c =
            __builtin_amdgcn_alignbit(hi, lo, (unsigned)sd.shift) & (unsigned)(sd.ncent - 1);
        const LT l = lut[sd.lut_off + c];
        const int ph = s & 3;
#pragma unroll
        for (int q = 0; q < QB; q++) {
          const float x = lv_get<QB>(l, q);
          dism[q] = (ph == 0) ? x : dism[q] + x;              // dism = l0; dism += l1..l3
          if (ph == 3) acc[q] = (s == 3) ? dism[q] : acc[q] + dism[q];  // dist += dism
        }
      }
    }
    const int64_t row = (tile0 + st * SCAN_WAVES) * TILE_ROWS + lane;
    const bool valid = row < r1;
#pragma unroll
    for (int q = 0; q < QB; q++) {
      wavesel_admit(sel[q], acc[q], (int)row, valid, lane);
      if (sel[q].cnt > admit_limit) wavesel_prune(sel[q], k, lane);
    }
  }

  const int nslots = p.n_slices * SCAN_WAVES;
  const int slot = slice * SCAN_WAVES + wave;
#pragma unroll
  for (int q = 0; q < QB; q++) {
    const int x = qbatch * QB + q;
    if (x < p.nq) {
      const size_t o = ((size_t)x * nslots + slot) * k;
      wavesel_flush(sel[q], k, lane, p.part_d + o, p.part_id + o);
    }
  }
}

size_t scan_lds_bytes(int layout, int M, int lut_floats, int qb, int kcap) {
  size_t lut = (size_t)(layout == LAYOUT_BYTES ? M * 256 : lut_floats) * 4 * qb;
  lut = (lut + 15) & ~(size_t)15;
  return lut + (size_t)SCAN_WAVES * qb * kcap * 8;
}

int scan_wg_step_rows(int layout, int M) {
  if (layout == LAYOUT_BYTES) return SCAN_THREADS * (M < 16 ? 16 / M : 1);
  return SCAN_THREADS;
}

// candidate rows one wave can admit per query between two prune checks
int scan_admit_per_step(int layout, int M) {
  if (layout == LAYOUT_BYTES) return 64 * (M < 16 ? 16 / M : 1);
  return 64;
}

template <typename K>
static hipError_t launch_scan_kernel(K kernel, const ScanParams &p, size_t lds, int grid,
                                     hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(SCAN_THREADS), lds, st, p);
  return hipGetLastError();
}

#define VAQ_DISPATCH_QB(KERNEL, A)                                                   \
  switch (p.qb) {                                                                    \
  case 1: return launch_scan_kernel(KERNEL<A, 1>, p, lds, grid, st);                 \
  case 2: return launch_scan_kernel(KERNEL<A, 2>, p, lds, grid, st);                 \
  case 4: return launch_scan_kernel(KERNEL<A, 4>, p, lds, grid, st);                 \
  default: return hipErrorInvalidValue;                                              \
  }

hipError_t launch_scan(const ScanParams &p, int *grid_out, hipStream_t st) {
  const int nqb = (p.nq + p.qb - 1) / p.qb;
  const int total = nqb * p.n_slices;
  const int grid = ((total + 7) / 8) * 8;
  if (grid_out) *grid_out = grid;
  if (total == 0) return hipSuccess;
  const size_t lds = scan_lds_bytes(p.layout, p.M, p.lut_floats, p.qb, p.kcap);
  if (p.layout == LAYOUT_BYTES) {
    switch (p.M) {
    case 8:  VAQ_DISPATCH_QB(scan_bytes_kernel, 8)
    case 16: VAQ_DISPATCH_QB(scan_bytes_kernel, 16)
    case 32: VAQ_DISPATCH_QB(scan_bytes_kernel, 32)
    default: return hipErrorInvalidValue;
    }
  }
  switch (p.W) {
  case 1: VAQ_DISPATCH_QB(scan_bits_kernel, 1)
  case 2: VAQ_DISPATCH_QB(scan_bits_kernel, 2)
  case 3: VAQ_DISPATCH_QB(scan_bits_kernel, 3)
  case 4: VAQ_DISPATCH_QB(scan_bits_kernel, 4)
  case 5: VAQ_DISPATCH_QB(scan_bits_kernel, 5)
  case 6: VAQ_DISPATCH_QB(scan_bits_kernel, 6)
  case 7: VAQ_DISPATCH_QB(scan_bits_kernel, 7)
  case 8: VAQ_DISPATCH_QB(scan_bits_kernel, 8)
  default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------
// Final k-min over the candidate lists of one query, output in the order
// heap_reorder produces (ascending; utils/Heap.hpp:322-349), empty slots
// -1 / FLT_MAX.  One workgroup per query; candidates are folded through a
// 2048-entry LDS buffer: [kept | new chunk] -> bitonic sort -> keep k.
// ---------------------------------------------------------------------------
constexpr int MERGE_THREADS = 256;
constexpr int MERGE_CAP = 2048;

__global__ __launch_bounds__(MERGE_THREADS) void merge_kernel(
    const float *__restrict__ part_d, const int *__restrict__ part_id, int n_lists,
    int64_t list_stride, int64_t query_stride, int k, int64_t id_base, int in_final,
    int32_t *__restrict__ labels, float *__restrict__ dist) {
  __shared__ float sd[MERGE_CAP];
  __shared__ int si[MERGE_CAP];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int64_t total = (int64_t)n_lists * k;
  int kept = 0;
  int64_t pos = 0;
  while (pos < total) {
    int take = MERGE_CAP - kept;
    if ((int64_t)take > total - pos) take = (int)(total - pos);
    for (int i = tid; i < take; i += MERGE_THREADS) {
      int64_t c = pos + i;
      int64_t l = c / k;
      int j = (int)(c - l * k);
      size_t a = (size_t)(l * list_stride + (int64_t)q * query_stride + j);
      float d = part_d[a];
      int id = part_id[a];
      if (in_final && id < 0) { d = INFINITY; id = ID_SENTINEL; }
      sd[kept + i] = d;
      si[kept + i] = id;
    }
    int n = kept + take;
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += MERGE_THREADS) { sd[i] = INFINITY; si[i] = ID_SENTINEL; }
    __syncthreads();
    bitonic_sort<true>(sd, si, P, tid, MERGE_THREADS);
    kept = n < k ? n : k;
    pos += take;
  }
  __syncthreads();
  for (int i = tid; i < k; i += MERGE_THREADS) {
    bool ok = i < kept && si[i] != ID_SENTINEL;
    labels[(size_t)q * k + i] = ok ? (int32_t)(si[i] + id_base) : -1;
    dist[(size_t)q * k + i] = ok ? sd[i] : FLT_MAX;
  }
}

hipError_t launch_merge(const float *part_d, const int *part_id, int n_lists, int64_t list_stride,
                        int64_t query_stride, int nq, int k, int64_t id_base, int in_final,
                        int32_t *labels, float *dist, hipStream_t st) {
  if (nq == 0 || k == 0) return hipSuccess;
  hipLaunchKernelGGL(merge_kernel, dim3(nq), dim3(MERGE_THREADS), 0, st, part_d, part_id, n_lists,
                     list_stride, query_stride, k, id_base, in_final, labels, dist);
  return hipGetLastError();
}

} // namespace vaq
