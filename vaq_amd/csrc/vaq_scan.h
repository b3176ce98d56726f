// vaq_scan.h -- device code shared by the scan translation units: the k-min selection state,
// the workgroup scaffolding (ScanCtx) and the two scan bodies (byte codes, bit-packed codes).
// vaq_scan_bytes.hip and vaq_scan_bits.hip instantiate the kernels (split so that the two
// halves compile in parallel); vaq_kernels.hip holds everything else and launch_scan().
// Everything here is a template or an inline device function.
#ifndef VAQ_SCAN_H_
#define VAQ_SCAN_H_

#include "vaq_kernels.h"

#include <float.h>
#include <limits.h>
#include <math.h>

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
// Bitonic networks on (distance, id) pairs in LDS, ascending by (distance, id).
// WG = false: one wavefront, no barriers (LDS is in-order per wave);
// WG = true : the whole workgroup with __syncthreads.
// ---------------------------------------------------------------------------
template <bool WG>
__device__ __forceinline__ void bitonic_stage(float *d, int *id, int P, int size, int stride,
                                              int tid, int nthreads) {
  for (int p = tid; p < (P >> 1); p += nthreads) {
    const int i = 2 * p - (p & (stride - 1));
    const int j = i + stride;
    const bool asc = (i & size) == 0;
    const float di = d[i], dj = d[j];
    const int ii = id[i], ij = id[j];
    const bool gt = pair_less(dj, ij, di, ii);
    if (gt == asc) {
      d[i] = dj; d[j] = di;
      id[i] = ij; id[j] = ii;
    }
  }
  if (WG) __syncthreads();
  else wave_lds_sync();
}

// full sort of P (power of two) entries
template <bool WG>
__device__ __forceinline__ void bitonic_sort(float *d, int *id, int P, int tid, int nthreads) {
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1)
      bitonic_stage<WG>(d, id, P, size, stride, tid, nthreads);
}

// P entries forming a bitonic sequence -> ascending
template <bool WG>
__device__ __forceinline__ void bitonic_merge(float *d, int *id, int P, int tid, int nthreads) {
  for (int stride = P >> 1; stride > 0; stride >>= 1)
    bitonic_stage<WG>(d, id, P, P << 1, stride, tid, nthreads);
}

// ---------------------------------------------------------------------------
// Running k-min of VAQ::searchHeap (VAQ.cpp:1750-1753 with
// utils/Heap.hpp:115-169), ONE per (workgroup, query), in LDS:
//   header   lock, ncand, nbest, thr_d (float bits), thr_id
//   [0, kp)        current best list, ascending, sentinel-padded
//                  (kp = power of two >= k)
//   [kp, kp+ccap)  rows admitted since the last fold (unsorted)
// A row is admitted iff it is strictly below the threshold (thr_d, thr_id) in
// (distance, id) order -- the reference admits iff heap_top > dist, i.e.
// strictly better than its current k-th.  The initial threshold FLT_MAX
// reproduces heap_heapify's neutral element (utils/Heap.hpp:211-235): a
// distance >= FLT_MAX is never admitted.  The threshold is an upper bound on
// the final k-th best of the query, so it may be tightened from ANY source
// (other workgroups): rows at or above it can never be in the result.
//
// Admissions are rare (about k*ln(rows/k) per query over a whole scan), so
// they are serialised by a workgroup lock: a wave that has candidates takes
// the lock, re-tests them against the exact threshold, folds the candidate
// region into the best list when it would overflow, appends, and releases.
// Waves only READ the threshold word while scanning.
// ---------------------------------------------------------------------------
enum { SEL_LOCK = 0, SEL_NCAND = 1, SEL_NBEST = 2, SEL_THR_D = 3, SEL_THR_ID = 4, SEL_HDR_WORDS = 8 };

struct SelView {
  unsigned *hdr;
  float *d;
  int *id;
};

__device__ __forceinline__ size_t sel_bytes(int kp, int ccap) {
  return (size_t)SEL_HDR_WORDS * 4 + (size_t)(kp + ccap) * 8;
}

__device__ __forceinline__ SelView sel_view(unsigned char *base, int kp, int ccap) {
  SelView v;
  v.hdr = reinterpret_cast<unsigned *>(base);
  v.d = reinterpret_cast<float *>(base + SEL_HDR_WORDS * 4);
  v.id = reinterpret_cast<int *>(v.d + kp + ccap);
  return v;
}

__device__ __forceinline__ float bits_to_float(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned float_to_bits(float f) { return __builtin_bit_cast(unsigned, f); }

// Byte B of a code dword, times 1 << SH (the byte offset of its lookup-table entry), in ONE vector
// instruction: the shifter's operand is selected by byte (SDWA).  The compiler writes v_bfe_u32 +
// v_lshl_add_u32 for the same value; in the scan loops that was one instruction in five.
#ifndef VAQ_NO_SDWA
template <int B, int SH> __device__ __forceinline__ unsigned byte_shl(const unsigned c) {
  unsigned r;
  if (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "n"(SH), "v"(c));
  if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "n"(SH), "v"(c));
  if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "n"(SH), "v"(c));
  if (B == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "n"(SH), "v"(c));
  return r;
}
#else
template <int B, int SH> __device__ __forceinline__ unsigned byte_shl(const unsigned c) { return ((c >> (8 * B)) & 0xffu) << SH; }
#endif
// Entry (byte B of c) of the 256-entry float table at LDS byte offset `table`: the address is formed as
// an INTEGER, so that the table's offset lands in the instruction's immediate field -- through a
// pointer into the dynamic LDS block the compiler adds the block's (link-time) base with one more
// instruction per lookup.  The callers' tables start at LDS offset 0 (lds_base_is_zero).
typedef __attribute__((address_space(3))) const float lds_cfloat;
template <int B> __device__ __forceinline__ float lds_lut(const unsigned table, const unsigned c) {
  return *reinterpret_cast<lds_cfloat *>((uintptr_t)(byte_shl<B, 2>(c) + table));
}
// the kernels that address LDS by integer keep their lookup tables first in the dynamic block and
// have no static LDS: the block starts at 0.  Checked once per workgroup (a trap, not a wrong answer).
__device__ __forceinline__ void lds_base_is_zero(const void *dynamic_lds) {
  if ((unsigned)(uintptr_t)dynamic_lds != 0u) __builtin_trap();
}

// (The spin is wave-uniform: only lane 0 tries the lock, but every lane goes round the loop.
//  A spin under `if (lane == 0)` is not a reconvergence point the compiler has to respect --
//  a prototype with such a spin inside a retry loop had lanes 1..63 run ahead of lane 0.)
__device__ __forceinline__ void sel_lock(const SelView &v, int lane) {
  for (;;) {
    unsigned held = 1u;
    if (lane == 0) held = atomicCAS(&v.hdr[SEL_LOCK], 0u, 1u);
    if (__builtin_amdgcn_readfirstlane((int)held) == 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void sel_unlock(const SelView &v, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(&v.hdr[SEL_LOCK], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Fold the candidate region into the best list (caller holds the lock, one
// wave): sort the candidates (padded to cp = power of two >= ncand), take
// min(best[kp-1-j], cand[j]) -- the kp smallest of the union, as a bitonic
// sequence -- and bitonic-merge it back to ascending.  Updates nbest, ncand
// and the threshold.  Returns true when the threshold moved.
__device__ __forceinline__ bool sel_fold(const SelView &v, int k, int kp, int lane) {
  const int ncand = (int)v.hdr[SEL_NCAND];
  if (ncand == 0) return false;
  float *cd = v.d + kp;
  int *ci = v.id + kp;
  int cp = 2;
  while (cp < ncand) cp <<= 1;
  for (int i = ncand + lane; i < cp; i += 64) {
    cd[i] = INFINITY;
    ci[i] = ID_SENTINEL;
  }
  wave_lds_sync();
  bitonic_sort<false>(cd, ci, cp, lane, 64);
  const int n = cp < kp ? cp : kp;  // candidates beyond the kp best of them cannot matter
  for (int j = lane; j < n; j += 64) {
    const int i = kp - 1 - j;
    const float db = v.d[i], dc = cd[j];
    const int ib = v.id[i], ic = ci[j];
    if (pair_less(dc, ic, db, ib)) {
      v.d[i] = dc;
      v.id[i] = ic;
    }
  }
  wave_lds_sync();
  bitonic_merge<false>(v.d, v.id, kp, lane, 64);
  int nb = (int)v.hdr[SEL_NBEST] + ncand;
  nb = nb < k ? nb : k;
  bool moved = false;
  if (nb == k) {
    const float td = v.d[k - 1];
    const int ti = v.id[k - 1];
    const float od = bits_to_float(v.hdr[SEL_THR_D]);
    const int oi = (int)v.hdr[SEL_THR_ID];
    moved = pair_less(td, ti, od, oi);
    if (moved && lane == 0) {
      v.hdr[SEL_THR_D] = float_to_bits(td);
      v.hdr[SEL_THR_ID] = (unsigned)ti;
    }
  }
  if (lane == 0) {
    v.hdr[SEL_NBEST] = (unsigned)nb;
    v.hdr[SEL_NCAND] = 0u;
  }
  wave_lds_sync();
  return moved;
}

// XCD-aware workgroup -> (slice, query batch) mapping.  Workgroups are dealt
// round-robin over the 8 XCDs, so b and b+8 share an L2; giving XCD x the
// contiguous range [x*G/8, (x+1)*G/8) of virtual ids makes the workgroups that
// stream the same code slice (consecutive query batches) share that L2.  Speed
// only: any placement is correct.
__device__ __forceinline__ int xcd_virtual_id(int b, int G) { return (b & 7) * (G >> 3) + (b >> 3); }

constexpr int SCAN_MAX_THREADS = SCAN_MAX_WAVES * 64;
// A CU admits waves by SGPR allocation too: above 80 SGPRs only 6-7 waves fit a
// SIMD instead of 8 (MI355X_MICROARCH.md, "Residency").  The scan kernels are
// latency-bound, so cap them and let the compiler keep the overflow in VGPR lanes.
#ifndef VAQ_SCAN_SGPR_CAP
#define VAQ_SCAN_SGPR_CAP 80
#endif
// (not the in-place form: it is HBM-bound with waves to spare and only pays for the spills)
#define VAQ_SCAN_SGPRS __attribute__((amdgpu_num_sgpr(VAQ_SCAN_SGPR_CAP)))
#ifndef VAQ_STREAM_RING
#define VAQ_STREAM_RING 3  // code items in flight per wave of the streaming (every-bucket) form
#endif
#ifndef VAQ_PREFETCH
#define VAQ_PREFETCH 2
#endif
#ifndef VAQ_CCAP
#define VAQ_CCAP 128
#endif
constexpr int PREFETCH = VAQ_PREFETCH;  // items loaded ahead of the one being processed
constexpr int PHASE_A_SUBS = 2;    // subspaces summed before the first survivor test
#ifndef VAQ_THR_EVERY
#define VAQ_THR_EVERY 8
#endif
constexpr int THR_LOCAL_EVERY = VAQ_THR_EVERY; // steps between reads of the workgroup threshold
constexpr int THR_GLOBAL_EVERY = 64;
#ifndef VAQ_HOT_MAX
#define VAQ_HOT_MAX 32
#endif
constexpr int HOT_MAX = VAQ_HOT_MAX;   // buckets scanned best-first
constexpr int HOT_MAX_BUCKETS = 4096;  // best-first needs 1 << bits[0] <= this (rank scratch, mask)
#ifndef VAQ_HOT_SEG
#define VAQ_HOT_SEG 16
#endif
constexpr int HOT_SEG_STEPS = VAQ_HOT_SEG;  // wave steps per best-first work unit (byte codes: 128..64 rows a step)
// bit-packed rows come 64 a step: shorter units balance the waves better (C3: 2.76 -> 2.70 ms;
// 8 was worse for byte codes, 32 for both)
constexpr int HOT_SEG_STEPS_BITS = VAQ_HOT_SEG / 2;
constexpr int GMIN_MAX_BITS = 4;  // at most this many bits of the second code extend the bucket key
// hot bucket ids, unit prefix, their row ranges, one mask bit per bucket, the ticket
__host__ __device__ inline int hot_mask_words(int n_buckets) { return (n_buckets + 31) / 32; }
__host__ __device__ inline size_t hot_bytes(int n_buckets) {
  return ((size_t)(HOT_MAX * 4 + 1 + hot_mask_words(n_buckets) + 1) * 4 + 15) & ~(size_t)15;
}

// TI form: begin / end / centre distance / farthest member per visited cluster + unit prefix
__host__ __device__ inline size_t ti_lds_bytes(int n_clusters) {
  return ((size_t)n_clusters * 20 + 4 + 15) & ~(size_t)15;
}
// Slack of the TI bound: the reference prunes when bsfK <= qToCCDist - mCodeToCCDist
// (VAQ.cpp:1566); both distances and the row sums carry fp32 rounding (a few ulp per
// summed dimension), so the kernel only prunes when the bound clears the threshold by
// 2^-13 of the operands -- then no admissible row can be lost and the result is the
// exact k-min of the visited rows, whatever order the waves ran in.
constexpr float TI_SLACK = 1.0f / 8192.0f;

// largest x with sqrtf(x) <= t (t >= 0): row sums are compared against it so that the
// partial-sum tests agree exactly with a comparison of square roots.  fl(t * t) is within
// half an ulp of t^2 and the answer within ~2 ulps above it.
__device__ __forceinline__ float sq_bound(float t) {
  if (!(t < 1.8446742e19f)) return FLT_MAX;  // t * t would overflow (includes the neutral FLT_MAX)
  float c = t * t;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float n = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c) + 1u);
    if (sqrtf(n) <= t) c = n;
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (sqrtf(c) > t && c > 0.0f) c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, c) - 1u);
  return c;
}

// VAQ_STATS (diagnostic builds only): per-wave event counts and cycle totals, added into
// ScanParams::stats at the end of the wave.
#ifdef VAQ_STATS
#define STAT_ADD(i, v) cx.st[i] += (unsigned long long)(v)
#define STAT_T0(name) const unsigned long long name = __builtin_readcyclecounter()
#define STAT_T1(i, name) cx.st[i] += __builtin_readcyclecounter() - name
#else
#define STAT_ADD(i, v)
#define STAT_T0(name)
#define STAT_T1(i, name)
#endif
enum { ST_STEPS = 0, ST_ALIVE_A, ST_ALIVE_A2, ST_DRAINS, ST_ADMITS, ST_FOLDS, ST_CYC_TOTAL, ST_CYC_ADMIT,
       ST_CYC_DRAIN, ST_BUCKETS_TESTED, ST_BUCKETS_VISITED, ST_CYC_SETUP, ST_CYC_STEPLOAD, ST_CYC_FOLD,
       ST_CYC_LOCKWAIT, ST_CYC_BOOT, ST_CYC_PREP, ST_CYC_FINAL, ST_CYC_SETUP_LUT, ST_N };

// Shared scaffolding of the two scan kernels: LDS carve-up, threshold
// exchange, survivor queue, admission, result write-out.
// LDS: [LUT][QB x selection state][per wave: survivor queue]
template <int QB, bool SQ> struct ScanCtx {
  typedef typename LutVec<QB>::T LT;
  LT *lut;
  SelView sel[QB];
  float thr_d[QB];   // wave-uniform cached copy of each query's threshold distance (>= exact)
  // TI form: the reference stores and compares sqrt(distance) (VAQ.cpp:1583-1587), and sqrt
  // merges neighbouring floats, so the k-min is kept on (sqrt(dist), label): the selection
  // state and the shared thresholds hold square roots, thr_s caches the threshold itself and
  // thr_d the largest row sum whose square root does not exceed it (for the partial-sum tests)
  static constexpr bool sq = SQ;  // (the TI kernels)
#ifdef VAQ_STATS
  unsigned long long st[ST_N];
#endif
  float thr_s[QB];
  int qi[QB];
  int *q_id;         // survivor queue (wave-private): row id
  float *q_p;        // [QB][qcap]: sum of the row's first group of four subspaces
  uint32_t *q_cw;    // [q_cw_words][qcap]: the row's code dwords 1.. (byte codes, M <= 16)
  int qcap, qcnt;
  int lane, wave, nwaves;
  int k, kp, ccap;
  bool multi_slice;
  unsigned *g_thr;
  const uint32_t *perm;
  // best-first phase: the n_hot buckets whose first term is smallest for this
  // query batch are scanned before the rest (so the thresholds are near-final
  // when the remaining buckets are tested for skipping)
  LT *lb;              // [n_buckets] per-bucket lower bound of the first term (== lut when shift == 0)
  int bshift;          // bucket = first code >> bshift ...
  int bt;              // ... or (bshift == 0) first code << bt | top bt bits of the second code
  unsigned *gmin;      // [QB][1 << bt] float bits: smallest second term of each group of second codes
  int *hot_bucket;     // [HOT_MAX] bucket ids in ascending-key order, -1 = none
  int *hot_pre;        // [HOT_MAX + 1] prefix of segment counts
  int *hot_bs, *hot_be;  // [HOT_MAX] the bucket's rows inside the slice (no global read per segment)
  unsigned *hot_mask;  // [hot_mask_words(n_buckets)] bit b set = bucket b is handled by the hot phase
  unsigned *hot_ticket;
  int n_hot;
  // triangle-inequality form: the query's visiting list (QB == 1)
  int *ti_begin;       // [nv] first index row of the i-th visited cluster
  int *ti_end;         // [nv] one past the last row taken from it
  float *ti_q;         // [nv] query-to-centre distance (qToCCDist)
  float *ti_x0;        // [nv] centre distance of the cluster's first (= farthest) row taken
  int *ti_pre;         // [nv + 1] prefix of work-unit counts
  int ti_nv;           // entries staged (one chunk of at most ti_cap of the visiting list)
  int ti_rows_before;  // rows of the clusters visited in earlier chunks (wave 0 keeps it)

  // Rank the buckets of the slice [r0, r1) by key = min over the batch's queries of
  // the first LUT term and keep the n_hot best.  Uses the (not yet staged) LUT
  // region as scratch: one packed word (key's high bits | bucket) per bucket.
  __device__ __forceinline__ void pick_hot(unsigned char *smem, const ScanParams &p, int r0, int r1,
                                           int seg_rows, int wstep, int tid, int nthreads) {
    const int K0 = p.n_buckets;
    int K0p = 2;
    while (K0p < K0) K0p <<= 1;
    const unsigned idx_mask = (unsigned)K0p - 1u;
    unsigned *tmp = reinterpret_cast<unsigned *>(smem);
    const int *__restrict__ bstart = p.bucket_start;
    for (int b = tid; b < K0p; b += nthreads) {
      unsigned key = 0xffffffffu;
      if (b < K0) {
        const int s0 = bstart[b] > r0 ? bstart[b] : r0;
        const int e0 = bstart[b + 1] < r1 ? bstart[b + 1] : r1;
        if (e0 > s0) {
          // smallest first term (bt > 0: first + second term) any row of the bucket can have,
          // over the batch's queries
          float m = INFINITY;
          if (bt > 0) {
#pragma unroll
            for (int q = 0; q < QB; q++) {
              const float x = p.lut[(size_t)qi[q] * p.lut_floats + (b >> bt)] +
                              bits_to_float(gmin[(q << bt) + (b & ((1 << bt) - 1))]);
              m = x < m ? x : m;
            }
          } else {
            for (int c = b << p.bucket_shift; c < ((b + 1) << p.bucket_shift); c++) {
#pragma unroll
              for (int q = 0; q < QB; q++) {
                const float x = p.lut[(size_t)qi[q] * p.lut_floats + c];
                m = x < m ? x : m;
              }
            }
          }
          key = (float_to_bits(m) & ~idx_mask) | (unsigned)b;  // m >= 0: bit order == value order
          if (m != m) key = 0xffffffffu;
        }
      }
      tmp[b] = key;
    }
    __syncthreads();
    for (int size = 2; size <= K0p; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < (K0p >> 1); t += nthreads) {
          const int i = 2 * t - (t & (stride - 1));
          const int j = i + stride;
          const unsigned a = tmp[i], c = tmp[j];
          if ((a > c) == ((i & size) == 0)) { tmp[i] = c; tmp[j] = a; }
        }
        __syncthreads();
      }
    for (int i = tid; i < HOT_MAX; i += nthreads) {
      const unsigned k = (i < K0p) ? tmp[i] : 0xffffffffu;
      hot_bucket[i] = (i < p.n_hot && k != 0xffffffffu) ? (int)(k & idx_mask) : -1;
    }
    for (int w = tid; w < hot_mask_words(K0); w += nthreads) hot_mask[w] = 0u;
    __syncthreads();
    if (tid < 64) {  // wave 0: one hot bucket per lane, segment counts prefix-summed across lanes
      static_assert(HOT_MAX <= 64, "one lane per hot bucket");
      const int b = tid < HOT_MAX ? hot_bucket[tid] : -1;
      int segs = 0;
      if (b >= 0) {
        const int s0 = bstart[b] > r0 ? bstart[b] : r0;
        const int e0 = bstart[b + 1] < r1 ? bstart[b + 1] : r1;
        segs = (e0 - (s0 & ~(wstep - 1)) + seg_rows - 1) / seg_rows;
        atomicOr(&hot_mask[b >> 5], 1u << (b & 31));
        hot_bs[tid] = s0;
        hot_be[tid] = e0;
      }
      int inc = segs;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o);
        if (tid >= o) inc += v;
      }
      if (tid < HOT_MAX) hot_pre[tid] = inc - segs;
      if (tid == HOT_MAX - 1) hot_pre[HOT_MAX] = inc;
      if (tid == 0) *hot_ticket = 0u;
    }
    __syncthreads();
  }

  __device__ __forceinline__ bool is_hot(int b) const {
    return n_hot > 0 && ((hot_mask[b >> 5] >> (b & 31)) & 1u);
  }

  __device__ __forceinline__ void setup(unsigned char *smem, const ScanParams &p, int lut_entries,
                                        int qbatch, int tid, int nthreads) {
    lane = tid & 63;
    wave = tid >> 6;
    nwaves = nthreads >> 6;
    k = p.k;
    kp = p.kp;
    ccap = p.ccap;
    qcap = p.qcap;
    qcnt = 0;
    multi_slice = p.share_thr != 0;
    g_thr = p.g_thr;
    perm = p.perm;
    lut = reinterpret_cast<LT *>(smem);
    size_t off = ((size_t)lut_entries * sizeof(LT) + 15) & ~(size_t)15;
    const size_t sb = (sel_bytes(p.kp, p.ccap) + 15) & ~(size_t)15;
#pragma unroll
    for (int q = 0; q < QB; q++) {
      const int x = qbatch * QB + q;
      const int xc = x < p.nq ? x : p.nq - 1;
      qi[q] = p.qorder ? p.qorder[xc] : xc;  // (similar queries are grouped into a pass)
      sel[q] = sel_view(smem + off + (size_t)q * sb, p.kp, p.ccap);
      thr_d[q] = FLT_MAX;
      thr_s[q] = FLT_MAX;
      for (int i = tid; i < p.kp; i += nthreads) {
        sel[q].d[i] = INFINITY;
        sel[q].id[i] = ID_SENTINEL;
      }
      if (tid == q) {
        unsigned td = float_to_bits(FLT_MAX);
        int ti = INT_MIN;
        if (multi_slice) {
          const unsigned g = __hip_atomic_load(&g_thr[qi[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (g < td) { td = g; ti = INT_MAX; }
        }
        sel[q].hdr[SEL_LOCK] = 0u;
        sel[q].hdr[SEL_NCAND] = 0u;
        sel[q].hdr[SEL_NBEST] = 0u;
        sel[q].hdr[SEL_THR_D] = td;
        sel[q].hdr[SEL_THR_ID] = (unsigned)ti;
      }
    }
    off += QB * sb;
    n_hot = p.n_hot;
    bshift = p.bucket_shift;
    bt = p.bucket_t;
    lb = (bshift > 0 || bt > 0) ? reinterpret_cast<LT *>(smem + off) : lut;
    if (bshift > 0 || bt > 0) off += ((size_t)p.n_buckets * sizeof(LT) + 15) & ~(size_t)15;
    hot_bucket = reinterpret_cast<int *>(smem + off);
    hot_pre = hot_bucket + HOT_MAX;
    hot_bs = hot_pre + HOT_MAX + 1;
    hot_be = hot_bs + HOT_MAX;
    hot_mask = reinterpret_cast<unsigned *>(hot_be + HOT_MAX);
    hot_ticket = hot_mask + hot_mask_words(p.n_buckets);
    // (borrows the first query's candidate slots: 4 << GMIN_MAX_BITS <= ccap words, consumed by
    //  pick_hot / stage_lut before the first admission writes there)
    gmin = reinterpret_cast<unsigned *>(sel[0].d + p.kp);
    off += hot_bytes(p.n_buckets);
    ti_nv = 0;
    ti_rows_before = 0;
    if (p.ti) {
      ti_begin = reinterpret_cast<int *>(smem + off);
      ti_end = ti_begin + p.ti_cap;
      ti_q = reinterpret_cast<float *>(ti_end + p.ti_cap);
      ti_x0 = ti_q + p.ti_cap;
      ti_pre = reinterpret_cast<int *>(ti_x0 + p.ti_cap);
      off += ti_lds_bytes(p.ti_cap);
    }
    const size_t q_bytes = (size_t)p.qcap * 4 * (1 + QB + p.q_cw_words);
    unsigned char *qb = smem + off + (size_t)wave * q_bytes;
    q_id = reinterpret_cast<int *>(qb);
    q_p = reinterpret_cast<float *>(qb + (size_t)p.qcap * 4);
    q_cw = reinterpret_cast<uint32_t *>(q_p + (size_t)QB * p.qcap);
  }

  // TI form (VAQ::searchTriangleInequality, VAQ.cpp:1548-1560): the clusters this query
  // visits, in order, with the rows taken from each (all of them, or what is left of the
  // row budget) cut into work units of seg_rows rows aligned to the wave step.  The list is
  // staged ti_cap entries at a time starting at entry c0 (one chunk is the normal case: the
  // host sizes ti_cap for int(T * visit); only the until-k-rows rule can make it longer).
  __device__ __forceinline__ void stage_ti(const ScanParams &p, int c0, int seg_rows, int wstep, int tid,
                                           int nthreads) {
    const int T = p.n_buckets;
    const int q = qi[0];
    int nv = p.ti_nvisit[q] - c0;
    if (nv > p.ti_cap) nv = p.ti_cap;
    ti_nv = nv;
    for (int i = tid; i < nv; i += nthreads) {
      const int c = p.ti_order[(size_t)q * T + c0 + i];
      const int b = p.bucket_start[c], e = p.bucket_start[c + 1];
      ti_begin[i] = b;
      ti_end[i] = e;
      ti_q[i] = p.ti_qcc[(size_t)q * T + c0 + i];
      ti_x0[i] = e > b ? p.ti_xcc[b] : 0.0f;
    }
    __syncthreads();
    if (wave == 0) {
      int carry_rows = ti_rows_before, carry_units = 0;
      for (int base = 0; base < nv; base += 64) {
        const int i = base + lane;
        const int b = i < nv ? ti_begin[i] : 0;
        const int n = i < nv ? ti_end[i] - b : 0;
        int inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(inc, o);
          if (lane >= o) inc += v;
        }
        const int before = carry_rows + inc - n;  // rows of the clusters visited earlier
        const int room = p.ti_rowcap > before ? p.ti_rowcap - before : 0;
        const int take = n < room ? n : room;
        const int e = b + take;
        const int units = take > 0 ? (e - (b & ~(wstep - 1)) + seg_rows - 1) / seg_rows : 0;
        int uinc = units;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(uinc, o);
          if (lane >= o) uinc += v;
        }
        if (i < nv) {
          ti_end[i] = e;
          ti_pre[i] = carry_units + uinc - units;
        }
        carry_rows += __builtin_amdgcn_readlane(inc, 63);
        carry_units += __builtin_amdgcn_readlane(uinc, 63);
      }
      ti_rows_before = carry_rows;
      if (lane == 0) {
        ti_pre[nv] = carry_units;
        *hot_ticket = 0u;
      }
    }
    __syncthreads();
  }

  // bt > 0: the minimum of the second table over each group of 1 << (bits1 - bt) codes, per
  // query (entries are >= 0, so their bit patterns order like the values)
  __device__ __forceinline__ void stage_gmin(const ScanParams &p, int off1, int ncent1, int tid, int nthreads) {
    if (bt == 0) return;
    for (int i = tid; i < (QB << bt); i += nthreads) gmin[i] = 0x7f800000u;
    __syncthreads();
    const int w = 31 - __builtin_clz((unsigned)ncent1) - bt;  // log2 of the group size
#pragma unroll
    for (int q = 0; q < QB; q++)
      for (int e = tid; e < ncent1; e += nthreads)
        atomicMin(&gmin[(q << bt) + (e >> w)], float_to_bits(p.lut[(size_t)qi[q] * p.lut_floats + off1 + e]));
    __syncthreads();
  }

  // copy the batch's LUTs into LDS, interleaved per entry (after pick_hot, which borrows the region)
  __device__ __forceinline__ void stage_lut(const ScanParams &p, int lut_entries, int tid, int nthreads) {
    for (int e = tid; e < lut_entries; e += nthreads) {
      LT val;
#pragma unroll
      for (int q = 0; q < QB; q++) lv_set<QB>(val, q, p.lut[(size_t)qi[q] * p.lut_floats + e]);
      lut[e] = val;
    }
    if (bshift > 0) {  // per-bucket lower bounds of the first term
      __syncthreads();
      for (int b = tid; b < p.n_buckets; b += nthreads) {
        LT m = lut[b << bshift];
        for (int c = (b << bshift) + 1; c < ((b + 1) << bshift); c++) {
          const LT x = lut[c];
#pragma unroll
          for (int q = 0; q < QB; q++)
            if (lv_get<QB>(x, q) < lv_get<QB>(m, q)) lv_set<QB>(m, q, lv_get<QB>(x, q));
        }
        lb[b] = m;
      }
    } else if (bt > 0) {  // first term + the smallest second term of the bucket's group
      __syncthreads();
      for (int b = tid; b < p.n_buckets; b += nthreads) {
        const LT l0 = lut[b >> bt];
        LT m;
#pragma unroll
        for (int q = 0; q < QB; q++)
          lv_set<QB>(m, q, lv_get<QB>(l0, q) + bits_to_float(gmin[(q << bt) + (b & ((1 << bt) - 1))]));
        lb[b] = m;
      }
    }
  }

  // re-read the workgroup thresholds; now and then pull in what other
  // workgroups scanning other slices of the same queries have published
  __device__ __forceinline__ void refresh(int64_t st) {
    if ((st & (THR_LOCAL_EVERY - 1)) != 0) return;
#pragma unroll
    for (int q = 0; q < QB; q++) {
      unsigned t = __hip_atomic_load(&sel[q].hdr[SEL_THR_D], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // (every 8 steps while the scan is young: that is when other workgroups' thresholds move most)
      if (multi_slice && wave == 0 && ((st & (THR_GLOBAL_EVERY - 1)) == 0 || st < THR_GLOBAL_EVERY)) {
        const unsigned g = __hip_atomic_load(&g_thr[qi[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g < t) {
          sel_lock(sel[q], lane);
          if (g < sel[q].hdr[SEL_THR_D] && lane == 0) {
            sel[q].hdr[SEL_THR_D] = g;
            sel[q].hdr[SEL_THR_ID] = (unsigned)INT_MAX;  // ties at g stay admissible
          }
          sel_unlock(sel[q], lane);
          t = g;
        }
      }
      set_thr(q, bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)t)));
    }
  }

  // cache a (wave-uniform) threshold read from the selection state
  __device__ __forceinline__ void set_thr(int q, float t) {
    if (!sq) {
      thr_d[q] = t;
      return;
    }
    if (t == thr_s[q]) return;
    thr_s[q] = t;
    thr_d[q] = sq_bound(t);
  }

  // true unless the partial sums rule the row out for every query of the
  // batch: all LUT entries are >= 0 and fp32 addition is monotone, so a
  // partial sum is a lower bound of the final distance (the reference's early
  // abandon, VAQ.cpp:1708, uses the same bound at group granularity).
  __device__ __forceinline__ bool survives(const float (&part)[QB]) const {
    bool a = false;
#pragma unroll
    for (int q = 0; q < QB; q++) a = a || !(part[q] > thr_d[q]);
    return a;
  }

  // final distances of up to 64 rows (one per lane, `srow` = row in the index's
  // bucketed order): admit those strictly below the query's threshold.  Labels
  // are ORIGINAL rows (perm), so ties break as the contract says.
  __device__ __forceinline__ void admit(const float (&dist)[QB], int srow, bool ok) {
    // cheap pre-test against the cached (never tighter than exact) thresholds
    if (__ballot(ok && survives(dist)) == 0ull) return;
#ifdef VAQ_STATS
    const unsigned long long t_adm = __builtin_readcyclecounter();
    st[ST_ADMITS]++;
#endif
    const int rid = (ok && survives(dist) && perm) ? (int)perm[srow] : srow;
    // (wait for that load here: left to the compiler its wait lands after the lock's spin loop,
    //  and the scan loop this rare path rejoins loses its exact load counts -- vaq_scan_bf.h flush())
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt, lgkmcnt untouched
#pragma unroll
    for (int q = 0; q < QB; q++) {
      if (__ballot(ok && !(dist[q] > thr_d[q])) == 0ull) continue;
      const float dq = sq ? sqrtf(dist[q]) : dist[q];
      const SelView &v = sel[q];
#ifdef VAQ_STATS
      const unsigned long long t_lock = __builtin_readcyclecounter();
#endif
      sel_lock(v, lane);
#ifdef VAQ_STATS
      st[ST_CYC_LOCKWAIT] += __builtin_readcyclecounter() - t_lock;
#endif
      float td = bits_to_float(v.hdr[SEL_THR_D]);
      int ti = (int)v.hdr[SEL_THR_ID];
      bool pass = ok && pair_less(dq, rid, td, ti);
      unsigned long long m = __ballot(pass);
      if (m != 0ull) {
        int ncand = (int)v.hdr[SEL_NCAND];
        if (ncand + __popcll(m) > ccap) {
#ifdef VAQ_STATS
          st[ST_FOLDS]++;
          const unsigned long long t_fold = __builtin_readcyclecounter();
          const bool moved_ = sel_fold(v, k, kp, lane);
          st[ST_CYC_FOLD] += __builtin_readcyclecounter() - t_fold;
          if (moved_) {
#else
          if (sel_fold(v, k, kp, lane)) {
#endif
            td = bits_to_float(v.hdr[SEL_THR_D]);
            ti = (int)v.hdr[SEL_THR_ID];
            if (multi_slice && lane == 0) atomicMin(&g_thr[qi[q]], float_to_bits(td));
            pass = pass && pair_less(dq, rid, td, ti);
            m = __ballot(pass);
          }
          ncand = 0;
        }
        if (m != 0ull) {
          const int pos = kp + ncand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (pass) {
            v.d[pos] = dq;
            v.id[pos] = rid;
          }
          if (lane == 0) v.hdr[SEL_NCAND] = (unsigned)(ncand + __popcll(m));
        }
      }
      sel_unlock(v, lane);
      set_thr(q, bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(td))));
    }
#ifdef VAQ_STATS
    st[ST_CYC_ADMIT] += __builtin_readcyclecounter() - t_adm;
#endif
  }

  // compact the lanes with `alive` set into the survivor queue (NCW code dwords ride along)
  template <int NCW>
  __device__ __forceinline__ void push(bool alive, int rid, const float (&acc)[QB], const uint32_t *cw) {
    const unsigned long long m = __ballot(alive);
    if (m != 0ull) {
      const int pos = qcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      if (alive) {
        q_id[pos] = rid;
#pragma unroll
        for (int q = 0; q < QB; q++) q_p[q * qcap + pos] = acc[q];
#pragma unroll
        for (int i = 0; i < NCW; i++) q_cw[i * qcap + pos] = cw[i];
      }
      qcnt += __popcll(m);
    }
  }

  // after every wave is done: wave q folds query q's leftovers and writes the
  // workgroup's k best (sentinel-padded) for the merge kernel
  __device__ __forceinline__ void write_out(const ScanParams &p, int slice, int qbatch) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QB; q++) {
      const int x = qbatch * QB + q;
      if (wave == (q % nwaves) && x < p.nq) {
        if (sel_fold(sel[q], k, kp, lane) && multi_slice && lane == 0)
          atomicMin(&g_thr[qi[q]], sel[q].hdr[SEL_THR_D]);
        if (p.final_labels) {
          // one slice per query: this list IS the result -- write it in the API's
          // format (heap_reorder's: ascending, empty slots -1 / FLT_MAX) and skip the merge
          const size_t o = (size_t)qi[q] * k;
          for (int i = lane; i < k; i += 64) {
            const int id = sel[q].id[i];
            const bool ok = id != ID_SENTINEL;
            p.final_labels[o + i] = ok ? (int32_t)(id + p.id_base) : -1;
            p.final_dist[o + i] = ok ? sel[q].d[i] : FLT_MAX;
          }
        } else {
          const size_t o = ((size_t)qi[q] * p.n_slices + slice) * k;
          for (int i = lane; i < k; i += 64) {
            p.part_d[o + i] = sel[q].d[i];
            p.part_id[o + i] = sel[q].id[i];
          }
          if (lane == 0) p.part_cnt[(size_t)qi[q] * p.n_slices + slice] = (int)sel[q].hdr[SEL_NBEST];
        }
      }
    }
  }
};

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
// VAQ::searchHeap / searchEarlyAbandon for 8-bit codes (VAQ.cpp:1694-1758).
// Per row:  dist = 0; for each group of 4 subspaces:
//             dism = l0; dism += l1; dism += l2; dism += l3; dist += dism
// (:1737-1748), plain fp32 adds.
// A workgroup stages the LUTs of QB queries in LDS, interleaved per entry
// ([entry][query], so one ds_read_b32/b64/b128 serves all QB queries), and
// its four wavefronts stream the workgroup's row slice with coalesced 16-byte
// loads (wave w takes every 4th KiB), two items prefetched ahead.
//
// EA = true (default) is the GPU form of searchEarlyAbandon:
//   A   every lane: dism = l0 + l1 (the two highest-variance subspaces after
//       PCA); a row whose partial sum already exceeds the threshold of every
//       query of the batch is dead
//   A2  live lanes only (EXEC-masked): dism += l2; dism += l3 -> the first
//       group's sum; test again
//   Q   rows still alive are compacted (row id + group sum) into a
//       wave-private LDS queue
//   B   whenever 64 survivors are queued, one per lane: re-read the row's code
//       words (L2-resident, just streamed), add the remaining groups in the
//       reference's order, abandoning after each, and admit to the k-min.
// Results are identical to EA = false, which sums every row completely.
// ---------------------------------------------------------------------------
// STREAM = true: the measurement form of the in-place kernel (`bucket_skip` = 0): every bucket is
// visited, so one launch streams the whole code array once; a separate instantiation so that
// profilers list it under its own name
template <int M, int QB, int EA, bool TI, bool STREAM = false>
__device__ __forceinline__ void scan_bytes_body(const ScanParams &p) {
  typedef typename LutVec<QB>::T LT;
  typedef BytesItem<M> Item;
  constexpr int WPR = Item::WPR;
  // survivors queue the rest of their row (up to 3 dwords) so that phase B reads LDS, not L2
  constexpr int QCW = (EA == EA_QUEUE && M <= 16) ? WPR - 1 : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int nqb = (p.nq + QB - 1) / QB;
  const int total = nqb * p.n_slices;
  const int v = xcd_virtual_id(blockIdx.x, gridDim.x);
  if (v >= total) return;
  const int vslice = v / nqb;
  const int qbatch = v - vslice * nqb;
  // best-first: the vslice-th most promising slice of this batch (or the vslice-th in row order)
  const int slice = p.slice_order ? p.slice_order[(size_t)qbatch * p.n_slices + vslice] : vslice;

  // slice = [r0, r0 + slice_rows); slice_rows is a multiple of the largest
  // workgroup step and the code buffer is padded to a multiple of it, so every
  // load is in bounds; rows >= n_rows are masked out.
  const int r0 = (int)((int64_t)slice * p.slice_stride);
  const int64_t r1l = (int64_t)r0 + p.slice_rows;
  const int r1 = (int)(r1l > p.n_rows ? p.n_rows : r1l);

  ScanCtx<QB, TI> cx;
#ifdef VAQ_STATS
  for (int i = 0; i < ST_N; i++) cx.st[i] = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  cx.setup(smem, p, M * 256, qbatch, tid, nthreads);
  if (!TI && EA != EA_NONE) cx.stage_gmin(p, 256, 256, tid, nthreads);
  if (!TI && EA != EA_NONE && cx.n_hot > 0)
    cx.pick_hot(smem, p, r0, r1, HOT_SEG_STEPS * 64 * Item::ROWS, 64 * Item::ROWS, tid, nthreads);
  if (TI) cx.stage_ti(p, 0, HOT_SEG_STEPS * 64 * Item::ROWS, 64 * Item::ROWS, tid, nthreads);
  cx.stage_lut(p, M * 256, tid, nthreads);
  const LT *lut = cx.lut;
  const int lane = cx.lane, wave = cx.wave;
  __syncthreads();
  cx.refresh(0);  // pick up the seeded / already published thresholds before the first bucket test
#ifdef VAQ_STATS
  cx.st[ST_CYC_SETUP] = __builtin_readcyclecounter() - t_begin;
#endif

  const int step_items = nthreads;  // items per workgroup step
  const int64_t item0 = r0 / Item::ROWS + wave * 64 + lane;
  const int step_rows = step_items * Item::ROWS;
  const int n_steps = (r1 > r0) ? (r1 - r0 + step_rows - 1) / step_rows : 0;
  const uint32_t *__restrict__ codes = p.codes;

  // subspaces [first, last) of group g: dism = l0; dism += l1; dism += l2; dism += l3
  auto group_sum = [&](const uint32_t c4, const int g, const int first, const int last,
                       float (&dism)[QB]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (j < first || j >= last) continue;
      const LT l = lut[(g * 4 + j) * 256 + ((c4 >> (8 * j)) & 0xffu)];
#pragma unroll
      for (int q = 0; q < QB; q++) dism[q] = (j == 0) ? lv_get<QB>(l, q) : dism[q] + lv_get<QB>(l, q);
    }
  };

  // groups 1.. of a row whose first group's sum is in acc[]; then admission
  auto finish = [&](const uint32_t (&cw)[WPR], float (&acc)[QB], const int rid, bool alive) {
#pragma unroll
    for (int g = 1; g < WPR; g++) {
      if (!EA || alive) {
        float dism[QB];
        group_sum(cw[g], g, 0, 4, dism);
#pragma unroll
        for (int q = 0; q < QB; q++) acc[q] = acc[q] + dism[q];  // dist += dism
        if (EA) alive = cx.survives(acc);
      }
    }
    cx.admit(acc, rid, alive);
  };

  // phase B: the top n (<= 64) queue entries, one per lane
  auto drain = [&](const int n) {
    STAT_T0(t_dr);
    STAT_ADD(ST_DRAINS, 1);
    const int base = cx.qcnt - n;
    const bool ok = lane < n;
    const int slot = base + (ok ? lane : 0);
    const int rid = cx.q_id[slot];
    float acc[QB];
#pragma unroll
    for (int q = 0; q < QB; q++) acc[q] = cx.q_p[q * cx.qcap + slot];
    cx.qcnt = base;
    uint32_t cw[WPR];
    cw[0] = 0u;
#pragma unroll
    for (int i = 1; i < WPR; i++)
      cw[i] = QCW > 0 ? cx.q_cw[(i - 1) * cx.qcap + slot] : codes[(int64_t)rid * WPR + i];
    finish(cw, acc, rid, ok);
    STAT_T1(ST_CYC_DRAIN, t_dr);
  };

  if (EA == EA_NONE) {
    Item pf[PREFETCH];
#pragma unroll
    for (int i = 0; i < PREFETCH; i++)
      if (i < n_steps) pf[i].load(p.codes, item0 + (int64_t)i * step_items);
    for (int st = 0; st < n_steps; st++) {
      const Item cur = pf[0];
#pragma unroll
      for (int i = 0; i + 1 < PREFETCH; i++) pf[i] = pf[i + 1];
      if (st + PREFETCH < n_steps)
        pf[PREFETCH - 1].load(p.codes, item0 + (int64_t)(st + PREFETCH) * step_items);
      cx.refresh(st);
      const int row0 = (int)(item0 + (int64_t)st * step_items) * Item::ROWS;
#pragma unroll
      for (int r = 0; r < Item::ROWS; r++) {
        uint32_t cw[WPR];
#pragma unroll
        for (int i = 0; i < WPR; i++) cw[i] = cur.word(r, i);
        float acc[QB];
        group_sum(cw[0], 0, 0, 4, acc);  // dist = 0; dist += dism
        finish(cw, acc, row0 + r, row0 + r < r1);
      }
    }
  } else {
    // Early abandon over the bucketed row order: each wave walks a contiguous
    // part of the slice bucket by bucket (all rows of a bucket share code 0).
    constexpr int WSTEP = 64 * Item::ROWS;  // rows per wave step
    const int per_wave = ((r1 - r0 + cx.nwaves * WSTEP - 1) / (cx.nwaves * WSTEP)) * WSTEP;
    const int w0 = r0 + wave * per_wave;
    const int w1 = (w0 + per_wave < r1) ? w0 + per_wave : r1;
    const int *__restrict__ bstart = p.bucket_start;
    constexpr int SEG_ROWS = HOT_SEG_STEPS * WSTEP;
    // work units (bucket b, rows [pos, be)): first the best-first segments, pulled by
    // ticket so that the waves share them, then this wave's own part of the slice in
    // natural order (minus the buckets already done)
    bool hot_phase = !TI && cx.n_hot > 0;
    const int hot_total = hot_phase ? cx.hot_pre[HOT_MAX] : 0;
    int ti_total = TI ? cx.ti_pre[cx.ti_nv] : 0;
    const int ti_all = TI ? p.ti_nvisit[cx.qi[0]] : 0;
    int ti_cur = 0, ti_c0 = 0;
    const float *__restrict__ xcc = p.ti_xcc;
    int wb = 0, wpos = w0, stepno = 0;
    // bucket starts are read 64 at a time (lane i: the start of bucket cbase + i) and picked
    // out with v_readlane: one global read per 64 buckets instead of a dependent one per bucket
    int cbase = 0, cval = 0;
    if (!TI && w0 < w1) {
      // largest b with bstart[b] <= w0, by two 64-way steps (n_buckets <= 4096)
      const int K0 = p.n_buckets;
      const int stride = (K0 + 63) >> 6;
      int i1 = lane * stride;
      const bool le1 = i1 < K0 && bstart[i1] <= w0;  // monotone in the lane: a prefix of lanes is true
      const int blk = __popcll(__ballot(le1)) - 1;   // bstart[0] = 0 <= w0, so blk >= 0
      int i2 = blk * stride + lane;
      const bool le2 = lane < stride && i2 < K0 && bstart[i2] <= w0;
      wb = blk * stride + __popcll(__ballot(le2)) - 1;
      cbase = wb;
      cval = bstart[(cbase + lane) < K0 ? cbase + lane : K0];
    }
    for (;;) {
      int b = 0, pos, be;
      float qc = 0.0f;  // TI: distance from the query to the centre of the unit's cluster
      if (TI) {
        // work units of the visiting list, nearest clusters first, shared by ticket
        // between the waves (and, unit u = ticket * n_slices + slice, between the
        // workgroups serving this query)
        int t = 0;
        if (lane == 0) t = (int)atomicAdd(cx.hot_ticket, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        const int64_t u64 = (int64_t)t * p.n_slices + slice;
        if (u64 >= ti_total) {
          // this chunk of the visiting list is done; every wave gets here once per chunk
          if (ti_c0 + p.ti_cap >= ti_all) break;
          __syncthreads();
          ti_c0 += p.ti_cap;
          cx.stage_ti(p, ti_c0, SEG_ROWS, WSTEP, tid, nthreads);
          ti_total = cx.ti_pre[cx.ti_nv];
          ti_cur = 0;
          continue;
        }
        const int u = (int)u64;
        int lo = ti_cur, hi = cx.ti_nv;  // largest i with ti_pre[i] <= u
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (cx.ti_pre[mid] <= u) lo = mid; else hi = mid;
        }
        ti_cur = lo;
        const int bs = cx.ti_begin[lo], bend = cx.ti_end[lo];
        qc = cx.ti_q[lo];
        // the whole cluster is out of reach (its farthest member gives the smallest bound):
        // decided from LDS, without touching the unit's rows or their centre distances
        if ((qc - cx.ti_x0[lo]) - TI_SLACK * (qc + cx.ti_x0[lo]) >= cx.thr_s[0]) continue;
        const int al = bs & ~(WSTEP - 1);
        const int j = u - cx.ti_pre[lo];
        pos = al + j * SEG_ROWS;
        if (pos < bs) pos = bs;
        be = al + (j + 1) * SEG_ROWS;
        if (be > bend) be = bend;
        pos = __builtin_amdgcn_readfirstlane(pos);  // (LDS reads land in VGPRs: tell the compiler
        be = __builtin_amdgcn_readfirstlane(be);    //  these are wave-uniform)
        qc = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(qc)));
      } else if (hot_phase) {
        int t = 0;
        if (lane == 0) t = (int)atomicAdd(cx.hot_ticket, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= hot_total) { hot_phase = false; continue; }
        int i = 0;
        while (cx.hot_pre[i + 1] <= t) i++;
        b = cx.hot_bucket[i];
        const int bs = cx.hot_bs[i], bend = cx.hot_be[i];
        const int al = bs & ~(WSTEP - 1);
        const int j = t - cx.hot_pre[i];
        pos = al + j * SEG_ROWS;
        if (pos < bs) pos = bs;
        be = al + (j + 1) * SEG_ROWS;
        if (be > bend) be = bend;
      } else {
        if (wpos >= w1) break;
        {
          int ci = wb + 1 - cbase;
          if (ci >= 64) {
            cbase = wb + 1;
            cval = bstart[(cbase + lane) < p.n_buckets ? cbase + lane : p.n_buckets];
            ci = 0;
          }
          be = __builtin_amdgcn_readlane(cval, __builtin_amdgcn_readfirstlane(ci));
        }
        if (be > w1) be = w1;
        if (be <= wpos) { wb++; continue; }
        b = wb;
        pos = wpos;
        wpos = be;
        wb++;
        if (cx.is_hot(b)) continue;
      }
      {
        {
          // the bucket's first term dism = l0 (or its lower bound) is wave-uniform
          float l0[QB];
#pragma unroll
          for (int q = 0; q < QB; q++) l0[q] = 0.0f;
          float lbq[QB];  // lower bound of every row sum of the bucket
#pragma unroll
          for (int q = 0; q < QB; q++) lbq[q] = 0.0f;
          if (!TI) {
            const LT lbv = cx.lb[b];
            const LT l0v = cx.bt > 0 ? lut[b >> cx.bt] : lbv;  // the rows' (shared) first term
#pragma unroll
            for (int q = 0; q < QB; q++) {
              lbq[q] = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(lv_get<QB>(lbv, q))));
              l0[q] = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(lv_get<QB>(l0v, q))));
            }
          }
          STAT_ADD(ST_BUCKETS_TESTED, 1);
          if (TI || STREAM || p.no_skip || cx.survives(lbq)) {  // otherwise no row of the bucket can be admitted: skip its codes
            STAT_ADD(ST_BUCKETS_VISITED, 1);
            const int base0 = pos & ~(WSTEP - 1);
            const int nst = (be - base0 + WSTEP - 1) / WSTEP;  // wave steps in this bucket segment
            // one wave step: the item's rows against the query batch; false = the unit is over (TI)
            auto do_step = [&](const Item &cur, const float xcur, const int t) -> bool {
              const int base = base0 + t * WSTEP;
              STAT_ADD(ST_STEPS, 1);
              cx.refresh(stepno++);
              if (TI) {
                // VAQ.cpp:1564-1568: rows of a cluster come farthest from the centre first, so
                // the bound qc - xcc only grows from here on: once it clears the threshold the
                // rest of the unit cannot hold an admissible row
                const float bound = (qc - xcur) - TI_SLACK * (qc + xcur);
                if (bound >= cx.thr_s[0]) return false;
              }
              const int row0 = base + lane * Item::ROWS;
              // A: dism = l0; dism += l1, every row of the item, all lanes
              float part[Item::ROWS][QB];
              bool alive[Item::ROWS];
              if (!TI && cx.bshift == 0) {
#pragma unroll
                for (int r = 0; r < Item::ROWS; r++) {
#pragma unroll
                  for (int q = 0; q < QB; q++) part[r][q] = l0[q];
                  group_sum(cur.word(r, 0), 0, 1, PHASE_A_SUBS, part[r]);
                }
              } else {  // coarse buckets: the first term is gathered per row
#pragma unroll
                for (int r = 0; r < Item::ROWS; r++) group_sum(cur.word(r, 0), 0, 0, PHASE_A_SUBS, part[r]);
              }
#pragma unroll
              for (int r = 0; r < Item::ROWS; r++) {
                alive[r] = (row0 + r >= pos) && (row0 + r < be) && cx.survives(part[r]);
                STAT_ADD(ST_ALIVE_A, __popcll(__ballot(alive[r])));
              }
              // A2 + Q per row
#pragma unroll
              for (int r = 0; r < Item::ROWS; r++) {
                bool live = alive[r];
                if (live) {
                  group_sum(cur.word(r, 0), 0, PHASE_A_SUBS, 4, part[r]);
                  live = cx.survives(part[r]);
                }
                STAT_ADD(ST_ALIVE_A2, __popcll(__ballot(live)));
                if (EA == EA_QUEUE) {
                  uint32_t rest[WPR];
#pragma unroll
                  for (int i = 1; i < WPR; i++) rest[i - 1] = cur.word(r, i);
                  cx.template push<QCW>(live, row0 + r, part[r], rest);
                  while (cx.qcnt >= 64) drain(64);  // (per row: the queue never holds more than 127)
                } else {
                  // EA_INPLACE: the live lanes finish their rows where they stand
                  uint32_t cw[WPR];
#pragma unroll
                  for (int i = 0; i < WPR; i++) cw[i] = cur.word(r, i);
                  finish(cw, part[r], row0 + r, live);
                }
              }
              return true;
            };
            if constexpr (STREAM && !TI && VAQ_STREAM_RING > 0) {
              // VAQ_STREAM_RING items in flight per wave in registers of their own, every path
              // issuing the same number of loads (a load past the segment's end re-reads its last
              // item): the steps wait for vmcnt(VAQ_STREAM_RING - 1), not for every load (the
              // two-deep copy loop below has one load in flight while a step computes)
              Item ring[VAQ_STREAM_RING > 0 ? VAQ_STREAM_RING : 1];
              const int last = nst - 1;
#pragma unroll
              for (int u = 0; u < VAQ_STREAM_RING; u++)
                ring[u].load(p.codes, (int64_t)((base0 + (u < last ? u : last) * WSTEP) / Item::ROWS) + lane);
              int t = 0;
              for (; t + VAQ_STREAM_RING < nst; t += VAQ_STREAM_RING) {
#pragma unroll
                for (int u = 0; u < VAQ_STREAM_RING; u++) {
                  do_step(ring[u], 0.0f, t + u);
                  const int nx = t + u + VAQ_STREAM_RING;
                  ring[u].load(p.codes, (int64_t)((base0 + (nx < last ? nx : last) * WSTEP) / Item::ROWS) + lane);
                }
              }
#pragma unroll
              for (int u = 0; u < VAQ_STREAM_RING; u++)
                if (t + u < nst) do_step(ring[u], 0.0f, t + u);
            } else {
              Item pf[PREFETCH];
              float xpf[PREFETCH];  // TI: centre distance of each step's first row (wave-uniform)
#pragma unroll
              for (int i = 0; i < PREFETCH; i++)
                if (i < nst) {
                  pf[i].load(p.codes, (int64_t)((base0 + i * WSTEP) / Item::ROWS) + lane);
                  if (TI) xpf[i] = xcc[i == 0 ? pos : base0 + i * WSTEP];
                }
              for (int t = 0; t < nst; t++) {
                STAT_T0(t_ld);
                const Item cur = pf[0];
#ifdef VAQ_STATS
                asm volatile("" ::"v"(cur.w[0].x));  // the wait for this step's item lands here
                STAT_T1(ST_CYC_STEPLOAD, t_ld);
#endif
                const float xcur =
                    TI ? bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(xpf[0]))) : 0.0f;
#pragma unroll
                for (int i = 0; i + 1 < PREFETCH; i++) {
                  pf[i] = pf[i + 1];
                  if (TI) xpf[i] = xpf[i + 1];
                }
                if (t + PREFETCH < nst) {
                  pf[PREFETCH - 1].load(p.codes, (int64_t)((base0 + (t + PREFETCH) * WSTEP) / Item::ROWS) + lane);
                  if (TI) xpf[PREFETCH - 1] = xcc[base0 + (t + PREFETCH) * WSTEP];
                }
                if (!do_step(cur, xcur, t)) break;
              }
            }
          }
        }
      }
    }
  }
  if (EA == EA_QUEUE && cx.qcnt > 0) drain(cx.qcnt);
  cx.write_out(p, slice, qbatch);
#ifdef VAQ_STATS
  cx.st[ST_CYC_TOTAL] = __builtin_readcyclecounter() - t_begin;
  if (p.stats && lane == 0)
    for (int i = 0; i < ST_N; i++) atomicAdd(&p.stats[i], cx.st[i]);
#endif
}

// ---------------------------------------------------------------------------
// VAQ::searchHeap / searchEarlyAbandon for arbitrary 1..15-bit codes (the
// variance-aware non-uniform allocation).  Same arithmetic and the same
// phases as scan_bytes_kernel; codes are bit-packed (LAYOUT_BITS: planar
// 64-row tiles), the LUT is packed with per-subspace offsets.  W = dwords per
// row; one row per lane per step.  The first group's four fields span at most
// 60 bits, i.e. dwords 0 and 1.
// ---------------------------------------------------------------------------
template <int W> struct BitsItem {
  uint32_t w[W];
  __device__ __forceinline__ void load(const uint32_t *codes, int64_t tile, int lane) {
    const uint32_t *tp = codes + tile * (int64_t)(TILE_ROWS * W) + lane;
#pragma unroll
    for (int i = 0; i < W; i++) w[i] = tp[i * TILE_ROWS];
  }
};

template <int W, int QB, int EA, bool TAIL, bool TI>
__device__ __forceinline__ void scan_bits_body(const ScanParams &p) {
  typedef typename LutVec<QB>::T LT;
  typedef BitsItem<W> Item;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int nqb = (p.nq + QB - 1) / QB;
  const int total = nqb * p.n_slices;
  const int v = xcd_virtual_id(blockIdx.x, gridDim.x);
  if (v >= total) return;
  const int vslice = v / nqb;
  const int qbatch = v - vslice * nqb;
  // best-first: the vslice-th most promising slice of this batch (or the vslice-th in row order)
  const int slice = p.slice_order ? p.slice_order[(size_t)qbatch * p.n_slices + vslice] : vslice;

  const int r0 = (int)((int64_t)slice * p.slice_stride);
  const int64_t r1l = (int64_t)r0 + p.slice_rows;
  const int r1 = (int)(r1l > p.n_rows ? p.n_rows : r1l);

  ScanCtx<QB, TI> cx;
  cx.setup(smem, p, p.lut_lds_entries, qbatch, tid, nthreads);
  if (!TI && EA != EA_NONE && p.M > 1) cx.stage_gmin(p, p.sub[1].lut_off, p.sub[1].ncent, tid, nthreads);
  if (!TI && EA != EA_NONE && cx.n_hot > 0)
    cx.pick_hot(smem, p, r0, r1, HOT_SEG_STEPS_BITS * TILE_ROWS, TILE_ROWS, tid, nthreads);
  if (TI) cx.stage_ti(p, 0, HOT_SEG_STEPS_BITS * TILE_ROWS, TILE_ROWS, tid, nthreads);
  cx.stage_lut(p, p.lut_lds_entries, tid, nthreads);
  const LT *lut = cx.lut;
  const int lane = cx.lane, wave = cx.wave, nwaves = cx.nwaves;
  __syncthreads();
  cx.refresh(0);  // pick up the seeded / already published thresholds before the first bucket test

  const int64_t tile0 = r0 / TILE_ROWS + wave;
  const int n_steps = (r1 > r0) ? (r1 - r0 + nthreads - 1) / nthreads : 0;
  const int M = p.M;
  const SubDesc *__restrict__ sub = p.sub;
  const int *__restrict__ first_sub = p.first_sub;
  const uint32_t *__restrict__ codes = p.codes;

  // one more subspace of the reference's chain.
  //   grouped (VAQ::searchHeap, VAQ.cpp:1737-1748): dism = l0; dism += l1..l3; dist += dism
  //   sequential (BitVecEngine::queryLUT, BitVecEngine.hpp:1296-1300): dist += l_s
  //     (dism mirrors dist so the survivor tests read the same variable in both modes)
  const bool seq = p.seq != 0;
  // table s is staged in LDS when s < lds_subs; larger allocations (e.g. 32 subspaces of up
  // to 13 bits) keep their tail tables in global memory -- only early-abandon survivors ever
  // reach those, and they sit in L2
  const int lds_subs = p.lds_subs;
  // (TAIL = false: every table is resident and this is a plain LDS read)
  auto lookup = [&](const SubDesc &sd, const int s, const uint32_t c) -> LT {
    if (!TAIL || s < lds_subs) return lut[sd.lut_off + c];
    LT v;
#pragma unroll
    for (int q = 0; q < QB; q++) lv_set<QB>(v, q, p.lut[(size_t)cx.qi[q] * p.lut_floats + sd.lut_off + c]);
    return v;
  };
  auto chain = [&](const int s, const LT l, float (&acc)[QB], float (&dism)[QB]) {
    const int ph = s & 3;
#pragma unroll
    for (int q = 0; q < QB; q++) {
      const float x = lv_get<QB>(l, q);
      if (seq) {
        acc[q] = (s == 0) ? x : acc[q] + x;
        dism[q] = acc[q];
      } else {
        dism[q] = (ph == 0) ? x : dism[q] + x;
        if (ph == 3) acc[q] = (s == 3) ? dism[q] : acc[q] + dism[q];
      }
    }
  };

  // phase B: the top n (<= 64) queue entries, one per lane; groups 1.. are
  // extracted from the row's code words, re-read from the planar tiles
  auto drain = [&](const int n) {
    const int base = cx.qcnt - n;
    const bool ok = lane < n;
    const int slot = base + (ok ? lane : 0);
    const int rid = cx.q_id[slot];
    float acc[QB], dism[QB];
#pragma unroll
    for (int q = 0; q < QB; q++) {
      acc[q] = cx.q_p[q * cx.qcap + slot];
      dism[q] = 0.0f;
    }
    cx.qcnt = base;
    const uint32_t *rp = codes + (int64_t)(rid / TILE_ROWS) * (TILE_ROWS * W) + (rid % TILE_ROWS);
    bool alive = ok;
    int cur_word = -1;
    uint32_t lo = 0, hi = 0;
    for (int s = 4; s < M; s++) {
      const SubDesc sd = sub[s];
      if (sd.word != cur_word) {
        cur_word = sd.word;
        lo = rp[cur_word * TILE_ROWS];
        hi = (cur_word + 1 < W) ? rp[(cur_word + 1) * TILE_ROWS] : 0u;
      }
      if (alive) {
        const uint32_t c =
            __builtin_amdgcn_alignbit(hi, lo, (unsigned)sd.shift) & (unsigned)(sd.ncent - 1);
        chain(s, lookup(sd, s, c), acc, dism);
        if (seq || (s & 3) == 3) alive = cx.survives(acc);
      }
    }
    cx.admit(acc, rid, alive);
  };

  // the rest of a row's chain from the words the lane holds (subspaces >= s_from)
  auto tail_inplace = [&](const Item &it, int s_from, float (&acc)[QB], float (&dism)[QB], bool alive,
                          const bool ea) -> bool {
    int s = s_from;
#pragma unroll
    for (int wi = 0; wi < W; wi++) {
      const uint32_t lo = it.w[wi];
      const uint32_t hi = (wi + 1 < W) ? it.w[wi + 1 < W ? wi + 1 : wi] : 0u;
      const int s_end = first_sub[wi + 1];
      if (s < first_sub[wi]) s = first_sub[wi];
      for (; s < s_end; s++) {
        if (s < s_from) continue;
        const SubDesc sd = sub[s];
        if (!ea || alive) {
          const uint32_t c =
              __builtin_amdgcn_alignbit(hi, lo, (unsigned)sd.shift) & (unsigned)(sd.ncent - 1);
          chain(s, lookup(sd, s, c), acc, dism);
          if (ea && (seq || (s & 3) == 3)) alive = cx.survives(acc);
        }
      }
    }
    return alive;
  };

  if (EA == EA_NONE) {
    Item pf[PREFETCH];
#pragma unroll
    for (int i = 0; i < PREFETCH; i++)
      if (i < n_steps) pf[i].load(p.codes, tile0 + (int64_t)i * nwaves, lane);
    for (int st = 0; st < n_steps; st++) {
      const Item cur = pf[0];
#pragma unroll
      for (int i = 0; i + 1 < PREFETCH; i++) pf[i] = pf[i + 1];
      if (st + PREFETCH < n_steps)
        pf[PREFETCH - 1].load(p.codes, tile0 + (int64_t)(st + PREFETCH) * nwaves, lane);
      cx.refresh(st);
      const int row = (int)(tile0 + (int64_t)st * nwaves) * TILE_ROWS + lane;
      float acc[QB], dism[QB];
#pragma unroll
      for (int q = 0; q < QB; q++) { acc[q] = 0.0f; dism[q] = 0.0f; }
      tail_inplace(cur, 0, acc, dism, true, false);
      cx.admit(acc, row, row < r1);
    }
  } else {
    // Early abandon over the bucketed row order (see scan_bytes_kernel)
    const int per_wave = ((r1 - r0 + nwaves * TILE_ROWS - 1) / (nwaves * TILE_ROWS)) * TILE_ROWS;
    const int w0 = r0 + wave * per_wave;
    const int w1 = (w0 + per_wave < r1) ? w0 + per_wave : r1;
    const int *__restrict__ bstart = p.bucket_start;
    // subspaces 1..3 complete the first group (sequential mode may have fewer than 4)
    const SubDesc s0c = sub[0];
    const SubDesc s1 = sub[M > 1 ? 1 : 0], s2 = sub[M > 2 ? 2 : 0], s3 = sub[M > 3 ? 3 : 0];
    constexpr int WSTEP = TILE_ROWS;
    constexpr int SEG_ROWS = HOT_SEG_STEPS_BITS * WSTEP;
    bool hot_phase = !TI && cx.n_hot > 0;
    const int hot_total = hot_phase ? cx.hot_pre[HOT_MAX] : 0;
    int ti_total = TI ? cx.ti_pre[cx.ti_nv] : 0;
    const int ti_all = TI ? p.ti_nvisit[cx.qi[0]] : 0;
    int ti_cur = 0, ti_c0 = 0;
    const float *__restrict__ xcc = p.ti_xcc;
    int wb = 0, wpos = w0, stepno = 0;
    // bucket starts are read 64 at a time (lane i: the start of bucket cbase + i) and picked
    // out with v_readlane: one global read per 64 buckets instead of a dependent one per bucket
    int cbase = 0, cval = 0;
    if (!TI && w0 < w1) {
      // largest b with bstart[b] <= w0, by two 64-way steps (n_buckets <= 4096)
      const int K0 = p.n_buckets;
      const int stride = (K0 + 63) >> 6;
      int i1 = lane * stride;
      const bool le1 = i1 < K0 && bstart[i1] <= w0;  // monotone in the lane: a prefix of lanes is true
      const int blk = __popcll(__ballot(le1)) - 1;   // bstart[0] = 0 <= w0, so blk >= 0
      int i2 = blk * stride + lane;
      const bool le2 = lane < stride && i2 < K0 && bstart[i2] <= w0;
      wb = blk * stride + __popcll(__ballot(le2)) - 1;
      cbase = wb;
      cval = bstart[(cbase + lane) < K0 ? cbase + lane : K0];
    }
    for (;;) {
      int b = 0, pos, be;
      float qc = 0.0f;
      if (TI) {  // see scan_bytes_body
        int t = 0;
        if (lane == 0) t = (int)atomicAdd(cx.hot_ticket, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        const int64_t u64 = (int64_t)t * p.n_slices + slice;
        if (u64 >= ti_total) {
          // this chunk of the visiting list is done; every wave gets here once per chunk
          if (ti_c0 + p.ti_cap >= ti_all) break;
          __syncthreads();
          ti_c0 += p.ti_cap;
          cx.stage_ti(p, ti_c0, SEG_ROWS, WSTEP, tid, nthreads);
          ti_total = cx.ti_pre[cx.ti_nv];
          ti_cur = 0;
          continue;
        }
        const int u = (int)u64;
        int lo = ti_cur, hi = cx.ti_nv;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (cx.ti_pre[mid] <= u) lo = mid; else hi = mid;
        }
        ti_cur = lo;
        const int bs = cx.ti_begin[lo], bend = cx.ti_end[lo];
        qc = cx.ti_q[lo];
        // the whole cluster is out of reach (its farthest member gives the smallest bound):
        // decided from LDS, without touching the unit's rows or their centre distances
        if ((qc - cx.ti_x0[lo]) - TI_SLACK * (qc + cx.ti_x0[lo]) >= cx.thr_s[0]) continue;
        const int al = bs & ~(WSTEP - 1);
        const int j = u - cx.ti_pre[lo];
        pos = al + j * SEG_ROWS;
        if (pos < bs) pos = bs;
        be = al + (j + 1) * SEG_ROWS;
        if (be > bend) be = bend;
        pos = __builtin_amdgcn_readfirstlane(pos);  // (LDS reads land in VGPRs: tell the compiler
        be = __builtin_amdgcn_readfirstlane(be);    //  these are wave-uniform)
        qc = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(qc)));
      } else if (hot_phase) {
        int t = 0;
        if (lane == 0) t = (int)atomicAdd(cx.hot_ticket, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= hot_total) { hot_phase = false; continue; }
        int i = 0;
        while (cx.hot_pre[i + 1] <= t) i++;
        b = cx.hot_bucket[i];
        const int bs = cx.hot_bs[i], bend = cx.hot_be[i];
        const int al = bs & ~(WSTEP - 1);
        const int j = t - cx.hot_pre[i];
        pos = al + j * SEG_ROWS;
        if (pos < bs) pos = bs;
        be = al + (j + 1) * SEG_ROWS;
        if (be > bend) be = bend;
      } else {
        if (wpos >= w1) break;
        {
          int ci = wb + 1 - cbase;
          if (ci >= 64) {
            cbase = wb + 1;
            cval = bstart[(cbase + lane) < p.n_buckets ? cbase + lane : p.n_buckets];
            ci = 0;
          }
          be = __builtin_amdgcn_readlane(cval, __builtin_amdgcn_readfirstlane(ci));
        }
        if (be > w1) be = w1;
        if (be <= wpos) { wb++; continue; }
        b = wb;
        pos = wpos;
        wpos = be;
        wb++;
        if (cx.is_hot(b)) continue;
      }
      {
        {
          float l0[QB];
#pragma unroll
          for (int q = 0; q < QB; q++) l0[q] = 0.0f;
          float lbq[QB];
#pragma unroll
          for (int q = 0; q < QB; q++) lbq[q] = 0.0f;
          if (!TI) {
            const LT lbv = cx.lb[b];  // lower bound of the bucket's row sums
            // the rows' shared first term (subspace 0's table starts the packed LUT), or its bound
            const LT l0v = cx.bt > 0 ? lut[b >> cx.bt] : lbv;
#pragma unroll
            for (int q = 0; q < QB; q++) {
              lbq[q] = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(lv_get<QB>(lbv, q))));
              l0[q] = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(lv_get<QB>(l0v, q))));
            }
          }
          if (TI || p.no_skip || cx.survives(lbq)) {
            const int base0 = pos & ~(TILE_ROWS - 1);
            const int nst = (be - base0 + TILE_ROWS - 1) / TILE_ROWS;
            Item pf[PREFETCH];
            float xpf[PREFETCH];
#pragma unroll
            for (int i = 0; i < PREFETCH; i++)
              if (i < nst) {
                pf[i].load(p.codes, base0 / TILE_ROWS + i, lane);
                if (TI) xpf[i] = xcc[i == 0 ? pos : base0 + i * TILE_ROWS];
              }
            for (int t = 0; t < nst; t++) {
              const Item cur = pf[0];
              const float xcur =
                  TI ? bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(xpf[0]))) : 0.0f;
#pragma unroll
              for (int i = 0; i + 1 < PREFETCH; i++) {
                pf[i] = pf[i + 1];
                if (TI) xpf[i] = xpf[i + 1];
              }
              if (t + PREFETCH < nst) {
                pf[PREFETCH - 1].load(p.codes, base0 / TILE_ROWS + t + PREFETCH, lane);
                if (TI) xpf[PREFETCH - 1] = xcc[base0 + (t + PREFETCH) * TILE_ROWS];
              }
              const int base = base0 + t * TILE_ROWS;
              cx.refresh(stepno++);
              if (TI) {  // VAQ.cpp:1564-1568, see scan_bytes_body
                const float bound = (qc - xcur) - TI_SLACK * (qc + xcur);
                if (bound >= cx.thr_s[0]) break;
              }
              const int row = base + lane;
              const uint32_t w0w = cur.w[0];
              const uint32_t w1w = W > 1 ? cur.w[W > 1 ? 1 : 0] : 0u;
              float acc[QB], dism[QB];
#pragma unroll
              for (int q = 0; q < QB; q++) { acc[q] = l0[q]; dism[q] = l0[q]; }  // dism = l0 / dist = l0
              if (TI || cx.bshift > 0)  // coarse buckets / TI clusters: gather the row's own first term
                chain(0, lut[w0w & (unsigned)(s0c.ncent - 1)], acc, dism);
              // A: dism += l1 (field 1 lies inside dword 0)
              if (M > 1)
                chain(1, lookup(s1, 1, (w0w >> s1.shift) & (unsigned)(s1.ncent - 1)), acc, dism);
              bool live = (row >= pos) && (row < be) && cx.survives(dism);
              if (live && M > 2) {
                // A2: fields 2 and 3 (dwords 0..1) complete the first group
                const uint32_t c2 = (s2.word == 0 ? __builtin_amdgcn_alignbit(w1w, w0w, (unsigned)s2.shift)
                                                  : (w1w >> s2.shift)) & (unsigned)(s2.ncent - 1);
                const uint32_t c3 = (s3.word == 0 ? __builtin_amdgcn_alignbit(w1w, w0w, (unsigned)s3.shift)
                                                  : (w1w >> s3.shift)) & (unsigned)(s3.ncent - 1);
                chain(2, lookup(s2, 2, c2), acc, dism);
                if (M > 3) chain(3, lookup(s3, 3, c3), acc, dism);
                live = cx.survives(seq ? acc : (M > 3 ? acc : dism));
              }
              if (EA == EA_QUEUE) {
                cx.template push<0>(live, row, acc, nullptr);
                while (cx.qcnt >= 64) drain(64);
              } else {
                live = tail_inplace(cur, 4, acc, dism, live, true);
                cx.admit(acc, row, live);
              }
            }
          }
        }
      }
    }
  }
  if (EA == EA_QUEUE && cx.qcnt > 0) drain(cx.qcnt);
  cx.write_out(p, slice, qbatch);
}

template <typename K>
inline hipError_t launch_scan_kernel(K kernel, const ScanParams &p, size_t lds, int grid,
                                     hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(p.nwaves * 64), lds, st, p);
  return hipGetLastError();
}


// byte-code / bit-packed halves of launch_scan (vaq_scan_bytes.hip, vaq_scan_bits.hip);
// `lds` already includes the TI list bytes when p.ti is set
hipError_t launch_scan_bytes(const ScanParams &p, size_t lds, int grid, hipStream_t st);
hipError_t launch_scan_bits(const ScanParams &p, size_t lds, int grid, hipStream_t st);

} // namespace vaq
#endif
