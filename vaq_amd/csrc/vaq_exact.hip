// vaq_exact.hip -- the reference's own choice among rows of EQUAL distance (option "exact_ties").
//
// VAQ::searchHeap (VAQ.cpp:1729-1758) keeps its k best in a binary max-heap (utils/Heap.hpp:115-169:
// pop takes the RIGHT child on equal children, push sifts up on strict >) and admits row i iff
// heap_top > dist_i, rows in ORIGINAL order.  Which of several rows tying at the k-th distance
// survive, and the order heap_reorder (:322-349) returns equal distances in, depend on the heap's
// shape, i.e. on every row it ever admitted -- a rule of the form "smallest label wins" (the scan
// kernels' order) cannot reproduce it.  So for the queries that HAVE ties the admission sequence is
// replayed:
//   exact_flag_kernel    the scan ran with k + 1: a query whose k + 1 smallest distances are all
//                        distinct has a unique answer, already in heap_reorder's order -- copied out.
//                        Any two equal neighbours (inside the top k, or the k-th and the (k+1)-th:
//                        a boundary tie) put the query on the replay list.
//   exact_replay_kernel  one workgroup per listed query walks the rows in ORIGINAL order (through the
//                        inverse of the bucketed order's permutation).  Waves 1.. evaluate a chunk of
//                        consecutive rows -- the complete row sum in the reference's order, abandoned
//                        once a partial sum reaches the heap top (VAQ::searchEarlyAbandon's test, :1708)
//                        -- into an LDS buffer; wave 0 meanwhile replays the PREVIOUS chunk: 64 rows at
//                        a time, ballot of dist < top, and for each set bit in row order the reference's
//                        own statements: if (top > dist) { heap_pop; heap_push }.  The top the evaluating
//                        waves abandon against is always one the heap had BEFORE the rows they look at,
//                        so nothing the reference would admit is ever dropped.  At the end heap_reorder.
// The heap functions below are the reference's (utils/Heap.hpp), statement for statement, as restated
// in oracle/vaq_oracle.c -- which is pinned against the compiled reference heap (tests/test_oracle_golden.py).
#include "vaq_scan.h"

namespace vaq {

constexpr int EX_THREADS = 512;
constexpr int EX_CHUNK = 2048;  // rows per chunk (a multiple of 64)

// utils/Heap.hpp:115-144 (1-based sift-down of the last element from the root; on equal children the
// comparison is false, so the RIGHT child is taken)
__device__ __forceinline__ void ex_heap_pop(const int k, float *val0, int *ids0) {
  float *val = val0 - 1;
  int *ids = ids0 - 1;
  const float v = val[k];
  int i = 1;
  for (;;) {
    const int i1 = i << 1, i2 = i1 + 1;
    if (i1 > k) break;
    if (i2 == k + 1 || val[i1] > val[i2]) {
      if (v > val[i1]) break;
      val[i] = val[i1];
      ids[i] = ids[i1];
      i = i1;
    } else {
      if (v > val[i2]) break;
      val[i] = val[i2];
      ids[i] = ids[i2];
      i = i2;
    }
  }
  val[i] = val[k];
  ids[i] = ids[k];
}

// utils/Heap.hpp:151-169 (sift-up from slot k)
__device__ __forceinline__ void ex_heap_push(const int k, float *val0, int *ids0, const float v, const int id) {
  float *val = val0 - 1;
  int *ids = ids0 - 1;
  int i = k;
  while (i > 1) {
    const int f = i >> 1;
    if (!(v > val[f])) break;
    val[i] = val[f];
    ids[i] = ids[f];
    i = f;
  }
  val[i] = v;
  ids[i] = id;
}

struct ExactParams {
  const uint32_t *codes;
  int layout, M, W;
  const SubDesc *sub;
  const uint32_t *inv;  // original row -> row of the bucketed order (nullptr = identity)
  const unsigned short *row_bucket;  // original row -> its bucket (nullptr: no bucket pruning)
  int n_buckets, bucket_shift, bucket_t;
  int64_t n_rows;
  const float *lut;     // [nq][lut_floats]
  int lut_floats;
  int lut_in_lds;
  int seq;
  int k;
  int64_t id_base;
  // the scan's result for k + 1
  const int32_t *in_labels;  // [nq][k + 1]
  const float *in_dist;
  int32_t *labels;           // [nq][k]
  float *dist;
  int *list;                 // [nq] queries to replay
  unsigned *count;
  int nq;
};

__global__ __launch_bounds__(256) void exact_flag_kernel(ExactParams p) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= p.nq) return;
  const int k = p.k, k1 = k + 1;
  const int32_t *il = p.in_labels + (size_t)q * k1;
  const float *id = p.in_dist + (size_t)q * k1;
  bool tie = false;
  for (int i = lane; i < k; i += 64) {
    const int32_t a = il[i], b = il[i + 1];
    tie = tie || (a >= 0 && b >= 0 && id[i] == id[i + 1]);
    p.labels[(size_t)q * k + i] = a;
    p.dist[(size_t)q * k + i] = id[i];
  }
  if (__ballot(tie) != 0ull && lane == 0) p.list[atomicAdd(p.count, 1u)] = q;
}

// complete sum of sorted row r in the reference's order (groups of four, VAQ.cpp:1737-1748),
// abandoned (-> +inf) once a partial sum is no longer below t
template <bool BYTES>
__device__ __forceinline__ float ex_row_dist(const ExactParams &p, const float *lut, const int64_t r, const float t) {
  float dist = 0.0f;
  if (BYTES) {
    const int WPR = p.M / 4;
    const uint32_t *rp = p.codes + r * WPR;
    for (int g = 0; g < WPR; g++) {
      const uint32_t c4 = rp[g];
      float dism = lut[(g * 4 + 0) * 256 + (c4 & 0xffu)];
      dism += lut[(g * 4 + 1) * 256 + ((c4 >> 8) & 0xffu)];
      dism += lut[(g * 4 + 2) * 256 + ((c4 >> 16) & 0xffu)];
      dism += lut[(g * 4 + 3) * 256 + (c4 >> 24)];
      dist = g == 0 ? dism : dist + dism;
      if (!(dist < t)) return INFINITY;
    }
    return dist;
  }
  const int W = p.W;
  const uint32_t *tp = p.codes + (r / TILE_ROWS) * (int64_t)(TILE_ROWS * W) + (r % TILE_ROWS);
  float dism = 0.0f;
  for (int s = 0; s < p.M; s++) {
    const SubDesc d = p.sub[s];
    const uint32_t lo = tp[(int64_t)d.word * TILE_ROWS];
    const uint32_t hi = d.word + 1 < W ? tp[(int64_t)(d.word + 1) * TILE_ROWS] : 0u;
    const uint32_t c = __builtin_amdgcn_alignbit(hi, lo, (unsigned)d.shift) & ((1u << d.bits) - 1u);
    const float l = lut[d.lut_off + c];
    dism = (s & 3) == 0 ? l : dism + l;
    if ((s & 3) == 3) {
      dist = s == 3 ? dism : dist + dism;
      if (!(dist < t)) return INFINITY;
    }
  }
  return dist;
}

template <bool BYTES>
__global__ __launch_bounds__(EX_THREADS) void exact_replay_kernel(ExactParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ex_smem[];
  const int e = blockIdx.x;
  if ((unsigned)e >= *p.count) return;
  const int q = p.list[e];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = p.k;
  // LDS: heap values, heap ids, two chunk buffers, [the query's lookup tables]
  float *hval = reinterpret_cast<float *>(ex_smem);
  int *hid = reinterpret_cast<int *>(hval + k);
  float *buf = reinterpret_cast<float *>(hid + k);  // [2][EX_CHUNK]
  float *lbound = buf + 2 * EX_CHUNK;              // [n_buckets] lower bound of the row sums of each bucket (row_bucket)
  float *lds_lut = lbound + (p.row_bucket ? p.n_buckets : 0);
  const float *glut = p.lut + (size_t)q * p.lut_floats;
  const float *lut = glut;
  if (p.lut_in_lds) {
    for (int i = tid; i < p.lut_floats; i += EX_THREADS) lds_lut[i] = glut[i];
    lut = lds_lut;
  }
  // heap_heapify (utils/Heap.hpp:211-235): neutral FLT_MAX, ids -1
  for (int i = tid; i < k; i += EX_THREADS) {
    hval[i] = FLT_MAX;
    hid[i] = -1;
  }
  // the heap top after the last COMPLETE pop + push, for the evaluating waves (the root itself passes
  // through values below the new top while a pop is under way)
  __shared__ float s_top;
  __shared__ unsigned s_gmin[1 << GMIN_MAX_BITS];
  if (tid == 0) s_top = FLT_MAX;
  if (p.row_bucket) {
    // Per bucket the smallest sum its rows can have -- the first table term, plus the smallest second
    // term of the bucket's group of second codes, or the minimum over a coarse bucket's first codes:
    // the bound the scan kernels order the buckets by.  A row whose bucket's bound is not below the
    // heap top cannot be admitted (every further term is >= 0 and fp32 addition is monotone), so its
    // codes are not even read.
    const int bt = p.bucket_t, bsh = p.bucket_shift;
    if (tid < (1 << GMIN_MAX_BITS)) s_gmin[tid] = 0x7f800000u;
    __syncthreads();
    if (bt > 0) {
      const int off1 = p.sub[1].lut_off, n1 = p.sub[1].ncent;
      const int w = 31 - __builtin_clz((unsigned)n1) - bt;
      for (int e = tid; e < n1; e += EX_THREADS) atomicMin(&s_gmin[e >> w], float_to_bits(glut[off1 + e]));
      __syncthreads();
    }
    for (int b = tid; b < p.n_buckets; b += EX_THREADS) {
      float m;
      if (bt > 0) {
        m = glut[b >> bt] + bits_to_float(s_gmin[b & ((1 << bt) - 1)]);
      } else {
        m = INFINITY;
        for (int c = b << bsh; c < ((b + 1) << bsh); c++) {
          const float x = glut[c];
          m = x < m ? x : m;
        }
      }
      lbound[b] = m == m ? m : -INFINITY;  // (a NaN table: never prune by it)
    }
  }
  __syncthreads();
  const int64_t N = p.n_rows;
  const int64_t nchunks = (N + EX_CHUNK - 1) / EX_CHUNK;
  for (int64_t c = 0; c <= nchunks; c++) {
    if (wave > 0) {
      if (c < nchunks) {
        // (the top as it is NOW: the heap has only seen rows before this chunk, so it is at least the
        //  top any row of the chunk will meet -- an admissible row is never abandoned)
        const float t = __hip_atomic_load(&s_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        float *out = buf + (c & 1) * EX_CHUNK;
        constexpr int RPT = (EX_CHUNK + EX_THREADS - 64 - 1) / (EX_THREADS - 64);
        // first the cheap test for all of the thread's rows (one coalesced halfword + one LDS word each),
        // then the gathers of the survivors, all issued before the first sum
        bool live[RPT];
        int64_t src[RPT];
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          const int j = tid - 64 + i * (EX_THREADS - 64);
          const int64_t row = c * EX_CHUNK + j;
          live[i] = j < EX_CHUNK && row < N;
          if (live[i] && p.row_bucket) live[i] = lbound[p.row_bucket[row]] < t;
        }
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          const int64_t row = c * EX_CHUNK + (tid - 64 + i * (EX_THREADS - 64));
          src[i] = live[i] ? (p.inv ? (int64_t)p.inv[row] : row) : 0;
        }
#pragma unroll
        for (int i = 0; i < RPT; i++) {
          const int j = tid - 64 + i * (EX_THREADS - 64);
          if (j < EX_CHUNK) out[j] = live[i] ? ex_row_dist<BYTES>(p, lut, src[i], t) : INFINITY;
        }
      }
    } else if (c > 0) {
      // wave 0: the previous chunk through the reference's loop (VAQ.cpp:1750-1753), 64 rows at a time
      const float *in = buf + ((c - 1) & 1) * EX_CHUNK;
      const int64_t base = (c - 1) * EX_CHUNK;
      for (int j0 = 0; j0 < EX_CHUNK; j0 += 64) {
        const float d = in[j0 + lane];
        float top = hval[0];
        unsigned long long m = __ballot(d < top);
        while (m != 0ull) {
          const int src = __builtin_ctzll(m);
          m &= m - 1ull;
          const float dv = bits_to_float((unsigned)__builtin_amdgcn_readlane((int)float_to_bits(d), src));
          if (top > dv) {  // if (heap_dis[0] > dist)
            if (lane == 0) {
              ex_heap_pop(k, hval, hid);
              ex_heap_push(k, hval, hid, dv, (int)(base + j0 + src));
              __hip_atomic_store(&s_top, hval[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            wave_lds_sync();
            top = hval[0];
            m &= __ballot(d < top);  // (rows the new top already excludes)
          }
        }
      }
    }
    __syncthreads();
  }
  // heap_reorder (utils/Heap.hpp:322-349): pop the maxima into the tail -> ascending; entries of id -1
  // are dropped, the tail refilled with FLT_MAX / -1
  if (tid == 0) {
    int ii = 0;
    for (int i = 0; i < k; i++) {
      const float v = hval[0];
      const int id = hid[0];
      ex_heap_pop(k - i, hval, hid);
      hval[k - ii - 1] = v;
      hid[k - ii - 1] = id;
      if (id != -1) ii++;
    }
    // (memmove of the ii kept entries to the front, then the neutral tail -- done by the copy below)
    reinterpret_cast<int *>(buf)[0] = ii;
  }
  __syncthreads();
  const int nel = reinterpret_cast<int *>(buf)[0];
  for (int i = tid; i < k; i += EX_THREADS) {
    const bool ok = i < nel;
    const int id = ok ? hid[k - nel + i] : -1;
    p.labels[(size_t)q * k + i] = ok ? (int32_t)(id + p.id_base) : -1;
    p.dist[(size_t)q * k + i] = ok ? hval[k - nel + i] : FLT_MAX;
  }
}

// inv[original row] = row of the bucketed order; row_bucket[original row] = its bucket (optional)
__global__ void inverse_perm_kernel(const uint32_t *__restrict__ perm, int64_t n, uint32_t *__restrict__ inv,
                                    const int *__restrict__ bucket_start, int n_buckets,
                                    unsigned short *__restrict__ row_bucket) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t lab = perm ? perm[r] : (uint32_t)r;
  inv[lab] = (uint32_t)r;
  if (row_bucket) {
    int lo = 0, hi = n_buckets;  // largest b with bucket_start[b] <= r
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)bucket_start[mid] <= r) lo = mid;
      else hi = mid;
    }
    row_bucket[lab] = (unsigned short)lo;
  }
}

hipError_t launch_inverse_perm(const uint32_t *perm, int64_t n, uint32_t *inv, const int *bucket_start, int n_buckets,
                               unsigned short *row_bucket, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(inverse_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, perm, n, inv, bucket_start,
                     n_buckets, row_bucket);
  return hipGetLastError();
}

// in_labels / in_dist: the scan's result for k + 1 per query; labels / dist: the caller's k per query
hipError_t launch_exact_ties(const uint32_t *codes, int layout, int M, int W, const SubDesc *sub, const uint32_t *inv,
                             const unsigned short *row_bucket, int n_buckets, int bucket_shift, int bucket_t,
                             int64_t n_rows, const float *lut, int lut_floats, int nq, int k, int64_t id_base,
                             const int32_t *in_labels, const float *in_dist, int32_t *labels, float *dist, int *list,
                             unsigned *count, hipStream_t st) {
  if (nq <= 0) return hipSuccess;
  ExactParams p;
  p.codes = codes;
  p.layout = layout;
  p.M = M;
  p.W = W;
  p.sub = sub;
  p.inv = inv;
  p.row_bucket = row_bucket;
  p.n_buckets = n_buckets;
  p.bucket_shift = bucket_shift;
  p.bucket_t = bucket_t;
  p.n_rows = n_rows;
  p.lut = lut;
  p.lut_floats = lut_floats;
  p.seq = 0;
  p.k = k;
  p.id_base = id_base;
  p.in_labels = in_labels;
  p.in_dist = in_dist;
  p.labels = labels;
  p.dist = dist;
  p.list = list;
  p.count = count;
  p.nq = nq;
  hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(exact_flag_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, p);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  size_t lds = (size_t)k * 8 + (size_t)2 * EX_CHUNK * 4 + (row_bucket ? (size_t)n_buckets * 4 : 0);
  p.lut_in_lds = (size_t)lut_floats * 4 + lds <= 96 * 1024 ? 1 : 0;
  if (p.lut_in_lds) lds += (size_t)lut_floats * 4;
  if (layout == LAYOUT_BYTES) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(exact_replay_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(exact_replay_kernel<true>, dim3(nq), dim3(EX_THREADS), lds, st, p);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(exact_replay_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(exact_replay_kernel<false>, dim3(nq), dim3(EX_THREADS), lds, st, p);
  }
  return hipGetLastError();
}

} // namespace vaq
