// vaq_scan_bm.hip -- the bucket-major second pass of the early-abandon scan (VAQ::searchEarlyAbandon,
// VAQ.cpp:1694-1727; results identical to VAQ::searchHeap, :1729-1758) for a database that is
// streamed from HBM by MANY queries at once (BASELINE configs C4 / C5: 100M-1B rows, 10 k queries).
//
// The best-first form (vaq_scan_bf.h) gives every query its own workgroup, which streams the
// buckets in that query's reach: 10 k queries x 3 % of 1B rows x 16 B = 5 TB through the memory
// fabric for a 16 GB code array, and nothing is shared between the ~300 queries that read the same
// bucket.  Here the work is turned round:
//   pass A  (vaq_scan_bf.h, ScanParams::bm_done) every query scans its nearest buckets, a capped
//           number of work units; it leaves its k best so far in the result arrays, the buckets it
//           has finished (done_key) and its k-th distance as a threshold (g_thr).
//   plan    bm_mark_kernel   per query: the buckets still in reach (bound <= threshold, key >
//                            done_key: the SAME key the best-first form orders by), one mask bit
//                            each, counted per bucket; the query's histogram starts from pass A's rows
//           bm_order_kernel  buckets by descending work (rows x query groups), dealt to the eight
//                            XCD queues; prefix of the work items of each queue; list offsets
//           bm_fill_kernel   per bucket the list of its queries
//   pass B  scan_bm_kernel   persistent workgroups, one queue per XCD.  A work item = (bucket,
//                            group of QB queries): the QB lookup tables are staged in LDS interleaved
//                            per entry (one ds_read_b128 serves four queries), the bucket's rows are
//                            streamed once for the group, all groups of a bucket run back to back on
//                            the same XCD so that its rows come from HBM once and from that L2 after.
//                            Phases A / A2 / survivor queue / tail are the best-first form's, per
//                            (row, query).  A row whose complete sum is not above its query's
//                            threshold is APPENDED to that query's candidate buffer (one global
//                            atomic), and counted in the query's 64-bin histogram over [0, H]: the
//                            upper edge of the bin where the running count reaches k has >= k real
//                            rows at or below it -- a valid threshold for everybody (g_thr).
//   select  bm_select_kernel per query: k best of (pass A's list + candidates), ascending by
//                            (distance, label); a query whose buffer overflowed is handed to the
//                            best-first form's second launch (ScanParams::defer_mode), which scans
//                            its remaining buckets on its own and merges (launch_defer_merge).
// Nothing depends on the order in which candidates arrive: the result is the k smallest (distance,
// label) pairs of the rows, as VAQ::searchHeap's is (DESIGN.md "Ties" for equal distances).
// Arithmetic per row: dism = l0; dism += l1; dism += l2; dism += l3; dist (+)= dism, group by group
// (VAQ.cpp:1737-1748).
#include "vaq_scan_bf.h"

namespace vaq {

constexpr int BM_QCAP = 128;          // survivor queue of a wave: at most 63 left over + 64 pushed
#ifndef VAQ_BM_RING
#define VAQ_BM_RING 4  // (3 / 4 / 6: 125M rows 16.9 / 16.6 / 16.5 ms, 1B 66.9 / 66.0 / 65.7)
#endif
constexpr int BM_RING = VAQ_BM_RING;  // code items in flight per wave
constexpr int BM_THR_EVERY = 16;      // wave steps between reads of the workgroup's thresholds (LDS)
constexpr int BM_THR_GLOBAL_EVERY = 64;  // ... and between wave 0's reads of the shared words
constexpr int BM_TRIGGER = 8;         // every this many candidates of a query its histogram is read
constexpr int BM_MAX_THREADS = 1024;

__host__ __device__ inline int bm_ioff_stride(int n_buckets) { return n_buckets / BM_XCDS + 2; }

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
// One workgroup per query.  The bucket key is scan_bf_body's bucket_key_of for one slice covering
// every row (r0 = 0, r1 = n_rows) and bucket_shift == 0, bit for bit: pass A's done_key is compared
// with it.
__global__ __launch_bounds__(256) void bm_mark_kernel(BmParams p) {
  __shared__ unsigned gmin[1 << GMIN_MAX_BITS];
  __shared__ unsigned smask[BF_MAX_BUCKETS / 32];
  __shared__ unsigned shist[BM_HIST_BINS];
  __shared__ unsigned s_cnt[2];
  __shared__ unsigned long long s_near[2 << GMIN_MAX_BITS];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int K0 = p.n_buckets, bt = p.bucket_t;
  const int nwords = K0 / 32;
  const unsigned done = p.done_key[q];
  const bool fresh = p.fresh[q] != 0u;
  if (tid == 0) p.cand_cnt[q] = 0u;
  if (tid < BM_HIST_BINS) shist[tid] = 0u;
  if (tid < nwords) smask[tid] = 0u;
  if (tid < 2) s_cnt[tid] = 0u;
  // (a search's first round: the word starts from what the first pass / the sample left in g_thr)
  const unsigned long long t64 = p.init64 ? (((unsigned long long)p.g_thr[q] << 32) | 0x7fffffffull) : p.thr64[q];
  if (p.init64 && tid == 0) p.thr64[q] = t64;
  const unsigned thr_bits = (unsigned)(t64 >> 32);
  const float H = bits_to_float(thr_bits);
  const float scale = (done != 0xffffffffu && H > 0.0f && H < FLT_MAX) ? (float)BM_HIST_BINS / H : 0.0f;
  unsigned next = 0xffffffffu;
  if (done != 0xffffffffu) {  // (workgroup-uniform: the query still has buckets in reach)
    const float *__restrict__ l = p.lut + (size_t)q * p.lut_floats;
    if (bt > 0) {
      if (tid < (1 << bt)) gmin[tid] = 0x7f800000u;
      __syncthreads();
      const int ncent1 = 256;
      const int w = 8 - bt;  // log2 of the group size
      const int seg = w < 6 ? 1 << w : 64;
      for (int e0 = tid - lane; e0 < ncent1; e0 += blockDim.x) {
        const int e = e0 + lane;
        unsigned v = e < ncent1 ? float_to_bits(l[256 + e]) : 0x7f800000u;
        for (int o = 1; o < seg; o <<= 1) {
          const unsigned x = (unsigned)__shfl_xor((int)v, o);
          v = x < v ? x : v;
        }
        if ((lane & (seg - 1)) == 0 && e < ncent1) atomicMin(&gmin[e >> w], v);
      }
    }
    __syncthreads();
    {
      // per group of second codes: the query's nearest and second nearest code of the group.  Queries
      // that agree on them lie close together in the second subspace and reach the same runs of a
      // bucket; the bucket lists are ordered by this key (bm_sort_kernel).
      const int w = 8 - bt, ng = 1 << bt;
      if (tid < 2 * ng) s_near[tid] = ~0ull;
      __syncthreads();
      const unsigned long long mine = ((unsigned long long)float_to_bits(l[256 + tid]) << 32) | (unsigned)tid;  // (256 threads: one second code each)
      atomicMin(&s_near[tid >> w], mine);
      __syncthreads();
      if (mine != s_near[tid >> w]) atomicMin(&s_near[ng + (tid >> w)], mine);
      __syncthreads();
      if (tid < ng) {
        const unsigned a = (unsigned)(s_near[tid] & 0xffu) & ((1u << w) - 1u);
        const unsigned b2 = (unsigned)(s_near[ng + tid] & 0xffu) & ((1u << w) - 1u);
        p.qkey[(size_t)q * ng + tid] = (unsigned short)((a << 8) | b2);
      }
    }
    const unsigned idx_mask = (unsigned)K0 - 1u;  // (K0 is a power of two >= 32)
    const unsigned empty_key = ~idx_mask;
    // the thread's buckets: their keys when still in reach, else 0xffffffff
    unsigned mykey[BF_MAX_BUCKETS / 256];
    int mine = 0;
#pragma unroll
    for (int e = 0; e < BF_MAX_BUCKETS / 256; e++) {
      const int b = tid + e * 256;
      unsigned key = 0xffffffffu;
      if (b < K0) {
        const int bs = p.bucket_start[b], be = p.bucket_start[b + 1];
        if (be > bs) {
          float m;
          if (bt > 0) {
            m = l[b >> bt] + bits_to_float(gmin[b & ((1 << bt) - 1)]);
          } else {
            const float x = l[b];
            m = x < INFINITY ? x : INFINITY;
          }
          if (m == m) {
            const unsigned kk = (float_to_bits(m) & ~idx_mask) | (unsigned)b;
            if ((kk & empty_key) != empty_key && (fresh || kk > done) && (kk & ~idx_mask) <= thr_bits) key = kk;
          }
        }
      }
      mykey[e] = key;
      mine += key != 0xffffffffu ? 1 : 0;
    }
    if (mine) atomicAdd(&s_cnt[0], (unsigned)mine);
    __syncthreads();
    const int total = (int)s_cnt[0];
    unsigned cut = 0xfffffffeu;  // keys <= cut join this round
    if (p.limit > 0 && total > p.limit) {
      // the limit-th smallest key in reach (keys are distinct): bisection on its bits
      unsigned lo = 0u, hi = 0xfffffffeu;
      while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int e = 0; e < BF_MAX_BUCKETS / 256; e++) c += mykey[e] <= mid ? 1 : 0;
        unsigned *ctr = &s_cnt[1];
        __syncthreads();
        if (tid == 0) *ctr = 0u;
        __syncthreads();
        if (c) atomicAdd(ctr, (unsigned)c);
        __syncthreads();
        if ((int)*ctr >= p.limit) hi = mid;
        else lo = mid + 1u;
      }
      cut = lo;
      next = cut;
    }
#pragma unroll
    for (int e = 0; e < BF_MAX_BUCKETS / 256; e++) {
      if (mykey[e] <= cut) {
        const int b = (int)(mykey[e] & idx_mask);
        atomicOr(&smask[b >> 5], 1u << (b & 31));
        atomicAdd(&p.cnt[b], 1);
      }
    }
    // the rows found so far are the first entries of the histogram
    if (scale != 0.0f) {
      for (int i = tid; i < p.k; i += blockDim.x) {
        if (p.labels[(size_t)q * p.k + i] >= 0) {
          const unsigned bin = (unsigned)(p.dist[(size_t)q * p.k + i] * scale);
          atomicAdd(&shist[bin < BM_HIST_BINS - 1 ? bin : BM_HIST_BINS - 1], 1u);
        }
      }
    }
  }
  __syncthreads();
  if (tid < nwords) p.mask[(size_t)q * nwords + tid] = smask[tid];
  if (tid < BM_HIST_BINS) p.hist[(size_t)q * BM_HIST_BINS + tid] = shist[tid];
  if (tid == 0) {
    p.scale[q] = scale;
    p.done_next[q] = next;
  }
}

// Instead of a best-first first pass: a threshold per query from a SAMPLE of its nearest rows.  One
// workgroup per query: the nearest non-empty bucket (smallest key), inside it the nearest non-empty
// run (smallest second term; the bucket's first row when the rows of a bucket are not ordered), from
// there up to BM_BOOT_ROWS rows summed completely; their k-th smallest sum has k real rows at or below
// it -- an upper bound of the final k-th distance (FLT_MAX when the sample has fewer than k rows).
// The query's result list starts empty and none of its buckets is finished.
#ifndef VAQ_BM_BOOT_ROWS
#define VAQ_BM_BOOT_ROWS 4096
#endif
constexpr int BM_BOOT_ROWS = VAQ_BM_BOOT_ROWS;
template <int M>
__global__ __launch_bounds__(256) void bm_boot_kernel(BmParams p) {
  constexpr int WPR = M / 4;
  extern __shared__ __attribute__((aligned(16))) float blut[];  // [M * 256]
  __shared__ unsigned gmin[1 << GMIN_MAX_BITS];
  __shared__ unsigned s_min, s_cnt;
  __shared__ unsigned long long s_run;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int K0 = p.n_buckets, bt = p.bucket_t, k = p.k;
  const float *__restrict__ l = p.lut + (size_t)q * p.lut_floats;
  for (int e = tid; e < M * 64; e += 256) reinterpret_cast<float4 *>(blut)[e] = reinterpret_cast<const float4 *>(l)[e];
  for (int i = tid; i < k; i += 256) {
    p.labels[(size_t)q * k + i] = -1;
    p.dist[(size_t)q * k + i] = FLT_MAX;
  }
  if (tid == 0) {
    s_min = 0xffffffffu;
    s_run = ~0ull;
    p.done_key[q] = 0u;
    p.fresh[q] = 1u;
  }
  if (tid < (1 << bt)) gmin[tid] = 0x7f800000u;
  __syncthreads();
  if (bt > 0) {
    const int w = 8 - bt;
    const int seg = w < 6 ? 1 << w : 64;
    for (int e0 = tid - lane; e0 < 256; e0 += 256) {
      const int e = e0 + lane;
      unsigned v = float_to_bits(blut[256 + e]);
      for (int o = 1; o < seg; o <<= 1) {
        const unsigned x = (unsigned)__shfl_xor((int)v, o);
        v = x < v ? x : v;
      }
      if ((lane & (seg - 1)) == 0) atomicMin(&gmin[e >> w], v);
    }
    __syncthreads();
  }
  const unsigned idx_mask = (unsigned)K0 - 1u;
  unsigned km = 0xffffffffu;
  for (int b = tid; b < K0; b += 256) {
    if (p.bucket_start[b + 1] <= p.bucket_start[b]) continue;
    float m;
    if (bt > 0) {
      m = blut[b >> bt] + bits_to_float(gmin[b & ((1 << bt) - 1)]);
    } else {
      const float x = blut[b];
      m = x < INFINITY ? x : INFINITY;
    }
    if (!(m == m)) continue;
    const unsigned key = (float_to_bits(m) & ~idx_mask) | (unsigned)b;
    km = key < km ? key : km;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned x = (unsigned)__shfl_xor((int)km, o);
    km = x < km ? x : km;
  }
  if (lane == 0) atomicMin(&s_min, km);
  __syncthreads();
  unsigned thr = 0x7f7fffffu;  // FLT_MAX: heap_heapify's neutral (utils/Heap.hpp:211-235)
  const unsigned kmin = s_min;
  if (kmin != 0xffffffffu && (kmin & ~idx_mask) < 0x7f800000u) {  // (workgroup-uniform)
    const int b = (int)(kmin & idx_mask);
    const int bs = p.bucket_start[b], be = p.bucket_start[b + 1];
    int rs = bs;
    if (p.sub_start) {
      const int R = 256 >> bt;
      const int f0 = b << (8 - bt), c1base = (b & ((1 << bt) - 1)) << (8 - bt);
      for (int j = tid; j < R; j += 256) {
        if (p.sub_start[f0 + j + 1] > p.sub_start[f0 + j])
          atomicMin(&s_run, ((unsigned long long)float_to_bits(blut[256 + c1base + j]) << 32) | (unsigned)j);
      }
      __syncthreads();
      const unsigned long long best = s_run;
      if (best != ~0ull) rs = p.sub_start[f0 + (int)(best & 0xffffffffull)];
    }
    int r1 = rs + BM_BOOT_ROWS < be ? rs + BM_BOOT_ROWS : be;
    if (r1 - rs < BM_BOOT_ROWS) rs = r1 - BM_BOOT_ROWS > bs ? r1 - BM_BOOT_ROWS : bs;  // (a short tail: take the rows before it)
    const int ns = r1 - rs;
    unsigned v[BM_BOOT_ROWS / 256];
#pragma unroll
    for (int e = 0; e < BM_BOOT_ROWS / 256; e++) {
      const int r = rs + tid + e * 256;
      unsigned db = 0x7f800000u;
      if (r < r1) {
        const uint32_t *rp = p.codes + (int64_t)r * WPR;
        float acc = 0.0f;
#pragma unroll
        for (int gq = 0; gq < WPR; gq++) {
          const uint32_t c4 = rp[gq];
          float dism = blut[(gq * 4 + 0) * 256 + (c4 & 0xffu)];
          dism = dism + blut[(gq * 4 + 1) * 256 + ((c4 >> 8) & 0xffu)];
          dism = dism + blut[(gq * 4 + 2) * 256 + ((c4 >> 16) & 0xffu)];
          dism = dism + blut[(gq * 4 + 3) * 256 + (c4 >> 24)];
          acc = gq == 0 ? dism : acc + dism;
        }
        if (acc == acc) db = float_to_bits(acc);  // (sums are >= 0: bit order == value order; NaN counts as +inf)
      }
      v[e] = db;
    }
    if (ns >= k) {
      unsigned lo = 0u, hi = 0x7f800000u;  // smallest t with count(v <= t) >= k
      while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int e = 0; e < BM_BOOT_ROWS / 256; e++) c += __popcll(__ballot(v[e] <= mid));
        __syncthreads();
        if (tid == 0) s_cnt = 0u;
        __syncthreads();
        if (lane == 0 && c) atomicAdd(&s_cnt, (unsigned)c);
        __syncthreads();
        if ((int)s_cnt >= k) hi = mid;
        else lo = mid + 1u;
      }
      if (lo < 0x7f800000u) thr = lo < thr ? lo : thr;
    }
  }
  if (tid == 0) p.g_thr[q] = thr;
}

// One workgroup of 1024 threads: thread b owns bucket b.
__global__ __launch_bounds__(BM_MAX_THREADS) void bm_order_kernel(BmParams p) {
  __shared__ unsigned long long keys[BF_MAX_BUCKETS];
  __shared__ int pre[BF_MAX_BUCKETS];
  const int tid = threadIdx.x, K0 = p.n_buckets, QB = p.qb;
  const int c = tid < K0 ? p.cnt[tid] : 0;
  const long long rows = tid < K0 ? (long long)p.bucket_start[tid + 1] - p.bucket_start[tid] : 0;
  const long long G = (c + QB - 1) / QB;
  const unsigned long long work = (unsigned long long)(rows > 0 ? rows : 0) * (unsigned long long)G;  // < 2^44
  keys[tid] = tid < K0 ? ((((1ull << 44) - 1ull - work) << 10) | (unsigned)tid) : ~0ull;
  pre[tid] = c;
  __syncthreads();
  // buckets by descending work (ties by bucket)
  for (int size = 2; size <= BF_MAX_BUCKETS; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (tid < BF_MAX_BUCKETS / 2) {
        const int i = 2 * tid - (tid & (stride - 1));
        const int j = i + stride;
        const unsigned long long a = keys[i], d = keys[j];
        if ((a > d) == ((i & size) == 0)) { keys[i] = d; keys[j] = a; }
      }
      __syncthreads();
    }
  if (tid < K0) p.border[tid] = (int)(keys[tid] & 1023ull);
  // inclusive prefix of the per-bucket query counts
  for (int o = 1; o < BF_MAX_BUCKETS; o <<= 1) {
    const int v = tid >= o ? pre[tid - o] : 0;
    __syncthreads();
    pre[tid] += v;
    __syncthreads();
  }
  if (tid < K0) {
    p.qoff[tid + 1] = pre[tid];
    p.fill[tid] = 0;
  }
  if (tid == 0) p.qoff[0] = 0;
  // XCD x works through the buckets at positions x, x + 8, ... of the order: prefix of their items
  if (tid < BM_XCDS) {
    const int stride = bm_ioff_stride(K0);
    const int nb = (K0 - tid + BM_XCDS - 1) / BM_XCDS;
    int acc = 0;
    for (int j = 0; j < nb; j++) {
      p.ioff[tid * stride + j] = acc;
      const int b = (int)(keys[tid + BM_XCDS * j] & 1023ull);
      acc += (p.cnt[b] + QB - 1) / QB;
    }
    p.ioff[tid * stride + nb] = acc;
    p.tickets[tid] = 0u;
  }
}

// One wave per query: lane w owns word w of the query's bucket mask.
__global__ __launch_bounds__(64) void bm_fill_kernel(BmParams p) {
  const int q = blockIdx.x, lane = threadIdx.x;
  const int nwords = p.n_buckets / 32;
  if (lane >= nwords) return;
  unsigned bits = p.mask[(size_t)q * nwords + lane];
  while (bits) {
    const int b = lane * 32 + __builtin_ctz(bits);
    bits &= bits - 1u;
    const int pos = atomicAdd(&p.fill[b], 1);
    // (key << 14 | query: bm_sort_kernel orders the bucket's list by it and leaves the query alone)
    p.qlist[p.qoff[b] + pos] = (int)(((unsigned)p.qkey[(size_t)q * (1 << p.bucket_t) + (b & ((1 << p.bucket_t) - 1))] << 14) | (unsigned)q);
  }
}

// One workgroup per bucket: its list ordered by (key, query) -- similar queries next to each other --
// and stripped of the key.  Bitonic sort in LDS (a list holds at most one entry per query).
constexpr int BM_SORT_THREADS = 256;
__global__ __launch_bounds__(BM_SORT_THREADS) void bm_sort_kernel(BmParams p) {
  extern __shared__ unsigned bs_keys[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = p.cnt[b];
  if (n <= 0) return;
  int *list = p.qlist + p.qoff[b];
  int P = 2;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += BM_SORT_THREADS) bs_keys[i] = i < n ? (unsigned)list[i] : 0xffffffffu;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += BM_SORT_THREADS) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const unsigned a = bs_keys[i], c = bs_keys[j];
        if ((a > c) == ((i & size) == 0)) { bs_keys[i] = c; bs_keys[j] = a; }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += BM_SORT_THREADS) list[i] = (int)(bs_keys[i] & 0x3fffu);
}

// ---------------------------------------------------------------------------
// pass B
// ---------------------------------------------------------------------------
template <int QB> struct BmVec;
template <> struct BmVec<2> { typedef f32x2 T; };
template <> struct BmVec<4> { typedef f32x4 T; };

// candidates a wave gathers before it appends them (each append is a returning memory-side atomic the wave
// has to wait for: gathered, one wait serves a batch)
#ifndef VAQ_BM_BATCH
#define VAQ_BM_BATCH 24
#endif
constexpr int BM_BATCH = VAQ_BM_BATCH;
constexpr int BM_CB_CAP = BM_BATCH - 1 + 64;  // one finishing pass adds at most 64
__host__ __device__ inline size_t bm_lds_bytes(int M, int qb, int nwaves) {
  const int qcw = M <= 16 ? M / 4 - 1 : 0;
  return (size_t)M * 256 * qb * 4 + (size_t)nwaves * (BM_QCAP * 4 * (3 + qcw) + BM_CB_CAP * 12);
}

// The QB entries (one per query of the group) of code byte B in table T of the interleaved tables at LDS
// offset 0: byte_shl puts the byte's offset (16 or 8 bytes per code) in one instruction, the table's
// offset is the load's immediate (vaq_scan.h, lds_lut).
template <int QB, int B, typename VT> __device__ __forceinline__ VT lds_lut_vec(const unsigned table, const unsigned c) {
  typedef __attribute__((address_space(3))) const VT lds_cvt;
  constexpr int SH = QB == 4 ? 4 : 3;
  return *reinterpret_cast<lds_cvt *>((uintptr_t)(byte_shl<B, SH>(c) + table * (256u * QB * 4u)));
}
// one query's entry: i4 = 4 * (slot of the query in the group), per lane
template <int QB, int B> __device__ __forceinline__ float lds_lut_q(const unsigned table, const unsigned c, const unsigned i4) {
  constexpr int SH = QB == 4 ? 4 : 3;
  return *reinterpret_cast<lds_cfloat *>((uintptr_t)(byte_shl<B, SH>(c) + i4 + table * (256u * QB * 4u)));
}

constexpr int BM_MAX_RUNS = 256;  // second codes of a bucket (bucket_t == 0: all 256)
constexpr int BM_TABLES_BYTES = 8192;  // the scan kernel's per-workgroup tables, after bm_lds_bytes()

// SUB: the rows of a bucket are ordered by the second code (BmParams::sub_start): a RUN of rows shares
// its first two table terms, so l0 + l1 is one value per (run, query) -- the first early-abandon test
// of VAQ::searchEarlyAbandon (VAQ.cpp:1708) for a whole run at once.  Runs out of every query's reach
// are never read; the others start at the third term.
template <int M, int QB, bool SUB>
__global__ __launch_bounds__(BM_MAX_THREADS) void scan_bm_kernel(BmParams p) {
  typedef BfBytesItem<M, (M < 16 ? 16 : M)> Item;
  typedef typename BmVec<QB>::T VT;
  constexpr int ROWS = Item::ROWS;
  constexpr int WPR = M / 4;
  constexpr int QCW = (M <= 16) ? WPR - 1 : 0;
  constexpr int WSTEP = 64 * ROWS;
  // No static LDS: the lookup tables sit at LDS offset 0 and are addressed by integer (lds_lut_vec below:
  // a table's offset goes into the instruction's immediate field); the small per-workgroup tables follow the
  // waves' buffers (BM_TABLES_BYTES, bm_tables_*).
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  unsigned char *tb = smem + bm_lds_bytes(M, QB, (int)(blockDim.x >> 6));
  int *s_ioff = reinterpret_cast<int *>(tb);                       // [BF_MAX_BUCKETS / BM_XCDS + 2]
  int &s_ticket = *reinterpret_cast<int *>(tb + 528);
  unsigned *s_thr = reinterpret_cast<unsigned *>(tb + 544);        // [QB]
  int *s_q = reinterpret_cast<int *>(tb + 560);                    // [QB]
  // runs of the item's bucket (SUB): rows, prefix of the wave steps of the runs in reach, l0 + l1 per query
  int *s_run_s = reinterpret_cast<int *>(tb + 576);                // [BM_MAX_RUNS]
  int *s_run_e = s_run_s + BM_MAX_RUNS;                            // [BM_MAX_RUNS]
  int *s_cum = s_run_e + BM_MAX_RUNS;                              // [BM_MAX_RUNS + 1]
  float *s_p01 = reinterpret_cast<float *>(tb + 576 + 4 * (3 * BM_MAX_RUNS + 4));  // [BM_MAX_RUNS * QB]
  static_assert(576 + 4 * (3 * BM_MAX_RUNS + 4) + 4 * BM_MAX_RUNS * QB <= BM_TABLES_BYTES, "the tables' block");
  static_assert((BF_MAX_BUCKETS / BM_XCDS + 2) * 4 <= 528, "s_ioff");
  lds_base_is_zero(smem);
  // (the wave number through readfirstlane: the compiler then keeps everything derived from it -- step
  //  numbers, run cursors, row ranges -- in scalar registers instead of comparing vectors under EXEC masks)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
  float *lut = reinterpret_cast<float *>(smem);  // entry (t * 256 + c) of query i at (t * 256 + c) * QB + i
  unsigned char *wb = smem + (size_t)M * 256 * QB * 4 + (size_t)wave * (BM_QCAP * 4 * (3 + QCW) + BM_CB_CAP * 12);
  int *q_row = reinterpret_cast<int *>(wb);
  float *q_part = reinterpret_cast<float *>(q_row + BM_QCAP);
  int *q_i = reinterpret_cast<int *>(q_part + BM_QCAP);
  uint32_t *q_cw = reinterpret_cast<uint32_t *>(q_i + BM_QCAP);
  float *cb_d = reinterpret_cast<float *>(q_cw + (size_t)QCW * BM_QCAP);  // gathered candidates: distance, row, slot
  int *cb_row = reinterpret_cast<int *>(cb_d + BM_CB_CAP);
  int *cb_i = cb_row + BM_CB_CAP;
  const int K0 = p.n_buckets, bt = p.bucket_t, k = p.k;
  const uint32_t *__restrict__ codes = p.codes;
  const uint32_t *__restrict__ perm = p.perm;
  const int stride = bm_ioff_stride(K0);
  const int home = (int)(blockIdx.x & (BM_XCDS - 1));  // (workgroups are dealt round-robin over the XCDs)

  for (int hop = 0; hop < BM_XCDS; hop++) {
    // the home queue first; when it is empty, the others' leftovers (their buckets then come through
    // this XCD's L2 as well: a second copy, at the very end of the launch only)
    const int xq = (home + hop) & (BM_XCDS - 1);
    const int nb = (K0 - xq + BM_XCDS - 1) / BM_XCDS;
    __syncthreads();
    for (int i = tid; i <= nb; i += nthreads) s_ioff[i] = p.ioff[xq * stride + i];
    __syncthreads();
    const int total = s_ioff[nb];
    for (;;) {
      __syncthreads();  // the previous item is finished by every wave
      if (tid == 0) s_ticket = (int)atomicAdd(&p.tickets[xq], 1u);
      __syncthreads();
      const int t = s_ticket;
      if (t >= total) break;  // (every thread: the same t)
      int lo = 0, hi = nb;  // s_ioff[lo] <= t < s_ioff[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_ioff[mid] <= t) lo = mid;
        else hi = mid;
      }
      const int b = p.border[xq + BM_XCDS * lo];
      const int g = t - s_ioff[lo];
      const int nact_all = p.cnt[b] - g * QB;
      const int nact = nact_all < QB ? nact_all : QB;  // >= 1
      const int qbase = p.qoff[b] + g * QB;
      int qi[QB];
#pragma unroll
      for (int i = 0; i < QB; i++) qi[i] = __builtin_amdgcn_readfirstlane(p.qlist[qbase + (i < nact ? i : 0)]);
      const int bs = p.bucket_start[b], be = p.bucket_start[b + 1];
      if (tid < QB) {
        // (per-lane picks among the group's queries and thresholds go through LDS: a register array
        //  indexed by a lane value ends up in scratch memory)
        const int myq = p.qlist[qbase + (tid < nact ? tid : 0)];
        s_q[tid] = myq;
        s_thr[tid] = tid < nact ? (unsigned)(__hip_atomic_load(&p.thr64[myq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32)
                                : 0xbf800000u;  // -1: nothing passes (a slot past the end of the bucket's list)
      }
      __syncthreads();
      int nsteps;  // wave steps of the item
      if (SUB) {
        // ---- the bucket's runs: l0 + l1 per query, and which of them some query of the group can
        //      still reach (thresholds have moved since the plan was made) ----
        const int R = 256 >> bt;
        const int f0 = b << (8 - bt), c1base = (b & ((1 << bt) - 1)) << (8 - bt);
        for (int j = tid; j < R; j += nthreads) {
          const int rs = p.sub_start[f0 + j], re = p.sub_start[f0 + j + 1];
          bool any = false;
#pragma unroll
          for (int i = 0; i < QB; i++) {
            const float *__restrict__ gl = p.lut + (size_t)qi[i] * p.lut_floats;
            const float v = gl[b >> bt] + gl[256 + c1base + j];  // dism = l0; dism += l1
            s_p01[j * QB + i] = v;
            any = any || v <= bits_to_float(s_thr[i]);
          }
          s_run_s[j] = rs;
          s_run_e[j] = re;
          s_cum[j + 1] = (any && re > rs) ? (re - (rs & ~(WSTEP - 1)) + WSTEP - 1) / WSTEP : 0;
        }
        if (tid == 0) s_cum[0] = 0;
        __syncthreads();
        if (wave == 0) {  // inclusive prefix, 64 at a time
          int carry = 0;
          for (int base = 0; base < R; base += 64) {
            const int i = base + lane;
            int inc = i < R ? s_cum[i + 1] : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const int x = __shfl_up(inc, o);
              if (lane >= o) inc += x;
            }
            if (i < R) s_cum[i + 1] = carry + inc;
            carry += __builtin_amdgcn_readlane(inc, 63);
          }
        }
        __syncthreads();
        nsteps = s_cum[R];
        if (nsteps == 0) continue;  // (every thread alike) nothing of this bucket is in the group's reach any more
      } else {
        nsteps = (be - (bs & ~(WSTEP - 1)) + WSTEP - 1) / WSTEP;
      }
      // ---- the group's lookup tables, interleaved per entry ----
      {
        const float4 *g4[QB];
#pragma unroll
        for (int i = 0; i < QB; i++) g4[i] = reinterpret_cast<const float4 *>(p.lut + (size_t)qi[i] * p.lut_floats);
        // (SUB: the first two tables are never gathered -- their sum per run is already in s_p01)
        for (int e4 = tid + (SUB ? 128 : 0); e4 < M * 64; e4 += nthreads) {
          float4 v[QB];
#pragma unroll
          for (int i = 0; i < QB; i++) v[i] = g4[i][e4];
          VT *dst = reinterpret_cast<VT *>(lut) + (size_t)e4 * 4;
          VT o0, o1, o2, o3;
#pragma unroll
          for (int i = 0; i < QB; i++) {
            o0[i] = v[i].x;
            o1[i] = v[i].y;
            o2[i] = v[i].z;
            o3[i] = v[i].w;
          }
          dst[0] = o0;
          dst[1] = o1;
          dst[2] = o2;
          dst[3] = o3;
        }
      }
      __syncthreads();

      // ---- per-wave state ----
      float thr[QB], l0[QB];
#pragma unroll
      for (int i = 0; i < QB; i++) {
        thr[i] = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)s_thr[i]));
        l0[i] = SUB ? 0.0f : bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(lut[(size_t)(b >> bt) * QB + i])));
      }
      int qcnt = 0, ccnt = 0;

      auto pick_q = [&](const int i) -> int { return s_q[i]; };
      auto pick_thr = [&](const int i) -> float {
        return bits_to_float(__hip_atomic_load(&s_thr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      };
      auto refresh = [&](const int kstep) {
        if ((kstep & (BM_THR_EVERY - 1)) != 0) return;
        if (wave == 0 && (kstep & (BM_THR_GLOBAL_EVERY - 1)) == 0 && kstep > 0) {
          if (lane < nact) {
            const unsigned gt = (unsigned)(__hip_atomic_load(&p.thr64[pick_q(lane)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
            atomicMin(&s_thr[lane], gt);
          }
          wave_lds_sync();
        }
#pragma unroll
        for (int i = 0; i < QB; i++) {
          const unsigned tb = __hip_atomic_load(&s_thr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          thr[i] = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)tb));
        }
      };

      // up to 64 gathered candidates -> their queries' buffers
      auto append = [&](const float d, const int row, const int i, const bool ok) {
        if (__ballot(ok) == 0ull) return;
        int q = 0;
        float sc = 0.0f;
        bool trig = false;
        bool ok2 = ok;
        int label = 0;
        if (ok) {
          // the exact test, on (distance, label): the shared word may hold a label bound (after an
          // overflow among rows of EQUAL distance only the smallest labels are wanted)
          q = pick_q(i);
          label = perm ? (int)perm[row] : row;  // labels are ORIGINAL rows
          const unsigned long long t64 = __hip_atomic_load(&p.thr64[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok2 = (((unsigned long long)float_to_bits(d) << 32) | (unsigned)label) < t64;
        }
        if (ok2) {
          const unsigned pos = atomicAdd(&p.cand_cnt[q], 1u);
          sc = p.scale[q];
          if (sc != 0.0f) {
            const unsigned bin = (unsigned)(d * sc);
            atomicAdd(&p.hist[(size_t)q * BM_HIST_BINS + (bin < BM_HIST_BINS - 1 ? bin : BM_HIST_BINS - 1)], 1u);
          }
          if (pos < (unsigned)p.cap) {
            p.cand_d[(size_t)q * p.cap + pos] = d;
            p.cand_id[(size_t)q * p.cap + pos] = label;
          }
          trig = sc != 0.0f && ((pos + 1u) & (BM_TRIGGER - 1)) == 0u;
        }
        // (rare path: wait for its loads here, or the wait-count bookkeeping of every block it
        //  rejoins degrades to "all loads" -- vaq_scan_bf.h, flush())
        __builtin_amdgcn_s_waitcnt(0x0F70);
        unsigned long long m = __ballot(trig);
        while (m != 0ull) {
          // the query's histogram: the upper edge of the bin where the running count reaches k has
          // at least k rows (pass A's and appended ones: distinct rows) at or below it
          const int src = __builtin_ctzll(m);
          m &= m - 1ull;
          const int qq = __builtin_amdgcn_readlane(q, src);
          const int ii = __builtin_amdgcn_readlane(i, src);
          const float scq = bits_to_float((unsigned)__builtin_amdgcn_readlane((int)float_to_bits(sc), src));
          int inc = (int)__hip_atomic_load(&p.hist[(size_t)qq * BM_HIST_BINS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(inc, o);
            if (lane >= o) inc += x;
          }
          const unsigned long long reach = __ballot(inc >= k);
          if (reach != 0ull) {
            const int jb = __builtin_ctzll(reach);
            if (jb < BM_HIST_BINS - 1) {  // (the last bin also holds everything beyond H)
              const float edge = ((float)(jb + 1) / scq) * (1.0f + 1.0f / 1048576.0f);
              if (lane == 0) {
                atomicMin(&p.thr64[qq], ((unsigned long long)float_to_bits(edge) << 32) | 0x7fffffffull);
                atomicMin(&s_thr[ii], float_to_bits(edge));
              }
            }
          }
        }
      };

      auto flush = [&]() {
        while (ccnt > 0) {
          const int n = ccnt < 64 ? ccnt : 64;
          ccnt -= n;
          const bool has = lane < n;
          const int slot = ccnt + (has ? lane : 0);
          const float d = cb_d[slot];
          const int ii = cb_i[slot];
          // (the threshold may have moved since the row was gathered)
          append(d, cb_row[slot], ii, has && d <= pick_thr(ii));
        }
      };

      // phase B: the top n (<= 64) queue entries, one per lane: groups 1.. of the row, abandoning
      // after each (VAQ.cpp:1708)
      auto drain = [&](const int n) {
        qcnt -= n;
        const bool ok = lane < n;
        const int slot = qcnt + (ok ? lane : 0);
        const int row = q_row[slot];
        float acc = q_part[slot];
        const int i = q_i[slot];
        const unsigned i4 = (unsigned)i * 4u;
        const float tl = pick_thr(i);
        bool alive = ok;
        uint32_t cw[QCW > 0 ? QCW : 1];
#pragma unroll
        for (int w = 0; w < QCW; w++) cw[w] = q_cw[w * BM_QCAP + slot];
#pragma unroll
        for (int gq = 1; gq < WPR; gq++) {
          const uint32_t c4 = QCW > 0 ? cw[gq - 1] : codes[(int64_t)row * WPR + gq];
          if (alive) {
            const float e0 = lds_lut_q<QB, 0>(gq * 4 + 0, c4, i4), e1 = lds_lut_q<QB, 1>(gq * 4 + 1, c4, i4);
            const float e2 = lds_lut_q<QB, 2>(gq * 4 + 2, c4, i4), e3 = lds_lut_q<QB, 3>(gq * 4 + 3, c4, i4);
            float dism = e0 + e1;
            dism = dism + e2;
            dism = dism + e3;
            acc = acc + dism;  // dist += dism
            alive = acc <= tl;
          }
        }
        // a row whose complete sum is not above its query's threshold: gathered (no memory access yet)
        const unsigned long long m = __ballot(alive);
        if (m != 0ull) {
          const int pos = ccnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (alive) {
            cb_d[pos] = acc;
            cb_row[pos] = row;
            cb_i[pos] = i;
          }
          ccnt += __popcll(m);
          if (ccnt >= BM_BATCH) flush();
        }
      };

      // rows [rs, re) of the step's item at `base`; p01 (SUB): l0 + l1 of the run, per query
      // (SUB) is the cached run out of every query's reach?  Re-evaluated only when the run or the
      // thresholds change -- comparing two scalar floats takes vector instructions on this ISA
      bool run_dead = false;
      auto step = [&](const Item &cur, const int kstep, const int base, const int rs, const int re, const float *p01,
                      const bool run_changed) {
        refresh(kstep);
        if (SUB) {  // the run may have dropped out of reach since the item began
          if (run_changed || (kstep & (BM_THR_EVERY - 1)) == 0) {
            bool dead = true;
#pragma unroll
            for (int i = 0; i < QB; i++) dead = dead && !(p01[i] <= thr[i]);
            run_dead = __builtin_amdgcn_readfirstlane((int)dead) != 0;
          }
          if (run_dead) return;
        }
        const bool interior = base >= rs && base + WSTEP <= re;  // wave-uniform
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          const int row = base + lane * ROWS + r;
          const uint32_t c0 = cur.word(r, 0);
          float part[QB];
          bool alive[QB];
          // The first group's sum in a straight line -- SUB: dism (= l0 + l1, the run's) += l2; dism += l3;
          // otherwise dism = l0; dism += l1; dism += l2; dism += l3 -- and one test per query.  (A test after
          // l1 saved nothing: some lane of nearly every step passes it, as in the best-first form.)  On the
          // steps at a row range's edges the rows outside it get +inf in place of their sums: the
          // wave-uniform branch keeps the range tests off the interior steps, and the masks below are the
          // compares' own results.
          const VT v2 = lds_lut_vec<QB, 2, VT>(2, c0);
          const VT v3 = lds_lut_vec<QB, 3, VT>(3, c0);
          VT pv;  // (element-wise vector adds: the packed fp32 add does two of them per instruction)
          if (SUB) {
#pragma unroll
            for (int i = 0; i < QB; i++) pv[i] = p01[i];
          } else {
            const VT v1 = lds_lut_vec<QB, 1, VT>(1, c0);
#pragma unroll
            for (int i = 0; i < QB; i++) pv[i] = l0[i];
            pv = pv + v1;
          }
          pv = pv + v2;
          pv = pv + v3;
          if (!interior) {
            if (row < rs || row >= re) {
#pragma unroll
              for (int i = 0; i < QB; i++) pv[i] = INFINITY;
            }
          }
          unsigned long long any = 0ull;
#pragma unroll
          for (int i = 0; i < QB; i++) {
            part[i] = pv[i];
            alive[i] = part[i] <= thr[i];
            any |= __ballot(alive[i]);
          }
          if (any == 0ull) continue;
#pragma unroll
          for (int i = 0; i < QB; i++) {
            const unsigned long long m = __ballot(alive[i]);
            if (m != 0ull) {
              const int qp = qcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
              if (alive[i]) {
                q_row[qp] = row;
                q_part[qp] = part[i];
                q_i[qp] = i;
#pragma unroll
                for (int w = 0; w < QCW; w++) q_cw[w * BM_QCAP + qp] = cur.word(r, w + 1);
              }
              qcnt += __popcll(m);
              if (qcnt >= 64) drain(64);
            }
          }
        }
      };

      // ---- the item's wave steps: wave w takes steps w, w + nwaves, ... (the workgroup streams the
      //      rows front to back, as every other group of this bucket does at the same time) ----
      const int nmine = wave < nsteps ? (nsteps - wave + nwaves - 1) / nwaves : 0;
      if (nmine > 0) {
        const int base00 = bs & ~(WSTEP - 1);
        // step number -> first row of its item (SUB: through the prefix of the live runs' steps; the
        // cursor only moves forward) [+ the run's rows and l0 + l1]
        // A cursor keeps its run's step range and first item row in scalar registers: a step inside the
        // range costs one scalar compare and no LDS access (runs are tens to hundreds of steps long and a
        // wave takes every 16th step); only when a step leaves the range is the prefix walked on.
        struct Cursor {
          int run, lo, hi, row0;  // steps [lo, hi) belong to `run`, whose first item starts at row0
        };
        auto seek = [&](Cursor &c, const int sidx) {
          if (sidx < c.hi) return;  // (wave-uniform: the values come from readfirstlane)
          int r = c.run, hi = c.hi;
          do {
            r++;
            hi = __builtin_amdgcn_readfirstlane(s_cum[r + 1]);
          } while (hi <= sidx);
          c.run = r;
          c.hi = hi;
          c.lo = __builtin_amdgcn_readfirstlane(s_cum[r]);
          c.row0 = __builtin_amdgcn_readfirstlane(s_run_s[r]) & ~(WSTEP - 1);
        };
        auto step_base = [&](const int kk, Cursor &c) -> int {
          const int kc = kk < nmine - 1 ? kk : nmine - 1;
          const int sidx = wave + kc * nwaves;
          if (!SUB) return base00 + sidx * WSTEP;
          seek(c, sidx);
          return c.row0 + (sidx - c.lo) * WSTEP;
        };
        Cursor cur_l = {-1, 0, 0, 0}, cur_p = {-1, 0, 0, 0};  // of the loads (ahead) and of the steps
        int run_p = -1;                                           // run whose rows and l0 + l1 are cached
        int rs = bs, re = be;
        float p01[QB];
#pragma unroll
        for (int i = 0; i < QB; i++) p01[i] = 0.0f;
        auto do_step = [&](const Item &it, const int kk) {
          const int base = step_base(kk, cur_p);
          const bool run_changed = SUB && cur_p.run != run_p;
          if (run_changed) {
            run_p = cur_p.run;
            rs = __builtin_amdgcn_readfirstlane(s_run_s[run_p]);
            re = __builtin_amdgcn_readfirstlane(s_run_e[run_p]);
#pragma unroll
            for (int i = 0; i < QB; i++)
              p01[i] = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(s_p01[run_p * QB + i])));
          }
          step(it, kk, base, rs, re, p01, run_changed);
        };
        Item ring[BM_RING];
#pragma unroll
        for (int u = 0; u < BM_RING; u++) ring[u].load(codes, (int64_t)(step_base(u, cur_l) / ROWS) + lane);
        int kk = 0;
        for (; kk + BM_RING <= nmine; kk += BM_RING) {
#pragma unroll
          for (int u = 0; u < BM_RING; u++) {
            do_step(ring[u], kk + u);
            // (past the end: the last item again, a cache hit)
            ring[u].load(codes, (int64_t)(step_base(kk + u + BM_RING, cur_l) / ROWS) + lane);
          }
        }
#pragma unroll
        for (int u = 0; u < BM_RING; u++)
          if (kk + u < nmine) do_step(ring[u], kk + u);
      }
      while (qcnt > 0) drain(qcnt < 64 ? qcnt : 64);
      flush();  // (the slots of a candidate are this item's: nothing may stay gathered)
    }
  }
}

// ---------------------------------------------------------------------------
// select
// ---------------------------------------------------------------------------
// One workgroup per query: the k smallest (distance bits, label) keys of pass A's list and the
// candidates at or below the final threshold, ascending -- heap_reorder's output order
// (utils/Heap.hpp:322-349), empty slots -1 / FLT_MAX.
constexpr int BM_SELECT_THREADS = 256;
__global__ __launch_bounds__(BM_SELECT_THREADS) void bm_select_kernel(BmParams p, int P_max) {
  extern __shared__ unsigned long long sk[];  // [P_max]
  __shared__ unsigned s_n;
  const int q = blockIdx.x, tid = threadIdx.x, k = p.k;
  const unsigned done = p.done_key[q];
  if (done == 0xffffffffu) return;  // pass A finished this query: its list is the result
  const unsigned cnt_all = p.cand_cnt[q];
  const unsigned long long t64 = p.thr64[q];
  const unsigned thr = (unsigned)(t64 >> 32);
  const bool over = cnt_all > (unsigned)p.cap;
  const unsigned cnt = over ? (unsigned)p.cap : cnt_all;
  if (over && !p.retry) {
    // more candidates than slots, and no round left to try again: the best-first form finishes this
    // query on its own
    if (tid == 0) {
      const unsigned idx = atomicAdd(p.defer_count, 1u);
      if (idx < (unsigned)p.defer_cap) {
        DeferRec rec;
        rec.q = q;
        rec.done_key = done;
        rec.thr = thr;
        rec.pad = (int)p.fresh[q];  // (nothing of the query is finished yet)
        p.defer_list[idx] = rec;
        atomicMin(&p.g_thr[q], thr);  // (the word its workgroups share)
      }
      p.done_key[q] = 0xffffffffu;  // later rounds leave the query alone
    }
    return;
  }
  if (tid == 0) s_n = 0u;
  __syncthreads();
  for (int i = tid; i < k; i += BM_SELECT_THREADS) {
    const int32_t lab = p.labels[(size_t)q * k + i];
    if (lab >= 0)
      sk[atomicAdd(&s_n, 1u)] = ((unsigned long long)float_to_bits(p.dist[(size_t)q * k + i]) << 32) |
                                (unsigned)((int64_t)lab - p.id_base);
  }
  for (unsigned i = tid; i < cnt; i += BM_SELECT_THREADS) {
    const unsigned long long key = ((unsigned long long)float_to_bits(p.cand_d[(size_t)q * p.cap + i]) << 32) |
                                   (unsigned)p.cand_id[(size_t)q * p.cap + i];
    if (key < t64) sk[atomicAdd(&s_n, 1u)] = key;  // (distances are >= 0: bit order == value order)
  }
  __syncthreads();
  const int n = (int)s_n;
  int P = 2;
  while (P < n) P <<= 1;
  for (int i = n + tid; i < P; i += BM_SELECT_THREADS) sk[i] = ~0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += BM_SELECT_THREADS) {
        const int i = 2 * t - (t & (stride - 1));
        const int j = i + stride;
        const unsigned long long a = sk[i], c = sk[j];
        if ((a > c) == ((i & size) == 0)) { sk[i] = c; sk[j] = a; }
      }
      __syncthreads();
    }
  if (over) {
    // More candidates than slots: the stored ones are still real rows, so the k-th smallest key here
    // bounds the final k-th distance.  Nothing else is kept -- the list and done_key stay as they
    // were, and the NEXT round plans the same buckets again under the tighter threshold.
    // (the bound is the k-th pair itself, label included -- among rows of EQUAL distance only the
    //  smallest labels are wanted, or a query with thousands of identical rows would overflow for ever)
    if (tid == 0 && n >= k) atomicMin(&p.thr64[q], sk[k - 1] + 1ull);
    return;
  }
  for (int i = tid; i < k; i += BM_SELECT_THREADS) {
    const bool ok = i < n;
    const unsigned long long key = ok ? sk[i] : 0ull;
    p.labels[(size_t)q * k + i] = ok ? (int32_t)((int64_t)(int)(unsigned)(key & 0xffffffffull) + p.id_base) : -1;
    p.dist[(size_t)q * k + i] = ok ? bits_to_float((unsigned)(key >> 32)) : FLT_MAX;
  }
  if (tid == 0) {
    // the round's buckets are finished; the k-th distance found so far bounds the final one
    p.done_key[q] = p.done_next[q];
    p.fresh[q] = 0u;
    if (n >= k) atomicMin(&p.thr64[q], sk[k - 1] + 1ull);
  }
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
bool scan_bm_supported(int layout, int M, int n_buckets, int bucket_shift, int seq, int k) {
  if (layout != LAYOUT_BYTES || (M != 8 && M != 16 && M != 32)) return false;
  return bucket_shift == 0 && !seq && n_buckets >= 32 && n_buckets <= BF_MAX_BUCKETS &&
         (n_buckets & (n_buckets - 1)) == 0 && k >= 1 && k <= 1024;
}

size_t scan_bm_lds_bytes(int M, int qb, int nwaves) { return bm_lds_bytes(M, qb, nwaves); }

size_t bm_plan_small_words(int n_buckets) {
  // cnt, qoff (+1), fill, border, ioff, tickets
  return (size_t)n_buckets * 4 + 1 + (size_t)BM_XCDS * bm_ioff_stride(n_buckets) + BM_XCDS;
}

// Staged search: the thresholds (distance bits; >= 0, so ordered like the values) leave the index for the
// exchange between shards and come back as the minimum over all of them.  A threshold is an upper bound
// of the query's final k-th distance, and any shard's bound holds for every shard: the k rows behind it
// exist, wherever they are.  Label bounds are shard-local and are not exchanged (rows AT the exchanged
// distance stay admissible).
__global__ void bm_thresholds_kernel(BmParams p, const int32_t *__restrict__ thr_in, int32_t *__restrict__ thr_out, int init) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= p.nq) return;
  if (init) p.thr64[q] = ((unsigned long long)p.g_thr[q] << 32) | 0x7fffffffull;
  if (thr_in) {
    atomicMin(&p.thr64[q], ((unsigned long long)(unsigned)thr_in[q] << 32) | 0x7fffffffull);
    atomicMin(&p.g_thr[q], (unsigned)thr_in[q]);  // (what a first round still to come, and the best-first fallback, start from)
  }
  if (thr_out) thr_out[q] = (int32_t)(p.thr64[q] >> 32);
}

hipError_t launch_bm_thresholds(const BmParams &p, const int32_t *thr_in, int32_t *thr_out, int init, hipStream_t st) {
  if (p.nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(bm_thresholds_kernel, dim3((p.nq + 255) / 256), dim3(256), 0, st, p, thr_in, thr_out, init);
  return hipGetLastError();
}

hipError_t launch_bm_boot(const BmParams &p, int64_t n_rows, hipStream_t st) {
  if (p.nq <= 0) return hipSuccess;
  (void)n_rows;
  if (p.k > BM_BOOT_ROWS) return hipErrorInvalidValue;
  const size_t lds = (size_t)p.M * 256 * sizeof(float);
  switch (p.M) {
  case 8: hipLaunchKernelGGL(bm_boot_kernel<8>, dim3(p.nq), dim3(256), lds, st, p); break;
  case 16: hipLaunchKernelGGL(bm_boot_kernel<16>, dim3(p.nq), dim3(256), lds, st, p); break;
  case 32: hipLaunchKernelGGL(bm_boot_kernel<32>, dim3(p.nq), dim3(256), lds, st, p); break;
  default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_bm_plan(const BmParams &p, hipStream_t st) {
  if (p.nq <= 0) return hipSuccess;
  if (p.n_buckets > BF_MAX_BUCKETS || p.n_buckets < 32 || p.bucket_t > GMIN_MAX_BITS || (p.qb != 2 && p.qb != 4))
    return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(p.cnt, 0, (size_t)p.n_buckets * sizeof(int), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bm_mark_kernel, dim3(p.nq), dim3(256), 0, st, p);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  hipLaunchKernelGGL(bm_order_kernel, dim3(1), dim3(BM_MAX_THREADS), 0, st, p);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  hipLaunchKernelGGL(bm_fill_kernel, dim3(p.nq), dim3(64), 0, st, p);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  int P = 2;
  while (P < p.nq) P <<= 1;
  const size_t lds = (size_t)P * sizeof(unsigned);
  if (p.nq > (1 << 14)) return hipErrorInvalidValue;  // (a list entry packs the query into 14 bits)
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(bm_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
    return e;
  hipLaunchKernelGGL(bm_sort_kernel, dim3(p.n_buckets), dim3(BM_SORT_THREADS), lds, st, p);
  return hipGetLastError();
}

template <typename K>
static hipError_t launch_bm_kernel(K kernel, const BmParams &p, size_t lds, int grid, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(p.nwaves * 64), lds, st, p);
  return hipGetLastError();
}

hipError_t launch_scan_bm(const BmParams &p, int n_cu, hipStream_t st) {
  if (p.nq <= 0) return hipSuccess;
  if (p.nwaves < 1 || p.nwaves > 16) return hipErrorInvalidValue;
  const size_t lds = bm_lds_bytes(p.M, p.qb, p.nwaves) + BM_TABLES_BYTES;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  // persistent workgroups: as many as are resident at once, a multiple of the XCD count
  int per_cu = (int)((160 * 1024) / lds);
  const int by_waves = 32 / p.nwaves;
  per_cu = per_cu < by_waves ? per_cu : by_waves;
  per_cu = per_cu < 1 ? 1 : per_cu;
  const int grid = ((n_cu * per_cu + BM_XCDS - 1) / BM_XCDS) * BM_XCDS;
#define VAQ_BM_CASE(MM)                                                                               \
  case MM:                                                                                            \
    if (p.sub_start)                                                                                  \
      return p.qb == 4 ? launch_bm_kernel(scan_bm_kernel<MM, 4, true>, p, lds, grid, st)              \
                       : launch_bm_kernel(scan_bm_kernel<MM, 2, true>, p, lds, grid, st);             \
    return p.qb == 4 ? launch_bm_kernel(scan_bm_kernel<MM, 4, false>, p, lds, grid, st)               \
                     : launch_bm_kernel(scan_bm_kernel<MM, 2, false>, p, lds, grid, st);
  switch (p.M) {
    VAQ_BM_CASE(8)
    VAQ_BM_CASE(16)
    VAQ_BM_CASE(32)
  default: return hipErrorInvalidValue;
  }
#undef VAQ_BM_CASE
}

hipError_t launch_bm_select(const BmParams &p, hipStream_t st) {
  if (p.nq <= 0) return hipSuccess;
  int P = 2;
  while (P < p.k + p.cap) P <<= 1;
  const size_t lds = (size_t)P * sizeof(unsigned long long);
  if (lds > 128 * 1024) return hipErrorInvalidValue;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bm_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bm_select_kernel, dim3(p.nq), dim3(BM_SELECT_THREADS), lds, st, p, P);
  return hipGetLastError();
}

} // namespace vaq
