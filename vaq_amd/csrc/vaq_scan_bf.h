// vaq_scan_bf.h -- the best-first form of the early-abandon scan (VAQ::searchEarlyAbandon,
// VAQ.cpp:1694-1727; results identical to VAQ::searchHeap, :1729-1758), one query per
// workgroup.  Chosen by the host when a workgroup's row slice spans many buckets (rows sharing
// their first code): the cache-resident databases, where the scan is bound by instruction issue
// and by the serial k-min, not by memory.
//
//  * The workgroup sorts ALL buckets of its slice by the lower bound of their row sums (the
//    first lookup-table term) and cuts them into work units of BF_SEG_STEPS wave steps.  Waves
//    pull units from one LDS ticket, nearest bucket first.  Thresholds only move down and the
//    bounds only go up along that order, so the first unit whose bound exceeds the wave's
//    threshold ends the wave's scan: no bucket is tested twice, none is walked past, and the
//    waves finish together whatever the sizes of the buckets.
//  * Rows that survive every partial-sum test are appended to a WAVE-PRIVATE candidate buffer
//    (no lock, LDS is in order per wave).  When BF_FLUSH_AT have gathered, the wave reads their
//    labels in one go (one global-memory latency for the batch), takes the workgroup lock,
//    re-tests them against the exact (distance, label) threshold and appends the survivors to
//    the workgroup's pool: an unsorted k-min whose threshold moves only when it fills up
//    (pool_compact below).  The admission rule is vaq_scan.h's: admit iff strictly below the
//    threshold in (distance, label) order.
// Arithmetic per row is scan_bytes_body's: dism = l0; dism += l1; dism += l2; dism += l3;
// dist += dism, group by group (VAQ.cpp:1737-1748).
#ifndef VAQ_SCAN_BF_H_
#define VAQ_SCAN_BF_H_

#include "vaq_scan.h"

namespace vaq {

#ifndef VAQ_BF_SEG
#define VAQ_BF_SEG 8
#endif
#ifndef VAQ_BF_FLUSH
#define VAQ_BF_FLUSH 16
#endif
#ifndef VAQ_BF_BOOT
#define VAQ_BF_BOOT 4
#endif
#ifndef VAQ_BF_RING
#define VAQ_BF_RING 5
#endif
constexpr int BF_SEG_STEPS = VAQ_BF_SEG;        // wave steps per work unit
constexpr int BF_FLUSH_AT = VAQ_BF_FLUSH;       // candidates a wave gathers before it takes the lock
constexpr int BF_CB_CAP = BF_FLUSH_AT - 1 + 64; // one drain appends at most 64
constexpr int BF_RING = VAQ_BF_RING;            // code items in flight per wave (register sets)
constexpr int BF_QCAP = 128;                    // survivor queue: at most 63 left over + 64 pushed
constexpr int BF_MAX_BUCKETS = 1024;
constexpr int BF_BOOT_STEPS = VAQ_BF_BOOT;      // wave steps each wave samples to bootstrap the threshold (0 = off)

__host__ __device__ inline size_t bf_align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline int bf_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}
// slots of the k-min pool (below)
#ifndef VAQ_BF_POOL
#define VAQ_BF_POOL 512
#endif
__host__ __device__ inline int bf_pool_cap(int kp) { return 2 * kp < VAQ_BF_POOL ? VAQ_BF_POOL : 2 * kp; }
constexpr int BF_HIST_BINS = 64;  // one per lane
constexpr int BF_HDR_SCALE = 5;   // header word: float bits of bins / H, 0 = histogram off
// code dwords a survivor carries through the queue (the rest of its row, byte codes M <= 16)
__host__ __device__ inline int bf_queue_code_words(int M) { return M <= 16 ? M / 4 - 1 : 0; }
// LDS of one workgroup: [LUT][k-min][sorted bucket keys, row ranges, unit prefix, ticket, gmin]
// [per wave: survivor queue, candidate buffer]
__host__ __device__ inline size_t bf_lds_bytes(int lut_entries, int kp, int n_buckets, int nwaves, int qcw) {
  size_t b = bf_align16((size_t)lut_entries * 4);
  b += bf_align16((size_t)SEL_HDR_WORDS * 4 + (size_t)bf_pool_cap(kp) * 8) + (size_t)BF_HIST_BINS * 4;
  b += bf_align16((size_t)n_buckets * 12 + (size_t)(n_buckets + 1) * 4 + 12 + (size_t)(1 << GMIN_MAX_BITS) * 4);
  b += (size_t)nwaves * ((size_t)BF_QCAP * 4 * (2 + qcw) + (size_t)BF_CB_CAP * 8);
  return b;
}

// ---------------------------------------------------------------------------
// k-min of the best-first form: an UNSORTED pool of admitted rows in LDS.
//   header  lock, count, threshold (distance bits, label)       (SelView, vaq_scan.h)
//   [0, cap) (distance, label) pairs, cap = bf_pool_cap(kp) >= 2 kp
// A row is admitted iff it is strictly below the threshold in (distance, label) order --
// VAQ::searchHeap's rule (VAQ.cpp:1750-1753: push iff heap top > dist) with the heap top
// replaced by any upper bound of the final k-th best, which never changes the result.
// Appending is a few LDS words under the workgroup lock; the threshold only moves when the
// pool is full: pool_compact() finds the k-th smallest distance by bisection on its bits
// (counting with ballots), keeps the rows at or below it and makes it the threshold.
// In between, a 64-bin histogram of the admitted distances over [0, H] (H = the bootstrap
// threshold) gives a cheap bound: the upper edge of the bin in which the cumulative count
// reaches k has at least k admitted rows at or below it, so it is a valid threshold; one
// LDS word per lane and a wave prefix sum per batch of candidates instead of a sorted merge.
// The pool is cut to the exact k best and sorted once, when the scan is over.
// ---------------------------------------------------------------------------
// entries of d[0, n) with distance bits <= t (distances are >= 0 or NaN-free: bit order == value order)
__device__ __forceinline__ int pool_count_le(const float *d, const int n, const unsigned t, const int lane) {
  int c = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    c += __popcll(__ballot(i < n && float_to_bits(d[i]) <= t));
  }
  return c;
}

// One wave, lock held, pool holds n > k rows.  Keeps every row at or below the k-th smallest
// distance (>= k rows; more only when rows tie at that distance) and lowers the threshold to
// (that distance, INT_MAX).  If ties leave less than `room` free slots, the tie is cut exactly:
// the pool is sorted by (distance, label), the k smallest stay and the threshold becomes the
// k-th pair itself.  Returns the new count.
__device__ __forceinline__ int pool_compact(const SelView &sel, const int n, const int k, const int cap,
                                            const int room, const int lane) {
  unsigned lo = 0u, hi = 0x7f800000u;  // smallest t with count(bits <= t) >= k
  if (n <= 512) {
    // the distances in registers (8 per lane): a bisection step is compares and ballots only
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int i = e * 64 + lane;
      v[e] = i < n ? float_to_bits(sel.d[i]) : 0xffffffffu;
    }
    while (lo < hi) {
      const unsigned mid = lo + ((hi - lo) >> 1);
      int c = 0;
#pragma unroll
      for (int e = 0; e < 8; e++) c += __popcll(__ballot(v[e] <= mid));
      if (c >= k) hi = mid;
      else lo = mid + 1u;
    }
  } else {
    while (lo < hi) {
      const unsigned mid = lo + ((hi - lo) >> 1);
      if (pool_count_le(sel.d, n, mid, lane) >= k) hi = mid;
      else lo = mid + 1u;
    }
  }
  const unsigned t = lo;
  int w = 0;
  for (int base = 0; base < n; base += 64) {  // in-place, forwards: writes never pass the reads
    const int i = base + lane;
    const float di = i < n ? sel.d[i] : INFINITY;
    const int ii = i < n ? sel.id[i] : ID_SENTINEL;
    const bool keep = i < n && float_to_bits(di) <= t;
    const unsigned long long m = __ballot(keep);
    wave_lds_sync();
    if (keep) {
      const int pos = w + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      sel.d[pos] = di;
      sel.id[pos] = ii;
    }
    w += __popcll(m);
    wave_lds_sync();
  }
  float td = bits_to_float(t);
  int ti = INT_MAX;
  if (w + room > cap) {  // (rows tying at the k-th distance fill the pool: cut the tie by label)
    int P = 2;
    while (P < w) P <<= 1;
    for (int i = w + lane; i < P; i += 64) {
      sel.d[i] = INFINITY;
      sel.id[i] = ID_SENTINEL;
    }
    wave_lds_sync();
    bitonic_sort<false>(sel.d, sel.id, P, lane, 64);
    w = k;
    td = sel.d[k - 1];
    ti = sel.id[k - 1];
  }
  const float od = bits_to_float(sel.hdr[SEL_THR_D]);
  const int oi = (int)sel.hdr[SEL_THR_ID];
  if (lane == 0) {
    if (pair_less(td, ti, od, oi)) {
      sel.hdr[SEL_THR_D] = float_to_bits(td);
      sel.hdr[SEL_THR_ID] = (unsigned)ti;
    }
    sel.hdr[SEL_NCAND] = (unsigned)w;
  }
  wave_lds_sync();
  return w;
}

template <int M, bool UL0>
__device__ __forceinline__ void scan_bytes_bf_body(const ScanParams &p) {
  typedef BytesItem<M> Item;
  constexpr int WPR = Item::WPR;
  constexpr int QCW = (M <= 16) ? WPR - 1 : 0;
  constexpr int WSTEP = 64 * Item::ROWS;  // rows per wave step
  constexpr int SEG_ROWS = BF_SEG_STEPS * WSTEP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int total = p.nq * p.n_slices;
  const int v = xcd_virtual_id(blockIdx.x, gridDim.x);
  if (v >= total) return;
  const int slice = v / p.nq;
  const int qi = v - slice * p.nq;
  const int r0 = (int)((int64_t)slice * p.slice_stride);
  const int64_t r1l = (int64_t)r0 + p.slice_rows;
  const int r1 = (int)(r1l > p.n_rows ? p.n_rows : r1l);
  const int K0 = p.n_buckets;
  const int K0p = bf_pow2(K0);
  const unsigned idx_mask = (unsigned)K0p - 1u;
  const int k = p.k, kp = p.kp;
  const bool multi_slice = p.share_thr != 0;
  const int bshift = p.bucket_shift, bt = p.bucket_t;

  // ---- LDS carve-up (bf_lds_bytes) ----
  float *lut = reinterpret_cast<float *>(smem);
  size_t off = bf_align16((size_t)M * 256 * 4);
  const int cap = bf_pool_cap(kp);
  const SelView sel = sel_view(smem + off, cap, 0);
  off += bf_align16((size_t)SEL_HDR_WORDS * 4 + (size_t)cap * 8);
  unsigned *hist = reinterpret_cast<unsigned *>(smem + off);  // [64] admitted rows per distance bin
  off += (size_t)BF_HIST_BINS * 4;
  unsigned *keys = reinterpret_cast<unsigned *>(smem + off);  // [K0] sorted: bound bits | bucket
  int *s0a = reinterpret_cast<int *>(keys + K0);              // [K0] first row of the i-th best bucket
  int *e0a = s0a + K0;                                        // [K0] one past its last row (inside the slice)
  int *cum = e0a + K0;                                        // [K0 + 1] prefix of work units
  unsigned *ticket = reinterpret_cast<unsigned *>(cum + K0 + 1);
  unsigned *boot_cnt = ticket + 1;                            // bootstrap: rows counted below boot_thr
  unsigned *boot_thr = ticket + 2;                            //            float bits, max over the waves
  unsigned *gmin = ticket + 3;                                // [1 << bt] smallest second term per group
  off += bf_align16((size_t)K0 * 12 + (size_t)(K0 + 1) * 4 + 12 + (size_t)(1 << GMIN_MAX_BITS) * 4);
  unsigned *ktmp = reinterpret_cast<unsigned *>(smem + off);  // [K0] unsorted keys: borrows the waves' buffers
  unsigned char *wb = smem + off + (size_t)wave * ((size_t)BF_QCAP * 4 * (2 + QCW) + (size_t)BF_CB_CAP * 8);
  int *q_id = reinterpret_cast<int *>(wb);
  float *q_p = reinterpret_cast<float *>(q_id + BF_QCAP);
  uint32_t *q_cw = reinterpret_cast<uint32_t *>(q_p + BF_QCAP);
  float *cb_d = reinterpret_cast<float *>(q_cw + (size_t)QCW * BF_QCAP);
  int *cb_row = reinterpret_cast<int *>(cb_d + BF_CB_CAP);

#ifdef VAQ_STATS
  struct { unsigned long long st[ST_N]; } cx;
  for (int i = 0; i < ST_N; i++) cx.st[i] = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  // ---- setup ----
  const float *__restrict__ glut = p.lut + (size_t)qi * p.lut_floats;
  const int *__restrict__ bstart = p.bucket_start;
  for (int e = tid; e < M * 256; e += nthreads) lut[e] = glut[e];
  if (tid == 0) {
    unsigned td = float_to_bits(FLT_MAX);
    int ti = INT_MIN;
    if (multi_slice) {
      const unsigned g = __hip_atomic_load(&p.g_thr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g < td) { td = g; ti = INT_MAX; }
    }
    sel.hdr[SEL_LOCK] = 0u;
    sel.hdr[SEL_NCAND] = 0u;
    sel.hdr[SEL_NBEST] = 0u;
    sel.hdr[SEL_THR_D] = td;
    sel.hdr[SEL_THR_ID] = (unsigned)ti;
    sel.hdr[BF_HDR_SCALE] = 0u;
    *ticket = 0u;
    cum[0] = 0;
  }
  if (bt > 0) {  // the bucket key continues into the second code: its groups' smallest terms
    for (int i = tid; i < (1 << bt); i += nthreads) gmin[i] = 0x7f800000u;
    __syncthreads();
    const int w = 8 - bt;  // log2 of the group size (second code: 8 bits)
    for (int e = tid; e < 256; e += nthreads) atomicMin(&gmin[e >> w], float_to_bits(glut[256 + e]));
    __syncthreads();
  }
  // one packed word per bucket: lower bound of its row sums (low bits cut: still a lower bound) |
  // bucket -- unique, so ranks are positions.  Empty buckets and NaN tables sort last.
  const unsigned empty_key = ~idx_mask;
  for (int b = tid; b < K0; b += nthreads) {
    unsigned key = empty_key | (unsigned)b;
    const int s0 = bstart[b] > r0 ? bstart[b] : r0;
    const int e0 = bstart[b + 1] < r1 ? bstart[b + 1] : r1;
    if (e0 > s0) {
      float m;
      if (bt > 0) {
        m = glut[b >> bt] + bits_to_float(gmin[b & ((1 << bt) - 1)]);
      } else {
        m = INFINITY;
        for (int c = b << bshift; c < ((b + 1) << bshift); c++) {
          const float x = glut[c];
          m = x < m ? x : m;
        }
      }
      if (m == m) key = (float_to_bits(m) & ~idx_mask) | (unsigned)b;  // m >= 0: bit order == value order
    }
    ktmp[b] = key;
  }
  if (tid == 0) {
    *boot_cnt = 0u;
    *boot_thr = 0u;
  }
  if (tid < BF_HIST_BINS) hist[tid] = 0u;
  __syncthreads();
  // rank sort: position = number of smaller keys (every lane reads the same words: LDS broadcasts)
  for (int b = tid; b < K0; b += nthreads) {
    const unsigned key = ktmp[b];
    int rank = 0;
    for (int j = 0; j < K0; j += 4) {
      const uint4 o = *reinterpret_cast<const uint4 *>(ktmp + j);  // K0 is a power of two >= 16
      rank += (o.x < key) + (o.y < key) + (o.z < key) + (o.w < key);
    }
    int s0 = 0, e0 = 0, units = 0;
    if ((key & empty_key) != empty_key) {
      s0 = bstart[b] > r0 ? bstart[b] : r0;
      e0 = bstart[b + 1] < r1 ? bstart[b + 1] : r1;
      units = (e0 - (s0 & ~(WSTEP - 1)) + SEG_ROWS - 1) / SEG_ROWS;
    }
    keys[rank] = key;
    s0a[rank] = s0;
    e0a[rank] = e0;
    cum[rank + 1] = units;
  }
  __syncthreads();
  if (wave == 0) {  // inclusive prefix of the unit counts, 64 at a time
    int carry = 0;
    for (int base = 0; base < K0; base += 64) {
      const int i = base + lane;
      const int u = i < K0 ? cum[i + 1] : 0;
      int inc = u;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(inc, o);
        if (lane >= o) inc += x;
      }
      if (i < K0) cum[i + 1] = carry + inc;
      carry += __builtin_amdgcn_readlane(inc, 63);
    }
  }
  __syncthreads();
#ifdef VAQ_STATS
  cx.st[ST_CYC_SETUP] = __builtin_readcyclecounter() - t_begin;
#endif
  // ---- per-wave state ----
  float thr_d = FLT_MAX;  // wave-uniform cached copy of the threshold distance (never below the exact one)
  int qcnt = 0, ccnt = 0, stepno = 0;
  const uint32_t *__restrict__ codes = p.codes;
  const uint32_t *__restrict__ perm = p.perm;

  auto refresh = [&](const int st) {
    if ((st & (THR_LOCAL_EVERY - 1)) != 0) return;
    unsigned t = __hip_atomic_load(&sel.hdr[SEL_THR_D], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (multi_slice && wave == 0 && ((st & (THR_GLOBAL_EVERY - 1)) == 0 || st < THR_GLOBAL_EVERY)) {
      const unsigned g = __hip_atomic_load(&p.g_thr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g < t) {
        sel_lock(sel, lane);
        if (g < sel.hdr[SEL_THR_D] && lane == 0) {
          sel.hdr[SEL_THR_D] = g;
          sel.hdr[SEL_THR_ID] = (unsigned)INT_MAX;  // ties at g stay admissible
        }
        sel_unlock(sel, lane);
        t = g;
      }
    }
    thr_d = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)t));
  };

  // up to 64 gathered candidates -> the workgroup's best list
  auto flush = [&]() {
    const int n = ccnt < 64 ? ccnt : 64;
    ccnt -= n;
    const bool has = lane < n;
    const int slot = ccnt + (has ? lane : 0);
    const float d = cb_d[slot];
    const int srow = cb_row[slot];
    const bool cand = has && !(d > thr_d);  // the threshold may have moved since the row was gathered
    if (__ballot(cand) == 0ull) return;
    STAT_T0(t_fl);
    STAT_ADD(ST_ADMITS, 1);
    const int rid = (cand && perm) ? (int)perm[srow] : srow;  // labels are ORIGINAL rows: ties break by them
    STAT_T0(t_lk);
    sel_lock(sel, lane);
    STAT_T1(ST_CYC_LOCKWAIT, t_lk);
    float td = bits_to_float(sel.hdr[SEL_THR_D]);
    int ti = (int)sel.hdr[SEL_THR_ID];
    bool pass = cand && pair_less(d, rid, td, ti);
    unsigned long long m = __ballot(pass);
    if (m != 0ull) {
      int cnt = (int)sel.hdr[SEL_NCAND];
      if (cnt + __popcll(m) > cap) {
        STAT_T0(t_fo);
        STAT_ADD(ST_FOLDS, 1);
        cnt = pool_compact(sel, cnt, k, cap, 64, lane);
        td = bits_to_float(sel.hdr[SEL_THR_D]);
        ti = (int)sel.hdr[SEL_THR_ID];
        if (multi_slice && lane == 0) atomicMin(&p.g_thr[qi], float_to_bits(td));
        pass = pass && pair_less(d, rid, td, ti);
        m = __ballot(pass);
        STAT_T1(ST_CYC_FOLD, t_fo);
      }
      if (m != 0ull) {
        const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (pass) {
          sel.d[pos] = d;
          sel.id[pos] = rid;
        }
        if (lane == 0) sel.hdr[SEL_NCAND] = (unsigned)(cnt + __popcll(m));
        const float scale = bits_to_float(sel.hdr[BF_HDR_SCALE]);
        if (scale != 0.0f) {
          // count the admitted rows by distance bin; the bin where the running total reaches k
          // bounds the k-th best from above
          if (pass) {
            const unsigned b = (unsigned)(d * scale);
            atomicAdd(&hist[b < BF_HIST_BINS - 1 ? b : BF_HIST_BINS - 1], 1u);
          }
          wave_lds_sync();
          int inc = (int)hist[lane];
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int x = __shfl_up(inc, o);
            if (lane >= o) inc += x;
          }
          const unsigned long long reach = __ballot(inc >= k);
          if (reach != 0ull) {
            const int jb = __builtin_ctzll(reach);
            if (jb < BF_HIST_BINS - 1) {  // (the last bin also holds everything beyond H)
              // upper edge of bin jb, widened past any rounding of d * scale
              const float edge = ((float)(jb + 1) / scale) * (1.0f + 1.0f / 1048576.0f);
              if (pair_less(edge, INT_MAX, td, ti)) {
                td = edge;
                ti = INT_MAX;
                if (lane == 0) {
                  sel.hdr[SEL_THR_D] = float_to_bits(td);
                  sel.hdr[SEL_THR_ID] = (unsigned)ti;
                  if (multi_slice) atomicMin(&p.g_thr[qi], float_to_bits(td));
                }
              }
            }
          }
        }
      }
    }
    sel_unlock(sel, lane);
    STAT_T1(ST_CYC_ADMIT, t_fl);
    thr_d = bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(td)));
  };

  // rows whose complete sum is not above the cached threshold: gather, no lock
  auto gather = [&](const float dist, const int srow, const bool ok) {
    const unsigned long long m = __ballot(ok);
    if (m == 0ull) return;
    const int pos = ccnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (ok) {
      cb_d[pos] = dist;
      cb_row[pos] = srow;
    }
    ccnt += __popcll(m);
    if (ccnt >= BF_FLUSH_AT) flush();
  };

  auto lut4 = [&](const uint32_t c4, const int g) -> float {  // dism = l0; dism += l1; dism += l2; dism += l3
    float dism = lut[(g * 4 + 0) * 256 + (c4 & 0xffu)];
    dism = dism + lut[(g * 4 + 1) * 256 + ((c4 >> 8) & 0xffu)];
    dism = dism + lut[(g * 4 + 2) * 256 + ((c4 >> 16) & 0xffu)];
    dism = dism + lut[(g * 4 + 3) * 256 + (c4 >> 24)];
    return dism;
  };

  // phase B: the top n (<= 64) queue entries, one per lane: groups 1.. of the row, abandoning
  // after each (VAQ.cpp:1708)
  auto drain = [&](const int n) {
    STAT_T0(t_dr);
    STAT_ADD(ST_DRAINS, 1);
    qcnt -= n;
    const bool ok = lane < n;
    const int slot = qcnt + (ok ? lane : 0);
    const int rid = q_id[slot];
    float acc = q_p[slot];
    bool alive = ok;
#pragma unroll
    for (int g = 1; g < WPR; g++) {
      const uint32_t cw = QCW > 0 ? q_cw[(g - 1) * BF_QCAP + slot] : codes[(int64_t)rid * WPR + g];
      if (alive) {
        acc = acc + lut4(cw, g);  // dist += dism
        alive = !(acc > thr_d);
      }
    }
    STAT_T1(ST_CYC_DRAIN, t_dr);
    gather(acc, rid, alive);
  };

  auto take_ticket = [&]() -> int {
    int t = 0;
    if (lane == 0) t = (int)atomicAdd(ticket, 1u);
    return __builtin_amdgcn_readfirstlane(t);
  };

  const int total_units = cum[K0];
  // ---- bootstrap: a first threshold from the rows nearest the query ----
  // Wave w sums the rows of the first BF_BOOT_STEPS steps of work unit w completely (the
  // nearest buckets come first) and every lane keeps the smallest distance it saw: 64 distinct
  // rows per wave.  With q = ceil(k / waves), the q-th smallest of a wave's 64 vouches for q
  // rows at or below it, so the largest of the waves' values has at least k rows at or below
  // it: an upper bound of the final k-th distance, i.e. a valid admission threshold (with
  // label INT_MAX: rows AT that distance stay admissible).  Without it every wave's first
  // steps pass every row and the waves queue up on the lock folding them.
  if (BF_BOOT_STEPS > 0 && !p.no_skip) {
    const int nwaves = nthreads >> 6;
    const int q = (k + nwaves - 1) / nwaves;
    if (q <= 64 && wave < total_units) {
      int i = 0;
      while (cum[i + 1] <= wave) i++;
      const int bs = s0a[i], bend = e0a[i];
      const int j = wave - cum[i];
      const int al = bs & ~(WSTEP - 1);
      int pos = al + j * SEG_ROWS;
      if (pos < bs) pos = bs;
      int be = al + (j + 1) * SEG_ROWS;
      if (be > bend) be = bend;
      const int base0 = pos & ~(WSTEP - 1);
      int nst = (be - base0 + WSTEP - 1) / WSTEP;
      if (nst > BF_BOOT_STEPS) nst = BF_BOOT_STEPS;
      float best = INFINITY;
      for (int st = 0; st < nst; st++) {
        Item cur;
        cur.load(codes, (int64_t)((base0 + st * WSTEP) / Item::ROWS) + lane);
#pragma unroll
        for (int r = 0; r < Item::ROWS; r++) {
          const int row = base0 + st * WSTEP + lane * Item::ROWS + r;
          float acc = lut4(cur.word(r, 0), 0);
#pragma unroll
          for (int g = 1; g < WPR; g++) acc = acc + lut4(cur.word(r, g), g);
          if (row >= pos && row < be && acc < best) best = acc;
        }
      }
      int rank = 0;  // position of the lane's value among the wave's 64 (ties by lane)
      for (int o = 0; o < 64; o++) {
        const float dj = bits_to_float((unsigned)__builtin_amdgcn_readlane((int)float_to_bits(best), o));
        rank += (dj < best || (dj == best && o < lane)) ? 1 : 0;
      }
      const int have = __popcll(__ballot(best < INFINITY));
      const int qw = q < have ? q : have;
      if (qw > 0) {
        const unsigned long long at = __ballot(rank == qw - 1);
        const float tw = bits_to_float((unsigned)__builtin_amdgcn_readlane((int)float_to_bits(best), __builtin_ctzll(at)));
        if (lane == 0) {
          atomicAdd(boot_cnt, (unsigned)qw);
          atomicMax(boot_thr, float_to_bits(tw));  // distances are >= 0: bit order == value order
        }
      }
    }
    __syncthreads();
    if (tid == 0 && *boot_cnt >= (unsigned)k) {
      const unsigned tb = *boot_thr;
      if (tb < sel.hdr[SEL_THR_D]) {
        sel.hdr[SEL_THR_D] = tb;
        sel.hdr[SEL_THR_ID] = (unsigned)INT_MAX;
        if (multi_slice) atomicMin(&p.g_thr[qi], tb);
      }
      // histogram over [0, H], H = the threshold every admitted row is at or below from now on
      const float H = bits_to_float(sel.hdr[SEL_THR_D]);
      if (H > 0.0f && H < FLT_MAX) sel.hdr[BF_HDR_SCALE] = float_to_bits((float)BF_HIST_BINS / H);
    }
    __syncthreads();
  }
  refresh(0);  // the bootstrapped / seeded / already published threshold, before the first bound is tested
  // window of the unit prefix in registers: lane j holds cum[wbase + j + 1]
  int wbase = 0;
  int creg = (lane + 1 <= K0) ? cum[lane + 1] : INT_MAX;
  int t = take_ticket();
  while (t < total_units) {
    const int t_next = take_ticket();  // (its LDS round trip overlaps this unit's work)
    STAT_ADD(ST_BUCKETS_TESTED, 1);
    int cnt;
    for (;;) {  // i = largest index with cum[i] <= t
      cnt = __popcll(__ballot(creg <= t));
      if (cnt < 64) break;
      wbase += 64;
      creg = (wbase + lane + 1 <= K0) ? cum[wbase + lane + 1] : INT_MAX;
    }
    const int i = wbase + cnt;
    const unsigned key = (unsigned)__builtin_amdgcn_readfirstlane((int)keys[i]);
    const float lbv = bits_to_float(key & ~idx_mask);
    // buckets come in ascending order of their bound and thresholds only fall: nothing
    // from here on can hold an admissible row
    if (!p.no_skip && lbv > thr_d) break;
    STAT_ADD(ST_BUCKETS_VISITED, 1);
    const int b = (int)(key & idx_mask);
    const int bs = __builtin_amdgcn_readfirstlane(s0a[i]);
    const int bend = __builtin_amdgcn_readfirstlane(e0a[i]);
    const int j = t - __builtin_amdgcn_readfirstlane(cum[i]);
    const int al = bs & ~(WSTEP - 1);
    int pos = al + j * SEG_ROWS;
    if (pos < bs) pos = bs;
    int be = al + (j + 1) * SEG_ROWS;
    if (be > bend) be = bend;
    // the rows' shared first term (fine buckets)
    const float l0 = UL0 ? bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(lut[b >> bt]))) : 0.0f;
    const int base0 = pos & ~(WSTEP - 1);
    const int nst = (be - base0 + WSTEP - 1) / WSTEP;
    const int64_t item0 = (int64_t)(base0 / Item::ROWS) + lane;

    auto step = [&](const Item &cur, const int st) {
      const int base = base0 + st * WSTEP;
      STAT_ADD(ST_STEPS, 1);
      refresh(stepno++);
      const bool interior = base >= pos && base + WSTEP <= be;  // wave-uniform: no per-row range test
      const int row0 = base + lane * Item::ROWS;
      float part[Item::ROWS];
      bool alive[Item::ROWS];
#pragma unroll
      for (int r = 0; r < Item::ROWS; r++) {
        const uint32_t c0 = cur.word(r, 0);
        // A: dism = l0; dism += l1
        const float first = UL0 ? l0 : lut[c0 & 0xffu];
        part[r] = first + lut[256 + ((c0 >> 8) & 0xffu)];
        alive[r] = !(part[r] > thr_d);
        if (!interior) alive[r] = alive[r] && (row0 + r >= pos) && (row0 + r < be);
        STAT_ADD(ST_ALIVE_A, __popcll(__ballot(alive[r])));
      }
#pragma unroll
      for (int r = 0; r < Item::ROWS; r++) {
        bool live = alive[r];
        if (live) {  // A2: dism += l2; dism += l3 -> the first group's sum
          const uint32_t c0 = cur.word(r, 0);
          part[r] = part[r] + lut[512 + ((c0 >> 16) & 0xffu)];
          part[r] = part[r] + lut[768 + (c0 >> 24)];
          live = !(part[r] > thr_d);
        }
        STAT_ADD(ST_ALIVE_A2, __popcll(__ballot(live)));
        if (WPR == 1) {  // (M = 4 would end here; kept for completeness)
          gather(part[r], row0 + r, live);
        } else {
          const unsigned long long m = __ballot(live);
          if (m != 0ull) {
            const int qp = qcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (live) {
              q_id[qp] = row0 + r;
              q_p[qp] = part[r];
#pragma unroll
              for (int w = 0; w < QCW; w++) q_cw[w * BF_QCAP + qp] = cur.word(r, w + 1);
            }
            qcnt += __popcll(m);
            if (qcnt >= 64) drain(64);
          }
        }
      }
    };

    Item ring[BF_RING];
#pragma unroll
    for (int u = 0; u < BF_RING; u++)
      if (u < nst) ring[u].load(codes, item0 + (int64_t)u * 64);
    for (int st = 0; st < nst; st += BF_RING) {
#pragma unroll
      for (int u = 0; u < BF_RING; u++) {
        if (st + u < nst) {
          step(ring[u], st + u);
          if (st + u + BF_RING < nst) ring[u].load(codes, item0 + (int64_t)(st + u + BF_RING) * 64);
        }
      }
    }
    t = t_next;
  }
  STAT_T0(t_tail);
  while (qcnt > 0) drain(qcnt < 64 ? qcnt : 64);
  while (ccnt > 0) flush();

  // ---- results: wave 0 cuts the pool to the k best and sorts them ----
  __syncthreads();
#ifdef VAQ_STATS
  cx.st[ST_CYC_STEPLOAD] = __builtin_readcyclecounter() - t_tail;  // (tail: last drains / flushes + waiting for the other waves)
  cx.st[ST_CYC_TOTAL] = __builtin_readcyclecounter() - t_begin;
  if (p.stats && lane == 0)
    for (int i = 0; i < ST_N; i++) atomicAdd(&p.stats[i], cx.st[i]);
#endif
  if (wave == 0) {
    int n = (int)sel.hdr[SEL_NCAND];
    if (n > k) n = pool_compact(sel, n, k, cap, 0, lane);  // rows at or below the k-th distance
    int P = 2;
    while (P < n) P <<= 1;
    const int pad_to = P > kp ? P : kp;
    for (int i = n + lane; i < pad_to; i += 64) {
      sel.d[i] = INFINITY;
      sel.id[i] = ID_SENTINEL;
    }
    wave_lds_sync();
    bitonic_sort<false>(sel.d, sel.id, P, lane, 64);  // ascending by (distance, label)
    const int nbest = n < k ? n : k;
    if (p.final_labels) {
      // one slice per query: this list IS the result, in the API's format (heap_reorder's:
      // ascending, empty slots -1 / FLT_MAX, utils/Heap.hpp:322-349)
      const size_t o = (size_t)qi * k;
      for (int i = lane; i < k; i += 64) {
        const int id = sel.id[i];
        const bool ok = id != ID_SENTINEL;
        p.final_labels[o + i] = ok ? (int32_t)(id + p.id_base) : -1;
        p.final_dist[o + i] = ok ? sel.d[i] : FLT_MAX;
      }
    } else {
      const size_t o = ((size_t)qi * p.n_slices + slice) * k;
      for (int i = lane; i < k; i += 64) {
        p.part_d[o + i] = sel.d[i];
        p.part_id[o + i] = sel.id[i];
      }
      if (lane == 0) p.part_cnt[(size_t)qi * p.n_slices + slice] = nbest;
    }
  }
}

// bytes of LDS of the byte-code best-first kernel and its launch (vaq_scan_bf.hip)
hipError_t launch_scan_bf(const ScanParams &p, int grid, hipStream_t st);

} // namespace vaq
#endif
