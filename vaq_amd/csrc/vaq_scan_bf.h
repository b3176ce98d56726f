// vaq_scan_bf.h -- the best-first form of the early-abandon scan (VAQ::searchEarlyAbandon,
// VAQ.cpp:1694-1727; results identical to VAQ::searchHeap, :1729-1758), one query per
// workgroup.  Chosen by the host when a workgroup's row slice spans many buckets (rows sharing
// their first code): the cache-resident databases, where the scan is bound by instruction issue
// and by the serial k-min, not by memory.  Byte codes (BfBytes) and bit-packed rows (BfBits)
// share the scaffold scan_bf_body; DESIGN.md section 4, "Best-first form".
//
//  * Bootstrap: before anything is merged, the waves sum a few steps of the NEAREST bucket's
//    rows completely; the q-th smallest of each wave's 64 lane minima, maximised over the waves,
//    has at least k rows at or below it -- a valid first threshold.
//  * Order: every bucket whose lower bound (first lookup-table term, ...) is not above the
//    threshold joins a ROUND (at most BF_ROUND_BUCKETS, the nearest first when there are more),
//    rank-sorted by bound and cut into work units of BF_SEG_STEPS wave steps.  Waves pull units
//    from one LDS ticket.  Thresholds only move down and the bounds only go up along that order,
//    so the first unit whose bound exceeds a wave's threshold ends its scan: no bucket is tested
//    twice, none is walked past, and the waves finish together whatever the bucket sizes.
//  * k-min: rows that survive every partial-sum test are appended to a WAVE-PRIVATE candidate
//    buffer (no lock, LDS is in order per wave).  When BF_FLUSH_AT have gathered, the wave reads
//    their labels in one go (one global-memory latency for the batch), takes the workgroup lock
//    for a few LDS words, re-tests them against the exact (distance, label) threshold and appends
//    the survivors to the workgroup's unsorted pool; a 64-bin histogram of the admitted distances
//    moves the threshold between the (rare) exact compactions of the pool, and the pool is cut to
//    the k best and sorted once, at the end.  The admission rule is vaq_scan.h's: admit iff
//    strictly below the threshold in (distance, label) order.
// Arithmetic per row is scan_bytes_body's / scan_bits_body's: dism = l0; dism += l1; dism += l2;
// dism += l3; dist += dism, group by group (VAQ.cpp:1737-1748).
#ifndef VAQ_SCAN_BF_H_
#define VAQ_SCAN_BF_H_

#include "vaq_scan.h"

namespace vaq {

#ifndef VAQ_BF_SEG
#define VAQ_BF_SEG 64  // (with the prefetch working a unit's fixed cost is what counts: C2 8 / 16 / 32 / 64 / 128 steps 0.748 / 0.712 / 0.700 / 0.688 / 0.689 ms, C4 27.1 / - / 25.6 / 25.4 / 25.0)
#endif
#ifndef VAQ_BF_FLUSH
#define VAQ_BF_FLUSH 16
#endif
#ifndef VAQ_BF_BOOT
#define VAQ_BF_BOOT 2  // (with the ranked dispatch: 4 / 2 / 1 steps C2 0.546 / 0.539 / 0.697 ms, C3 1.085 / 0.990 / 1.217)
#endif
#ifndef VAQ_BF_RING
#define VAQ_BF_RING 3  // (3 / 4 / 5 measure the same once the steps wait for vmcnt(RING - 1); 3 leaves registers)
#endif
// Issue priority of a wave (s_setprio): the per-workgroup phases in which the other waves of the
// workgroup wait at a barrier or have nothing to do -- setup, bootstrap, ordering a round's
// buckets, the final cut and sort -- run above the scanning waves of the other workgroups on the
// SIMD, which hide the gap; their LDS and wave slots come free sooner.
#ifndef VAQ_BF_PRIO
#define VAQ_BF_PRIO 2
#endif
#define BF_PRIO_SERIAL() __builtin_amdgcn_s_setprio(VAQ_BF_PRIO)
#define BF_PRIO_SCAN() __builtin_amdgcn_s_setprio(0)
// VAQ_PHASES (diagnostic builds): cycles each wave spends in the phases of the best-first body, in
// scalar registers (the VAQ_STATS counters cost tens of VGPRs and distort what they measure)
#ifdef VAQ_PHASES
#define PH_MARK(i) do { const unsigned long long ph_now = __builtin_readcyclecounter(); ph[i] += ph_now - ph_t; ph_t = ph_now; } while (0)
#else
#define PH_MARK(i)
#endif
constexpr int BF_SEG_STEPS = VAQ_BF_SEG;        // wave steps per work unit
constexpr int BF_FLUSH_AT = VAQ_BF_FLUSH;       // candidates a wave gathers before it takes the lock
constexpr int BF_CB_CAP = BF_FLUSH_AT - 1 + 64; // one drain appends at most 64
constexpr int BF_RING = VAQ_BF_RING;            // code items in flight per wave (register sets)
constexpr int BF_QCAP = 128;                    // survivor queue: at most 63 left over + 64 pushed
constexpr int BF_MAX_BUCKETS = 1024;
constexpr int VAQ_BF_MAX_SUBS = 128;  // VAQHIP_MAX_SUBSPACES
#ifndef VAQ_BF_ROUND
#define VAQ_BF_ROUND 128  // (256 measures the same at C2 / C3 and costs 1 KB of LDS: with 128 an eighth C2 workgroup fits a CU)
#endif
constexpr int BF_ROUND_BUCKETS = VAQ_BF_ROUND;  // buckets one round orders and scans
constexpr int BF_BOOT_STEPS = VAQ_BF_BOOT;      // wave steps each wave samples to bootstrap the threshold (0 = off)

__host__ __device__ inline size_t bf_align16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ inline int bf_pow2(int n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}
// slots of the k-min pool (below): ScanParams::bf_pool, chosen by the planner between these two so
// that the pool never costs a resident workgroup (C2, 4 waves: 23 408 B of LDS with 512 slots is
// 6 workgroups per CU and 1.02 ms, 22 896 B with 448 is 7 and 0.94 ms).  Round 3: a full pool first drops
// the rows above the current threshold (one pass, flush()), and its size hardly matters any more -- C2
// with 448 / 384 / 256 slots 0.420 / 0.426 / 0.423 ms at 7 workgroups per CU; 256 slots and 128 buckets
// per round are 20 336 B: EIGHT workgroups per CU, 0.410 ms.
#ifndef VAQ_BF_POOL
#define VAQ_BF_POOL 512
#endif
#ifndef VAQ_BF_POOL_MIN
#define VAQ_BF_POOL_MIN 256
#endif
__host__ __device__ inline int bf_pool_min(int kp) { return 2 * kp < VAQ_BF_POOL_MIN ? VAQ_BF_POOL_MIN : 2 * kp; }
__host__ __device__ inline int bf_pool_max(int kp) { return 2 * kp < VAQ_BF_POOL ? VAQ_BF_POOL : 2 * kp; }
constexpr int BF_HIST_BINS = 64;  // one per lane
constexpr int BF_RANK_SORT_MAX = 256;  // final lists up to this long are ordered by counting, longer ones by a bitonic sort
constexpr int BF_HDR_SCALE = 5;   // header word: float bits of bins / H, 0 = histogram off
// code dwords a survivor carries through the queue (the rest of its row, byte codes M <= 16)
__host__ __device__ inline int bf_queue_code_words(int M) { return M <= 16 ? M / 4 - 1 : 0; }
// LDS of one workgroup: [LUT][k-min][sorted bucket keys, row ranges, unit prefix, ticket, gmin]
// [per wave: survivor queue, candidate buffer]
__host__ __device__ inline size_t bf_lds_bytes(int lut_entries, int pool, int n_buckets, int nwaves, int qcw,
                                               int extra_words) {
  size_t b = bf_align16((size_t)lut_entries * 4) + bf_align16((size_t)extra_words * 4);
  b += bf_align16((size_t)SEL_HDR_WORDS * 4 + (size_t)pool * 8) + (size_t)BF_HIST_BINS * 4;
  b += bf_align16((size_t)BF_ROUND_BUCKETS * 8 + 4 + 32 + (size_t)(1 << GMIN_MAX_BITS) * 4);
  (void)n_buckets;
  b += (size_t)nwaves * ((size_t)BF_QCAP * 4 * (2 + qcw) + (size_t)BF_CB_CAP * 8);
  return b;
}

// ---------------------------------------------------------------------------
// k-min of the best-first form: an UNSORTED pool of admitted rows in LDS.
//   header  lock, count, threshold (distance bits, label)       (SelView, vaq_scan.h)
//   [0, cap) (distance, label) pairs, cap = ScanParams::bf_pool >= 2 kp
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

// One wave, lock held: keeps the rows at or below (distance bits t, label tl), in place and forwards
// (writes never pass the reads); returns their number.  Does not touch the header.
__device__ __forceinline__ int pool_keep_le(const SelView &sel, const int cnt, const unsigned t, const int tl, const int lane) {
  int w = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    const float di = i < cnt ? sel.d[i] : INFINITY;
    const int ii = i < cnt ? sel.id[i] : ID_SENTINEL;
    const unsigned bi = float_to_bits(di);
    const bool keep = i < cnt && (bi < t || (bi == t && ii <= tl));
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
  return w;
}

// One wave, lock held, pool holds n > k rows.  Keeps every row at or below the k-th smallest
// distance (>= k rows; more only when rows tie at that distance) and lowers the threshold to
// (that distance, INT_MAX).  If ties leave less than `room` free slots, the tie is cut exactly:
// of the rows at that distance only the smallest labels stay, k rows in all, and the threshold
// becomes the k-th (distance, label) pair itself.  Returns the new count.
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
  auto keep_le = [&](const int cnt, const int tl) { return pool_keep_le(sel, cnt, t, tl, lane); };
  int w = keep_le(n, INT_MAX);
  const float td = bits_to_float(t);
  int ti = INT_MAX;
  if (w > k && w + room > cap) {
    // Rows tying at the k-th distance fill the pool: cut the tie by label.  The rows below t
    // stay; of the rows at t the (k - below) smallest labels do -- the same bisection, on the
    // labels (distinct, >= 0), so the pool needs no sorting and no power-of-two capacity.
    int below = 0;
    for (int base = 0; base < w; base += 64) {
      const int i = base + lane;
      below += __popcll(__ballot(i < w && float_to_bits(sel.d[i]) < t));
    }
    const int need = k - below;  // >= 1: fewer than k rows lie below the k-th smallest distance
    unsigned llo = 0u, lhi = 0x7fffffffu;
    while (llo < lhi) {
      const unsigned mid = llo + ((lhi - llo) >> 1);
      int c = 0;
      for (int base = 0; base < w; base += 64) {
        const int i = base + lane;
        c += __popcll(__ballot(i < w && float_to_bits(sel.d[i]) == t && (unsigned)sel.id[i] <= mid));
      }
      if (c >= need) lhi = mid;
      else llo = mid + 1u;
    }
    ti = (int)llo;
    w = keep_le(w, ti);  // == k
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

// ---------------------------------------------------------------------------
// Layout policies: what the best-first scaffold needs to know about a code layout.
//   Item            the registers one lane holds per wave step (ROWS rows)
//   first_two       A : dism = l0; dism += l1            (l0 wave-uniform when UL0)
//   rest_of_first   A2: dism += l2; dism += l3            -> the first group's sum
//   carry / tail    the row's remaining groups, from the words carried through the survivor
//                   queue (QCW of them) or re-read from the code array; abandons per group
//   full            the complete row sum (bootstrap sample)
// Arithmetic order is the reference's throughout (VAQ.cpp:1737-1748).
// ---------------------------------------------------------------------------
// (experiment knob: bytes of code one lane loads per wave step for 8-byte rows; 16 = 2 rows.  32 = 4 rows
//  per lane and step with a ring of 3 measured slower at C2, 0.755 vs 0.725 ms, and level at C4)
#ifndef VAQ_BF_M8_ITEM_BYTES
#define VAQ_BF_M8_ITEM_BYTES 16
#endif
template <int M, int BYTES_> struct BfBytesItem {
  static constexpr int BYTES = BYTES_;
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
template <int M> struct BfBytes {
  typedef BfBytesItem<M, (M == 8 ? VAQ_BF_M8_ITEM_BYTES : (M < 16 ? 16 : M))> Item;
  static constexpr int ROWS = Item::ROWS;
  static constexpr int WPR = Item::WPR;
  static constexpr int QCW = (M <= 16) ? WPR - 1 : 0;
  static constexpr bool HAS_TAIL = WPR > 1;
  // No test after the first two terms: a byte costs one instruction to take out of its dword, and on the
  // cache-resident databases this form serves 41 % of the rows of a visited bucket pass that test (C2) --
  // some lane of nearly every wave step does, so the wave looks the other two terms up anyway and the
  // test only adds a compare and a branch (C2 0.535 -> 0.517 ms).  The sums are the same sums.
#ifndef VAQ_BF_BYTES_TEST_A
#define VAQ_BF_BYTES_TEST_A 0
#endif
  static constexpr bool TEST_A = VAQ_BF_BYTES_TEST_A != 0;
  static constexpr int LDS_WORDS = 0;  // no per-workgroup tables besides the LUT
  const float *lut;
  // (the tables are addressed by their LDS byte offsets, lds_lut: they must start at offset 0)
  __device__ __forceinline__ void init(const ScanParams &, const float *lut_lds, unsigned *, int, int) {
    lut = lut_lds;
    lds_base_is_zero(lut_lds);
  }
  __device__ static __forceinline__ int lut_entries(const ScanParams &) { return M * 256; }
  __device__ static __forceinline__ void second_table(const ScanParams &, int &off1, int &ncent1) {
    off1 = 256;
    ncent1 = 256;
  }
  __device__ static __forceinline__ void load(Item &it, const uint32_t *codes, const int base_row, const int lane) {
    it.load(codes, (int64_t)(base_row / ROWS) + lane);
  }
  __device__ __forceinline__ float lut4(const uint32_t c4, const int g) const {
    // (the four entries first: one LDS round trip, not two)
    const float l0 = lds_lut<0>((g * 4 + 0) * 1024, c4), l1 = lds_lut<1>((g * 4 + 1) * 1024, c4);
    const float l2 = lds_lut<2>((g * 4 + 2) * 1024, c4), l3 = lds_lut<3>((g * 4 + 3) * 1024, c4);
    float dism = l0 + l1;
    dism = dism + l2;
    dism = dism + l3;
    return dism;
  }
  template <bool UL0> __device__ __forceinline__ float first_two(const Item &it, const int r, const float l0) const {
    const uint32_t c0 = it.word(r, 0);
    const float first = UL0 ? l0 : lds_lut<0>(0, c0);
    return first + lds_lut<1>(1024, c0);
  }
  __device__ __forceinline__ float rest_of_first(const Item &it, const int r, float part) const {
    const uint32_t c0 = it.word(r, 0);
    part = part + lds_lut<2>(2048, c0);
    part = part + lds_lut<3>(3072, c0);
    return part;
  }
  __device__ static __forceinline__ uint32_t carry(const Item &it, const int r, const int w) { return it.word(r, w + 1); }
  // cw: the QCW carried words (QCW == 0: the row is re-read from the code array)
  __device__ __forceinline__ float tail(float acc, const uint32_t *cw, const int rid, const uint32_t *codes,
                                        const float thr, bool &alive) const {
#pragma unroll
    for (int g = 1; g < WPR; g++) {
      const uint32_t c4 = QCW > 0 ? cw[g - 1] : codes[(int64_t)rid * WPR + g];
      if (alive) {
        acc = acc + lut4(c4, g);  // dist += dism
        alive = !(acc > thr);
      }
    }
    return acc;
  }
  __device__ __forceinline__ float full(const Item &it, const int r) const {
    float acc = lut4(it.word(r, 0), 0);
#pragma unroll
    for (int g = 1; g < WPR; g++) acc = acc + lut4(it.word(r, g), g);
    return acc;
  }
};

// Bit-packed rows (LAYOUT_BITS: W dwords per row in planar tiles of 64 rows; any 1..15-bit
// allocation, M a multiple of 4).  CARRY: every field of the groups after the first lies in the
// row's last dword, which then rides through the survivor queue (C3: 39 of 64 bits are the
// first group); otherwise phase B re-reads the row from the tiles.
template <int W, bool CARRY> struct BfBits {
  struct Item {
    uint32_t w[W];
  };
  static constexpr int ROWS = 1;
  static constexpr int QCW = CARRY ? 1 : 0;
  static constexpr bool HAS_TAIL = true;
  static constexpr bool TEST_A = true;  // (taking a field out of a packed row costs several instructions: C3 0.92 ms with the test, 1.04 without)
  static constexpr int LDS_WORDS = VAQ_BF_MAX_SUBS;  // one packed descriptor per subspace
  const float *lut;
  const unsigned *pd;  // LDS: word | shift << 3 | bits << 8 | lut_off << 12 per subspace
  int M;
  unsigned m0;               // mask of the first code
  unsigned sh1, m1, o1;      // shift / mask / table offset of fields 1..3 (fields 2, 3 may straddle dwords 0-1)
  unsigned sh2, m2, o2, w2;
  unsigned sh3, m3, o3, w3;
  // (the descriptors become readable after the next workgroup barrier)
  __device__ __forceinline__ void init(const ScanParams &p, const float *lut_lds, unsigned *lds_words, const int tid,
                                       const int nthreads) {
    lut = lut_lds;
    const SubDesc *__restrict__ sub = p.sub;
    M = p.M;
    for (int s = tid; s < M; s += nthreads) {
      const SubDesc x = sub[s];
      lds_words[s] = (unsigned)x.word | ((unsigned)x.shift << 3) | ((unsigned)x.bits << 8) | ((unsigned)x.lut_off << 12);
    }
    pd = lds_words;
    const SubDesc a = sub[0], b = sub[1], c = sub[2], d = sub[3];
    m0 = (unsigned)a.ncent - 1u;
    sh1 = (unsigned)b.shift; m1 = (unsigned)b.ncent - 1u; o1 = (unsigned)b.lut_off;
    sh2 = (unsigned)c.shift; m2 = (unsigned)c.ncent - 1u; o2 = (unsigned)c.lut_off; w2 = (unsigned)c.word;
    sh3 = (unsigned)d.shift; m3 = (unsigned)d.ncent - 1u; o3 = (unsigned)d.lut_off; w3 = (unsigned)d.word;
  }
  __device__ static __forceinline__ int lut_entries(const ScanParams &p) { return p.lut_floats; }
  __device__ static __forceinline__ void second_table(const ScanParams &p, int &off1, int &ncent1) {
    off1 = p.sub[1].lut_off;
    ncent1 = p.sub[1].ncent;
  }
  __device__ static __forceinline__ void load(Item &it, const uint32_t *codes, const int base_row, const int lane) {
    const uint32_t *tp = codes + (int64_t)(base_row / TILE_ROWS) * (TILE_ROWS * W) + lane;
#pragma unroll
    for (int i = 0; i < W; i++) it.w[i] = tp[i * TILE_ROWS];
  }
  template <bool UL0> __device__ __forceinline__ float first_two(const Item &it, const int, const float l0) const {
    const uint32_t w0 = it.w[0];
    const float first = UL0 ? l0 : lut[w0 & m0];
    return first + lut[o1 + ((w0 >> sh1) & m1)];  // field 1 lies inside dword 0 (two fields <= 30 bits)
  }
  __device__ __forceinline__ float rest_of_first(const Item &it, const int, float part) const {
    const uint32_t w0 = it.w[0];
    const uint32_t w1 = W > 1 ? it.w[W > 1 ? 1 : 0] : 0u;
    const uint32_t c2 = (w2 == 0 ? __builtin_amdgcn_alignbit(w1, w0, sh2) : (w1 >> sh2)) & m2;
    const uint32_t c3 = (w3 == 0 ? __builtin_amdgcn_alignbit(w1, w0, sh3) : (w1 >> sh3)) & m3;
    part = part + lut[o2 + c2];
    part = part + lut[o3 + c3];
    return part;
  }
  __device__ static __forceinline__ uint32_t carry(const Item &it, const int, const int) { return it.w[W - 1]; }
  // subspaces [s_from, M) of a row given as words: dism = l; dism += l x3; dist += dism per group
  template <typename GetWord>
  __device__ __forceinline__ float chain(float acc, const int s_from, GetWord word, const float thr, bool &alive,
                                         const bool ea) const {
    float dism = 0.0f;
    for (int s = s_from; s < M; s++) {
      const unsigned d = pd[s];
      if (!ea || alive) {
        const int wd = (int)(d & 7u);
        const uint32_t lo = word(wd);
        const uint32_t hi = wd + 1 < W ? word(wd + 1) : 0u;
        const uint32_t c = __builtin_amdgcn_alignbit(hi, lo, (d >> 3) & 31u) & ((1u << ((d >> 8) & 15u)) - 1u);
        const float l = lut[(d >> 12) + c];
        dism = (s & 3) == 0 ? l : dism + l;
        if ((s & 3) == 3) {
          acc = s == 3 ? dism : acc + dism;
          if (ea) alive = !(acc > thr);
        }
      }
    }
    return acc;
  }
  __device__ __forceinline__ float tail(float acc, const uint32_t *cw, const int rid, const uint32_t *codes,
                                        const float thr, bool &alive) const {
    if (CARRY) {
      const uint32_t last = cw[0];
      return chain(acc, 4, [&](const int) -> uint32_t { return last; }, thr, alive, true);
    }
    const uint32_t *rp = codes + (int64_t)(rid / TILE_ROWS) * (TILE_ROWS * W) + (rid % TILE_ROWS);
    return chain(acc, 4, [&](const int w) -> uint32_t { return rp[w * TILE_ROWS]; }, thr, alive, true);
  }
  __device__ __forceinline__ float full(const Item &it, const int) const {
    bool alive = true;
    return chain(0.0f, 0, [&](const int w) -> uint32_t {
      uint32_t x = it.w[0];
#pragma unroll
      for (int i = 1; i < W; i++) x = (w == i) ? it.w[i] : x;
      return x;
    }, FLT_MAX, alive, false);
  }
};

template <typename Pol, bool UL0>
__device__ __forceinline__ void scan_bf_body(const ScanParams &p) {
  typedef typename Pol::Item Item;
  constexpr int ROWS = Pol::ROWS;
  constexpr int QCW = Pol::QCW;
  constexpr int WSTEP = 64 * ROWS;  // rows per wave step
  constexpr int SEG_ROWS = BF_SEG_STEPS * WSTEP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
#ifdef VAQ_BF_VECTOR_WAVE
  const int lane = tid & 63, wave = tid >> 6;
#else
  // (through readfirstlane: what is derived from the wave number -- its buffers' addresses, its share of the
  //  bootstrap, `wave == 0` tests -- is then scalar for the compiler)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int total = p.nq * p.n_slices;
  // One workgroup per query on a cache-resident database: nothing is gained by giving an XCD a
  // contiguous range of queries (C2 without the ranking: 0.74 ms; block b -> query b: 0.69 ms); with
  // several slices per query the workgroups of a slice share its rows through their XCD's L2.
  const int v = (p.n_slices == 1 && p.defer_mode == 0) ? (int)blockIdx.x : xcd_virtual_id(blockIdx.x, gridDim.x);
  if (!p.qorder && v >= total) return;
  // (second launch of a deferring scan, ScanParams::defer_mode: the query comes from defer_list)
  const bool defer2 = p.defer_mode != 0;
  int slice, qi, ent = 0;
  unsigned rec_done = 0u, rec_thr = 0u;
  bool rec_none = false;  // (DeferRec::pad != 0: no bucket of the query is finished yet, done_key means nothing)
  if (defer2) {
    const unsigned asked = *p.defer_count;
    const int cnt = asked < (unsigned)p.defer_cap ? (int)asked : p.defer_cap;
    ent = v / p.n_slices;
    slice = v - ent * p.n_slices;
    if (ent >= cnt) return;
    const DeferRec rec = p.defer_list[ent];
    qi = __builtin_amdgcn_readfirstlane(rec.q);
    rec_done = (unsigned)__builtin_amdgcn_readfirstlane((int)rec.done_key);
    rec_thr = (unsigned)__builtin_amdgcn_readfirstlane((int)rec.thr);
    rec_none = __builtin_amdgcn_readfirstlane(rec.pad) != 0;
  } else if (p.qorder) {
    // one workgroup per query, expensive queries first (launch_cost_order): block b, dispatched b-th
    // and dealt to XCD b % 8, serves the b-th query of the ranking
    if ((int)blockIdx.x >= p.nq) return;
    slice = 0;
    qi = __builtin_amdgcn_readfirstlane(p.qorder[blockIdx.x]);
  } else {
    slice = v / p.nq;
    qi = v - slice * p.nq;
  }
  const int out_q = defer2 ? ent : qi;  // where this workgroup's list goes among the partial lists
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
  const int lut_entries = Pol::lut_entries(p);
  size_t off = bf_align16((size_t)lut_entries * 4);
  unsigned *pol_words = reinterpret_cast<unsigned *>(smem + off);
  off += bf_align16((size_t)Pol::LDS_WORDS * 4);
  const int cap = p.bf_pool;
  const SelView sel = sel_view(smem + off, cap, 0);
  off += bf_align16((size_t)SEL_HDR_WORDS * 4 + (size_t)cap * 8);
  unsigned *hist = reinterpret_cast<unsigned *>(smem + off);  // [64] admitted rows per distance bin
  off += (size_t)BF_HIST_BINS * 4;
  // the buckets a round works through: at most BF_ROUND_BUCKETS of them, nearest first
  unsigned *keys = reinterpret_cast<unsigned *>(smem + off);  // [BF_ROUND_BUCKETS] sorted: bound bits | bucket
  int *cum = reinterpret_cast<int *>(keys + BF_ROUND_BUCKETS); // [BF_ROUND_BUCKETS + 1] prefix of work units
  unsigned *ticket = reinterpret_cast<unsigned *>(cum + BF_ROUND_BUCKETS + 1);
  unsigned *boot_cnt = ticket + 1;   // bootstrap: rows counted below boot_thr
  unsigned *boot_thr = ticket + 2;   //            float bits, max over the waves
  unsigned *sh_min = ticket + 3;     // smallest key (the nearest bucket)
  unsigned *sh_cnt = ticket + 4;     // buckets eligible for the round
  unsigned *sh_rot = ticket + 5;     // [3] counters of the overflow bisection
  unsigned *gmin = ticket + 8;       // [1 << bt] smallest second term per group
  off += bf_align16((size_t)BF_ROUND_BUCKETS * 8 + 4 + 32 + (size_t)(1 << GMIN_MAX_BITS) * 4);
  unsigned *kuns = reinterpret_cast<unsigned *>(smem + off);  // [BF_ROUND_BUCKETS] unsorted: borrows the waves' buffers
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
#ifdef VAQ_WGTIME
  const unsigned long long wg_t0 = __builtin_readcyclecounter();
  int wg_drains = 0, wg_flushes = 0, wg_compacts = 0, wg_elig = 0, wg_rounds = 0;
#define WG_COUNT(x) (x)++
#else
#define WG_COUNT(x)
#endif
#ifdef VAQ_PHASES
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ph_t = __builtin_readcyclecounter();
#endif
  BF_PRIO_SERIAL();
  const float *__restrict__ glut = p.lut + (size_t)qi * p.lut_floats;
  const int *__restrict__ bstart = p.bucket_start;
  typedef const int __attribute__((address_space(4))) *const_int_ptr;
  const const_int_ptr bstart_k = (const_int_ptr)(uintptr_t)p.bucket_start;
  const int nwaves = nthreads >> 6;
  const int bs_first = tid < K0 ? bstart[tid] : 0, be_first = tid < K0 ? bstart[tid + 1] : 0;
  if (((lut_entries | p.lut_floats) & 3) == 0) {  // (16-byte loads: a quarter of the instructions for the same bytes)
    const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(glut);
    float4 *l4 = reinterpret_cast<float4 *>(lut);
    for (int e = tid; e < (lut_entries >> 2); e += nthreads) l4[e] = g4[e];
  } else {
    for (int e = tid; e < lut_entries; e += nthreads) lut[e] = glut[e];
  }
  Pol pol;
  pol.init(p, lut, pol_words, tid, nthreads);  // (its LDS tables are complete after the barriers below)
  if (tid == 0) {
    unsigned td = float_to_bits(FLT_MAX);
    int ti = INT_MIN;
    if (multi_slice) {
      const unsigned g = __hip_atomic_load(&p.g_thr[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g < td) { td = g; ti = INT_MAX; }
    }
    if (defer2 && rec_thr < td) { td = rec_thr; ti = INT_MAX; }  // (rows AT the first launch's threshold stay admissible)
    sel.hdr[SEL_LOCK] = 0u;
    sel.hdr[SEL_NCAND] = 0u;
    sel.hdr[SEL_NBEST] = 0u;
    sel.hdr[SEL_THR_D] = td;
    sel.hdr[SEL_THR_ID] = (unsigned)ti;
    sel.hdr[BF_HDR_SCALE] = 0u;
    *boot_cnt = 0u;
    *boot_thr = 0u;
    *sh_min = 0xffffffffu;
    cum[0] = 0;
  }
  if (tid < BF_HIST_BINS) hist[tid] = 0u;
  if (bt > 0) {  // the bucket key continues into the second code: its groups' smallest terms
    for (int i = tid; i < (1 << bt); i += nthreads) gmin[i] = 0x7f800000u;
    __syncthreads();
    int off1, ncent1;
    Pol::second_table(p, off1, ncent1);
    const int w = 31 - __builtin_clz((unsigned)ncent1) - bt;  // log2 of the group size
    // (a wave's 64 consecutive entries span whole groups or lie within one: reduce across the
    //  lanes of a group first, one LDS atomic per group and wave instead of one per entry)
    const int seg = w < 6 ? 1 << w : 64;
    for (int e0 = tid - lane; e0 < ncent1; e0 += nthreads) {
      const int e = e0 + lane;
      unsigned v = e < ncent1 ? float_to_bits(lut[off1 + e]) : 0x7f800000u;
      for (int o = 1; o < seg; o <<= 1) {
        const unsigned x = (unsigned)__shfl_xor((int)v, o);
        v = x < v ? x : v;
      }
      if ((lane & (seg - 1)) == 0 && e < ncent1) atomicMin(&gmin[e >> w], v);
    }
  }
  __syncthreads();
#ifdef VAQ_STATS
  cx.st[ST_CYC_SETUP_LUT] = __builtin_readcyclecounter() - t_begin;
#endif
  PH_MARK(0);
  // One packed word per bucket: lower bound of its row sums (low bits cut: still a lower bound) |
  // bucket -- unique.  Empty buckets and NaN tables get the largest keys and are never eligible.
  const unsigned empty_key = ~idx_mask;
  auto bucket_key_of = [&](const int b, const int bs_raw, const int be_raw) -> unsigned {
    const int s0 = bs_raw > r0 ? bs_raw : r0;
    const int e0 = be_raw < r1 ? be_raw : r1;
    if (e0 <= s0) return empty_key | (unsigned)b;
    float m;
    if (bt > 0) {
      m = lut[b >> bt] + bits_to_float(gmin[b & ((1 << bt) - 1)]);
    } else {
      m = INFINITY;
      for (int c = b << bshift; c < ((b + 1) << bshift); c++) {
        const float x = lut[c];
        m = x < m ? x : m;
      }
    }
    if (!(m == m)) return empty_key | (unsigned)b;
    return (float_to_bits(m) & ~idx_mask) | (unsigned)b;  // m >= 0: bit order == value order
  };
  auto bucket_key = [&](const int b) -> unsigned { return bucket_key_of(b, bstart[b], bstart[b + 1]); };
  auto units_of = [&](const unsigned key) -> int {
    const int b = (int)(key & idx_mask);
    const int s0 = bstart[b] > r0 ? bstart[b] : r0;
    const int e0 = bstart[b + 1] < r1 ? bstart[b + 1] : r1;
    return (e0 - (s0 & ~(WSTEP - 1)) + SEG_ROWS - 1) / SEG_ROWS;
  };
  // every bucket's key, kept in LDS (the waves' buffers, not yet in use) until the first round has
  // picked its buckets; a thread only ever reads back the entries it wrote
  unsigned *kall = kuns + BF_ROUND_BUCKETS;  // [K0]
  unsigned km0 = 0xffffffffu;
  {
    // (the row range of the thread's next bucket is fetched while this one's key is formed, and
    //  the first one before the tables arrive: one memory round trip is exposed, not one per key)
    int bs_n = bs_first, be_n = be_first;
#pragma unroll 1
    for (int b = tid; b < K0; b += nthreads) {
      const int bs_c = bs_n, be_c = be_n;
      const int bn = b + nthreads;
      if (bn < K0) {
        bs_n = bstart[bn];
        be_n = bstart[bn + 1];
      }
      const unsigned key = bucket_key_of(b, bs_c, be_c);
      kall[b] = key;
      km0 = key < km0 ? key : km0;
    }
  }
  {  // the nearest bucket (smallest key)
    unsigned km = km0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned x = (unsigned)__shfl_xor((int)km, o);
      km = x < km ? x : km;
    }
    if (lane == 0) atomicMin(sh_min, km);
  }
  __syncthreads();
#ifdef VAQ_STATS
  cx.st[ST_CYC_SETUP] = __builtin_readcyclecounter() - t_begin;
#endif
  PH_MARK(1);
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
    WG_COUNT(wg_flushes);
    STAT_T0(t_fl);
    STAT_ADD(ST_ADMITS, 1);
    const int rid = (cand && perm) ? (int)perm[srow] : srow;  // labels are ORIGINAL rows: ties break by them
    // Wait for that load HERE.  Left to the compiler, its wait lands after the lock's spin loop, and
    // the wait-count bookkeeping of every block this rarely taken path rejoins degrades to "all
    // loads" (seen in the ISA: s_waitcnt vmcnt(0) in front of each step of the scan loop below,
    // i.e. no prefetch; with this line the steps wait for vmcnt(BF_RING - 1)).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt, lgkmcnt untouched
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
        // The pool is full.  The threshold has moved down since most of its rows came in (the histogram
        // lowers it batch by batch): the rows above it are out of the k best whatever follows, and dropping
        // them is ONE pass.  Only if that frees too little is the k-th best found exactly, by bisection
        // (pool_compact: ~8 k cycles with the lock held -- a 384-slot pool cost 3 % of the C2 kernel, a
        // 256-slot one 9 %, before this pass).
        cnt = pool_keep_le(sel, cnt, float_to_bits(td), ti, lane);
        if (lane == 0) sel.hdr[SEL_NCAND] = (unsigned)cnt;
        wave_lds_sync();
      }
      if (cnt + __popcll(m) > cap) {
        STAT_T0(t_fo);
        STAT_ADD(ST_FOLDS, 1);
        WG_COUNT(wg_compacts);
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

  // phase B: the top n (<= 64) queue entries, one per lane: groups 1.. of the row, abandoning
  // after each (VAQ.cpp:1708)
  auto drain = [&](const int n) {
    WG_COUNT(wg_drains);
    STAT_T0(t_dr);
    STAT_ADD(ST_DRAINS, 1);
    qcnt -= n;
    const bool ok = lane < n;
    const int slot = qcnt + (ok ? lane : 0);
    const int rid = q_id[slot];
    float acc = q_p[slot];
    bool alive = ok;
    uint32_t cw[QCW > 0 ? QCW : 1];
#pragma unroll
    for (int w = 0; w < QCW; w++) cw[w] = q_cw[w * BF_QCAP + slot];
    acc = pol.tail(acc, cw, rid, codes, thr_d, alive);
    STAT_T1(ST_CYC_DRAIN, t_dr);
    gather(acc, rid, alive);
  };

  auto take_ticket = [&]() -> int {
    int t = 0;
    if (lane == 0) t = (int)atomicAdd(ticket, 1u);
    return __builtin_amdgcn_readfirstlane(t);
  };

  // ---- bootstrap: a first threshold from the rows nearest the query ----
  // The waves sum BF_BOOT_STEPS steps each of the nearest bucket's rows completely and every lane
  // keeps the smallest distance it saw: 64 distinct rows per wave.  With q = ceil(k / sampling
  // waves), the q-th smallest of a wave's 64 vouches for q rows at or below it, so the largest of
  // the waves' values has at least k rows at or below it: an upper bound of the final k-th
  // distance, i.e. a valid admission threshold (with label INT_MAX: rows AT that distance stay
  // admissible).  It also bounds which buckets can matter at all (below).
  const unsigned kmin = *sh_min;
  STAT_T0(t_boot);
  if (defer2) {
    // the first launch's threshold is the bootstrap: histogram over [0, H] as below
    if (tid == 0) {
      const float H = bits_to_float(sel.hdr[SEL_THR_D]);
      if (H > 0.0f && H < FLT_MAX) sel.hdr[BF_HDR_SCALE] = float_to_bits((float)BF_HIST_BINS / H);
    }
    __syncthreads();
  } else if (BF_BOOT_STEPS > 0 && !p.no_skip && (kmin & empty_key) != empty_key) {
    const int bb = (int)(kmin & idx_mask);
    const int bs = bstart[bb] > r0 ? bstart[bb] : r0;
    const int bend = bstart[bb + 1] < r1 ? bstart[bb + 1] : r1;
    const int base00 = bs & ~(WSTEP - 1);
    const int steps = (bend - base00 + WSTEP - 1) / WSTEP;
    int pw = (steps + BF_BOOT_STEPS - 1) / BF_BOOT_STEPS;  // waves that get rows to sample
    pw = pw < nwaves ? pw : nwaves;
    const int q = (k + pw - 1) / pw;
    if (q <= 64 && wave < pw) {
      const int pos = bs, be = bend;
      const int base0 = base00 + wave * BF_BOOT_STEPS * WSTEP;
      int nst = (be - base0 + WSTEP - 1) / WSTEP;
      if (nst > BF_BOOT_STEPS) nst = BF_BOOT_STEPS;
      float best = INFINITY;
      for (int st = 0; st < nst; st++) {
        Item cur;
        Pol::load(cur, codes, base0 + st * WSTEP, lane);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          const int row = base0 + st * WSTEP + lane * ROWS + r;
          const float acc = pol.full(cur, r);
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
  STAT_T1(ST_CYC_BOOT, t_boot);
  PH_MARK(2);
  // ---- rounds: the eligible buckets (bound not above the threshold, not done yet), nearest first,
  //      at most BF_ROUND_BUCKETS per round; thresholds fall while a round runs, so a second round
  //      is rarely anything but the check that nothing is left ----
  bool first_round = !defer2 || rec_none;
  bool defer_tried = false;
  bool bm_handed = false;  // (bucket-major second pass follows: publish the exact k-th distance)
  unsigned done_key = rec_done;  // buckets with keys <= done_key are finished (after the first round)
  for (;;) {
    BF_PRIO_SERIAL();
    PH_MARK(5);
    STAT_T0(t_prep);
    if (tid == 0) *sh_cnt = 0u;
    __syncthreads();
    // the round's threshold must be the SAME for every thread (the bisection below is a workgroup-
    // wide loop): read it from LDS between two barriers, where nothing scans and nothing moves it
    thr_d = bits_to_float(sel.hdr[SEL_THR_D]);
    const unsigned thr_bits = p.no_skip ? 0x7f800000u : float_to_bits(thr_d);
    auto eligible = [&](const unsigned key) -> bool {
      return (key & empty_key) != empty_key && (first_round || key > done_key) && (key & ~idx_mask) <= thr_bits;
    };
    unsigned hi_key = 0xffffffffu;  // (only keys <= hi_key join this round)
    if (!first_round) {
#pragma unroll 1
      for (int b = tid; b < K0; b += nthreads) kall[b] = bucket_key(b);
    }
#pragma unroll 1
    for (int b = tid; b < K0; b += nthreads) {
      const unsigned key = kall[b];
      if (eligible(key)) {
        const unsigned pos = atomicAdd(sh_cnt, 1u);
        if (pos < (unsigned)BF_ROUND_BUCKETS) kuns[pos] = key;
      }
    }
    __syncthreads();
    int n = (int)*sh_cnt;
    if (n == 0) break;
    if (p.defer_units > 0 && !first_round && !defer_tried) {
      // A first round that was cut short (below) and buckets still in reach after it: an expensive
      // query.  Hand the rest to the second launch, where several workgroups share it, instead of
      // keeping the launch waiting for this one (a full list: scan on here).
      defer_tried = true;
      if (tid == 0) {
        bool ok;
        if (p.bm_done) {
          // bucket-major second pass (vaq_scan_bm.hip): every query hands over, no list
          p.bm_done[qi] = done_key;
          atomicMin(&p.g_thr[qi], sel.hdr[SEL_THR_D]);  // (lowered to the exact k-th distance below)
          ok = true;
        } else {
          const unsigned idx = atomicAdd(p.defer_count, 1u);
          ok = idx < (unsigned)p.defer_cap;
          if (ok) {
            DeferRec rec;
            rec.q = qi;
            rec.done_key = done_key;
            rec.thr = sel.hdr[SEL_THR_D];
            rec.pad = 0;
            p.defer_list[idx] = rec;
            p.g_thr[qi] = rec.thr;  // the word the second launch's workgroups of this query share
          }
        }
        sh_rot[0] = ok ? 1u : 0u;
      }
      __syncthreads();
      const bool handed = sh_rot[0] != 0u;
      __syncthreads();
      if (handed) {
        bm_handed = p.bm_done != nullptr;
        break;
      }
    }
    STAT_ADD(ST_FOLDS, n);
#ifdef VAQ_WGTIME
    wg_elig += n;
    wg_rounds++;
#endif
    if (n > BF_ROUND_BUCKETS) {
      // more eligible buckets than a round holds (weak or no bootstrap threshold): bisect for the
      // largest key bound that lets at most BF_ROUND_BUCKETS of them in
      // (any bound that lets between half a round and a whole round in will do; three counters
      //  in rotation: one barrier per step)
      unsigned lo = 0u, hi = thr_bits | idx_mask;  // count(key <= lo) <= cap is maintained
      __syncthreads();
      if (tid == 0) { sh_rot[0] = 0u; sh_rot[1] = 0u; sh_rot[2] = 0u; }
      __syncthreads();
      int it = 0;
      while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1) + 1u;  // upper middle: lo < mid <= hi
        int c = 0;
#pragma unroll 1
        for (int b = tid; b < K0; b += nthreads) {
          const unsigned key = kall[b];
          c += (eligible(key) && key <= mid) ? 1 : 0;
        }
        unsigned *cur = sh_rot + (it % 3);
        if (c) atomicAdd(cur, (unsigned)c);
        if (tid == 0) sh_rot[(it + 1) % 3] = 0u;  // (last read two steps ago)
        __syncthreads();
        const int tot = (int)*cur;
        it++;
        if (tot <= BF_ROUND_BUCKETS) {
          lo = mid;
          if (tot >= BF_ROUND_BUCKETS / 2) break;
        } else {
          hi = mid - 1u;
        }
      }
      hi_key = lo;
      __syncthreads();
      if (tid == 0) *sh_cnt = 0u;
      __syncthreads();
#pragma unroll 1
      for (int b = tid; b < K0; b += nthreads) {
        const unsigned key = kall[b];
        if (eligible(key) && key <= hi_key) kuns[atomicAdd(sh_cnt, 1u)] = key;
      }
      __syncthreads();
      n = (int)*sh_cnt;  // >= 1: keys are unique, so the smallest eligible key alone always fits
    }
    // rank sort (n <= BF_ROUND_BUCKETS): position = number of smaller keys
    for (int i = tid; i < n; i += nthreads) {
      const unsigned key = kuns[i];
      const int units = units_of(key);  // (two reads of bucket_start: in flight while the rank is counted)
      int rank = 0;
      for (int j = 0; j < n; j++) rank += kuns[j] < key ? 1 : 0;
      keys[rank] = key;
      cum[rank + 1] = units;
    }
    if (tid == 0) *ticket = 0u;
    __syncthreads();
    if (wave == 0) {  // inclusive prefix of the unit counts, 64 at a time
      int carry = 0;
      for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const int u = i < n ? cum[i + 1] : 0;
        int inc = u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int x = __shfl_up(inc, o);
          if (lane >= o) inc += x;
        }
        if (i < n) cum[i + 1] = carry + inc;
        carry += __builtin_amdgcn_readlane(inc, 63);
      }
    }
    __syncthreads();
    STAT_T1(ST_CYC_PREP, t_prep);
    PH_MARK(3);
    BF_PRIO_SCAN();
    if (p.defer_units > 0 && first_round && cum[n] > p.defer_units) {
      // keep the nearest buckets whose units fit (at least one); the others wait for the next round,
      // or for the second launch
      int lo = 1, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cum[mid] <= p.defer_units) lo = mid;
        else hi = mid - 1;
      }
      n = lo;
      hi_key = keys[n - 1];
    }
    const int total_units = cum[n];
    // window of the unit prefix in registers: lane j holds cum[wbase + j + 1]
    int wbase = 0;
    int creg = (lane + 1 <= n) ? cum[lane + 1] : INT_MAX;
    // A unit's description -- sorted position, key, the bucket's row range -- is fetched one unit
    // ahead: the ticket's LDS round trip and the two reads of bucket_start (global memory) overlap
    // the scan of the unit before.
    auto locate = [&](const int t, int &i, unsigned &key, int &bs_raw, int &be_raw) {
      int cnt;
      for (;;) {  // i = largest index with cum[i] <= t (tickets only grow: the window moves forward)
        cnt = __popcll(__ballot(creg <= t));
        if (cnt < 64) break;
        wbase += 64;
        creg = (wbase + lane + 1 <= n) ? cum[wbase + lane + 1] : INT_MAX;
      }
      i = wbase + cnt;
      key = (unsigned)__builtin_amdgcn_readfirstlane((int)keys[i]);
      const int b = (int)(key & idx_mask);
      // (read through the constant address space: wave-uniform index, table not written by the
      //  kernel -> scalar loads, which count on lgkmcnt; as vector loads they are younger than the
      //  code items already in flight for this unit, and waiting for them would drain those)
      bs_raw = bstart_k[b];
      be_raw = bstart_k[b + 1];
    };
    // rows [pos, be) of unit tt (the (tt - cum[ii])-th of its bucket), the aligned row of its first
    // item and its wave steps
    auto geometry = [&](const int tt, const int ii, const int bsr, const int ber, int &pos, int &be, int &base0,
                        int &nst) {
      const int bs = __builtin_amdgcn_readfirstlane(bsr > r0 ? bsr : r0);
      const int bend = __builtin_amdgcn_readfirstlane(ber < r1 ? ber : r1);
      const int j = tt - __builtin_amdgcn_readfirstlane(cum[ii]);
      const int al = bs & ~(WSTEP - 1);
      pos = al + j * SEG_ROWS;
      if (pos < bs) pos = bs;
      be = al + (j + 1) * SEG_ROWS;
      if (be > bend) be = bend;
      base0 = pos & ~(WSTEP - 1);
      nst = (be - base0 + WSTEP - 1) / WSTEP;
    };
    // BF_RING code items in flight per wave, ACROSS units: while the last steps of a unit are
    // worked through, the slots they free take the first items of the next one, so only a round's
    // first unit waits for memory.  Every path issues the SAME number of loads (one past a unit's
    // end re-reads its last item -- a cache hit; the last unit of a round re-reads its own): a
    // load that is issued on some paths only leaves the compiler no count it can wait for but
    // zero, and every step then waits for the load issued just before it -- no prefetch at all
    // (seen in the ISA: s_waitcnt vmcnt(0) at each step; now it is vmcnt(BF_RING - 1)).
    Item ring[BF_RING];
    int t = take_ticket();
    int i = 0;
    unsigned key = 0u;
    int pos = 0, be = 0, base0 = 0, nst = 1;
    if (t < total_units) {
      int bs_raw, be_raw;
      locate(t, i, key, bs_raw, be_raw);
      geometry(t, i, bs_raw, be_raw, pos, be, base0, nst);
#pragma unroll
      for (int u = 0; u < BF_RING; u++) Pol::load(ring[u], codes, base0 + (u < nst - 1 ? u : nst - 1) * WSTEP, lane);
    }
    while (t < total_units) {
      const int t_next = take_ticket();
      STAT_ADD(ST_BUCKETS_TESTED, 1);
      const float lbv = bits_to_float(key & ~idx_mask);
      // steps [st_lo, st_lo + n_int) of the unit have all their rows inside [pos, be)
      const int st_lo = pos > base0 ? 1 : 0;
      const int st_hi = be < base0 + nst * WSTEP ? nst - 1 : nst;
      const int n_int = st_hi > st_lo ? st_hi - st_lo : 0;
      // buckets come in ascending order of their bound and thresholds only fall: nothing
      // from here on can hold an admissible row
      if (!p.no_skip && lbv > thr_d) break;
      STAT_ADD(ST_BUCKETS_VISITED, 1);
      const int b = (int)(key & idx_mask);
      // the rows' shared first term (fine buckets)
      const float l0 = UL0 ? bits_to_float((unsigned)__builtin_amdgcn_readfirstlane((int)float_to_bits(lut[b >> bt]))) : 0.0f;
      const bool has_next = t_next < total_units;
      int i_n = 0, bs_n = 0, be_n = 0;
      unsigned key_n = 0u;
      if (has_next) locate(t_next, i_n, key_n, bs_n, be_n);

      auto step = [&](const Item &cur, const int st) {
        const int base = base0 + st * WSTEP;
        STAT_ADD(ST_STEPS, 1);
        refresh(stepno++);
        // wave-uniform: no per-row range test on the steps that lie inside the unit -- all but, at most, its
        // first and its last one (st_lo, n_int: per unit, below; one subtraction and one compare per step)
        const bool interior = (unsigned)(st - st_lo) < (unsigned)n_int;
        const int row0 = base + lane * ROWS;
        float part[ROWS];
        // a row that survived its first group: to the survivor queue (or, a one-group row, to the candidates)
        auto keep = [&](const int r, const bool live) {
          STAT_ADD(ST_ALIVE_A2, __popcll(__ballot(live)));
          if (!Pol::HAS_TAIL) {  // (a row of one group ends here)
            gather(part[r], row0 + r, live);
            return;
          }
          const unsigned long long m = __ballot(live);
          if (m == 0ull) return;
          const int qp = qcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (live) {
            q_id[qp] = row0 + r;
            q_p[qp] = part[r];
  #pragma unroll
            for (int w = 0; w < QCW; w++) q_cw[w * BF_QCAP + qp] = Pol::carry(cur, r, w);
          }
          qcnt += __popcll(m);
          if (qcnt >= 64) drain(64);
        };
        if (!Pol::TEST_A) {
          // straight-line: the first group's four terms, one test.  On the steps at a unit's edges the rows
          // outside it get +inf in place of their sum (never at or below a threshold: thresholds are
          // finite) -- the wave-uniform branch keeps the range tests off the interior steps, and the
          // test's mask is the compare's own result
  #pragma unroll
          for (int r = 0; r < ROWS; r++) part[r] = pol.rest_of_first(cur, r, pol.template first_two<UL0>(cur, r, l0));
          if (!interior) {
  #pragma unroll
            for (int r = 0; r < ROWS; r++)
              if (row0 + r < pos || row0 + r >= be) part[r] = INFINITY;
          }
  #pragma unroll
          for (int r = 0; r < ROWS; r++) {
            STAT_ADD(ST_ALIVE_A, __popcll(__ballot(part[r] < INFINITY)));
            keep(r, !(part[r] > thr_d));
          }
        } else {
          bool alive[ROWS];
  #pragma unroll
          for (int r = 0; r < ROWS; r++) {
            part[r] = pol.template first_two<UL0>(cur, r, l0);  // A: dism = l0; dism += l1
            // (interior is wave-uniform: the row-range test folds into scalar mask logic)
            const bool in_range = interior || ((row0 + r >= pos) && (row0 + r < be));
            alive[r] = in_range && !(part[r] > thr_d);
            STAT_ADD(ST_ALIVE_A, __popcll(__ballot(alive[r])));
          }
  #pragma unroll
          for (int r = 0; r < ROWS; r++) {
            bool live = alive[r];
            if (live) {  // A2: dism += l2; dism += l3 -> the first group's sum
              part[r] = pol.rest_of_first(cur, r, part[r]);
              live = !(part[r] > thr_d);
            }
            keep(r, live);
          }
        }
      };

      const int last = nst - 1;  // (nst >= 1: a unit has rows)
      int st = 0;
      for (; st + BF_RING < nst; st += BF_RING) {
  #pragma unroll
        for (int u = 0; u < BF_RING; u++) {
          step(ring[u], st + u);
          const int nx = st + u + BF_RING;
          Pol::load(ring[u], codes, base0 + (nx < last ? nx : last) * WSTEP, lane);
        }
      }
      // the unit's last steps; each slot then takes its item of the next unit
      int pos_n = pos, be_n2 = be, base0_n = base0 + last * WSTEP, nst_n = 1;  // (no next unit: this one's last item again)
      if (has_next) geometry(t_next, i_n, bs_n, be_n, pos_n, be_n2, base0_n, nst_n);
  #pragma unroll
      for (int u = 0; u < BF_RING; u++) {
        if (st + u < nst) step(ring[u], st + u);
        Pol::load(ring[u], codes, base0_n + (u < nst_n - 1 ? u : nst_n - 1) * WSTEP, lane);
      }
      t = t_next;
      i = i_n;
      key = key_n;
      pos = pos_n;
      be = be_n2;
      base0 = base0_n;
      nst = nst_n;
    }
    PH_MARK(4);
    // the round's buckets are done (or out of reach); empty the wave's buffers, which the next
    // round's key list borrows
    if (hi_key != 0xffffffffu) {
      while (qcnt > 0) drain(qcnt < 64 ? qcnt : 64);
      while (ccnt > 0) flush();
    }
    __syncthreads();
    // Every eligible bucket was in this round and thresholds only fall: nothing is left.
    if (hi_key == 0xffffffffu) break;
    done_key = keys[n - 1];
    first_round = false;
    refresh(0);
  }
  BF_PRIO_SERIAL();
  PH_MARK(5);
  STAT_T0(t_tail);
  while (qcnt > 0) drain(qcnt < 64 ? qcnt : 64);
  while (ccnt > 0) flush();

  // ---- results: wave 0 cuts the pool to the k best and sorts them ----
  __syncthreads();
  PH_MARK(6);
#ifdef VAQ_STATS
  cx.st[ST_CYC_STEPLOAD] = __builtin_readcyclecounter() - t_tail;  // (tail: last drains / flushes + waiting for the other waves)
  cx.st[ST_CYC_TOTAL] = __builtin_readcyclecounter() - t_begin;
  if (p.stats && lane == 0)
    for (int i = 0; i < ST_N; i++) atomicAdd(&p.stats[i], cx.st[i]);
#endif
  // wave 0 cuts a long pool to exactly the k best (ties cut by label); the waves then share the ordering
  if (wave == 0) {
    int n0 = (int)sel.hdr[SEL_NCAND];
    PH_MARK(8);
    // The histogram of the admitted distances already brackets the k-th best: with jb the first bin at which
    // the running count reaches k, every one of the k best lies in a bin <= jb (k admitted rows -- rows of the
    // database, each counted once -- lie in bins <= jb, so a row of a later bin has k rows strictly below it;
    // rows the pool dropped on the way were above k kept ones and change nothing).  Keeping those bins' rows
    // -- one pass, no bisection -- leaves the k best and the few rows sharing the last bin with them, which
    // the ranking below orders; it was the bisection of pool_compact, run by this wave alone while the
    // others wait, that made the final cut 8 % of a C2 workgroup's life.
    const float hscale = bits_to_float(sel.hdr[BF_HDR_SCALE]);
    if (n0 > k && hscale != 0.0f) {
      int inc = (int)hist[lane];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(inc, o);
        if (lane >= o) inc += x;
      }
      const unsigned long long reach = __ballot(inc >= k);
      const int jb = reach != 0ull ? __builtin_ctzll(reach) : BF_HIST_BINS - 1;
      if (jb < BF_HIST_BINS - 1) {  // (the last bin also holds everything beyond H: nothing to drop)
        int w = 0;
        for (int base = 0; base < n0; base += 64) {
          const int i = base + lane;
          const float di = i < n0 ? sel.d[i] : INFINITY;
          const int ii = i < n0 ? sel.id[i] : ID_SENTINEL;
          const unsigned b = (unsigned)(di * hscale);  // the bin it was counted in (flush)
          const bool keep = i < n0 && (b < BF_HIST_BINS - 1 ? b : BF_HIST_BINS - 1) <= (unsigned)jb;
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
        n0 = w;
        if (lane == 0) sel.hdr[SEL_NCAND] = (unsigned)n0;
        wave_lds_sync();
      }
    }
    // (up to BF_RANK_SORT_MAX rows are ranked as they are, the k first written: no cut needed)
    if (n0 > k && n0 > BF_RANK_SORT_MAX) pool_compact(sel, n0, k, cap, cap, lane);  // (leaves the new count in the header)
    PH_MARK(9);
  }
  __syncthreads();
  {
    const int n = (int)sel.hdr[SEL_NCAND];
    // result slot i: the API's format when this list IS the result (one slice per query;
    // heap_reorder's: ascending, empty slots -1 / FLT_MAX, utils/Heap.hpp:322-349), else a
    // partial list for the merge
    const size_t o = p.final_labels ? (size_t)qi * k : ((size_t)out_q * p.n_slices + slice) * k;
    auto emit = [&](const int i, const float d, const int id) {
      if (p.final_labels) {
        const bool ok = id != ID_SENTINEL;
        p.final_labels[o + i] = ok ? (int32_t)(id + p.id_base) : -1;
        p.final_dist[o + i] = ok ? d : FLT_MAX;
      } else {
        p.part_d[o + i] = d;
        p.part_id[o + i] = id;
      }
    };
    if (n <= BF_RANK_SORT_MAX) {
      // Few rows: each lane counts the rows below its own -- (distance bits, label) compared as
      // one 64-bit key, distinct because labels are -- and writes its row at that position.
      // The other rows are broadcast reads of LDS, independent of each other: no chain of
      // dependent compare-exchange stages as in a sorting network (28 for 128 rows).  Wave w takes
      // rows [64 w, 64 w + 64), ...
      // (the keys packed once, in the waves' buffers -- idle by now: the counting loop is then one 8-byte
      //  broadcast read, one 64-bit compare and one add per row)
      unsigned long long *pk = reinterpret_cast<unsigned long long *>(kuns);
      for (int i = tid; i < n; i += nthreads) pk[i] = ((unsigned long long)float_to_bits(sel.d[i]) << 32) | (unsigned)sel.id[i];
      __syncthreads();
      for (int base = wave * 64; base < n; base += nwaves * 64) {
        const int i = base + lane;
        const bool valid = i < n;
        const unsigned long long ki = valid ? pk[i] : ~0ull;
        const float di = bits_to_float((unsigned)(ki >> 32));
        const int ii = valid ? (int)(unsigned)ki : ID_SENTINEL;
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < n; j++) rank += pk[j] < ki ? 1 : 0;
        if (valid && rank < k) emit(rank, di, ii);
        if (bm_handed && valid && rank == k - 1) atomicMin(&p.g_thr[qi], float_to_bits(di));
      }
      if (wave == nwaves - 1)
        for (int i = n + lane; i < k; i += 64) emit(i, INFINITY, ID_SENTINEL);
    } else if (wave == 0) {
      int P = 2;
      while (P < n) P <<= 1;
      const int pad_to = P > kp ? P : kp;
      for (int i = n + lane; i < pad_to; i += 64) {
        sel.d[i] = INFINITY;
        sel.id[i] = ID_SENTINEL;
      }
      wave_lds_sync();
      bitonic_sort<false>(sel.d, sel.id, P, lane, 64);  // ascending by (distance, label)
      for (int i = lane; i < k; i += 64) emit(i, sel.d[i], sel.id[i]);
      if (bm_handed && lane == 0 && n >= k) atomicMin(&p.g_thr[qi], float_to_bits(sel.d[k - 1]));
    }
    if (!p.final_labels && tid == 0) p.part_cnt[(size_t)out_q * p.n_slices + slice] = n < k ? n : k;
  }
#ifdef VAQ_WGTIME  // (experiment builds: the workgroup's lifetime and event counts in place of the last distances, tools/exp_lpt_oracle.py)
  __syncthreads();
  if (tid < 8) hist[tid] = 0u;
  __syncthreads();
  if (lane == 0) {
    atomicAdd(&hist[0], (unsigned)stepno);
    atomicAdd(&hist[1], (unsigned)wg_drains);
    atomicAdd(&hist[2], (unsigned)wg_flushes);
    atomicAdd(&hist[3], (unsigned)wg_compacts);
  }
  __syncthreads();
  if (wave == 0 && lane == 0 && p.final_dist) {
    float *o = p.final_dist + (size_t)qi * k;
    o[k - 1] = (float)(__builtin_readcyclecounter() - wg_t0);
    o[k - 2] = (float)hist[0];
    o[k - 3] = (float)hist[1];
    o[k - 4] = (float)hist[2];
    o[k - 5] = (float)hist[3];
    o[k - 6] = (float)wg_elig;
    o[k - 7] = (float)wg_rounds;
  }
#endif
#ifdef VAQ_PHASES
  PH_MARK(7);
  if (p.stats && lane == 0 && (blockIdx.x & 63) == 0) {  // (a sample: same-address atomics from every wave would be the slowest thing in the kernel)
    for (int i = 0; i < 10; i++) atomicAdd(&p.stats[i], ph[i]);
    atomicAdd(&p.stats[10], 1ull);
  }
#endif
}

// bytes of LDS of the byte-code best-first kernel and its launch (vaq_scan_bf.hip)
hipError_t launch_scan_bf(const ScanParams &p, int grid, hipStream_t st);

} // namespace vaq
#endif
