// vaqhip_api.cpp -- host side of the C ABI declared in include/vaqhip.h.
// Owns device memory, picks launch geometry, enqueues the gfx950 kernels of
// vaq_kernels.hip.  There is no CPU path here: every entry point needs a HIP
// device and fails with VAQHIP_ENODEVICE / VAQHIP_EHIP otherwise.
#include "vaqhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "vaq_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? VAQHIP_ENOMEM : VAQHIP_EHIP, "%s: %s", #expr, \
                  hipGetErrorString(e_));                                                 \
  } while (0)

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  // grow-only
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    else p = nullptr;
    return e;
  }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

constexpr size_t LDS_LIMIT = 160 * 1024;       // per CU on gfx950
#ifndef VAQ_BF_WAVES_PER_SIMD
#define VAQ_BF_WAVES_PER_SIMD 8
#endif
constexpr int BF_WAVES_PER_CU = 4 * VAQ_BF_WAVES_PER_SIMD;  // what the best-first kernels' register budget admits
constexpr size_t LDS_GRANULE = 1280;           // allocation unit assumed when counting resident workgroups
constexpr int QUERY_CHUNK = 16384;             // queries per internal launch set
constexpr int64_t MIN_SLICE_ROWS = 16384;      // do not cut slices finer than this
// Best-first form, one workgroup per query, option "defer_units" (OFF by default): a first round
// takes at most that many work units and what is still in reach after it is scanned by DEFER_SLICES
// workgroups per query in a second launch (at most DEFER_CAP queries; the others scan on in place).
// A query's cost spans 6x (C2: 318 wave steps on average, 1857 for the top 1 %) and a launch ends
// with its most expensive workgroups -- but the second launch has a tail of its own (a workgroup's
// setup, first round and final cut: ~0.1 ms with the chip nearly empty) and every handed-over query
// pays the per-workgroup costs twice more: C2 0.715 ms without, 0.74-0.77 with 48-128 units, 0.88
// with 24 (tools/exp_lpt_oracle.py has the cost statistics).  Kept as an option; -1 = the automatic
// rule below, which no default selects.
// (launch_cost_order pays from about one residency of workgroups on: 7 per CU)
constexpr int COST_ORDER_MIN_QUERIES = 1024;
constexpr double BM_QB2_MAX_BYTES = 4.5e9;  // bucket-major rounds, 16-byte rows: two queries per group up to this many code bytes
constexpr int DEFER_MIN_QUERIES = 4096, DEFER_UNITS = 96, DEFER_SLICES = 2, DEFER_CAP = 2048;
constexpr int64_t BUCKET_MIN_ROWS = 900;       // average rows per bucket the bucketed order aims for
constexpr int64_t BUCKET_MIN_ROWS_10 = 1900;   // ... before it takes a tenth key bit
constexpr int64_t UPLOAD_CHUNK_ROWS = 1 << 22; // rows per host->device staging chunk
constexpr int64_t SEED_MIN_ROWS = 1 << 21;     // below this a scan is too short to need seeding
constexpr int64_t SEED_MIN_SLICES = 256;       // fewer, longer slices warm themselves up
constexpr int BF_STREAMED_MIN_QUERIES = 128;    // from here on the best-first form also takes streamed databases
constexpr int INPLACE_MAX_BATCHES = 4;         // query batches per scan up to which EA_INPLACE is chosen
// Bucket-major second pass (vaq_scan_bm.hip): a streamed database and so many queries that every
// bucket is wanted by several of them.  Pass A (best-first, one workgroup per query) is cut after
// about one average bucket's worth of work units; BM_CAND_CAP candidate slots per query.
constexpr int BM_MIN_QUERIES = 8;   // (125M x 16 B: 1 / 2 / 8 / 32 queries 0.45 / 0.58 / 1.15 / 2.44 ms with the shared-stream forms,
                                    //  0.63 / 0.63 / 0.72 / 0.87 ms with the rounds -- a chain of ~20 launches is their floor)
constexpr int BM_CAND_CAP = 4096;
constexpr int BM_QB = 4, BM_NWAVES = 16;
constexpr int BM_BOOT_MIN_UNITS = 24;

} // namespace

struct StagedState {
  bool open = false;
  vaq::BmParams bp;
  vaq::ScanParams sp;
  struct { int chunk, n, cap, qb, units; } bi;
  int k = 0, defer_cap = 0, nr = 0, r_next = 0;
  int limits[4] = {0, 0, 0, 0};
  int32_t *labels = nullptr;
  float *dist = nullptr;
};

struct vaqhip_index {
  int D = 0, M = 0, L = 0;
  int max_bits = 0, min_bits = 0, total_bits = 0, W = 0, layout = 0, lut_floats = 0;
  int device = 0, n_cu = 256;
  std::vector<int> bits;
  std::vector<vaq::SubDesc> sub;
  DevBuf d_cent, d_cent_t, d_eig, d_sub, d_first_sub, d_codes, d_perm, d_bstart;
  // byte codes bucketed by the whole first code: rows of a bucket are ordered by the rest of the
  // second code too, d_sub holds the first row of every (first code, second code) run
  // (sub_fine = bits of the second code below the bucket key; 0 = no such order, e.g. after an append)
  DevBuf d_substart;
  int sub_fine = 0;
  bool has_eig = false;
  int seq = 0;  // 1: BitVecEngine::queryLUT's sequential row sum
  int bucket_shift = 0, bucket_t = 0, n_buckets = 1;  // bucketed row order (set with the codes)
  int64_t N = -1, id_base = 0;
  int64_t N_keyed = 0;  // rows the bucket key width was chosen for (appends rebuild once N outgrows it 4x)
  // triangle-inequality form (VAQ::clusterTI): rows grouped by cluster instead of by first
  // code; d_bstart then holds the cluster starts, n_buckets = ti_T, bucket_shift = 0
  int ti_T = 0, ti_seg = 0;
  float ti_visit = 1.0f;              // mVisit
  unsigned methods = VAQHIP_METHOD_HEAP;
  DevBuf d_ti_clusters, d_ti_clusters_t, d_ti_xcc, w_ti_order, w_ti_qcc, w_ti_nvisit;
  // workspace (grow-only, reused across searches)
  DevBuf w_q, w_qproj, w_lut, w_part_d, w_part_id, w_part_cnt, w_labels, w_dist, w_stage, w_lutref, w_thr, w_ms_d, w_ms_id, w_order, w_qorder;
  DevBuf w_cost;   // [nq] cost keys of launch_cost_order
  DevBuf w_defer;  // [0] entries asked for, then DEFER_CAP records (best-first form, queries cut in two)
  // bucket-major second pass: plan arrays, per-bucket query lists, candidates, per-query words
  DevBuf w_bm_small, w_bm_mask, w_bm_qlist, w_bm_cand_d, w_bm_cand_id, w_bm_query, w_bm_thr64;
  // option "exact_ties": original row -> bucketed row (built at the first such search after the codes change),
  // the scan's k + 1 results, the replay list
  DevBuf d_inv, d_rowbucket, w_ex_labels, w_ex_dist, w_ex_list;
  StagedState staged;  // vaqhip_search_begin_device .. vaqhip_search_finish_device
  bool inv_valid = false;
  hipStream_t stream = nullptr;
  // The workspaces above are shared by every call on this index.  Host-side enqueues are
  // serialised by `mu`, but `_device` entry points run on the caller's stream: the last enqueue
  // that used the workspaces leaves an event, and a call on a DIFFERENT stream makes its stream
  // wait for it first (same stream: in order anyway).
  hipEvent_t ws_event = nullptr;
  hipStream_t ws_stream = nullptr;
  bool ws_used = false;
  // options
  int opt_qb = 0, opt_slices = 0, opt_timing = 0, opt_ea = 3, opt_nwaves = 0, opt_seed = 1, opt_hot = 16, opt_seed_frac = 64, opt_order = 0, opt_bucket_bits = 0, opt_no_skip = 0, opt_bf = 1, opt_group = 1, opt_defer = 0, opt_cost_order = 1, opt_bm = 1, opt_bm_cap = 0, opt_bm_units = 0, opt_bm_qb = 0, opt_bm_nwaves = 0, opt_sub_order = 1, opt_bm_sub = 1, opt_bm_boot = 1, opt_bm_round = 6, opt_exact = 0;
  // timing: a ring of 5-event sets, one per search since the last read
  static constexpr int EV_SETS = 256;
  std::vector<hipEvent_t> ev;   // EV_SETS * 6, created on first use
  int ev_used = 0;              // searches recorded since the last vaqhip_last_timing
  vaqhip_timing last = {};
  std::mutex mu;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

struct Plan {
  int qb, ea, kp, ccap, qcap, nwaves, n_slices;
  int lds_subs, lut_lds_entries;  // LUT tables staged in LDS (a prefix of the subspaces)
  int ti_cap = 0;                 // TI form: visiting-list entries staged at a time
  int64_t slice_rows;
  size_t lds;
  // sampling pre-pass that seeds the shared thresholds (0 slices = none)
  int seed_slices;
  int64_t seed_rows, seed_stride;
  bool ordered;  // slices dispatched best-first per query batch
  bool bf = false;  // best-first scan form (vaq_scan_bf.h)
  int bf_carry = 0;
  int bf_pool = 0;
  int defer_units = 0;  // > 0: expensive queries are cut in two (ScanParams::defer_*)
  bool cost_order = false;  // one best-first workgroup per query: expensive queries are dispatched first
  bool bm = false;          // bucket-major rounds (vaq_scan_bm.hip)
  bool bm_boot = false;     //   thresholds from a sample instead of a capped best-first pass
  int bm_qb = 0, bm_nwaves = 0, bm_cap = 0;
};

int make_plan(const vaqhip_index *ix, int nq, int k, Plan *pl) {
  // early-abandon form: 1 = queue, 2 = in place, 3 = auto (in place when few
  // query batches stream a database that does not fit the 256 MB Infinity
  // Cache, i.e. the scan is HBM-bound rather than instruction-bound)
  int ea = ix->opt_ea;
  if (ea == 3) {
    const double stream_bytes = (double)ix->N * ((ix->total_bits + 7) / 8);
    const int nqb_est = (nq + 1) / 2;
    ea = (stream_bytes > 256e6 && nqb_est <= INPLACE_MAX_BATCHES) ? vaq::EA_INPLACE : vaq::EA_QUEUE;
  }
  // default queries per pass: 2 for byte codes (one ds_read_b64 serves both), 1 for the
  // bit-packed path (more whole buckets are skipped when only one query has to agree)
  // ... and 1 as well for byte codes that stay cache-resident (<= 128 MB): sharing the code
  // stream between two queries buys nothing there, per-query bucket skipping does
  const bool resident = (double)ix->N * ((ix->total_bits + 7) / 8) <= 128e6;
  // ... and 4 for a streamed (non-resident) byte-coded database once there are enough queries to
  // fill the passes (250M x 16 B: 256 queries 18.4 -> 15.4 ms, 32 queries 3.6 -> 3.4 ms; no gain
  // below)
  int qb = ix->opt_qb > 0 ? ix->opt_qb
                          : ((ix->layout == vaq::LAYOUT_BYTES && !resident) ? (nq >= 32 ? 4 : 2) : 1);
  // ... but with MANY queries the best-first form (one query per workgroup, vaq_scan_bf.h) wins on
  // streamed databases as well: each query reads only the buckets in its own reach, nearest first,
  // and concurrent workgroups share what they read through L2 / Infinity Cache.  1B x 16 B encoded:
  // 2048 queries 367 -> 201 ms, 10 k queries 1275 -> 807 ms; 250M, 256 queries 14.5 -> 9.7 ms; at 64
  // queries the two are level (8.0 vs 8.3 ms at 500M) and below that the shared stream wins.
  const bool bf_streamed = ix->opt_qb == 0 && ix->opt_bf && !resident && nq >= BF_STREAMED_MIN_QUERIES && ea == vaq::EA_QUEUE &&
                           ix->ti_T == 0 && !ix->opt_order &&
                           vaq::scan_bf_supported(ix->layout, ix->M, 1, ea, ix->n_buckets, ix->seq);
  if (bf_streamed) qb = 1;
  // ... and with MORE queries still, several of them want every bucket: after a capped best-first
  // pass (one workgroup per query, its nearest buckets) the rest is scanned bucket-major, each
  // bucket streamed once for all the queries that reach it (vaq_scan_bm.hip)
  const bool bm = ix->opt_bm && ix->opt_bf && ix->ti_T == 0 && !ix->opt_order && !ix->opt_no_skip && ix->opt_slices <= 1 &&
                  (ix->opt_qb == 0 || ix->opt_bm == 2) && (ea == vaq::EA_QUEUE || ix->opt_bm == 2) && k <= 256 && ix->N > 0 &&
                  ((!resident && nq >= BM_MIN_QUERIES) || ix->opt_bm == 2) &&
                  vaq::scan_bm_supported(ix->layout, ix->M, ix->n_buckets, ix->bucket_shift, ix->seq, k) &&
                  vaq::scan_bf_supported(ix->layout, ix->M, 1, vaq::EA_QUEUE, ix->n_buckets, ix->seq);
  if (bm) {
    qb = 1;
    ea = vaq::EA_QUEUE;
  }
  if (nq < qb) qb = nq >= 2 ? 2 : 1;
  // Pick the workgroup size that puts the most wavefronts on a CU: the LUT and
  // the selection state are per workgroup, the survivor queues per wave; a CU
  // holds 160 KB of LDS and 32 waves (the scan kernels stay within 64 VGPRs
  // for Qb <= 2; Qb = 4 needs about twice that, i.e. half the waves).
  const int wave_cap = qb <= 2 ? 32 : 24;
  // LUT tables are staged in LDS for a prefix of the subspaces (all of them whenever they
  // fit; the byte-code kernels need all).  The bit-packed kernel reads the tail tables from
  // global memory, so big allocations (32 x up to 13 bits) still run; only table 0
  // (bucket bounds) must be resident.
  const int need = ix->layout == vaq::LAYOUT_BYTES ? ix->M : 1;
  int best_nw = 0, best_waves = 0, subs = ix->M, entries = ix->lut_floats;
  for (;;) {
    for (subs = ix->M; subs >= need; subs--) {
      entries = subs == ix->M ? ix->lut_floats : ix->sub[subs].lut_off;
      best_nw = 0;
      best_waves = 0;
      for (int nw : {4, 8, 16}) {
        if (ix->opt_nwaves > 0 && nw != ix->opt_nwaves) continue;
        const size_t lds = vaq::scan_lds_bytes(ix->layout, ix->M, entries, qb, k, ea, nw, ix->n_buckets,
                                               ix->bucket_shift, ix->bucket_t);
        if (lds > LDS_LIMIT) continue;
        const int wgs = (int)std::min<size_t>(LDS_LIMIT / lds, (size_t)(wave_cap / nw));
        // 16 waves share one admission lock and one ticket: measured much slower than 8 at equal
        // residency (C3: 5.8 vs 3.1 ms), so they must buy > 1.5x the waves to be chosen
        // (not when every bucket is streamed in place, where the lock is taken for admitted rows
        //  only and the waves just keep loads in flight: 1B rows, 16 waves x 2 workgroups per CU
        //  2.78 ms, 8 x 3 2.90 ms; with bucket skipping on it is the other way round, 250M rows x 2
        //  queries 0.27 vs 0.23 ms)
        const bool streaming = ea == vaq::EA_INPLACE && ix->opt_no_skip;
        const int score = (nw == 16 && !streaming) ? (wgs * nw * 2) / 3 : wgs * nw;
        if (score > best_waves) { best_waves = score; best_nw = nw; }
      }
      if (best_nw) break;
    }
    if (best_nw && subs < ix->M && qb > 2) best_nw = 0;  // spilled tables: kernels exist for Qb <= 2 only
    if (best_nw) break;
    if (qb > 1) qb >>= 1;
    else
      return fail(VAQHIP_EUNSUPPORTED,
                  "the lookup tables of the first %d subspaces plus top-%d buffers do not fit %zu B of LDS",
                  need, k, LDS_LIMIT);
  }
  pl->lds_subs = subs;
  pl->lut_lds_entries = entries;
  pl->qb = qb;
  pl->ea = ea;
  pl->nwaves = best_nw;
  vaq::scan_geometry(ix->layout, ix->M, k, ea, &pl->kp, &pl->ccap, &pl->qcap);
  pl->lds = vaq::scan_lds_bytes(ix->layout, ix->M, entries, qb, k, ea, best_nw, ix->n_buckets,
                                ix->bucket_shift, ix->bucket_t);
  const int step = vaq::scan_wg_step_rows(ix->layout, ix->M);
  const int64_t N = ix->N;
  const int nqb = (nq + qb - 1) / qb;
  int64_t s;
  if (bm) s = 1;
  else if (ix->opt_slices > 0) s = ix->opt_slices;
  else {
    // workgroups wanted in flight; the best-first form on a streamed database likes four times as
    // many (shorter workgroups: a query's cost varies tenfold and the launch ends with the longest;
    // 1B rows: 2048 queries x 1 / 2 / 4 / 8 / 16 slices 337 / 254 / 201 / 205 / 228 ms) and no slice
    // of more than 2^29 rows (10 k queries x 1 / 2 / 4 slices: 887 / 807 / 823 ms)
    const int64_t target = (int64_t)ix->n_cu * (bf_streamed ? 32 : 8);
    s = (target + nqb - 1) / nqb;
    if (bf_streamed) s = std::max<int64_t>(s, (N + ((int64_t)1 << 29) - 1) >> 29);
    const int64_t max_s = std::max<int64_t>(1, N / MIN_SLICE_ROWS);
    s = std::min(s, max_s);
  }
  s = std::max<int64_t>(1, s);
  int64_t rows = (N + s - 1) / s;
  rows = std::max<int64_t>(step, ((rows + step - 1) / step) * step);
  s = N > 0 ? (N + rows - 1) / rows : 1;
  pl->n_slices = (int)s;
  pl->slice_rows = rows;
  // Threshold seeding: when a query's rows are split over several workgroups,
  // each would otherwise warm its admission threshold up on its own slice
  // (k-th best of the few rows it has seen).  A pre-pass scans ~1/64 of the
  // rows, spread evenly, merges its top-k and publishes the k-th distance as
  // the starting threshold of every workgroup of the full scan: an upper
  // bound of the final k-th, so results are unchanged.
  pl->seed_slices = 0;
  pl->seed_rows = pl->seed_stride = 0;
  // best-first slice order (slice_order_kernel): an alternative to the pre-pass, off by default --
  // measured slower (250M rows, 2 queries: 1.21 vs 0.70 ms; 32 queries: 8.7 vs 5.6 ms): the first
  // wave of workgroups all starts cold, and batches no longer share a slice's rows through L2
  pl->ordered = ea && ix->opt_order && s > 1 && s <= 4096 && ix->n_buckets <= 4096 && ix->bucket_t == 0;
  if (ea && ix->opt_seed && !pl->ordered && s >= SEED_MIN_SLICES && N >= SEED_MIN_ROWS) {
    const int64_t sample = std::max<int64_t>(N / ix->opt_seed_frac, (int64_t)16 * k);
    // small workgroups (4 waves) and many slices: the pre-pass runs with cold
    // thresholds, where the waves of a workgroup queue on its admission lock
    int64_t ss = std::min<int64_t>(1024, std::max<int64_t>(8, sample / 8192));
    int64_t srows = ((sample / ss + step - 1) / step) * step;
    int64_t stride = (N / ss / step) * step;
    if (stride >= srows && srows > 0) {
      pl->seed_slices = (int)ss;
      pl->seed_rows = srows;
      pl->seed_stride = stride;
    }
  }
  // Best-first form: when a workgroup's slice spans many buckets (the cache-resident databases), all
  // of them are visited in ascending order of their bound with work units handed out by ticket.
  pl->bf = false;
  if (ix->opt_bf && ea == vaq::EA_QUEUE && qb == 1 && !pl->ordered && subs == ix->M &&
      vaq::scan_bf_supported(ix->layout, ix->M, qb, ea, ix->n_buckets, ix->seq) && ix->n_buckets >= 16 &&
      pl->slice_rows >= 8 * (N / ix->n_buckets + 1)) {
    int bnw = 0, bscore = 0, bpool = 0;
    size_t blds = 0;
    int pool_lo, pool_hi;
    vaq::scan_bf_pool_range(k, &pool_lo, &pool_hi);
    // bit-packed rows: when every field after the first group lies in the last dword, it is queued
    const int carry = (ix->layout == vaq::LAYOUT_BITS && ix->M > 4 && ix->sub[4].word == ix->W - 1) ? 1 : 0;
    for (int nw : {4, 8, 16}) {
      if (ix->opt_nwaves > 0 && nw != ix->opt_nwaves) continue;
      size_t lds = vaq::scan_bf_lds_bytes(ix->layout, ix->M, entries, pool_lo, nw, ix->n_buckets, carry);
      if (lds > LDS_LIMIT) continue;
      const int wgs = (int)std::min<size_t>(LDS_LIMIT / lds, (size_t)(32 / nw));
      // the largest k-min pool that keeps that many workgroups resident (LDS is handed out in
      // LDS_GRANULE pieces; at 72 VGPRs a SIMD holds 7 waves, so 4-wave workgroups stop at 7)
      const size_t budget = LDS_LIMIT / (size_t)std::min(wgs, std::max(1, BF_WAVES_PER_CU / nw)) / LDS_GRANULE * LDS_GRANULE;
      int pool = pool_lo;
      while (pool + 64 <= pool_hi &&
             vaq::scan_bf_lds_bytes(ix->layout, ix->M, entries, pool + 64, nw, ix->n_buckets, carry) <= budget)
        pool += 64;
      lds = vaq::scan_bf_lds_bytes(ix->layout, ix->M, entries, pool, nw, ix->n_buckets, carry);
      // small workgroups win here even at lower residency: setup, bootstrap and the final sort
      // are per workgroup and leave its other waves idle (C2: 4 waves x 6 workgroups per CU
      // 1.02 ms, 8 x 4 1.15 ms, 16 x 2 1.8 ms)
      const int score = nw == 4 ? wgs * nw * 10 : nw == 8 ? wgs * nw * 7 : wgs * nw * 4;
      if (score > bscore) { bscore = score; bnw = nw; blds = lds; bpool = pool; }
    }
    if (bnw) {
      pl->bf = true;
      pl->bf_carry = carry;
      pl->bf_pool = bpool;
      if (s == 1 && k <= 256 && (ix->opt_defer > 0 || (ix->opt_defer < 0 && nq >= DEFER_MIN_QUERIES)))
        pl->defer_units = ix->opt_defer > 0 ? ix->opt_defer : DEFER_UNITS;
      // more queries than workgroups resident at a time: start the expensive ones first
      // (calls of more than QUERY_CHUNK queries are served chunk by chunk, each ranked on its own)
      pl->cost_order = s == 1 && ix->opt_cost_order && nq >= COST_ORDER_MIN_QUERIES &&
                       (ix->sub[0].ncent >> ix->bucket_shift) <= 1024;
      pl->nwaves = bnw;
      pl->lds = blds;
      if (bm && s == 1) {
        pl->bm = true;
        pl->bm_qb = ix->opt_bm_qb > 0 ? ix->opt_bm_qb : BM_QB;
        // 16-byte rows: four queries' tables are 64 KB, one 16-wave workgroup per CU.  Two queries per group
        // in 8-wave workgroups are two workgroups per CU at twice the passes over a bucket's rows -- that
        // pays while those passes come out of L2, i.e. on shards up to about 4 GB of codes (10 k queries:
        // 62.5M / 125M / 250M rows 13.2 -> 12.2 / 16.6 -> 15.6 / 23.8 -> 23.3 ms; 500M 37.7 -> 39.0, 1B 65.6 -> 82.1)
        if (ix->opt_bm_qb <= 0 && ix->M == 16 && (double)N * 16.0 <= BM_QB2_MAX_BYTES) pl->bm_qb = 2;
        pl->bm_nwaves = ix->opt_bm_nwaves > 0 ? ix->opt_bm_nwaves : BM_NWAVES;
        if (ix->opt_bm_nwaves <= 0) {
          // 16 or 8 waves per workgroup: whichever keeps more waves resident on a CU, and on a tie the
          // smaller workgroups (more items in flight, shorter waits at an item's barriers).  8-byte rows:
          // 16 waves need 90 KB (one workgroup), 8 waves 65 KB (two): C4 7.9 -> 6.8 ms; 16-byte rows have
          // room for one workgroup either way, and 16 waves are 66 ms at 1B rows where 8 are 92.
          int best_res = 0;
          for (int nw : {16, 8}) {
            const size_t lds = vaq::scan_bm_lds_bytes(ix->M, pl->bm_qb, nw) + 8192;
            if (lds > LDS_LIMIT) continue;
            const int res = std::min<int>(32, nw * (int)(LDS_LIMIT / lds));
            if (res >= best_res) { best_res = res; pl->bm_nwaves = nw; }
          }
        }
        pl->bm_cap = ix->opt_bm_cap > 0 ? ix->opt_bm_cap : BM_CAND_CAP;
        while (vaq::scan_bm_lds_bytes(ix->M, pl->bm_qb, pl->bm_nwaves) + 8192 > LDS_LIMIT && pl->bm_nwaves > 4) pl->bm_nwaves >>= 1;
        // pass A: about one average bucket per query (a work unit = 64 wave steps)
        const int64_t unit_rows = 64 * (int64_t)(vaq::scan_wg_step_rows(ix->layout, ix->M) / vaq::SCAN_MAX_WAVES);
        const int64_t avg = N / ix->n_buckets + 1;
        const int64_t bucket_units = (avg + unit_rows - 1) / unit_rows;
        pl->defer_units = ix->opt_bm_units > 0 ? ix->opt_bm_units
                                               : (int)std::min<int64_t>(4096, std::max<int64_t>(8, bucket_units));
        // Small buckets: a best-first pass over each query's nearest one is cheap and leaves a better
        // threshold than a sample (100M x 8 B, 10 k queries: 9.4 ms against 13.8).  Large buckets: that
        // pass streams 10 k buckets from HBM with nothing shared (1B x 16 B: 63 ms of 161), so a
        // sampled threshold and the nearest bucket as the first bucket-major round (11 ms).
        pl->bm_boot = ix->opt_bm_boot == 1 ? bucket_units >= BM_BOOT_MIN_UNITS : ix->opt_bm_boot != 0;
      }
    }
  }
  return VAQHIP_OK;
}

// Launch geometry of the TI form: one query per workgroup (Qb = 1, survivors queued), each
// query's work units spread over `n_slices` workgroups when there are few queries.
int make_ti_plan(const vaqhip_index *ix, int nq, int k, Plan *pl) {
  const int qb = 1, ea = vaq::EA_QUEUE;
  // the visiting list is int(T * visit) clusters long unless the until-k-rows rule extends it:
  // stage that many (rounded up to a wave's worth) at a time; longer lists go in chunks
  const int max_visit = ix->ti_visit < 1.0f ? (int)((float)ix->ti_T * ix->ti_visit) : ix->ti_T;
  pl->ti_cap = std::min(ix->ti_T, std::max(64, ((max_visit + 63) / 64) * 64));
  const size_t ti_bytes = vaq::scan_ti_lds_bytes(pl->ti_cap);
  const int need = ix->layout == vaq::LAYOUT_BYTES ? ix->M : 1;
  int best_nw = 0, best_waves = 0, subs = ix->M, entries = ix->lut_floats;
  for (subs = ix->M; subs >= need; subs--) {
    entries = subs == ix->M ? ix->lut_floats : ix->sub[subs].lut_off;
    best_nw = 0;
    best_waves = 0;
    for (int nw : {4, 8, 16}) {
      if (ix->opt_nwaves > 0 && nw != ix->opt_nwaves) continue;
      const size_t lds = vaq::scan_lds_bytes(ix->layout, ix->M, entries, qb, k, ea, nw, ix->ti_T, 0, 0) + ti_bytes;
      if (lds > LDS_LIMIT) continue;
      const int wgs = (int)std::min<size_t>(LDS_LIMIT / lds, (size_t)(32 / nw));
      if (wgs * nw > best_waves) { best_waves = wgs * nw; best_nw = nw; }
    }
    if (best_nw) break;
  }
  if (!best_nw)
    return fail(VAQHIP_EUNSUPPORTED,
                "the lookup tables of the first %d subspaces, top-%d buffers and %d clusters do not fit "
                "%zu B of LDS", need, k, ix->ti_T, LDS_LIMIT);
  pl->lds_subs = subs;
  pl->lut_lds_entries = entries;
  pl->qb = qb;
  pl->ea = ea;
  pl->nwaves = best_nw;
  vaq::scan_geometry(ix->layout, ix->M, k, ea, &pl->kp, &pl->ccap, &pl->qcap);
  pl->lds = vaq::scan_lds_bytes(ix->layout, ix->M, entries, qb, k, ea, best_nw, ix->ti_T, 0, 0) + ti_bytes;
  int64_t s;
  if (ix->opt_slices > 0) s = ix->opt_slices;
  else {
    // enough workgroups to fill the chip, but no more than the visited rows give work units
    // (one unit = 16 wave steps) to two rounds of a workgroup's waves
    const int64_t target = (int64_t)ix->n_cu * 8;
    s = (target + nq - 1) / nq;
    const int64_t unit_rows = 16 * (vaq::scan_wg_step_rows(ix->layout, ix->M) / vaq::SCAN_MAX_WAVES);
    const double frac = ix->ti_visit < 1.0f ? std::max(ix->ti_visit, 1.0f / ix->ti_T) : 1.0;
    const int64_t units = (int64_t)(frac * ((double)ix->N / unit_rows + ix->ti_T));
    s = std::min<int64_t>(s, std::max<int64_t>(1, units / (2 * best_nw)));
  }
  pl->n_slices = (int)std::max<int64_t>(1, std::min<int64_t>(s, 4096));
  pl->slice_rows = 0;
  pl->seed_slices = 0;
  pl->seed_rows = pl->seed_stride = 0;
  pl->ordered = false;
  return VAQHIP_OK;
}

// before / after enqueueing work that touches the index's shared workspaces on stream `st`
int ws_acquire(vaqhip_index *ix, hipStream_t st) {
  if (!ix->ws_event) HIP_TRY(hipEventCreateWithFlags(&ix->ws_event, hipEventDisableTiming));
  if (ix->ws_used && st != ix->ws_stream) HIP_TRY(hipStreamWaitEvent(st, ix->ws_event, 0));
  return VAQHIP_OK;
}
int ws_release(vaqhip_index *ix, hipStream_t st) {
  HIP_TRY(hipEventRecord(ix->ws_event, st));
  ix->ws_stream = st;
  ix->ws_used = true;
  return VAQHIP_OK;
}

int ensure_events(vaqhip_index *ix) {
  if (!ix->ev.empty()) return VAQHIP_OK;
  std::vector<hipEvent_t> ev(vaqhip_index::EV_SETS * 6);
  for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
  ix->ev.swap(ev);
  return VAQHIP_OK;
}

// ---- bucket-major rounds (vaq_scan_bm.hip): helpers shared by the one-call search and the staged one ----
struct BmRoundInfo {
  int chunk, n, cap, qb, units;
};

// rounds [r0, r1) of nr: plan, scan, select
int bm_run_rounds(vaqhip_index *ix, vaq::BmParams &bp, const int *limits, int r0, int r1, int nr, const BmRoundInfo &bi,
                  hipStream_t st) {
  for (int r = r0; r < r1; r++) {
    bp.retry = r + 1 < nr ? 1 : 0;
    bp.limit = limits[r];
    bp.init64 = r == 0 ? 1 : 0;
    HIP_TRY(vaq::launch_bm_plan(bp, st));
    HIP_TRY(vaq::launch_scan_bm(bp, ix->n_cu, st));
    HIP_TRY(vaq::launch_bm_select(bp, st));
    if (getenv("VAQHIP_BM_DEBUG")) {  // diagnostic (synchronises): what the round planned and appended
      const int chunk = bi.chunk, n = bi.n;
      std::vector<unsigned> hq((size_t)3 * chunk);
      std::vector<int> hcnt((size_t)ix->n_buckets);
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipMemcpy(hq.data(), ix->w_bm_query.p, hq.size() * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(hcnt.data(), bp.cnt, hcnt.size() * 4, hipMemcpyDeviceToHost));
      unsigned long long handed = 0, appended = 0, over = 0, maxc = 0, pairs = 0, groups = 0, work = 0;
      for (int i = 0; i < n; i++) {
        if (hq[i] != 0xffffffffu) handed++;
        const unsigned c = hq[(size_t)chunk + i];
        appended += c;
        over += c > (unsigned)bi.cap;
        maxc = std::max<unsigned long long>(maxc, c);
      }
      std::vector<int> hb((size_t)ix->n_buckets + 1);
      HIP_TRY(hipMemcpy(hb.data(), ix->d_bstart.p, hb.size() * 4, hipMemcpyDeviceToHost));
      for (int b = 0; b < ix->n_buckets; b++) {
        pairs += hcnt[b];
        const unsigned long long g = (hcnt[b] + bi.qb - 1) / bi.qb;
        groups += g;
        work += g * (unsigned long long)(hb[b + 1] - hb[b]);
      }
      std::fprintf(stderr, "[VAQHIP_BM_DEBUG] round %d (limit %d): queries %d still open after it %llu; (query, bucket) pairs %llu, items %llu, row-steps x QB "
                           "%.3e (= %.2f %% of rows per query slot); candidates appended %llu (max %llu per query), overflowed "
                           "queries %llu; pass A units %d\n",
                   r, limits[r], n, handed, pairs, groups, (double)work * bi.qb, 100.0 * (double)work * bi.qb / ((double)n * (double)ix->N),
                   appended, maxc, over, bi.units);
    }
  }
  return VAQHIP_OK;
}

// what the expensive / overflowed queries have left: the best-first form's second launch, DEFER_SLICES
// workgroups per listed query (those beyond the list's length return at once), merged into the results
int bm_fallback(vaqhip_index *ix, const vaq::ScanParams &sp, int defer_cap, int k, int32_t *labels, float *dist,
                hipStream_t st) {
  vaq::ScanParams s2 = sp;
  s2.defer_units = 0;
  s2.defer_mode = 1;
  s2.bm_done = nullptr;
  s2.qorder = nullptr;
  s2.nq = defer_cap;
  s2.n_slices = DEFER_SLICES;
  const int step2 = vaq::scan_wg_step_rows(ix->layout, ix->M);
  int64_t rows2 = (ix->N + DEFER_SLICES - 1) / DEFER_SLICES;
  rows2 = std::max<int64_t>(step2, ((rows2 + step2 - 1) / step2) * step2);
  s2.slice_rows = rows2;
  s2.slice_stride = rows2;
  s2.share_thr = 1;
  s2.final_labels = nullptr;
  s2.final_dist = nullptr;
  int grid2 = 0;
  HIP_TRY(vaq::launch_scan(s2, &grid2, st));
  HIP_TRY(vaq::launch_defer_merge(sp.defer_count, defer_cap, sp.defer_list, DEFER_SLICES, k, sp.part_d, sp.part_id,
                                  sp.part_cnt, ix->id_base, labels, dist, st));
  return VAQHIP_OK;
}

// core: device pointers in, device pointers out, enqueue only
int search_core(vaqhip_index *ix, const float *d_queries, int nq, int k, int projected,
                int32_t *d_labels, float *d_dist, hipStream_t st, int32_t *stage_thr_out = nullptr) {
  if (ix->N < 0) return fail(VAQHIP_ESTATE, "search before codes were set");
  if (ix->staged.open)
    return fail(VAQHIP_ESTATE, "a staged search is open on this index: call vaqhip_search_finish_device first");
  if (((ix->methods & VAQHIP_METHOD_TI) != 0) != (ix->ti_T > 0))
    return fail(VAQHIP_ESTATE, ix->ti_T > 0 ? "the rows are grouped by TI cluster: the method must include TI"
                                             : "method TI needs vaqhip_index_set_ti_clusters first");
  if (nq < 0 || k <= 0) return fail(VAQHIP_EINVAL, "nq=%d k=%d", nq, k);
  if (k > VAQHIP_MAX_K) return fail(VAQHIP_EUNSUPPORTED, "k=%d > %d", k, VAQHIP_MAX_K);
  if (nq == 0) return VAQHIP_OK;
  if (!d_queries || !d_labels || !d_dist) return fail(VAQHIP_EINVAL, "null pointer");

  {
    int rc = ws_acquire(ix, st);
    if (rc) return rc;
  }
  bool timing = ix->opt_timing != 0;
  hipEvent_t *ev = nullptr;
  if (timing) {
    int rc = ensure_events(ix);
    if (rc) return rc;
    if (ix->ev_used >= vaqhip_index::EV_SETS) timing = false;  // ring full: stop recording
    else ev = ix->ev.data() + (size_t)ix->ev_used * 6;
  }
  vaqhip_timing tm = {};
  tm.deferred_queries = -1;
  Plan pl;
  const bool ti = ix->ti_T > 0;
  {
    int rc = ti ? make_ti_plan(ix, std::min(nq, QUERY_CHUNK), k, &pl)
                : make_plan(ix, std::min(nq, QUERY_CHUNK), k, &pl);
    if (rc) return rc;
  }
  const int chunk = std::min(nq, QUERY_CHUNK);
  if (stage_thr_out) {
    if (!(pl.bm && pl.bf && pl.n_slices == 1 && ix->N > 0 && nq <= QUERY_CHUNK && !ti))
      return fail(VAQHIP_EUNSUPPORTED, "a staged search needs the bucket-major rounds (streamed byte codes, >= 8 queries, "
                                       "at most %d per call)", QUERY_CHUNK);
    timing = false;
  }
  // (BitVecEngine::queryLUT projects with checking, BitVecEngine.hpp:1226: non-finite coordinates -> 0,
  //  which needs a pass over the queries even without a rotation)
  const bool do_project = !projected && (ix->has_eig || ix->seq);
  if (do_project) HIP_TRY(ix->w_qproj.ensure((size_t)chunk * ix->D * sizeof(float)));
  HIP_TRY(ix->w_lut.ensure((size_t)chunk * ix->lut_floats * sizeof(float)));
  const int nslots = std::max(pl.n_slices, pl.seed_slices);
  HIP_TRY(ix->w_part_d.ensure((size_t)chunk * nslots * k * sizeof(float)));
  HIP_TRY(ix->w_part_id.ensure((size_t)chunk * nslots * k * sizeof(int)));
  HIP_TRY(ix->w_part_cnt.ensure((size_t)chunk * nslots * sizeof(int)));
  HIP_TRY(ix->w_thr.ensure((size_t)chunk * sizeof(unsigned)));
  if (pl.cost_order) {
    HIP_TRY(ix->w_qorder.ensure((size_t)chunk * sizeof(int)));
    HIP_TRY(ix->w_cost.ensure((size_t)chunk * sizeof(unsigned long long)));
  }
  if (pl.bm) {
    const size_t K0 = (size_t)ix->n_buckets;
    HIP_TRY(ix->w_bm_small.ensure(vaq::bm_plan_small_words(ix->n_buckets) * 4));
    HIP_TRY(ix->w_bm_mask.ensure((size_t)chunk * (K0 / 32) * 4));
    HIP_TRY(ix->w_bm_qlist.ensure((size_t)chunk * K0 * sizeof(int)));
    HIP_TRY(ix->w_bm_cand_d.ensure((size_t)chunk * pl.bm_cap * sizeof(float)));
    HIP_TRY(ix->w_bm_cand_id.ensure((size_t)chunk * pl.bm_cap * sizeof(int)));
    // per query: done_key, candidate count, scale, next done_key, fresh, histogram, list keys (16 x 2 bytes)
    HIP_TRY(ix->w_bm_query.ensure((size_t)chunk * (5 + vaq::BM_HIST_BINS + 8) * 4));
    HIP_TRY(ix->w_bm_thr64.ensure((size_t)chunk * sizeof(unsigned long long)));
    // overflowed queries are finished by the best-first form's second launch
    HIP_TRY(ix->w_defer.ensure(16 + (size_t)chunk * sizeof(vaq::DeferRec)));
    HIP_TRY(ix->w_part_d.ensure((size_t)chunk * DEFER_SLICES * k * sizeof(float)));
    HIP_TRY(ix->w_part_id.ensure((size_t)chunk * DEFER_SLICES * k * sizeof(int)));
    HIP_TRY(ix->w_part_cnt.ensure((size_t)chunk * DEFER_SLICES * sizeof(int)));
  } else if (pl.defer_units > 0) {
    HIP_TRY(ix->w_defer.ensure(16 + (size_t)DEFER_CAP * sizeof(vaq::DeferRec)));
    HIP_TRY(ix->w_part_d.ensure((size_t)DEFER_CAP * DEFER_SLICES * k * sizeof(float)));
    HIP_TRY(ix->w_part_id.ensure((size_t)DEFER_CAP * DEFER_SLICES * k * sizeof(int)));
    HIP_TRY(ix->w_part_cnt.ensure((size_t)DEFER_CAP * DEFER_SLICES * sizeof(int)));
  }
  {
    const size_t ms = vaq::merge_scratch_elems(nslots, chunk, k);
    HIP_TRY(ix->w_ms_d.ensure(std::max<size_t>(ms, 1) * sizeof(float)));
    HIP_TRY(ix->w_ms_id.ensure(std::max<size_t>(ms, 1) * sizeof(int)));
  }

  if (ti) {
    HIP_TRY(ix->w_ti_order.ensure((size_t)chunk * ix->ti_T * sizeof(int)));
    HIP_TRY(ix->w_ti_qcc.ensure((size_t)chunk * ix->ti_T * sizeof(float)));
    HIP_TRY(ix->w_ti_nvisit.ensure((size_t)chunk * sizeof(int)));
  }

  if (timing && nq > chunk)
    return fail(VAQHIP_EUNSUPPORTED, "timing supports at most %d queries per call", QUERY_CHUNK);

  for (int q0 = 0; q0 < nq; q0 += chunk) {
    const int n = std::min(chunk, nq - q0);
    const float *dq = d_queries + (size_t)q0 * ix->D;
    const float *qp = dq;
    if (timing) HIP_TRY(hipEventRecord(ev[0], st));
    if (do_project) {
      HIP_TRY(vaq::launch_project(dq, n, ix->D, ix->has_eig ? ix->d_eig.as<float>() : nullptr, ix->w_qproj.as<float>(), st,
                                  ix->seq ? 1 : 0));
      qp = ix->w_qproj.as<float>();
    }
    if (timing) HIP_TRY(hipEventRecord(ev[1], st));
    HIP_TRY(vaq::launch_lut_build(qp, n, ix->D, ix->M, ix->L, ix->d_sub.as<vaq::SubDesc>(),
                                  ix->d_cent_t.as<float>(), ix->lut_floats, 1 << ix->max_bits, ix->w_lut.as<float>(), st,
                                  1 << ix->min_bits));
    if (timing) HIP_TRY(hipEventRecord(ev[2], st));
    vaq::ScanParams sp;
    sp.codes = ix->d_codes.as<uint32_t>();
    sp.n_rows = ix->N;
    sp.layout = ix->layout;
    sp.M = ix->M;
    sp.W = ix->W;
    sp.sub = ix->d_sub.as<vaq::SubDesc>();
    sp.first_sub = ix->d_first_sub.as<int>();
    sp.perm = ix->d_perm.as<uint32_t>();
    sp.bucket_start = ix->d_bstart.as<int>();
    sp.n_buckets = ix->n_buckets;
    sp.bucket_shift = ix->bucket_shift;
    sp.bucket_t = ix->bucket_t;
    sp.n_hot = 0;
    sp.bf = 0;
    sp.bf_carry = 0;
    sp.bf_pool = 0;
    sp.defer_units = 0;
    sp.defer_mode = 0;
    sp.defer_cap = 0;
    sp.defer_count = nullptr;
    sp.defer_list = nullptr;
    sp.bm_done = nullptr;
    sp.no_skip = ix->opt_no_skip;
    sp.stats = nullptr;
#if defined(VAQ_STATS) || defined(VAQ_PHASES)
    static unsigned long long *d_stats = nullptr;
    if (!d_stats) HIP_TRY(hipMalloc(&d_stats, 24 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d_stats, 0, 24 * sizeof(unsigned long long), st));
    sp.stats = d_stats;
#endif
    sp.lut = ix->w_lut.as<float>();
    sp.lut_floats = ix->lut_floats;
    sp.lds_subs = pl.lds_subs;
    sp.lut_lds_entries = pl.lut_lds_entries;
    sp.nq = n;
    sp.k = k;
    sp.kp = pl.kp;
    sp.ccap = pl.ccap;
    sp.qcap = pl.qcap;
    sp.ea = pl.ea;
    sp.seq = ix->seq;
    sp.nwaves = pl.nwaves;
    sp.g_thr = ix->w_thr.as<unsigned>();
    sp.qb = pl.qb;
    sp.part_d = ix->w_part_d.as<float>();
    sp.part_id = ix->w_part_id.as<int>();
    sp.part_cnt = ix->w_part_cnt.as<int>();
    sp.final_labels = nullptr;
    sp.final_dist = nullptr;
    sp.slice_order = nullptr;
    sp.qorder = nullptr;
    sp.id_base = ix->id_base;
    sp.ti = 0;
    sp.ti_order = nullptr;
    sp.ti_qcc = nullptr;
    sp.ti_nvisit = nullptr;
    sp.ti_xcc = nullptr;
    sp.ti_rowcap = 0x7fffffff;
    sp.ti_cap = 0;
    sp.sqrt_out = 0;
    int grid = 0;
    // Multi-query passes over a streamed database: put queries with the same nearest first and
    // second codes into the same pass (a pass visits the union of its queries' buckets).
    // "group_queries": 1 = when it pays (streamed codes, several passes), 2 = always, 0 = never.
    if (!ti && pl.qb > 1 && ix->M > 1 && n <= 16384 && !pl.ordered &&
        (ix->opt_group == 2 ||
         (ix->opt_group == 1 && n >= 4 * pl.qb && (double)ix->N * ((ix->total_bits + 7) / 8) > 256e6))) {
      HIP_TRY(ix->w_qorder.ensure((size_t)n * sizeof(int)));
      HIP_TRY(vaq::launch_query_order(ix->w_lut.as<float>(), ix->lut_floats, n, ix->sub[0].ncent, ix->sub[1].lut_off,
                                      ix->sub[1].ncent, ix->w_qorder.as<int>(), st));
      sp.qorder = ix->w_qorder.as<int>();
    }
    // shared admission thresholds start at heap_heapify's neutral FLT_MAX (0x7f7fffff); a query
    // served by ONE workgroup and no pre-pass never reads the word (share_thr = 0 below)
    if (pl.n_slices > 1 || pl.seed_slices > 0 || pl.bm)
      HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ix->w_thr.p), 0x7f7fffff, n, st));
    if (ti) {
      // VAQ::search's TI branch (VAQ.cpp:799-826) then VAQ::searchTriangleInequality (:1540-1692)
      const int T = ix->ti_T;
      const int max_visit = ix->ti_visit < 1.0f ? (int)((float)T * ix->ti_visit) : T;  // :1548-1551
      HIP_TRY(vaq::launch_ti_plan(qp, n, ix->D, ix->ti_seg * ix->L, ix->d_ti_clusters_t.as<float>(), T,
                                  ix->d_bstart.as<int>(), max_visit, k, ix->w_ti_order.as<int>(),
                                  ix->w_ti_qcc.as<float>(), ix->w_ti_nvisit.as<int>(), st));
      if (timing) HIP_TRY(hipEventRecord(ev[3], st));
      sp.ti = 1;
      sp.ti_order = ix->w_ti_order.as<int>();
      sp.ti_qcc = ix->w_ti_qcc.as<float>();
      sp.ti_nvisit = ix->w_ti_nvisit.as<int>();
      sp.ti_xcc = ix->d_ti_xcc.as<float>();
      // without EA the reference never admits a row after the first k of the visiting order
      // (bsfKSquared stays 0, VAQ.cpp:1617-1686): reproduce that by taking only those rows
      sp.ti_rowcap = (ix->methods & VAQHIP_METHOD_EA) ? 0x7fffffff : k;
      sp.ti_cap = pl.ti_cap;
      sp.sqrt_out = 1;
      sp.n_slices = pl.n_slices;
      sp.slice_rows = 0;
      sp.slice_stride = 0;
      sp.share_thr = pl.n_slices > 1;
      const bool direct = pl.n_slices == 1 && ix->N > 0;
      if (direct) {
        sp.final_labels = d_labels + (size_t)q0 * k;
        sp.final_dist = d_dist + (size_t)q0 * k;
      }
      if (ix->N > 0) HIP_TRY(vaq::launch_scan(sp, &grid, st));
      if (timing) HIP_TRY(hipEventRecord(ev[4], st));
      if (!direct)
        HIP_TRY(vaq::launch_merge(sp.part_d, sp.part_id, nullptr, ix->N > 0 ? pl.n_slices : 0, k,
                                  (int64_t)pl.n_slices * k, n, k, ix->id_base, 0,
                                  d_labels + (size_t)q0 * k, d_dist + (size_t)q0 * k, nullptr,
                                  ix->w_ms_d.as<float>(), ix->w_ms_id.as<int>(), st));
      if (timing) HIP_TRY(hipEventRecord(ev[5], st));
      tm.seed_slices = 0;
      tm.early_abandon = pl.ea;
      tm.queries_per_pass = 1;
      tm.slices = pl.n_slices;
      tm.workgroups = grid;
      tm.passes = n;
      tm.lds_bytes = (int)pl.lds;
      continue;
    }
    if (ix->N > 0 && pl.seed_slices > 0) {
      sp.n_slices = pl.seed_slices;
      sp.slice_rows = pl.seed_rows;
      sp.slice_stride = pl.seed_stride;
      sp.share_thr = 1;
      sp.nwaves = 4;
      HIP_TRY(vaq::launch_scan(sp, nullptr, st));
      sp.nwaves = pl.nwaves;
      // (the pre-pass ran cold: its lists are full, so the plain 16-way tree, not the compacting level)
      HIP_TRY(vaq::launch_merge(sp.part_d, sp.part_id, nullptr, pl.seed_slices, k, (int64_t)pl.seed_slices * k, n,
                                k, 0, 0, nullptr, nullptr, ix->w_thr.as<unsigned>(),
                                ix->w_ms_d.as<float>(), ix->w_ms_id.as<int>(), st));
    }
    // (the ranking of the queries counts as a pre-pass in the timing: "seed_ms")
    if (pl.bf && pl.cost_order && pl.n_slices == 1 && ix->N > 0 && !ti && n >= COST_ORDER_MIN_QUERIES) {
      HIP_TRY(vaq::launch_cost_order(ix->w_lut.as<float>(), ix->lut_floats, n, ix->sub[0].ncent, ix->bucket_shift,
                                     ix->w_cost.as<unsigned long long>(), ix->w_qorder.as<int>(), st));
      sp.qorder = ix->w_qorder.as<int>();
    }
    if (timing) HIP_TRY(hipEventRecord(ev[3], st));
    sp.n_slices = pl.n_slices;
    sp.slice_rows = pl.slice_rows;
    sp.slice_stride = pl.slice_rows;
    sp.share_thr = pl.n_slices > 1;
    if (pl.ordered && ix->N > 0) {
      const int nqb = (n + pl.qb - 1) / pl.qb;
      HIP_TRY(ix->w_order.ensure((size_t)nqb * pl.n_slices * sizeof(int)));
      HIP_TRY(vaq::launch_slice_order(ix->w_lut.as<float>(), ix->lut_floats, n, pl.qb, ix->d_bstart.as<int>(),
                                      ix->n_buckets, ix->bucket_shift, pl.slice_rows, pl.n_slices, ix->N,
                                      ix->w_order.as<int>(), st));
      sp.slice_order = ix->w_order.as<int>();
    }
    const bool direct = pl.n_slices == 1 && ix->N > 0;  // the single list per query is the result
    if (direct) {
      sp.final_labels = d_labels + (size_t)q0 * k;
      sp.final_dist = d_dist + (size_t)q0 * k;
    }
    // best-first buckets pay when a workgroup's slice spans many buckets
    // (its ranking scratch, one word per bucket, borrows the LDS region of the lookup tables)
    sp.n_hot = (ix->opt_hot && sp.n_buckets >= 16 && sp.n_buckets <= 4096 &&
                (int64_t)sp.n_buckets <= (int64_t)(ix->layout == vaq::LAYOUT_BYTES ? ix->M * 256 : pl.lut_lds_entries) * pl.qb &&
                pl.slice_rows >= 8 * (ix->N / sp.n_buckets + 1)) ? ix->opt_hot : 0;
    sp.bf = pl.bf ? 1 : 0;
    sp.bf_carry = pl.bf_carry;
    sp.bf_pool = pl.bf_pool;
    const bool bm = pl.bm && pl.bf && direct && !ti;
    const bool defer = pl.bf && pl.defer_units > 0 && direct && !bm;
    const int defer_cap = bm ? n : DEFER_CAP;
    vaq::BmParams bp = {};
    if (bm) {
      // pass A: every query's nearest buckets, capped; what is left in reach is handed over
      unsigned *qw = ix->w_bm_query.as<unsigned>();
      sp.defer_units = pl.defer_units;
      sp.defer_cap = defer_cap;
      sp.defer_count = ix->w_defer.as<unsigned>();
      sp.defer_list = reinterpret_cast<vaq::DeferRec *>(ix->w_defer.as<unsigned char>() + 16);
      sp.bm_done = qw;
      HIP_TRY(hipMemsetAsync(sp.defer_count, 0, sizeof(unsigned), st));
      HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(qw), 0xffffffffu, n, st));
      bp.codes = sp.codes;
      bp.perm = sp.perm;
      bp.bucket_start = sp.bucket_start;
      bp.n_buckets = sp.n_buckets;
      bp.bucket_t = sp.bucket_t;
      bp.sub_start = (ix->sub_fine > 0 && ix->sub_fine + ix->bucket_t == 8 && ix->opt_bm_sub) ? ix->d_substart.as<int>() : nullptr;
      bp.M = ix->M;
      bp.lut = sp.lut;
      bp.lut_floats = sp.lut_floats;
      bp.nq = n;
      bp.k = k;
      bp.qb = pl.bm_qb;
      bp.nwaves = pl.bm_nwaves;
      bp.g_thr = sp.g_thr;
      bp.thr64 = ix->w_bm_thr64.as<unsigned long long>();
      bp.init64 = 0;
      bp.done_key = qw;
      bp.cand_cnt = qw + (size_t)chunk;
      bp.scale = reinterpret_cast<float *>(qw + (size_t)2 * chunk);
      bp.done_next = qw + (size_t)3 * chunk;
      bp.fresh = qw + (size_t)4 * chunk;
      bp.hist = qw + (size_t)5 * chunk;
      bp.qkey = reinterpret_cast<unsigned short *>(qw + (size_t)(5 + vaq::BM_HIST_BINS) * chunk);
      bp.limit = 0;
      bp.retry = 0;
      HIP_TRY(hipMemsetAsync(bp.fresh, 0, (size_t)n * 4, st));
      bp.mask = ix->w_bm_mask.as<unsigned>();
      {
        int *sm = ix->w_bm_small.as<int>();
        const int K0 = ix->n_buckets;
        bp.cnt = sm;
        bp.qoff = sm + K0;
        bp.fill = sm + 2 * K0 + 1;
        bp.border = sm + 3 * K0 + 1;
        bp.ioff = sm + 4 * K0 + 1;
        bp.tickets = reinterpret_cast<unsigned *>(sm + vaq::bm_plan_small_words(K0) - vaq::BM_XCDS);
      }
      bp.qlist = ix->w_bm_qlist.as<int>();
      bp.cap = pl.bm_cap;
      bp.cand_d = ix->w_bm_cand_d.as<float>();
      bp.cand_id = ix->w_bm_cand_id.as<int>();
      bp.labels = d_labels + (size_t)q0 * k;
      bp.dist = d_dist + (size_t)q0 * k;
      bp.id_base = ix->id_base;
      bp.defer_count = sp.defer_count;
      bp.defer_list = sp.defer_list;
      bp.defer_cap = defer_cap;
    }
    if (defer) {
      sp.defer_units = pl.defer_units;
      sp.defer_cap = DEFER_CAP;
      sp.defer_count = ix->w_defer.as<unsigned>();
      sp.defer_list = reinterpret_cast<vaq::DeferRec *>(ix->w_defer.as<unsigned char>() + 16);
      HIP_TRY(hipMemsetAsync(sp.defer_count, 0, sizeof(unsigned), st));
    }
    const bool bm_boot = bm && pl.bm_boot;
    if (bm_boot) {
      // no best-first pass: a threshold per query from a sample of its nearest rows, then the
      // nearest bucket of every query is the first bucket-major round
      HIP_TRY(vaq::launch_bm_boot(bp, ix->N, st));
    } else if (ix->N > 0) {
      HIP_TRY(vaq::launch_scan(sp, &grid, st));
    }
    if (bm) {
      // rounds of plan, scan, select (vaq_scan_bm.hip): each query's nearest bucket [after a sampled
      // threshold], its next few, then everything still in reach -- thresholds are near their final
      // values before the bulk of the rows is met.  Queries whose candidate buffer overflows join
      // the defer list.
      // A query whose candidate buffer overflows in a round keeps its place: what was stored tightens
      // its threshold and the next round plans the same buckets again; one more round (nothing to do
      // when no buffer overflowed) gives the last regular round that second try too, and only what
      // overflows THERE is left to the best-first form.
      int limits[4], nr = 0;
      if (bm_boot) limits[nr++] = 1;
      if (ix->opt_bm_round > 0) limits[nr++] = ix->opt_bm_round;
      limits[nr++] = 0;
      limits[nr++] = 0;
      const BmRoundInfo bi = {chunk, n, pl.bm_cap, pl.bm_qb, pl.defer_units};
      if (stage_thr_out) {
        // staged search (vaqhip_search_begin_device): the limited rounds now; the thresholds they leave go
        // to the caller, who exchanges them with the other shards; vaqhip_search_finish_device goes on
        int r_split = 0;
        while (r_split < nr && limits[r_split] > 0) r_split++;
        int rc = bm_run_rounds(ix, bp, limits, 0, r_split, nr, bi, st);
        if (rc) return rc;
        // (no limited round in this plan: the 64-bit words still have to be made from g_thr)
        HIP_TRY(vaq::launch_bm_thresholds(bp, nullptr, stage_thr_out, r_split == 0 ? 1 : 0, st));
        StagedState &ss = ix->staged;
        ss.open = true;
        ss.bp = bp;
        ss.sp = sp;
        ss.bi.chunk = bi.chunk; ss.bi.n = bi.n; ss.bi.cap = bi.cap; ss.bi.qb = bi.qb; ss.bi.units = bi.units;
        ss.k = k;
        ss.defer_cap = defer_cap;
        ss.nr = nr;
        ss.r_next = r_split;
        for (int r = 0; r < 4; r++) ss.limits[r] = limits[r];
        ss.labels = d_labels;
        ss.dist = d_dist;
        ix->last = tm;
        return ws_release(ix, st);
      }
      int rc = bm_run_rounds(ix, bp, limits, 0, nr, nr, bi, st);
      if (rc) return rc;
    }
    if (defer || bm) {
      int rc = bm_fallback(ix, sp, defer_cap, k, d_labels + (size_t)q0 * k, d_dist + (size_t)q0 * k, st);
      if (rc) return rc;
    }
#ifdef VAQ_PHASES
    if (pl.bf) {
      unsigned long long h[11];
      HIP_TRY(hipMemcpyAsync(h, sp.stats, sizeof h, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      const double w = (double)std::max<unsigned long long>(h[10], 1);  // reporting waves
      std::fprintf(stderr,
                   "[VAQ_PHASES] cycles per wave: tables %.0f keys %.0f bootstrap %.0f round prep %.0f scan %.0f "
                   "round end %.0f tail %.0f final (wave 0 works): read count %.0f cut %.0f order + write %.0f | sum %.0f\n",
                   h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w, h[8] / w, h[9] / w, h[7] / w,
                   (h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + h[6] + h[7] + h[8] + h[9]) / w);
    }
#endif
#ifdef VAQ_STATS
    {
      unsigned long long h[24];
      HIP_TRY(hipMemcpyAsync(h, sp.stats, sizeof h, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      const double w = (double)grid * sp.nwaves;
      std::fprintf(stderr,
                   "[VAQ_STATS] per wave: steps %.1f alive_A %.1f alive_A2 %.1f drains %.2f admits %.2f folds %.2f "
                   "buckets tested %.1f visited %.1f | cycles total %.0f setup %.0f stepload-wait %.0f admit %.0f "
                   "(fold %.0f lock-wait %.0f) drain %.0f | best-first: bootstrap %.0f round prep %.0f final (wave 0) %.0f setup up to the tables %.0f\n",
                   h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[9] / w, h[10] / w, h[6] / w, h[11] / w,
                   h[12] / w, h[7] / w, h[13] / w, h[14] / w, h[8] / w, h[15] / w, h[16] / w, (double)h[17] / grid, h[18] / w);
    }
#endif
    if (timing) HIP_TRY(hipEventRecord(ev[4], st));
    const int lists = ix->N > 0 ? pl.n_slices : 0;
    if (!direct)
      // after a seeded scan most lists are empty: let the first merge level gather by the counts
      HIP_TRY(vaq::launch_merge(sp.part_d, sp.part_id, (pl.seed_slices > 0 || pl.ordered) ? sp.part_cnt : nullptr,
                                lists, k,
                                (int64_t)pl.n_slices * k, n, k,
                                ix->id_base, 0, d_labels + (size_t)q0 * k, d_dist + (size_t)q0 * k,
                                nullptr, ix->w_ms_d.as<float>(), ix->w_ms_id.as<int>(), st));
    if (timing) HIP_TRY(hipEventRecord(ev[5], st));
    tm.seed_slices = pl.seed_slices;
    tm.early_abandon = pl.ea;
    tm.best_first = pl.bf ? 1 : 0;
    tm.deferred_queries = (defer || bm) ? 0 : -1;  // (bucket-major: queries whose candidate buffer overflowed)
    tm.bucket_major = bm ? 1 : 0;
    tm.queries_per_pass = pl.qb;
    tm.slices = pl.n_slices;
    tm.workgroups = grid;
    tm.passes = (n + pl.qb - 1) / pl.qb;
    tm.lds_bytes = (int)pl.lds;
  }
  tm.n_searches = 0;
  ix->last = tm;
  if (timing) ix->ev_used++;
  return ws_release(ix, st);
}

// Option "exact_ties": the scan runs with k + 1; queries whose k + 1 smallest distances are distinct
// have a unique answer and are copied out, the others are replayed through the reference's heap in
// original row order (vaq_exact.hip).  One internal launch set (<= QUERY_CHUNK queries) at a time: the
// replay reads that set's lookup tables.
int search_device_locked(vaqhip_index *ix, const float *d_queries, int nq, int k, int projected,
                         int32_t *d_labels, float *d_dist, hipStream_t st) {
  const bool exact = ix->opt_exact && ix->ti_T == 0 && !ix->seq && nq > 0 && k > 0 && k < VAQHIP_MAX_K && ix->N >= 0 &&
                     d_queries && d_labels && d_dist;
  if (!exact) return search_core(ix, d_queries, nq, k, projected, d_labels, d_dist, st);
  const int chunk = std::min(nq, QUERY_CHUNK);
  HIP_TRY(ix->w_ex_labels.ensure((size_t)chunk * (k + 1) * sizeof(int32_t)));
  HIP_TRY(ix->w_ex_dist.ensure((size_t)chunk * (k + 1) * sizeof(float)));
  HIP_TRY(ix->w_ex_list.ensure((size_t)chunk * sizeof(int) + 16));
  if (ix->N > 0 && !ix->inv_valid) {
    HIP_TRY(ix->d_inv.ensure((size_t)ix->N * sizeof(uint32_t)));
    HIP_TRY(ix->d_rowbucket.ensure((size_t)ix->N * sizeof(unsigned short)));
    HIP_TRY(vaq::launch_inverse_perm(ix->d_perm.as<uint32_t>(), ix->N, ix->d_inv.as<uint32_t>(), ix->d_bstart.as<int>(),
                                     ix->n_buckets, ix->d_rowbucket.as<unsigned short>(), st));
    ix->inv_valid = true;
  }
  for (int q0 = 0; q0 < nq; q0 += chunk) {
    const int n = std::min(chunk, nq - q0);
    int rc = search_core(ix, d_queries + (size_t)q0 * ix->D, n, k + 1, projected, ix->w_ex_labels.as<int32_t>(),
                         ix->w_ex_dist.as<float>(), st);
    if (rc) return rc;
    rc = ws_acquire(ix, st);
    if (rc) return rc;
    HIP_TRY(vaq::launch_exact_ties(ix->d_codes.as<uint32_t>(), ix->layout, ix->M, ix->W, ix->d_sub.as<vaq::SubDesc>(),
                                   ix->d_inv.as<uint32_t>(), ix->N > 0 ? ix->d_rowbucket.as<unsigned short>() : nullptr,
                                   ix->n_buckets, ix->bucket_shift, ix->bucket_t, ix->N, ix->w_lut.as<float>(), ix->lut_floats, n,
                                   k, ix->id_base,
                                   ix->w_ex_labels.as<int32_t>(), ix->w_ex_dist.as<float>(), d_labels + (size_t)q0 * k,
                                   d_dist + (size_t)q0 * k, reinterpret_cast<int *>(ix->w_ex_list.as<unsigned char>() + 16),
                                   ix->w_ex_list.as<unsigned>(), st));
    rc = ws_release(ix, st);
    if (rc) return rc;
  }
  return VAQHIP_OK;
}

// second half of a staged search: take over the exchanged thresholds, the remaining rounds, the fallback
int search_finish_locked(vaqhip_index *ix, const int32_t *d_thr_in, hipStream_t st) {
  StagedState &ss = ix->staged;
  if (!ss.open) return fail(VAQHIP_ESTATE, "no staged search is open on this index");
  {
    int rc = ws_acquire(ix, st);
    if (rc) return rc;
  }
  ss.open = false;
  if (d_thr_in) HIP_TRY(vaq::launch_bm_thresholds(ss.bp, d_thr_in, nullptr, 0, st));
  const BmRoundInfo bi = {ss.bi.chunk, ss.bi.n, ss.bi.cap, ss.bi.qb, ss.bi.units};
  int rc = bm_run_rounds(ix, ss.bp, ss.limits, ss.r_next, ss.nr, ss.nr, bi, st);
  if (rc) return rc;
  rc = bm_fallback(ix, ss.sp, ss.defer_cap, ss.k, ss.labels, ss.dist, st);
  if (rc) return rc;
  return ws_release(ix, st);
}

} // namespace

extern "C" {

const char *vaqhip_last_error(void) { return g_err.c_str(); }
int vaqhip_version(void) { return VAQHIP_VERSION; }

int vaqhip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(VAQHIP_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int vaqhip_index_create(vaqhip_index **out, int D, int M, const int *bits,
                        const float *const *centroids, const float *eig, int device_id) {
  return vaqhip_index_create_ex(out, D, M, bits, centroids, eig, device_id, 0u);
}

int vaqhip_index_create_ex(vaqhip_index **out, int D, int M, const int *bits,
                           const float *const *centroids, const float *eig, int device_id,
                           unsigned flags) {
  if (!out) return fail(VAQHIP_EINVAL, "out is null");
  const bool seq = (flags & VAQHIP_SUM_SEQUENTIAL) != 0;
  if (flags & ~(unsigned)VAQHIP_SUM_SEQUENTIAL) return fail(VAQHIP_EINVAL, "unknown flags 0x%x", flags);
  *out = nullptr;
  if (D <= 0 || M <= 0 || !bits || !centroids) return fail(VAQHIP_EINVAL, "bad D/M/bits/centroids");
  if (M > VAQHIP_MAX_SUBSPACES) return fail(VAQHIP_EUNSUPPORTED, "M=%d > %d", M, VAQHIP_MAX_SUBSPACES);
  if (M % 4 != 0 && !seq)
    return fail(VAQHIP_EINVAL, "M=%d: the reference scan reads 4 codes per step (VAQ.cpp:1741-1746)", M);
  if (D % M != 0) return fail(VAQHIP_EINVAL, "D=%d is not a multiple of M=%d", D, M);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(VAQHIP_ENODEVICE, "no HIP device available (this library has no CPU path)");
  if (device_id < 0 || device_id >= ndev) return fail(VAQHIP_EINVAL, "device_id=%d of %d", device_id, ndev);

  vaqhip_index *ix = new (std::nothrow) vaqhip_index();
  if (!ix) return fail(VAQHIP_ENOMEM, "host allocation");
  struct Cleanup {
    vaqhip_index *p;
    ~Cleanup() { if (p) vaqhip_index_destroy(p); }
  } cleanup{ix};
  ix->D = D;
  ix->M = M;
  ix->L = D / M;
  ix->device = device_id;
  ix->bits.assign(bits, bits + M);
  ix->sub.resize(M);
  int bit_off = 0, lut_off = 0, cent_off = 0, maxb = 0;
  bool all8 = true;
  for (int s = 0; s < M; s++) {
    const int b = bits[s];
    if (b < 1 || b > VAQHIP_MAX_BITS) return fail(VAQHIP_EINVAL, "bits[%d]=%d outside 1..%d", s, b, VAQHIP_MAX_BITS);
    if (!centroids[s]) return fail(VAQHIP_EINVAL, "centroids[%d] is null", s);
    vaq::SubDesc &sd = ix->sub[s];
    sd.ncent = 1 << b;
    sd.bits = b;
    sd.bit_off = bit_off;
    sd.lut_off = lut_off;
    sd.cent_off = cent_off;
    sd.word = bit_off / 32;
    sd.shift = bit_off % 32;
    sd.pad = 0;
    bit_off += b;
    lut_off += sd.ncent;
    cent_off += sd.ncent * ix->L;
    maxb = std::max(maxb, b);
    all8 = all8 && b == 8;
  }
  ix->max_bits = maxb;
  ix->min_bits = *std::min_element(bits, bits + M);
  ix->total_bits = bit_off;
  ix->lut_floats = lut_off;
  ix->W = (bit_off + 31) / 32;
  ix->seq = seq ? 1 : 0;
  ix->layout = (!seq && all8 && (M == 8 || M == 16 || M == 32)) ? vaq::LAYOUT_BYTES : vaq::LAYOUT_BITS;
  if (ix->layout == vaq::LAYOUT_BITS && ix->W > 8)
    return fail(VAQHIP_EUNSUPPORTED, "%d code bits per row; this build packs at most 256", bit_off);
  std::vector<int> first_sub(ix->W + 1, M);
  first_sub[0] = 0;
  for (int w = 1; w <= ix->W; w++) {
    int f = M;
    for (int s = 0; s < M; s++)
      if (ix->sub[s].bit_off >= 32 * w) { f = s; break; }
    first_sub[w] = f;
  }

  DeviceGuard g(device_id);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", device_id);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
    ix->n_cu = prop.multiProcessorCount;
  HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  HIP_TRY(ix->d_cent.ensure((size_t)cent_off * sizeof(float)));
  for (int s = 0; s < M; s++)
    HIP_TRY(hipMemcpy(ix->d_cent.as<float>() + ix->sub[s].cent_off, centroids[s],
                      (size_t)ix->sub[s].ncent * ix->L * sizeof(float), hipMemcpyHostToDevice));
  {
    // the same matrices dimension-major (the reference keeps mCentroidsPerSubsCMajor for its
    // AVX loads, VAQ.cpp:655-660): the LUT build reads them one centroid per lane, coalesced
    std::vector<float> t((size_t)cent_off);
    for (int s = 0; s < M; s++) {
      const int K = ix->sub[s].ncent;
      for (int c = 0; c < K; c++)
        for (int j = 0; j < ix->L; j++)
          t[(size_t)ix->sub[s].cent_off + (size_t)j * K + c] = centroids[s][(size_t)c * ix->L + j];
    }
    HIP_TRY(ix->d_cent_t.ensure((size_t)cent_off * sizeof(float)));
    HIP_TRY(hipMemcpy(ix->d_cent_t.p, t.data(), (size_t)cent_off * sizeof(float), hipMemcpyHostToDevice));
  }
  HIP_TRY(ix->d_sub.ensure(M * sizeof(vaq::SubDesc)));
  HIP_TRY(hipMemcpy(ix->d_sub.p, ix->sub.data(), M * sizeof(vaq::SubDesc), hipMemcpyHostToDevice));
  HIP_TRY(ix->d_first_sub.ensure(first_sub.size() * sizeof(int)));
  HIP_TRY(hipMemcpy(ix->d_first_sub.p, first_sub.data(), first_sub.size() * sizeof(int),
                    hipMemcpyHostToDevice));
  if (eig) {
    HIP_TRY(ix->d_eig.ensure((size_t)D * D * sizeof(float)));
    HIP_TRY(hipMemcpy(ix->d_eig.p, eig, (size_t)D * D * sizeof(float), hipMemcpyHostToDevice));
    ix->has_eig = true;
  }
  cleanup.p = nullptr;
  *out = ix;
  return VAQHIP_OK;
}

void vaqhip_index_destroy(vaqhip_index *ix) {
  if (!ix) return;
  {
    DeviceGuard g(ix->device);
    if (ix->stream) {
      (void)hipStreamSynchronize(ix->stream);
      (void)hipStreamDestroy(ix->stream);
    }
    for (auto &e : ix->ev) (void)hipEventDestroy(e);
    if (ix->ws_event) (void)hipEventDestroy(ix->ws_event);
    for (DevBuf *b : {&ix->d_cent, &ix->d_cent_t, &ix->d_eig, &ix->d_sub, &ix->d_first_sub, &ix->d_codes, &ix->d_perm,
                      &ix->d_bstart, &ix->w_q,
                      &ix->w_qproj, &ix->w_lut, &ix->w_part_d, &ix->w_part_id, &ix->w_part_cnt, &ix->w_labels,
                      &ix->w_dist, &ix->w_stage, &ix->w_lutref, &ix->w_thr, &ix->w_ms_d, &ix->w_ms_id, &ix->w_order, &ix->w_qorder})
      b->release();
  }
  delete ix;
}

// Order the N rows of the device matrix d_u16 (CodebookType layout) -- by first code, or by TI
// cluster when clusters are set -- and pack them.  Synchronises the stream.
static int build_rows(vaqhip_index *ix, const uint16_t *d_u16, int64_t N, hipStream_t st) {
  const int step = vaq::scan_wg_step_rows(ix->layout, ix->M);
  const int64_t padded = std::max<int64_t>(step, ((N + step - 1) / step) * step);
  const int64_t words = vaq::packed_words(padded, ix->M, ix->layout, ix->W);
  HIP_TRY(ix->d_codes.ensure((size_t)words * sizeof(uint32_t)));
  const vaq::SubDesc *dsub = ix->d_sub.as<vaq::SubDesc>();
  int shift = 0, bt = 0, K0 = 1, fine = 0;
  if (ix->ti_T > 0) {
    K0 = ix->ti_T;
  } else {
    // bucket key = the top bits of the first code, continued -- when the whole first code is
    // used up -- by up to 4 top bits of the second: as many key bits as keep ~900 rows per
    // bucket on average, at most 10 (measured on 250M rows x 16 B: 10 bits beat 8, 11 and 12
    // for 2, 32 and 256 queries; the option accepts up to 12); a tenth bit from the second code
    // wants ~1900 rows per bucket (8 B rows, 10 k queries, best-first form: 1M rows 0.86 / 0.75 /
    // 0.83 ms with 8 / 9 / 10 bits, 2M 1.41 / 1.11 / 1.09, 8M 4.98 / 3.36 / 2.96), one from the
    // first code does not (12-bit first code, 1M rows: 1.54 ms with 10 bits, 1.83 with 9)
    int want = 4;
    while (want < 10 && ((int64_t)2 << want) * BUCKET_MIN_ROWS <= std::max<int64_t>(N, 1)) want++;
    if (ix->opt_bucket_bits > 0) want = ix->opt_bucket_bits;
    const int kb = std::min(want, ix->bits[0]);
    shift = ix->bits[0] - kb;
    // Continuing into the second code: always where the best-first form will scan the rows (its
    // per-bucket bookkeeping is a key in LDS), else only on large databases -- 250M rows, 32
    // queries: 4.0 vs 5.3 ms, but 1M rows, 10 bits, shared-stream form: 2.0 vs 1.45 ms; an
    // explicit "bucket_bits" option is obeyed as given
    if (shift == 0 && ix->M > 1) {
      int want_c = want;  // (a tenth bit taken from the SECOND code wants more rows per bucket)
      if (want_c == 10 && kb < 10 && ix->opt_bucket_bits <= 0 && N < (int64_t)1024 * BUCKET_MIN_ROWS_10) want_c = 9;
      const int cont = std::min(std::min(want_c - kb, 4), ix->bits[1]);
      const bool bf_form = ix->opt_bf && cont > 0 &&
                           vaq::scan_bf_supported(ix->layout, ix->M, 1, vaq::EA_QUEUE, 1 << (kb + cont), ix->seq);
      if (N >= ((int64_t)1 << 24) || ix->opt_bucket_bits > 0 || bf_form) bt = std::max(cont, 0);
    }
    K0 = 1 << (kb + bt);
  }
  HIP_TRY(ix->d_bstart.ensure((size_t)(K0 + 1) * sizeof(int)));
  HIP_TRY(ix->d_perm.ensure(std::max<size_t>((size_t)N, 1) * sizeof(uint32_t)));
  if (ix->ti_T > 0) HIP_TRY(ix->d_ti_xcc.ensure(std::max<size_t>((size_t)padded, 1) * sizeof(float)));
  std::vector<int> bstart((size_t)K0 + 1, (int)N);
  if (N == 0) {
    HIP_TRY(hipMemsetAsync(ix->d_codes.p, 0, (size_t)words * sizeof(uint32_t), st));
  } else {
    if (ix->ti_T > 0)
      HIP_TRY(vaq::ti_group_rows(d_u16, N, ix->M, ix->L, ix->ti_seg, dsub, ix->d_cent.as<float>(),
                                 ix->d_ti_clusters.as<float>(), ix->ti_T, ix->d_perm.as<uint32_t>(),
                                 ix->d_bstart.as<int>(), ix->d_ti_xcc.as<float>(), st));
    else {
      // (byte codes keyed by the whole first code: order each bucket by the rest of the second code)
      fine = (ix->layout == vaq::LAYOUT_BYTES && shift == 0 && ix->M > 1 && ix->opt_sub_order) ? ix->bits[1] - bt : 0;
      if (fine > 0) HIP_TRY(ix->d_substart.ensure((((size_t)K0 << fine) + 1) * sizeof(int)));
      HIP_TRY(vaq::sort_by_first_code(d_u16, N, ix->M, ix->bits[0], shift, ix->M > 1 ? ix->bits[1] : 0, bt,
                                      ix->d_perm.as<uint32_t>(), ix->d_bstart.as<int>(), st, fine,
                                      fine > 0 ? ix->d_substart.as<int>() : nullptr));
      if (fine > 0) {
        std::vector<int> ss(((size_t)K0 << fine) + 1);
        HIP_TRY(hipMemcpy(ss.data(), ix->d_substart.p, ss.size() * sizeof(int), hipMemcpyDeviceToHost));
        ss[ss.size() - 1] = (int)N;
        for (int64_t f = (int64_t)ss.size() - 2; f >= 0; f--)
          if (ss[f] < 0) ss[f] = ss[f + 1];  // runs that do not occur: empty
        HIP_TRY(hipMemcpy(ix->d_substart.p, ss.data(), ss.size() * sizeof(int), hipMemcpyHostToDevice));
      }
    }
    HIP_TRY(hipMemcpy(bstart.data(), ix->d_bstart.p, (size_t)(K0 + 1) * sizeof(int), hipMemcpyDeviceToHost));
    bstart[K0] = (int)N;
    for (int b = K0 - 1; b >= 0; b--)
      if (bstart[b] < 0) bstart[b] = bstart[b + 1];  // codes / clusters that do not occur: empty
    HIP_TRY(vaq::launch_pack_codes(d_u16, 0, N, padded, ix->M, ix->layout, ix->W, dsub,
                                   ix->d_perm.as<uint32_t>(), ix->d_codes.as<uint32_t>(), st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  HIP_TRY(hipMemcpy(ix->d_bstart.p, bstart.data(), (size_t)(K0 + 1) * sizeof(int), hipMemcpyHostToDevice));
  ix->N = N;
  ix->N_keyed = N;
  ix->bucket_shift = shift;
  ix->bucket_t = bt;
  ix->n_buckets = K0;
  ix->sub_fine = N > 0 ? fine : 0;
  ix->inv_valid = false;
  return VAQHIP_OK;
}

static int set_codes_common(vaqhip_index *ix, const uint16_t *codes, bool on_device, int64_t N,
                            int64_t id_base, hipStream_t st) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (N < 0 || (N > 0 && !codes)) return fail(VAQHIP_EINVAL, "bad codes/N");
  if (id_base < 0) return fail(VAQHIP_EINVAL, "id_base < 0");
  if (N > 0x7fffffffLL - 1 || id_base + N > 0x7fffffffLL)
    return fail(VAQHIP_ERANGE, "labels are 32-bit ints (utils/Types.hpp:100): id_base+N = %lld",
                (long long)(id_base + N));
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  // all rows must be resident to sort them: stage a host matrix on the device first
  DevBuf staged;
  const uint16_t *d_u16 = codes;
  if (!on_device && N > 0) {
    HIP_TRY(staged.ensure((size_t)N * ix->M * sizeof(uint16_t)));
    for (int64_t r = 0; r < N; r += UPLOAD_CHUNK_ROWS) {
      const int64_t e = std::min(N, r + UPLOAD_CHUNK_ROWS);
      HIP_TRY(hipMemcpyAsync(staged.as<uint16_t>() + r * ix->M, codes + r * ix->M,
                             (size_t)(e - r) * ix->M * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    }
    d_u16 = staged.as<uint16_t>();
  }
  // (a search enqueued on another stream may still be scanning the rows this call rewrites)
  if (int rc = ws_acquire(ix, st)) return rc;
  int rc = build_rows(ix, d_u16, N, st);  // synchronises: `staged` is freed on return
  if (rc) return rc;
  ix->id_base = id_base;
  return ws_release(ix, st);
}

// append to a bucketed (non-TI) index: sort and pack the NEW rows only, then merge them into the
// existing order bucket by bucket (launch_merge_rows).  O(N) bytes are copied once -- the packed
// rows and their labels -- but nothing is unpacked and nothing is re-sorted; temporaries are
// O(n_new) plus the new packed buffer.
static int append_rows_bucketed(vaqhip_index *ix, const uint16_t *d_new, int64_t n_new, hipStream_t st) {
  const int64_t n_old = ix->N, N = n_old + n_new;
  // (rows ordered inside the buckets too: merge run by run, so that the order survives -- the runs
  //  are the buckets of a finer key, ix->d_substart their starts)
  const int fine = ix->sub_fine;
  const int KB = ix->n_buckets;
  const int K0 = KB << fine;
  const int step = vaq::scan_wg_step_rows(ix->layout, ix->M);
  const vaq::SubDesc *dsub = ix->d_sub.as<vaq::SubDesc>();
  // the new rows in bucketed order among themselves
  DevBuf new_perm, new_start, new_bstart, new_codes, out_codes, out_perm;
  HIP_TRY(new_perm.ensure((size_t)n_new * sizeof(uint32_t)));
  HIP_TRY(new_start.ensure((size_t)(K0 + 1) * sizeof(int)));
  HIP_TRY(new_bstart.ensure((size_t)(KB + 1) * sizeof(int)));
  HIP_TRY(vaq::sort_by_first_code(d_new, n_new, ix->M, ix->bits[0], ix->bucket_shift, ix->M > 1 ? ix->bits[1] : 0,
                                  ix->bucket_t, new_perm.as<uint32_t>(), fine > 0 ? new_bstart.as<int>() : new_start.as<int>(), st,
                                  fine, fine > 0 ? new_start.as<int>() : nullptr));
  std::vector<int> ns((size_t)K0 + 1), os((size_t)K0 + 1), ts((size_t)K0 + 1);
  HIP_TRY(hipMemcpy(ns.data(), new_start.p, (size_t)(K0 + 1) * sizeof(int), hipMemcpyDeviceToHost));
  ns[K0] = (int)n_new;
  for (int b = K0 - 1; b >= 0; b--)
    if (ns[b] < 0) ns[b] = ns[b + 1];
  HIP_TRY(hipMemcpy(new_start.p, ns.data(), (size_t)(K0 + 1) * sizeof(int), hipMemcpyHostToDevice));
  const int *d_old_start = fine > 0 ? ix->d_substart.as<int>() : ix->d_bstart.as<int>();
  HIP_TRY(hipMemcpy(os.data(), d_old_start, (size_t)(K0 + 1) * sizeof(int), hipMemcpyDeviceToHost));
  const int64_t new_padded = std::max<int64_t>(step, ((n_new + step - 1) / step) * step);
  HIP_TRY(new_codes.ensure((size_t)vaq::packed_words(new_padded, ix->M, ix->layout, ix->W) * sizeof(uint32_t)));
  HIP_TRY(vaq::launch_pack_codes(d_new, 0, n_new, new_padded, ix->M, ix->layout, ix->W, dsub, new_perm.as<uint32_t>(),
                                 new_codes.as<uint32_t>(), st));
  // the merged buffers
  const int64_t padded = std::max<int64_t>(step, ((N + step - 1) / step) * step);
  const int64_t words = vaq::packed_words(padded, ix->M, ix->layout, ix->W);
  HIP_TRY(out_codes.ensure((size_t)words * sizeof(uint32_t)));
  HIP_TRY(out_perm.ensure((size_t)N * sizeof(uint32_t)));
  HIP_TRY(hipMemsetAsync(out_codes.p, 0, (size_t)words * sizeof(uint32_t), st));  // (the padding rows must be zero)
  HIP_TRY(vaq::launch_merge_rows(ix->d_codes.as<uint32_t>(), ix->d_perm.as<uint32_t>(), d_old_start,
                                 new_codes.as<uint32_t>(), new_perm.as<uint32_t>(), new_start.as<int>(), K0, n_old, N,
                                 ix->M, ix->layout, ix->W, out_codes.as<uint32_t>(), out_perm.as<uint32_t>(), st));
  for (int b = 0; b <= K0; b++) ts[b] = os[b] + ns[b];
  HIP_TRY(hipStreamSynchronize(st));
  std::swap(ix->d_codes.p, out_codes.p);
  std::swap(ix->d_codes.cap, out_codes.cap);
  std::swap(ix->d_perm.p, out_perm.p);
  std::swap(ix->d_perm.cap, out_perm.cap);
  if (fine > 0) {
    HIP_TRY(hipMemcpy(ix->d_substart.p, ts.data(), (size_t)(K0 + 1) * sizeof(int), hipMemcpyHostToDevice));
    std::vector<int> tb((size_t)KB + 1);
    for (int b = 0; b <= KB; b++) tb[b] = ts[(size_t)b << fine];
    HIP_TRY(hipMemcpy(ix->d_bstart.p, tb.data(), (size_t)(KB + 1) * sizeof(int), hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy(ix->d_bstart.p, ts.data(), (size_t)(K0 + 1) * sizeof(int), hipMemcpyHostToDevice));
  }
  ix->N = N;
  ix->inv_valid = false;
  return VAQHIP_OK;
}

// append: a bucketed index merges the new rows in (above); a TI-grouped index (rows ordered by
// cluster and distance to the centre) and an empty index are rebuilt: recover the rows already
// packed (original order), put the new ones behind them, regroup everything
static int add_codes_common(vaqhip_index *ix, const uint16_t *codes, bool on_device, int64_t n_new,
                            hipStream_t st) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (n_new < 0 || (n_new > 0 && !codes)) return fail(VAQHIP_EINVAL, "bad codes/N");
  std::lock_guard<std::mutex> lk(ix->mu);
  const int64_t n_old = ix->N < 0 ? 0 : ix->N;
  const int64_t N = n_old + n_new;
  if (N > 0x7fffffffLL - 1 || ix->id_base + N > 0x7fffffffLL)
    return fail(VAQHIP_ERANGE, "labels are 32-bit ints (utils/Types.hpp:100): id_base+N = %lld",
                (long long)(ix->id_base + N));
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  if (n_new == 0 && ix->N >= 0) return VAQHIP_OK;
  {
    int rc = ws_acquire(ix, st);  // (a search on another stream may still be reading the codes)
    if (rc) return rc;
  }
  if (ix->ti_T == 0 && n_old > 0 && n_new > 0 && N < 4 * std::max<int64_t>(ix->N_keyed, 4096)) {
    DevBuf staged;
    const uint16_t *d_new = codes;
    if (!on_device) {
      HIP_TRY(staged.ensure((size_t)n_new * ix->M * sizeof(uint16_t)));
      HIP_TRY(hipMemcpyAsync(staged.p, codes, (size_t)n_new * ix->M * sizeof(uint16_t), hipMemcpyHostToDevice, st));
      d_new = staged.as<uint16_t>();
    }
    return append_rows_bucketed(ix, d_new, n_new, st);  // synchronises
  }
  DevBuf rows;
  HIP_TRY(rows.ensure(std::max<size_t>((size_t)N * ix->M * sizeof(uint16_t), 16)));
  if (n_old > 0)
    HIP_TRY(vaq::launch_unpack_codes(ix->d_codes.as<uint32_t>(), n_old, ix->M, ix->layout, ix->W,
                                     ix->d_sub.as<vaq::SubDesc>(), ix->d_perm.as<uint32_t>(),
                                     rows.as<uint16_t>(), st));
  if (n_new > 0)
    HIP_TRY(hipMemcpyAsync(rows.as<uint16_t>() + n_old * ix->M, codes, (size_t)n_new * ix->M * sizeof(uint16_t),
                           on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  return build_rows(ix, rows.as<uint16_t>(), N, st);  // synchronises
}

int vaqhip_index_add_codes_u16(vaqhip_index *ix, const uint16_t *codes, int64_t n_new) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  return add_codes_common(ix, codes, false, n_new, ix->stream);
}

int vaqhip_index_add_codes_u16_device(vaqhip_index *ix, const uint16_t *d_codes, int64_t n_new, void *stream) {
  return add_codes_common(ix, d_codes, true, n_new, static_cast<hipStream_t>(stream));
}

int vaqhip_index_set_codes_u16(vaqhip_index *ix, const uint16_t *codes, int64_t N, int64_t id_base) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  int rc = set_codes_common(ix, codes, false, N, id_base, ix->stream);
  if (rc) return rc;
  DeviceGuard g(ix->device);
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return VAQHIP_OK;
}

int vaqhip_index_set_codes_u16_device(vaqhip_index *ix, const uint16_t *d_codes, int64_t N,
                                      int64_t id_base, void *stream) {
  return set_codes_common(ix, d_codes, true, N, id_base, static_cast<hipStream_t>(stream));
}

int vaqhip_search_staged_supported(vaqhip_index *ix, int nq, int k) {
  if (!ix || nq <= 0 || k <= 0 || k > VAQHIP_MAX_K) return 0;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->N <= 0 || ix->ti_T > 0 || nq > QUERY_CHUNK || ix->opt_exact) return 0;
  Plan pl;
  if (make_plan(ix, nq, k, &pl)) return 0;
  return (pl.bm && pl.bf && pl.n_slices == 1) ? 1 : 0;
}

int vaqhip_search_begin_device(vaqhip_index *ix, const float *d_queries, int nq, int k, int projected,
                               int32_t *d_labels, float *d_distances, int32_t *d_thresholds_out, void *stream) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (!d_thresholds_out) return fail(VAQHIP_EINVAL, "null pointer");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  if (ix->opt_exact) return fail(VAQHIP_EUNSUPPORTED, "exact_ties is a property of ONE index; shards merge by (distance, label)");
  return search_core(ix, d_queries, nq, k, projected, d_labels, d_distances, static_cast<hipStream_t>(stream),
                     d_thresholds_out);
}

int vaqhip_search_finish_device(vaqhip_index *ix, const int32_t *d_thresholds_in, void *stream) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  return search_finish_locked(ix, d_thresholds_in, static_cast<hipStream_t>(stream));
}

int vaqhip_search_device(vaqhip_index *ix, const float *d_queries, int nq, int k, int projected,
                         int32_t *d_labels, float *d_dist, void *stream) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  return search_device_locked(ix, d_queries, nq, k, projected, d_labels, d_dist,
                              static_cast<hipStream_t>(stream));
}

static int search_host(vaqhip_index *ix, const float *queries, int nq, int k, int projected,
                       int32_t *labels, float *distances) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (nq < 0 || k <= 0) return fail(VAQHIP_EINVAL, "nq=%d k=%d", nq, k);
  if (nq == 0) return VAQHIP_OK;
  if (!queries || !labels || !distances) return fail(VAQHIP_EINVAL, "null pointer");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  const size_t qbytes = (size_t)nq * ix->D * sizeof(float);
  const size_t rbytes = (size_t)nq * k * sizeof(float);
  HIP_TRY(ix->w_q.ensure(qbytes));
  HIP_TRY(ix->w_labels.ensure(rbytes));
  HIP_TRY(ix->w_dist.ensure(rbytes));
  {
    int rc = ws_acquire(ix, ix->stream);
    if (rc) return rc;
  }
  HIP_TRY(hipMemcpyAsync(ix->w_q.p, queries, qbytes, hipMemcpyHostToDevice, ix->stream));
  int rc = search_device_locked(ix, ix->w_q.as<float>(), nq, k, projected, ix->w_labels.as<int32_t>(),
                                ix->w_dist.as<float>(), ix->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(labels, ix->w_labels.p, rbytes, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipMemcpyAsync(distances, ix->w_dist.p, rbytes, hipMemcpyDeviceToHost, ix->stream));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return VAQHIP_OK;
}

int vaqhip_search(vaqhip_index *ix, const float *queries, int nq, int k, int32_t *labels,
                  float *distances) {
  return search_host(ix, queries, nq, k, 0, labels, distances);
}

int vaqhip_search_projected(vaqhip_index *ix, const float *qproj, int nq, int k, int32_t *labels,
                            float *distances) {
  return search_host(ix, qproj, nq, k, 1, labels, distances);
}

int vaqhip_project(vaqhip_index *ix, const float *X, int64_t n, float *out) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (n < 0 || (n > 0 && (!X || !out))) return fail(VAQHIP_EINVAL, "bad arguments");
  if (n == 0) return VAQHIP_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  if (!ix->has_eig) {
    std::memcpy(out, X, (size_t)n * ix->D * sizeof(float));
    return VAQHIP_OK;
  }
  const int64_t chunk = std::min<int64_t>(n, 1 << 20);
  HIP_TRY(ix->w_q.ensure((size_t)chunk * ix->D * sizeof(float)));
  HIP_TRY(ix->w_qproj.ensure((size_t)chunk * ix->D * sizeof(float)));
  {
    int rc = ws_acquire(ix, ix->stream);
    if (rc) return rc;
  }
  for (int64_t r = 0; r < n; r += chunk) {
    const int64_t m = std::min(chunk, n - r);
    const size_t bytes = (size_t)m * ix->D * sizeof(float);
    HIP_TRY(hipMemcpyAsync(ix->w_q.p, X + r * ix->D, bytes, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(vaq::launch_project(ix->w_q.as<float>(), m, ix->D, ix->d_eig.as<float>(),
                                ix->w_qproj.as<float>(), ix->stream));
    HIP_TRY(hipMemcpyAsync(out + r * ix->D, ix->w_qproj.p, bytes, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
  }
  return VAQHIP_OK;
}

int vaqhip_build_lut(vaqhip_index *ix, const float *queries, int nq, int projected, float *lut_out) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (nq < 0 || (nq > 0 && (!queries || !lut_out))) return fail(VAQHIP_EINVAL, "bad arguments");
  if (nq == 0) return VAQHIP_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  const int ksub = 1 << ix->max_bits;
  const size_t per_q = (size_t)ix->M * ksub;
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)nq, ((size_t)256 << 20) / (per_q * 4)));
  HIP_TRY(ix->w_q.ensure((size_t)chunk * ix->D * sizeof(float)));
  HIP_TRY(ix->w_qproj.ensure((size_t)chunk * ix->D * sizeof(float)));
  HIP_TRY(ix->w_lut.ensure((size_t)chunk * ix->lut_floats * sizeof(float)));
  HIP_TRY(ix->w_lutref.ensure((size_t)chunk * per_q * sizeof(float)));
  hipStream_t st = ix->stream;
  {
    int rc = ws_acquire(ix, st);
    if (rc) return rc;
  }
  for (int q0 = 0; q0 < nq; q0 += chunk) {
    const int n = std::min(chunk, nq - q0);
    HIP_TRY(hipMemcpyAsync(ix->w_q.p, queries + (size_t)q0 * ix->D, (size_t)n * ix->D * sizeof(float),
                           hipMemcpyHostToDevice, st));
    const float *qp = ix->w_q.as<float>();
    if (!projected && ix->has_eig) {
      HIP_TRY(vaq::launch_project(qp, n, ix->D, ix->d_eig.as<float>(), ix->w_qproj.as<float>(), st));
      qp = ix->w_qproj.as<float>();
    }
    HIP_TRY(vaq::launch_lut_build(qp, n, ix->D, ix->M, ix->L, ix->d_sub.as<vaq::SubDesc>(),
                                  ix->d_cent_t.as<float>(), ix->lut_floats, 1 << ix->max_bits, ix->w_lut.as<float>(), st,
                                  1 << ix->min_bits));
    HIP_TRY(vaq::launch_lut_expand(ix->w_lut.as<float>(), n, ix->M, ix->d_sub.as<vaq::SubDesc>(),
                                   ix->lut_floats, ksub, ix->w_lutref.as<float>(), st));
    HIP_TRY(hipMemcpyAsync(lut_out + (size_t)q0 * per_q, ix->w_lutref.p, (size_t)n * per_q * sizeof(float),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return VAQHIP_OK;
}

int vaqhip_merge_topk_device(int device_id, const float *d_dist_lists, const int32_t *d_label_lists,
                             int n_lists, int nq, int k, int32_t *d_labels_out, float *d_dist_out,
                             void *stream) {
  return vaqhip_merge_topk_strided_device(device_id, d_dist_lists, d_label_lists, n_lists,
                                          (int64_t)nq * k, k, nq, k, d_labels_out, d_dist_out, stream);
}

int vaqhip_merge_topk_strided_device(int device_id, const float *d_dist_lists,
                                     const int32_t *d_label_lists, int n_lists, int64_t list_stride,
                                     int64_t query_stride, int nq, int k, int32_t *d_labels_out,
                                     float *d_dist_out, void *stream) {
  if (n_lists < 0 || nq < 0 || k <= 0) return fail(VAQHIP_EINVAL, "bad sizes");
  if (k > VAQHIP_MAX_K) return fail(VAQHIP_EUNSUPPORTED, "k=%d > %d", k, VAQHIP_MAX_K);
  if (nq == 0) return VAQHIP_OK;
  if ((n_lists > 0 && (!d_dist_lists || !d_label_lists)) || !d_labels_out || !d_dist_out)
    return fail(VAQHIP_EINVAL, "null pointer");
  DeviceGuard g(device_id);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", device_id);
  if (n_lists > 16) return fail(VAQHIP_EUNSUPPORTED, "at most 16 lists per merge");
  if (list_stride < 0 || query_stride < 0) return fail(VAQHIP_EINVAL, "negative stride");
  HIP_TRY(vaq::launch_merge(d_dist_lists, d_label_lists, nullptr, n_lists, list_stride, query_stride, nq, k, 0, 1,
                            d_labels_out, d_dist_out, nullptr, nullptr, nullptr,
                            static_cast<hipStream_t>(stream)));
  return VAQHIP_OK;
}

// core of vaqhip_encode*: caller holds ix->mu and has the device current
static int encode_device_locked(vaqhip_index *ix, const float *d_X, int64_t n, int projected, uint16_t *d_codes,
                                hipStream_t st) {
  // (BitVecEngine::queryLUT projects with checking, BitVecEngine.hpp:1226: non-finite coordinates -> 0,
  //  which needs a pass over the queries even without a rotation)
  const bool do_project = !projected && (ix->has_eig || ix->seq);
  const int64_t chunk = std::min<int64_t>(n, 1 << 20);
  if (do_project) HIP_TRY(ix->w_qproj.ensure((size_t)chunk * ix->D * sizeof(float)));
  {
    int rc = ws_acquire(ix, st);
    if (rc) return rc;
  }
  for (int64_t r = 0; r < n; r += chunk) {
    const int64_t m = std::min(chunk, n - r);
    const float *xp = d_X + r * ix->D;
    if (do_project) {
      HIP_TRY(vaq::launch_project(xp, m, ix->D, ix->d_eig.as<float>(), ix->w_qproj.as<float>(), st));
      xp = ix->w_qproj.as<float>();
    }
    HIP_TRY(vaq::launch_encode(xp, m, ix->D, ix->M, ix->L, ix->d_sub.as<vaq::SubDesc>(),
                               ix->d_cent.as<float>(), d_codes + r * ix->M, st));
  }
  return ws_release(ix, st);
}

int vaqhip_encode_device(vaqhip_index *ix, const float *d_X, int64_t n, int projected,
                         uint16_t *d_codes, void *stream) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (n < 0 || (n > 0 && (!d_X || !d_codes))) return fail(VAQHIP_EINVAL, "bad arguments");
  if (n == 0) return VAQHIP_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  return encode_device_locked(ix, d_X, n, projected, d_codes, static_cast<hipStream_t>(stream));
}

int vaqhip_encode(vaqhip_index *ix, const float *X, int64_t n, int projected, uint16_t *codes) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (n < 0 || (n > 0 && (!X || !codes))) return fail(VAQHIP_EINVAL, "bad arguments");
  if (n == 0) return VAQHIP_OK;
  const int64_t chunk = std::min<int64_t>(n, 1 << 20);
  // the staging buffers (w_q, w_stage) are the index's: hold its lock across upload, encode and
  // download, as search_host does
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  HIP_TRY(ix->w_q.ensure((size_t)chunk * ix->D * sizeof(float)));
  HIP_TRY(ix->w_stage.ensure((size_t)chunk * ix->M * sizeof(uint16_t)));
  for (int64_t r = 0; r < n; r += chunk) {
    const int64_t m = std::min(chunk, n - r);
    {
      int rc = ws_acquire(ix, ix->stream);  // (w_q may still be read by a search on another stream)
      if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(ix->w_q.p, X + r * ix->D, (size_t)m * ix->D * sizeof(float), hipMemcpyHostToDevice,
                           ix->stream));
    int rc = encode_device_locked(ix, ix->w_q.as<float>(), m, projected, ix->w_stage.as<uint16_t>(), ix->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(codes + r * ix->M, ix->w_stage.p, (size_t)m * ix->M * sizeof(uint16_t),
                           hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
  }
  return VAQHIP_OK;
}

int vaqhip_refine_device(int device_id, const float *d_queries, int nq, int D, const float *d_dataset,
                         const int32_t *d_labels_in, int R, int k, int32_t *d_labels_out,
                         float *d_dist_out, void *stream) {
  if (nq < 0 || D <= 0 || R <= 0 || k <= 0) return fail(VAQHIP_EINVAL, "bad sizes");
  if (R > 2048 || k > R) return fail(VAQHIP_EUNSUPPORTED, "need k <= R <= 2048 (R=%d k=%d)", R, k);
  if (nq == 0) return VAQHIP_OK;
  if (!d_queries || !d_dataset || !d_labels_in || !d_labels_out || !d_dist_out)
    return fail(VAQHIP_EINVAL, "null pointer");
  DeviceGuard g(device_id);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", device_id);
  HIP_TRY(vaq::launch_refine(d_queries, nq, D, d_dataset, nullptr, d_labels_in, R, k, d_labels_out,
                             d_dist_out, static_cast<hipStream_t>(stream)));
  return VAQHIP_OK;
}

int vaqhip_refine(int device_id, const float *queries, int nq, int D, const float *dataset, int64_t N,
                  const int32_t *labels_in, int R, int k, int32_t *labels_out, float *dist_out) {
  if (nq < 0 || D <= 0 || R <= 0 || k <= 0 || N < 0) return fail(VAQHIP_EINVAL, "bad sizes");
  if (R > 2048 || k > R) return fail(VAQHIP_EUNSUPPORTED, "need k <= R <= 2048 (R=%d k=%d)", R, k);
  if (nq == 0) return VAQHIP_OK;
  if (!queries || !dataset || !labels_in || !labels_out || !dist_out) return fail(VAQHIP_EINVAL, "null pointer");
  DeviceGuard g(device_id);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed (no CPU path)", device_id);
  // gather the candidate rows on the host, re-rank on the GPU, in chunks of queries
  const size_t per_q = (size_t)R * D;
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)nq, ((size_t)256 << 20) / (per_q * 4)));
  DevBuf d_q, d_rows, d_lab, d_ol, d_od;
  HIP_TRY(d_q.ensure((size_t)chunk * D * 4));
  HIP_TRY(d_rows.ensure((size_t)chunk * per_q * 4));
  HIP_TRY(d_lab.ensure((size_t)chunk * R * 4));
  HIP_TRY(d_ol.ensure((size_t)chunk * k * 4));
  HIP_TRY(d_od.ensure((size_t)chunk * k * 4));
  std::vector<float> rows((size_t)chunk * per_q);
  for (int q0 = 0; q0 < nq; q0 += chunk) {
    const int n = std::min(chunk, nq - q0);
    for (int q = 0; q < n; q++)
      for (int i = 0; i < R; i++) {
        const int32_t lab = labels_in[(size_t)(q0 + q) * R + i];
        float *dst = rows.data() + ((size_t)q * R + i) * D;
        if (lab >= 0 && (int64_t)lab < N) std::memcpy(dst, dataset + (size_t)lab * D, (size_t)D * 4);
        else if (lab >= 0) return fail(VAQHIP_EINVAL, "label %d outside the dataset", lab);
        else std::memset(dst, 0, (size_t)D * 4);
      }
    HIP_TRY(hipMemcpy(d_q.p, queries + (size_t)q0 * D, (size_t)n * D * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_rows.p, rows.data(), (size_t)n * per_q * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_lab.p, labels_in + (size_t)q0 * R, (size_t)n * R * 4, hipMemcpyHostToDevice));
    HIP_TRY(vaq::launch_refine(d_q.as<float>(), n, D, nullptr, d_rows.as<float>(), d_lab.as<int32_t>(), R, k,
                               d_ol.as<int32_t>(), d_od.as<float>(), nullptr));
    HIP_TRY(hipMemcpy(labels_out + (size_t)q0 * k, d_ol.p, (size_t)n * k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dist_out + (size_t)q0 * k, d_od.p, (size_t)n * k * 4, hipMemcpyDeviceToHost));
  }
  return VAQHIP_OK;
}

int vaqhip_index_set_ti_clusters(vaqhip_index *ix, const float *clusters, int T, int seg_num) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (T < 0 || (T > 0 && !clusters)) return fail(VAQHIP_EINVAL, "bad clusters/T");
  if (T > VAQHIP_MAX_TI_CLUSTERS)
    return fail(VAQHIP_EUNSUPPORTED, "T=%d > %d clusters", T, VAQHIP_MAX_TI_CLUSTERS);
  if (T > 0 && (seg_num < 1 || seg_num > ix->M))
    return fail(VAQHIP_EINVAL, "seg_num=%d outside 1..%d", seg_num, ix->M);
  if (T > 0 && (int64_t)seg_num * ix->L > 1024)
    return fail(VAQHIP_EUNSUPPORTED, "TI centres of %d dims (> 1024)", seg_num * ix->L);
  if (T > 0 && ix->seq) return fail(VAQHIP_EINVAL, "TI is a VAQ::search method, not a queryLUT one");
  std::lock_guard<std::mutex> lk(ix->mu);
  DeviceGuard g(ix->device);
  if (!g.ok) return fail(VAQHIP_ENODEVICE, "hipSetDevice(%d) failed", ix->device);
  if (T == 0 && ix->ti_T == 0) return VAQHIP_OK;
  hipStream_t st = ix->stream;
  // (a search enqueued on another stream may still be scanning the rows this call regroups)
  if (int rc = ws_acquire(ix, st)) return rc;
  // rows already handed over: recover them in original order, then regroup
  DevBuf rows;
  if (ix->N > 0) {
    HIP_TRY(rows.ensure((size_t)ix->N * ix->M * sizeof(uint16_t)));
    HIP_TRY(vaq::launch_unpack_codes(ix->d_codes.as<uint32_t>(), ix->N, ix->M, ix->layout, ix->W,
                                     ix->d_sub.as<vaq::SubDesc>(), ix->d_perm.as<uint32_t>(),
                                     rows.as<uint16_t>(), st));
  }
  if (T > 0) {
    const size_t bytes = (size_t)T * seg_num * ix->L * sizeof(float);
    HIP_TRY(ix->d_ti_clusters.ensure(bytes));
    HIP_TRY(hipMemcpyAsync(ix->d_ti_clusters.p, clusters, bytes, hipMemcpyHostToDevice, st));
    // dimension-major copy for the per-query plan (one centre per lane, coalesced)
    const int dd = seg_num * ix->L;
    std::vector<float> t((size_t)T * dd);
    for (int c = 0; c < T; c++)
      for (int j = 0; j < dd; j++) t[(size_t)j * T + c] = clusters[(size_t)c * dd + j];
    HIP_TRY(ix->d_ti_clusters_t.ensure(bytes));
    HIP_TRY(hipMemcpy(ix->d_ti_clusters_t.p, t.data(), bytes, hipMemcpyHostToDevice));
  }
  ix->ti_T = T;
  ix->ti_seg = T > 0 ? seg_num : 0;
  if (T > 0) ix->methods |= VAQHIP_METHOD_TI;
  else {
    ix->methods &= ~VAQHIP_METHOD_TI;
    if (!ix->methods) ix->methods = VAQHIP_METHOD_HEAP;
  }
  if (ix->N >= 0) {
    int rc = build_rows(ix, rows.as<uint16_t>(), ix->N, st);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(st));
  return VAQHIP_OK;
}

int vaqhip_index_set_method(vaqhip_index *ix, unsigned methods, float visit) {
  if (!ix) return fail(VAQHIP_EINVAL, "index is null");
  if (methods & ~(VAQHIP_METHOD_EA | VAQHIP_METHOD_TI | VAQHIP_METHOD_HEAP))
    return fail(VAQHIP_EUNSUPPORTED, "method bits 0x%x: only HEAP, EA and TI are on this path", methods);
  if (!(methods & (VAQHIP_METHOD_EA | VAQHIP_METHOD_TI | VAQHIP_METHOD_HEAP)))
    return fail(VAQHIP_EUNSUPPORTED, "no search method selected (SORT is not provided)");
  if (!(visit > 0.0f)) return fail(VAQHIP_EINVAL, "visit must be > 0");
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->methods = methods;
  ix->ti_visit = visit;
  return VAQHIP_OK;
}

int vaqhip_index_info(const vaqhip_index *ix, vaqhip_info *out) {
  if (!ix || !out) return fail(VAQHIP_EINVAL, "null pointer");
  out->D = ix->D;
  out->M = ix->M;
  out->L = ix->L;
  out->max_bits = ix->max_bits;
  out->total_bits = ix->total_bits;
  out->code_bytes = ix->layout == vaq::LAYOUT_BYTES ? ix->M : ix->W * 4;
  out->algo_code_bytes = (ix->total_bits + 7) / 8;
  out->lut_floats = ix->lut_floats;
  out->N = ix->N < 0 ? 0 : ix->N;
  out->id_base = ix->id_base;
  out->device_id = ix->device;
  out->layout = ix->layout;
  out->ti_clusters = ix->ti_T;
  out->ti_segments = ix->ti_seg;
  out->methods = ix->methods;
  out->visit = ix->ti_visit;
  return VAQHIP_OK;
}

int vaqhip_set_option(vaqhip_index *ix, const char *key, int64_t value) {
  if (!ix || !key) return fail(VAQHIP_EINVAL, "null pointer");
  std::lock_guard<std::mutex> lk(ix->mu);
  const std::string k(key);
  if (k == "queries_per_pass") {
    if (value != 0 && value != 1 && value != 2 && value != 4)
      return fail(VAQHIP_EINVAL, "queries_per_pass must be 0, 1, 2 or 4");
    ix->opt_qb = (int)value;
  } else if (k == "slices") {
    if (value < 0 || value > (1 << 20)) return fail(VAQHIP_EINVAL, "slices out of range");
    ix->opt_slices = (int)value;
  } else if (k == "timing") {
    ix->opt_timing = value != 0;
    if (ix->opt_timing) {  // create the event ring now, not inside the first timed search
      DeviceGuard g(ix->device);
      int rc = ensure_events(ix);
      if (rc) return rc;
    }
  } else if (k == "early_abandon") {
    if (value < 0 || value > 3) return fail(VAQHIP_EINVAL, "early_abandon must be 0..3");
    ix->opt_ea = (int)value;
  } else if (k == "ordered_slices") {
    ix->opt_order = value != 0;
  } else if (k == "seed_fraction") {
    if (value < 2 || value > 65536) return fail(VAQHIP_EINVAL, "seed_fraction must be 2..65536");
    ix->opt_seed_frac = (int)value;
  } else if (k == "hot_buckets") {
    if (value < 0 || value > 32) return fail(VAQHIP_EINVAL, "hot_buckets must be 0..32");
    ix->opt_hot = (int)value;
  } else if (k == "bucket_skip") {
    ix->opt_no_skip = value == 0;
  } else if (k == "bucket_bits") {
    if (value < 0 || value > 12) return fail(VAQHIP_EINVAL, "bucket_bits must be 0..12");
    ix->opt_bucket_bits = (int)value;  // takes effect when the codes are (re)set
  } else if (k == "group_queries") {
    if (value < 0 || value > 2) return fail(VAQHIP_EINVAL, "group_queries must be 0, 1 or 2");
    ix->opt_group = (int)value;
  } else if (k == "best_first") {
    ix->opt_bf = value != 0;
  } else if (k == "cost_order") {
    ix->opt_cost_order = value != 0;
  } else if (k == "defer_units") {
    if (value < -1 || value > 1 << 20) return fail(VAQHIP_EINVAL, "defer_units must be -1 (automatic), 0 (off) or a number of work units");
    ix->opt_defer = (int)value;
  } else if (k == "exact_ties") {
    ix->opt_exact = value != 0;
  } else if (k == "bm_boot") {
    if (value < 0 || value > 2) return fail(VAQHIP_EINVAL, "bm_boot must be 0 (never), 1 (automatic) or 2 (always)");
    ix->opt_bm_boot = (int)value;
  } else if (k == "bm_round") {
    if (value < 0 || value > 1024) return fail(VAQHIP_EINVAL, "bm_round must be 0..1024 buckets");
    ix->opt_bm_round = (int)value;
  } else if (k == "bm_runs") {
    ix->opt_bm_sub = value != 0;  // 0: the bucket-major pass ignores the order inside the buckets (every row of a bucket read)
  } else if (k == "sub_order") {
    ix->opt_sub_order = value != 0;  // takes effect when the codes are (re)set
  } else if (k == "bucket_major") {
    if (value < 0 || value > 2) return fail(VAQHIP_EINVAL, "bucket_major must be 0 (off), 1 (automatic) or 2 (whenever a kernel exists)");
    ix->opt_bm = (int)value;
  } else if (k == "bm_candidates") {
    if (value < 0 || value > 7168) return fail(VAQHIP_EINVAL, "bm_candidates must be 0 (default) or 1..7168");
    ix->opt_bm_cap = (int)value;
  } else if (k == "bm_units") {
    if (value < 0 || value > (1 << 20)) return fail(VAQHIP_EINVAL, "bm_units must be 0 (automatic) or a number of work units");
    ix->opt_bm_units = (int)value;
  } else if (k == "bm_queries_per_group") {
    if (value != 0 && value != 2 && value != 4) return fail(VAQHIP_EINVAL, "bm_queries_per_group must be 0, 2 or 4");
    ix->opt_bm_qb = (int)value;
  } else if (k == "bm_waves") {
    if (value != 0 && value != 4 && value != 8 && value != 16) return fail(VAQHIP_EINVAL, "bm_waves must be 0, 4, 8 or 16");
    ix->opt_bm_nwaves = (int)value;
  } else if (k == "seed_thresholds") {
    ix->opt_seed = value != 0;
  } else if (k == "waves_per_workgroup") {
    if (value != 0 && value != 4 && value != 8 && value != 16)
      return fail(VAQHIP_EINVAL, "waves_per_workgroup must be 0, 4, 8 or 16");
    ix->opt_nwaves = (int)value;
  } else {
    return fail(VAQHIP_EINVAL, "unknown option '%s'", key);
  }
  return VAQHIP_OK;
}

int vaqhip_last_timing(vaqhip_index *ix, vaqhip_timing *out) {
  if (!ix || !out) return fail(VAQHIP_EINVAL, "null pointer");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->ev_used > 0) {
    DeviceGuard g(ix->device);
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < ix->ev_used; i++) {
      hipEvent_t *ev = ix->ev.data() + (size_t)i * 6;
      HIP_TRY(hipEventSynchronize(ev[5]));
      for (int j = 0; j < 5; j++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev[j], ev[j + 1]));
        acc[j] += ms;
      }
    }
    const double n = ix->ev_used;
    ix->last.project_ms = (float)(acc[0] / n);
    ix->last.lut_ms = (float)(acc[1] / n);
    ix->last.seed_ms = (float)(acc[2] / n);
    ix->last.scan_ms = (float)(acc[3] / n);
    ix->last.merge_ms = (float)(acc[4] / n);
    ix->last.n_searches = ix->ev_used;
    ix->ev_used = 0;
    if (ix->last.deferred_queries >= 0 && ix->w_defer.p) {  // (the events above are past: the counter is final)
      unsigned asked = 0;
      HIP_TRY(hipMemcpy(&asked, ix->w_defer.p, sizeof asked, hipMemcpyDeviceToHost));
      ix->last.deferred_queries = ix->last.bucket_major ? (int)asked : (int)std::min<unsigned>(asked, (unsigned)DEFER_CAP);
    }
  }
  *out = ix->last;
  return VAQHIP_OK;
}

} // extern "C"
