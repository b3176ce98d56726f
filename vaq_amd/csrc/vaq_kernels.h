// vaq_kernels.h -- launch interface between the C-ABI host code
// (vaqhip_api.cpp) and the gfx950 kernels (vaq_kernels.hip).
#ifndef VAQ_KERNELS_H_
#define VAQ_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vaq {

// Per-subspace descriptor, one table per index in device memory.
struct SubDesc {
  int ncent;    // mCentroidsNum[s] = 1 << bits
  int bits;     // mBitsAlloc[s]
  int bit_off;  // first bit of the field inside a packed row (LSB-first)
  int lut_off;  // entry offset inside the packed per-query LUT
  int cent_off; // float offset inside the concatenated centroid buffer
  int word;     // bit_off / 32
  int shift;    // bit_off % 32
  int pad;
};

enum Layout { LAYOUT_BYTES = 0, LAYOUT_BITS = 1 };
// early-abandon forms of the scan (identical results):
//   EA_NONE    every row summed completely (VAQ::searchHeap as written)
//   EA_QUEUE   survivors of the first group compacted into an LDS queue and
//              finished 64 at a time (best when many query batches share the
//              code stream through cache: instruction-bound)
//   EA_INPLACE survivors finish in place under the EXEC mask (no queue, no
//              re-read of their codes: best for a single HBM-bound pass)
enum EarlyAbandon { EA_NONE = 0, EA_QUEUE = 1, EA_INPLACE = 2 };

// rows per planar tile of the bit-packed layout (one wavefront step)
constexpr int TILE_ROWS = 64;
// most wavefronts a scan workgroup may have (the host picks 4, 8 or 16)
constexpr int SCAN_MAX_WAVES = 16;
// sentinel id of an empty candidate slot (sorts after every real id)
constexpr int ID_SENTINEL = 0x7fffffff;

// one query handed from the first launch of the best-first form to the second (ScanParams::defer_*)
struct DeferRec {
  int q;              // query (index within the call)
  unsigned done_key;  // buckets with keys <= this were scanned by the first launch
  unsigned thr;       // its threshold when it stopped (float bits): an upper bound of the final k-th distance
  int pad;            // != 0: no bucket is finished yet (done_key is not a key)
};

struct ScanParams {
  const uint32_t *codes;  // packed codes (layout-specific)
  int64_t n_rows;         // local rows
  int layout;             // Layout
  int M;                  // subspaces
  int W;                  // dwords per row (bit-packed layout)
  const SubDesc *sub;     // [M]
  const uint32_t *perm;   // [n_rows] sorted row -> original row (labels), or nullptr = identity
  const int *bucket_start;// [n_buckets + 1] first sorted row of each subspace-0 code
  int n_buckets;          // 1 << (bits[0] - bucket_shift)
  int bucket_shift;       // bucket = first code >> bucket_shift
  int bucket_t;           // (bucket_shift == 0) bucket = first code << bucket_t | top bucket_t bits of code 1
  int n_hot;              // buckets a workgroup scans best-first before the rest (0 = off, <= 32)
  unsigned long long *stats; // diagnostic builds (-DVAQ_STATS): event counters, else nullptr
  int no_skip;            // 1: visit every bucket (measurement only: the streaming rate of the scan)
  const int *first_sub;   // [W+1] first subspace starting in word w (bit-packed layout)
  const float *lut;       // [nq][lut_floats]
  int lut_floats;
  int lds_subs;           // tables of subspaces [0, lds_subs) are staged in LDS ...
  int lut_lds_entries;    // ... = this many packed LUT entries; the rest is read from `lut`
  int nq;
  int k;
  int kp;                 // power of two >= k: slots of a workgroup's best list (per query)
  int ccap;               // slots of its candidate region
  int qcap;               // slots of a wave's survivor queue (early abandon)
  int q_cw_words;         // code dwords per queue entry (set by launch_scan)
  int nwaves;             // wavefronts per workgroup
  int ea;                 // EarlyAbandon
  unsigned *g_thr;        // [nq] shared threshold distances (float bits), preset to FLT_MAX
  int qb;                 // queries per pass (1, 2, 4)
  int n_slices;           // row slices per query batch
  int64_t slice_rows;     // rows per slice (multiple of the workgroup step)
  int64_t slice_stride;   // rows between slice starts (== slice_rows for a full scan,
                          // larger for the sampling pre-pass)
  int share_thr;          // 1: exchange thresholds between workgroups through g_thr
  const int *slice_order; // [query batches][n_slices] best-first slice order, or nullptr = row order
  const int *qorder;      // [nq] the i-th query a pass takes (similar queries share a pass), or nullptr = i
  int seq;                // 1: sequential row sum (BitVecEngine::queryLUT) instead of groups of 4
  int32_t *final_labels;  // non-null (needs n_slices == 1): results written directly, [nq][k]
  float *final_dist;
  int64_t id_base;
  int *part_cnt;          // [nq][n_slices] real entries of each list
  // triangle-inequality form (VAQ::searchTriangleInequality): rows are grouped by cluster
  // (bucket_start = cluster starts, n_buckets = clusters, farthest-from-centre first) and
  // each query visits its own list of clusters; needs qb == 1 and ea == EA_QUEUE
  int bf_carry;           // bit-packed best-first form: 1 = the groups after the first lie in the row's last
                          //    dword, which is carried through the survivor queue
  int bf_pool;            // best-first form: slots of the k-min pool (scan_bf_pool_range, multiple of 64)
  // best-first form, ONE workgroup per query (n_slices == 1): the expensive queries are cut in two.
  //   defer_units > 0   the first round takes at most this many work units (nearest buckets first);
  //                     what is still in reach when it is over goes to defer_list instead of a
  //                     second round, and a SECOND launch serves it with n_slices workgroups each
  //   defer_mode == 1   this is that second launch: workgroup v serves entry v / n_slices of
  //                     defer_list, rows of slice v % n_slices, starting from the entry's threshold
  //                     and skipping the buckets at or below its done_key; lists go to part_d/part_id
  int defer_units, defer_mode, defer_cap;
  unsigned *defer_count;  // entries asked for so far (beyond defer_cap: not handed over, scanned in place)
  struct DeferRec *defer_list;
  // bucket-major second pass (vaq_scan_bm.hip): with defer_units > 0 and bm_done set, EVERY query whose
  // capped first round left buckets in reach stores its done_key here (preset 0xffffffff = nothing
  // left) and its threshold -- lowered to its exact k-th distance when it has k rows -- in g_thr,
  // instead of joining defer_list
  unsigned *bm_done;
  int bf;                 // 1: best-first form (vaq_scan_bf.h): all buckets of the slice in ascending order
                          //    of their bound, work units by ticket (needs qb == 1, ea == EA_QUEUE, no TI)
  int ti;                 // 1: TI form
  const int *ti_order;    // [nq][n_buckets] clusters in visiting order
  const float *ti_qcc;    // [nq][n_buckets] query-to-centre distances, same order
  const int *ti_nvisit;   // [nq] clusters visited
  const float *ti_xcc;    // [n_rows] row-to-centre distances (index order)
  int ti_rowcap;          // rows taken from the visiting order (INT_MAX = all)
  int ti_cap;             // entries of the visiting list a workgroup stages in LDS at a time
  int sqrt_out;           // 1: the k-min is kept on sqrt(distance), as the reference stores it
                          //    (VAQ.cpp:1583); partial lists and g_thr then hold square roots
  float *part_d;          // [nq][n_slices][k]
  int *part_id;
};

// checked != 0: non-finite coordinates of the product become 0 (BitVecEngine.hpp:53-71); E == nullptr = identity
hipError_t launch_project(const float *X, int64_t n, int D, const float *E, float *out,
                          hipStream_t st, int checked = 0);
// cent_t: the codebooks dimension-major (entry j * ncent + c of subspace s at cent_off)
// min_ncent: the smallest codebook (subspaces of < 8 centroids take the reference's scalar branch)
hipError_t launch_lut_build(const float *qproj, int nq, int D, int M, int L,
                            const SubDesc *sub, const float *cent_t, int lut_floats, int max_ncent,
                            float *lut, hipStream_t st, int min_ncent = 1);
hipError_t launch_lut_expand(const float *lut_packed, int nq, int M, const SubDesc *sub,
                             int lut_floats, int ksub, float *lut_ref, hipStream_t st);
// VAQ::encodeImpl: Xp is n x D already in PCA space; codes is n x M uint16 row-major
hipError_t launch_encode(const float *Xp, int64_t n, int D, int M, int L, const SubDesc *sub,
                         const float *cent, uint16_t *codes, hipStream_t st);
// VAQ::refine: exact re-rank of R (<= 2048) candidates per query
hipError_t launch_refine(const float *Q, int nq, int D, const float *dataset, const float *rows,
                         const int32_t *labels_in, int R, int k, int32_t *labels, float *dist,
                         hipStream_t st);
// dwords the packed layout needs for `rows` rows
int64_t packed_words(int64_t rows, int M, int layout, int W);
// Pack rows [row_begin, row_end) (codes_u16 points at row_begin; row_begin a
// multiple of 64) and zero-fill the packed words up to out_row_end.
// perm (optional): packed row r is source row perm[r] (the bucketed order)
hipError_t launch_pack_codes(const uint16_t *codes_u16, int64_t row_begin, int64_t row_end,
                             int64_t out_row_end, int M, int layout, int W, const SubDesc *sub,
                             const uint32_t *perm, uint32_t *out, hipStream_t st);
// Stable sort of rows by their subspace-0 code: d_perm[n] (sorted row -> original
// row) and d_bucket_start[(1<<bits0)+1] (first sorted row of each code that
// occurs, -1 otherwise; the caller back-fills).  Synchronises the stream.
// fine > 0 (shift == 0): the rows of a bucket are also ordered by the next `fine` bits of the second
// code, and d_sub_start[((1 << kbits) << fine) + 1] receives the first row of every run (-1: back-fill)
hipError_t sort_by_first_code(const uint16_t *d_codes, int64_t n, int M, int bits0, int shift, int bits1,
                              int t, uint32_t *d_perm, int *d_bucket_start, hipStream_t st, int fine = 0,
                              int *d_sub_start = nullptr);
// Appending: merge the (separately sorted and packed) new rows into the bucketed order, bucket by
// bucket; *_start have K0 + 1 entries (back-filled); out_* are the new buffers
hipError_t launch_merge_rows(const uint32_t *old_codes, const uint32_t *old_perm, const int *old_start,
                             const uint32_t *new_codes, const uint32_t *new_perm, const int *new_start, int K0,
                             int64_t n_old, int64_t n_total, int M, int layout, int W, uint32_t *out_codes,
                             uint32_t *out_perm, hipStream_t st);
// Best-first slice order per query batch (n_slices <= 4096, n_buckets <= 4096)
hipError_t launch_slice_order(const float *lut, int lut_floats, int nq, int qb, const int *bstart,
                              int n_buckets, int bucket_shift, int64_t slice_rows, int n_slices,
                              int64_t n_rows, int *order, hipStream_t st);
// order[i] = the query the i-th pass slot takes: by (nearest first code, nearest second code); nq <= 16384
// best-first form, one workgroup per query: order[b] = the query block b serves, expensive queries
// first (keys: nq 64-bit scratch words; n0 = entries of table 0, shift = bucket_shift)
hipError_t launch_cost_order(const float *lut, int lut_floats, int nq, int n0, int shift, unsigned long long *keys,
                             int *order, hipStream_t st);
hipError_t launch_query_order(const float *lut, int lut_floats, int nq, int n0, int off1, int n1, int *order,
                              hipStream_t st);
// LDS geometry of a scan workgroup for top-k = k
void scan_geometry(int layout, int M, int k, int ea, int *kp, int *ccap, int *qcap);
// bytes of LDS a scan workgroup of `nwaves` wavefronts needs
size_t scan_lds_bytes(int layout, int M, int lut_floats, int qb, int k, int ea, int nwaves,
                      int n_buckets, int bucket_shift, int bucket_t);
// rows one step of the LARGEST workgroup covers: slice_rows and the code
// buffer's padding must be multiples of it
int scan_wg_step_rows(int layout, int M);
hipError_t launch_scan(const ScanParams &p, int *grid_out, hipStream_t st);
// best-first form (vaq_scan_bf.h): is there a kernel for this plan, and its LDS bytes
bool scan_bf_supported(int layout, int M, int qb, int ea, int n_buckets, int seq);
// `pool`: slots of the k-min pool (ScanParams::bf_pool), a multiple of 64 within scan_bf_pool_range(k)
size_t scan_bf_lds_bytes(int layout, int M, int lut_entries, int pool, int nwaves, int n_buckets, int bf_carry);
void scan_bf_pool_range(int k, int *lo, int *hi);
// in_final != 0: inputs use the API's -1 / FLT_MAX convention for empty slots
// candidate i of list l of query q sits at l*list_stride + q*query_stride + i
// labels == nullptr: no result is written (only thr_out is wanted).
// thr_out (optional): [nq] float bits, lowered to each query's k-th distance.
// More than 64 lists are folded in levels through scratch_d/scratch_id
// (merge_scratch_elems() elements each).
size_t merge_scratch_elems(int n_lists, int nq, int k);
// part_cnt (optional, scan partials only): real entries per list, [nq][n_lists]
// queries cut in two by the best-first form (ScanParams::defer_*): the first launch's result (already in
// labels/dist, the API's format) and the n_lists lists of the second -> the k best, in place
hipError_t launch_defer_merge(const unsigned *defer_count, int defer_cap, const DeferRec *defer_list, int n_lists, int k,
                              const float *part_d, const int *part_id, const int *part_cnt, int64_t id_base,
                              int32_t *labels, float *dist, hipStream_t st);
hipError_t launch_merge(const float *part_d, const int *part_id, const int *part_cnt, int n_lists,
                        int64_t list_stride, int64_t query_stride, int nq, int k,
                        int64_t id_base, int in_final, int32_t *labels, float *dist,
                        unsigned *thr_out, float *scratch_d, int *scratch_id, hipStream_t st);

// ---- bucket-major second pass of a streamed database with many queries (vaq_scan_bm.hip) ----
// After a capped best-first first pass (ScanParams::bm_done) every query knows the buckets it has
// finished (keys <= done_key) and an upper bound of its k-th distance (g_thr).  What is left in
// reach is turned round: per BUCKET the list of the queries that still want it, and workgroups
// that stream a bucket's rows once for QB queries whose lookup tables sit interleaved in LDS --
// the groups of one bucket run back to back on one XCD, so the bucket comes from HBM once and from
// that XCD's L2 for everybody else.  Survivors (complete sum not above the query's threshold) are
// appended to a per-query candidate buffer; a per-query histogram of the appended distances moves
// the shared threshold.  bm_select then takes the k best of (first pass list + candidates); a
// query whose buffer overflowed goes to the defer list and is finished by the best-first form.
constexpr int BM_HIST_BINS = 64;
constexpr int BM_XCDS = 8;
struct BmParams {
  const uint32_t *codes;
  const uint32_t *perm;      // sorted row -> label, or nullptr = identity
  const int *bucket_start;   // [n_buckets + 1]
  int n_buckets, bucket_t;   // bucket = first code << bucket_t | top bucket_t bits of the second (bucket_shift == 0)
  const int *sub_start;      // [(n_buckets << (8 - bucket_t)) + 1] first row of every (first code, second code) run
                             //    when the rows of a bucket are ordered by the second code, else nullptr
  int M;
  const float *lut;          // [nq][lut_floats]
  int lut_floats;
  int nq, k, qb;
  int nwaves;
  unsigned *g_thr;           // [nq] float bits: thresholds of the best-first form (rows AT them stay admissible):
                             //      what a best-first first pass leaves, and what its second launch (overflow) starts from
  unsigned long long *thr64; // [nq] the rounds' thresholds: distance bits << 32 | label bound -- a row is a candidate iff
                             //      its (distance bits << 32 | label) is BELOW this word; label bound 0x7fffffff keeps every
                             //      row at that distance.  init64: bm_mark sets it from g_thr first (a search's first round)
  int init64;
  unsigned *done_key;        // [nq] buckets with keys <= this are finished; 0xffffffff: the query is complete
  unsigned *done_next;       // [nq] what done_key becomes when the round's select succeeds
  unsigned *fresh;           // [nq] 1: no bucket of the query is finished yet (done_key is ignored: set by
                             //      launch_bm_boot, cleared by the query's first successful select)
  int limit;                 // buckets a query takes in this round, nearest first (<= 0: all in reach)
  int retry;                 // 1: a query whose candidate buffer overflows only gets a tighter threshold from what was
                             //    stored and tries the same buckets again next round; 0 (last round): defer list
  // plan (device, written by launch_bm_plan)
  unsigned *mask;            // [nq][n_buckets / 32] buckets still in reach
  int *cnt;                  // [n_buckets] queries per bucket
  int *qoff;                 // [n_buckets + 1]
  int *fill;                 // [n_buckets]
  int *qlist;                // [nq * n_buckets] (worst case) queries of bucket b at qoff[b], similar ones next to each other
  unsigned short *qkey;      // [nq][1 << bucket_t] per query and group of second codes: its nearest and second nearest
                             //      second code inside the group -- the bucket's list is ordered by it, so that the
                             //      queries of a group of QB want the same runs of the bucket
  int *border;               // [n_buckets] buckets by descending work
  int *ioff;                 // [BM_XCDS][n_buckets / BM_XCDS + 2] prefix of the work items of each XCD's buckets
  unsigned *tickets;         // [BM_XCDS]
  // candidates
  int cap;                   // slots per query
  float *cand_d;             // [nq][cap]
  int *cand_id;              // [nq][cap] labels (without id_base)
  unsigned *cand_cnt;        // [nq] candidates appended (beyond cap: not stored -> overflow)
  unsigned *hist;            // [nq][BM_HIST_BINS]
  float *scale;              // [nq] bins / H, 0 = histogram off
  // first-pass results (the API's format) and the final output, in place
  int32_t *labels;           // [nq][k]
  float *dist;
  int64_t id_base;
  // overflow -> defer list
  unsigned *defer_count;
  struct DeferRec *defer_list;
  int defer_cap;
};
bool scan_bm_supported(int layout, int M, int n_buckets, int bucket_shift, int seq, int k);
size_t scan_bm_lds_bytes(int M, int qb, int nwaves);
// words of the plan arrays above, for the caller's allocation: mask, cnt + qoff + fill + border + ioff + tickets
size_t bm_plan_small_words(int n_buckets);
// mark + count, order + prefix, fill (three launches)
hipError_t launch_bm_plan(const BmParams &p, hipStream_t st);
// instead of a best-first first pass: per query a threshold from a sample of its nearest rows (the k-th
// smallest complete sum of up to 4096 rows starting at the nearest run of the nearest bucket), an empty
// result list, nothing finished (the first round is then planned with BmParams::first = 1)
hipError_t launch_bm_boot(const BmParams &p, int64_t n_rows, hipStream_t st);
// staged search: thr64 = min(thr64, thr_in << 32 | 0x7fffffff) (thr_in optional); thr_out (optional) = distance bits
// of thr64; init: thr64 is first made from g_thr
hipError_t launch_bm_thresholds(const BmParams &p, const int32_t *thr_in, int32_t *thr_out, int init, hipStream_t st);
hipError_t launch_scan_bm(const BmParams &p, int n_cu, hipStream_t st);
hipError_t launch_bm_select(const BmParams &p, hipStream_t st);

// ---- the reference's own choice among rows of equal distance (vaq_exact.hip, option "exact_ties") ----
// inv[original row] = row of the bucketed order; row_bucket (optional): [original row] = its bucket
hipError_t launch_inverse_perm(const uint32_t *perm, int64_t n, uint32_t *inv, const int *bucket_start, int n_buckets,
                               unsigned short *row_bucket, hipStream_t st);
// in_labels / in_dist: the scan's result for k + 1 per query (labels carry id_base); labels / dist: the
// caller's k per query.  Queries whose k + 1 smallest distances are distinct are copied; the others
// are replayed through the reference's heap in original row order.  list: [nq] ints, count: one word.
hipError_t launch_exact_ties(const uint32_t *codes, int layout, int M, int W, const SubDesc *sub, const uint32_t *inv,
                             const unsigned short *row_bucket, int n_buckets, int bucket_shift, int bucket_t,
                             int64_t n_rows, const float *lut, int lut_floats, int nq, int k, int64_t id_base,
                             const int32_t *in_labels, const float *in_dist, int32_t *labels, float *dist, int *list,
                             unsigned *count, hipStream_t st);

// ---- triangle-inequality cluster pruning (vaq_ti.hip) ----------------------
// packed index rows -> uint16 N x M in original row order (inverse of launch_pack_codes)
hipError_t launch_unpack_codes(const uint32_t *packed, int64_t n, int M, int layout, int W,
                               const SubDesc *sub, const uint32_t *perm, uint16_t *out, hipStream_t st);
// VAQ::clusterTI's regrouping: d_perm[n], d_start[T+1] (-1 for clusters that do not occur;
// the caller back-fills), d_xcc_sorted[n].  Synchronises the stream.
hipError_t ti_group_rows(const uint16_t *d_codes, int64_t n, int M, int L, int seg, const SubDesc *sub,
                         const float *cent, const float *d_clusters, int T, uint32_t *d_perm,
                         int *d_start, float *d_xcc_sorted, hipStream_t st);
// per query: cluster visiting order, the matching distances, clusters visited
// (clusters_t: the centres dimension-major, T floats per dimension)
hipError_t launch_ti_plan(const float *qproj, int nq, int D, int d, const float *clusters_t, int T,
                          const int *start, int max_visit, int k, int *order, float *qcc, int *nvisit,
                          hipStream_t st);
// extra LDS bytes of a TI scan workgroup staging `cap` entries of its visiting list
size_t scan_ti_lds_bytes(int cap);

} // namespace vaq
#endif
