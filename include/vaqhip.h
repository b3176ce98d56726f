/*
 * vaqhip.h -- C ABI of the MI355X (gfx950) implementation of VAQ's
 * quantized-distance search path.
 *
 * The reference has no FFI layer: the path sits behind C++ member functions
 * called directly by its drivers.  Each entry point below names the reference
 * interface it replaces (file:line under the reference checkout).  A C++
 * adapter with the reference's own names (class VaqHip: search(),
 * parseMethodString(), public mCodebook-style members) is in
 * include/vaqhip.hpp; INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross this boundary
 *  - every function returns 0 on success or a negative VAQHIP_E* code; the
 *    process is never exit()ed or assert()ed (the reference prints and
 *    exits: VAQ.cpp:64-78, 1263-1266); vaqhip_last_error() gives the text
 *  - "host" entry points take host pointers and are synchronous;
 *    "_device" entry points take device pointers on the index's GPU plus a
 *    hipStream_t (passed as void*), enqueue only, and never synchronise
 *  - the library has no CPU fallback: without a usable HIP device every call
 *    fails with VAQHIP_ENODEVICE
 *  - one index may be used from several host threads and several streams: calls on the same
 *    index are serialised internally on the host, and since every call shares the index's
 *    workspaces (lookup tables, partial lists, thresholds), a "_device" call on a stream other
 *    than the one the previous call used makes its stream wait (hipStreamWaitEvent) for that
 *    call's work first -- searches on one index never overlap on the GPU; use one index per
 *    stream (or vaqhip_multi) for concurrency
 */
#ifndef VAQHIP_H_
#define VAQHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAQHIP_VERSION 104

/* error codes */
#define VAQHIP_OK            0
#define VAQHIP_EINVAL       -1   /* bad argument (null, negative size, ...)                    */
#define VAQHIP_EUNSUPPORTED -2   /* valid for the reference but outside this build's limits     */
#define VAQHIP_ENODEVICE    -3   /* no usable HIP device / HIP runtime error at init            */
#define VAQHIP_ENOMEM       -4   /* device or host allocation failed                            */
#define VAQHIP_EHIP         -5   /* HIP runtime error during a call                             */
#define VAQHIP_ERANGE       -6   /* label would not fit the reference's 32-bit int labels       */
#define VAQHIP_ESTATE       -7   /* call order (e.g. search before codes were added)            */

/* search method bits, numerically equal to VAQ::NNMethod (VAQ.hpp:38-49).
 * HEAP and EA run the same kernels because the reference's early abandon
 * returns results identical to HEAP (VAQ.cpp:1694-1727 vs 1729-1758).  TI is
 * the triangle-inequality cluster pruning (VAQ.cpp:799-826, 1540-1692); see
 * vaqhip_index_set_ti_clusters. */
#define VAQHIP_METHOD_EA   0x02u
#define VAQHIP_METHOD_TI   0x04u
#define VAQHIP_METHOD_HEAP 0x80u

/* limits of this build */
#define VAQHIP_MAX_SUBSPACES 128
#define VAQHIP_MAX_BITS      15    /* VAQ.cpp:787-798 dispatches CreateLUT<9..15> */
#define VAQHIP_MAX_K         1024
#define VAQHIP_MAX_TI_CLUSTERS 4096 /* mTIClusterNum; the paper's runs use 100..2000 */

typedef struct vaqhip_index vaqhip_index;

/* ---------------------------------------------------------------------------
 * Index state = the public members VAQ::search reads (VAQ.hpp:51-75):
 *   D                 mTotalDim (= M * mSubsLen)
 *   M                 mHighestSubs; must be a multiple of 4 (VAQ.cpp:1741-1746
 *                     reads four codes per step)
 *   bits[M]           mBitsAlloc; mCentroidsNum[s] = 1 << bits[s]
 *   centroids[s]      mCentroidsPerSubs[s], row-major (1<<bits[s]) x (D/M)
 *   eigvec_real       real part of mEigenVectors, row-major D x D, or NULL for
 *                     identity (queries already in PCA space)
 *   device_id         HIP device ordinal the index lives on
 * Replaces: the state half of `class VAQ` consumed by search(), VAQ.hpp:51-75.
 * ------------------------------------------------------------------------- */
int vaqhip_index_create(vaqhip_index **out, int D, int M, const int *bits,
                        const float *const *centroids_rowmajor,
                        const float *eigvec_real_rowmajor, int device_id);

/* Same with flags.  VAQHIP_SUM_SEQUENTIAL selects the row sum of the reference's other
 * entry on this path, BitVecEngine::queryLUT (BitVecEngine.hpp:1222-1343): one scalar
 * quantiser per PCA dimension (D == M, sub-vector length 1), LUT column stride 256, and
 * dist = ((l_0 + l_1) + l_2) + ... summed column by column (:1296-1300) instead of in
 * groups of four; M need not be a multiple of 4.  centroids[s] is then column s of the
 * engine's centroidsMat (1 << bits[s] values).  Queries are projected WITH CHECKING, as queryLUT
 * does (:1226 -> :53-71): a PCA coordinate that comes out NaN or infinite becomes 0 (one non-finite
 * component makes every coordinate of z * V non-finite, so such a query is answered as the zero
 * vector); VAQ::search's projection (VAQ.hpp:198-201) does not check and is left as it is. */
#define VAQHIP_SUM_SEQUENTIAL 0x1u
int vaqhip_index_create_ex(vaqhip_index **out, int D, int M, const int *bits,
                           const float *const *centroids_rowmajor,
                           const float *eigvec_real_rowmajor, int device_id, unsigned flags);

void vaqhip_index_destroy(vaqhip_index *ix);

/* mCodebook (CodebookType = RowMatrix<uint16_t>, utils/Types.hpp:31), N x M
 * row-major.  The codes are repacked on the GPU into the bit-packed device
 * layout (DESIGN.md "Data layout"); the caller keeps ownership of the input.
 * Replaces the `mCodebook` member filled by VAQ::encode (VAQ.cpp:663-726) or
 * loadCodebook (utils/IO.hpp:551-571).  Calling it again replaces the codes.
 * id_base: global row index of local row 0 (shard offset); labels
 * returned by search are id_base + local row and must stay < 2^31.
 * Both forms sort the rows by their first code on the GPU (bucketed order,
 * DESIGN.md section 3) and therefore synchronise; the _device form does so on `stream`. */
int vaqhip_index_set_codes_u16(vaqhip_index *ix, const uint16_t *codes_rowmajor,
                               int64_t N, int64_t id_base);
int vaqhip_index_set_codes_u16_device(vaqhip_index *ix, const uint16_t *d_codes_rowmajor,
                                      int64_t N, int64_t id_base, void *stream);

/* Append n_new rows (same layout) behind the rows already in the index; their labels
 * continue at id_base + N.  Stands for growing `mCodebook` and calling the setter again
 * (SURVEY.md section 8b lists the entry point as `add_codes`).  A bucketed index sorts and packs
 * the NEW rows only and merges them into the existing order bucket by bucket: the packed rows and
 * their labels are copied once, nothing is unpacked or re-sorted, temporaries are O(n_new) plus
 * the new packed buffer; the bucket key width stays the one chosen when the codes were set.  A
 * TI-grouped index (rows ordered by cluster and centre distance) is rebuilt as a whole.
 * Synchronises. */
int vaqhip_index_add_codes_u16(vaqhip_index *ix, const uint16_t *codes_rowmajor, int64_t n_new);
int vaqhip_index_add_codes_u16_device(vaqhip_index *ix, const uint16_t *d_codes_rowmajor,
                                      int64_t n_new, void *stream);

/* VAQ::clusterTI (VAQ.hpp:106, VAQ.cpp:878-999) from the point where mTIClusters
 * exists: `clusters` is mTIClusters, T x (seg_num * D/M) row-major, i.e. T
 * centres over the first seg_num subspaces (mTIClusterNum, mTISegmentNum); how
 * they were made (the reference: k-means over decoded codes, VAQ.cpp:897-900) is
 * the caller's business, like the codebooks.  Every code row joins its nearest
 * centre (VAQ.cpp:926-950), clusters are ordered farthest member first
 * (:972-979) and the packed codes are regrouped on the GPU (:984-996).  May be
 * called before or after the codes are set (the reference calls it after
 * encode); T = 0 returns the index to the exhaustive HEAP/EA form.  Sets /
 * clears VAQHIP_METHOD_TI in the index's method.  Synchronises.
 * Labels stay ORIGINAL row indices + id_base, as mTIClustersMember holds them. */
int vaqhip_index_set_ti_clusters(vaqhip_index *ix, const float *clusters_rowmajor, int T,
                                 int seg_num);

/* mMethods (VAQ::parseMethodString, VAQ.cpp:1205-1262) and mVisit (VAQ.hpp:84,
 * demo_vaq --visit-cluster).  With TI:
 *   - the clusters visited are the first  max(int(T * visit), shortest prefix
 *     holding k rows)  in ascending query-to-centre distance (VAQ.cpp:1548-1555,
 *     :1611); visit >= 1 visits all;
 *   - TI | EA returns the k best of the visited rows; distances are sqrt'ed
 *     (VAQ.cpp:1583);
 *   - TI without EA returns the first k rows of the visiting order, as the
 *     reference does (its bsfKSquared stays 0, VAQ.cpp:1617-1686).
 * HEAP / EA without TI: the exhaustive scan; `visit` is ignored. */
int vaqhip_index_set_method(vaqhip_index *ix, unsigned methods, float visit);

/* VAQ::search (VAQ.hpp:102, VAQ.cpp:776-847), HEAP / EA semantics:
 *   queries   nq x D row-major, unprojected (projected by eigvec on the GPU)
 *   labels    nq x k, ascending by (distance, label); unfilled slots -1
 *   distances nq x k squared L2 in PCA space (no sqrt, VAQ.cpp:1737-1753);
 *             unfilled slots FLT_MAX (utils/Heap.hpp:322-349)
 * Among rows of exactly equal distance the smaller label wins (the
 * reference's choice there depends on heap internals; DESIGN.md "Ties"). */
int vaqhip_search(vaqhip_index *ix, const float *queries_rowmajor, int nq, int k,
                  int32_t *labels, float *distances);
/* queries already in PCA space (skips ProjectOnEigenVectors, VAQ.hpp:198-201) */
int vaqhip_search_projected(vaqhip_index *ix, const float *qproj_rowmajor, int nq, int k,
                            int32_t *labels, float *distances);
/* device pointers, enqueue on `stream` */
int vaqhip_search_device(vaqhip_index *ix, const float *d_queries, int nq, int k,
                         int projected, int32_t *d_labels, float *d_distances,
                         void *stream);

/* Staged search for hosts that shard the rows over several GPUs, one process (or index) per GPU: every
 * shard finds ITS k best, so on its own its admission thresholds are looser than the global k-th
 * distance allows.  `begin` runs the first rounds of the search (each query's nearest buckets) and
 * writes the thresholds they leave -- nq distance bit patterns, int32, ordered like the distances --
 * to d_thresholds_out; the caller takes the element-wise MINIMUM over all shards (one all-reduce of
 * 4 * nq bytes: a threshold is an upper bound of the query's final k-th distance, and any shard's
 * bound holds everywhere, the k rows behind it exist) and hands it to `finish`, which scans what is
 * still in reach under it.  A shard may then return fewer than k rows for a query (slots -1 / FLT_MAX):
 * the merge of the shards' lists (vaqhip_merge_topk_*) is the global result, bit for bit what one
 * index over all rows returns.  d_labels / d_distances of `begin` hold intermediate lists until `finish`
 * returns; no other call on the index in between (VAQHIP_ESTATE).  VAQHIP_EUNSUPPORTED when the search
 * would not run the bucket-major rounds (few queries, cache-resident or bit-packed rows, TI): use
 * vaqhip_search_device then -- vaqhip_search_staged_supported tells beforehand, so that all shards can
 * agree.  d_thresholds_in may be NULL (no exchange).  Replaces nothing in the reference (its search is
 * one process on one host, VAQ.cpp:776); it is the exchange step SURVEY 8(e) allows for. */
int vaqhip_search_staged_supported(vaqhip_index *ix, int nq, int k);
int vaqhip_search_begin_device(vaqhip_index *ix, const float *d_queries, int nq, int k, int projected,
                               int32_t *d_labels, float *d_distances, int32_t *d_thresholds_out, void *stream);
int vaqhip_search_finish_device(vaqhip_index *ix, const int32_t *d_thresholds_in, void *stream);

/* Test hook for VAQ::CreateLUT<maxbit> (VAQ.hpp:128-167): writes, per query,
 * the reference LUTType (column-major ksub x M, ksub = 1 << max(bits), rows
 * >= 1<<bits[s] zero): lut_out[q*M*ksub + s*ksub + c]. */
int vaqhip_build_lut(vaqhip_index *ix, const float *queries_rowmajor, int nq,
                     int projected, float *lut_out);

/* VAQ::ProjectOnEigenVectors (VAQ.hpp:198-201): out = X * real(eigvec). */
int vaqhip_project(vaqhip_index *ix, const float *X_rowmajor, int64_t n, float *out);

/* VAQ::encode / encodeImpl (VAQ.cpp:663-748): per subspace, argmin over centroids of the
 * squared L2 to the row's sub-vector (strict <, first minimum wins).  codes_out is the
 * reference's CodebookType, n x M uint16 row-major.  projected != 0: X is already in
 * PCA space, which is what the reference's encode() expects (train() projects the
 * dataset in place, VAQ.cpp:294); projected == 0 applies eigvec first. */
int vaqhip_encode(vaqhip_index *ix, const float *X_rowmajor, int64_t n, int projected,
                  uint16_t *codes_out);
int vaqhip_encode_device(vaqhip_index *ix, const float *d_X, int64_t n, int projected,
                         uint16_t *d_codes_out, void *stream);

/* VAQ::refine (VAQ.cpp:849-876): exact squared L2 in the ORIGINAL space between each
 * query and its R candidate rows of the raw dataset, k best by the same k-min rule.
 * labels_in: nq x R (negative labels are skipped); R <= 2048.  The host form gathers
 * the candidate rows from `dataset_rowmajor` (N x D, raw, unprojected) itself. */
int vaqhip_refine(int device_id, const float *queries_rowmajor, int nq, int D,
                  const float *dataset_rowmajor, int64_t N, const int32_t *labels_in, int R,
                  int k, int32_t *labels_out, float *distances_out);
int vaqhip_refine_device(int device_id, const float *d_queries, int nq, int D,
                         const float *d_dataset, const int32_t *d_labels_in, int R, int k,
                         int32_t *d_labels_out, float *d_distances_out, void *stream);

/* Multi-GPU exchange step: after an all-gather of per-shard
 * results laid out [n_lists][nq][k] (labels already global), keep per query
 * the k smallest by (distance, label).  Device pointers.  The reference's
 * precedent for shard-and-merge: BitVecEngine.cpp:1034-1132 (:1114-1126). */
int vaqhip_merge_topk_device(int device_id, const float *d_dist_lists,
                             const int32_t *d_label_lists, int n_lists, int nq, int k,
                             int32_t *d_labels_out, float *d_dist_out, void *stream);

/* Same with explicit element strides: candidate i of list l of query q is read at
 * d_dist_lists[l*list_stride + q*query_stride + i] (labels likewise), so the lists may
 * sit inside one packed all-gather buffer (labels and distances gathered by a single
 * collective: vaq_amd/sharding.py). */
int vaqhip_merge_topk_strided_device(int device_id, const float *d_dist_lists,
                                     const int32_t *d_label_lists, int n_lists,
                                     int64_t list_stride, int64_t query_stride, int nq, int k,
                                     int32_t *d_labels_out, float *d_dist_out, void *stream);

/* ---------------------------------------------------------------------------
 * Multi-device index (north_star: "the code database shards
 * naturally across the 8 GPUs of one node with a final RCCL all-gather of per-shard top-k").
 * One process, one host thread per GPU.  set_codes cuts the rows into contiguous shards (shard g =
 * rows [g * ceil(N/G), (g+1) * ceil(N/G)), labels stay global row numbers); every device answers
 * all queries on its shard; the exchange step is ONE ncclAllGather (RCCL over xGMI) of the packed
 * per-shard results followed by the k-min merge by (distance, label): the result equals a single
 * index over all rows bit for bit.  The reference's precedent for shard-and-merge is
 * BitVecEngine.cpp:1034-1132 (merge at :1114-1126); its own search is single-threaded on one host
 * (VAQ.cpp:776-847), so this is the form `class VAQ` takes on a multi-GPU node.
 *   device_ids   HIP ordinals, one per shard.  Naming one GPU several times gives LOGICAL shards
 *                on that GPU (RCCL refuses duplicate devices, so the gather is then done with
 *                device-to-device copies; same buffers, same merge) -- how a one-GPU box tests
 *                the sharded path.
 * Options ("exchange": 0 auto, 1 RCCL, 2 copies; anything else is forwarded to every shard).
 * ------------------------------------------------------------------------- */
#define VAQHIP_MAX_DEVICES 16
typedef struct vaqhip_multi vaqhip_multi;
int vaqhip_multi_create(vaqhip_multi **out, int D, int M, const int *bits,
                        const float *const *centroids_rowmajor, const float *eigvec_real_rowmajor,
                        int n_devices, const int *device_ids, unsigned flags);
void vaqhip_multi_destroy(vaqhip_multi *mx);
/* mCodebook for the whole database (host pointer); sharded contiguously across the devices, every
 * shard uploaded, sorted and packed by its own host thread at the same time */
int vaqhip_multi_set_codes_u16(vaqhip_multi *mx, const uint16_t *codes_rowmajor, int64_t N, int64_t id_base);
/* append: the new rows continue the numbering, so they extend the LAST shard -- repeated appends pile
 * rows (memory and scan time) on one device; vaqhip_multi_get_info's shard_rows shows the skew, and
 * vaqhip_multi_set_codes_u16 with the whole matrix re-balances */
int vaqhip_multi_add_codes_u16(vaqhip_multi *mx, const uint16_t *codes_rowmajor, int64_t n_new);
/* VAQ::search on every shard + exchange + merge; host pointers, synchronous */
int vaqhip_multi_search(vaqhip_multi *mx, const float *queries_rowmajor, int nq, int k, int projected,
                        int32_t *labels, float *distances);
/* The same with device pointers, enqueue only: d_queries (nq x D) and the outputs live on device_ids[0];
 * the call returns once every shard's work is enqueued -- each shard copies the queries over the fabric
 * (hipMemcpyPeerAsync, no host staging), `stream` (a stream of device_ids[0]) is made to wait for the
 * merged result, the host is not.  Like the single-index "_device" entry points it never synchronises. */
int vaqhip_multi_search_device(vaqhip_multi *mx, const float *d_queries, int nq, int k, int projected,
                               int32_t *d_labels, float *d_distances, void *stream);
/* forwarded to every shard (each shard regroups its own rows under the same TI centres) */
int vaqhip_multi_set_ti_clusters(vaqhip_multi *mx, const float *clusters_rowmajor, int T, int seg_num);
int vaqhip_multi_set_method(vaqhip_multi *mx, unsigned methods, float visit);
int vaqhip_multi_set_option(vaqhip_multi *mx, const char *key, int64_t value);
typedef struct {
  int n_devices;
  int exchange;              /* what the last search used: 0 none (one shard), 1 RCCL all-gather, 2 copies */
  int64_t N, id_base;
  int device_ids[VAQHIP_MAX_DEVICES];
  int64_t shard_rows[VAQHIP_MAX_DEVICES];
  float last_search_ms;      /* device time on shard 0: upload + project + LUT + scan (+ merge of its slices) */
  float last_exchange_ms;    /*   the all-gather (or the copies), incl. waiting for the slowest shard       */
  float last_merge_ms;       /*   the G-way merge kernel                                                     */
} vaqhip_multi_info;
int vaqhip_multi_get_info(const vaqhip_multi *mx, vaqhip_multi_info *out);
/* shard g's single-device index (options, timing, info); owned by the multi index */
vaqhip_index *vaqhip_multi_shard(vaqhip_multi *mx, int g);
const char *vaqhip_multi_last_error(void);

/* ----- introspection / tuning -------------------------------------------- */
typedef struct {
  int D, M, L;
  int max_bits;        /* mMaxBitsPerSubs                                   */
  int total_bits;      /* sum of bits                                       */
  int code_bytes;      /* packed bytes per row on the device                */
  int algo_code_bytes; /* ceil(total_bits / 8): the roofline's byte figure  */
  int lut_floats;      /* sum of 1<<bits[s]: packed LUT entries per query   */
  int64_t N;
  int64_t id_base;
  int device_id;
  int layout;          /* 0 = one byte per subspace (all bits == 8), 1 = bit-packed */
  int ti_clusters;     /* mTIClusterNum, 0 = rows in the exhaustive (bucketed) order */
  int ti_segments;     /* mTISegmentNum                                     */
  unsigned methods;    /* VAQHIP_METHOD_* bits in force                     */
  float visit;         /* mVisit                                            */
} vaqhip_info;
int vaqhip_index_info(const vaqhip_index *ix, vaqhip_info *out);

/* Options (all optional; defaults chosen per launch):
 *   "queries_per_pass"  Qb in {0 = auto, 1, 2, 4}: queries served by one
 *                       streaming pass of a workgroup over its code slice
 *   "slices"            0 = auto, else number of row slices per query batch
 *   "timing"            1: record hipEvents (on the search's own stream) around
 *                       each kernel of every following search, up to 256
 *                       searches between two vaqhip_last_timing reads
 *   "hot_buckets"       0..32 (default 16): buckets (rows sharing their first code) each
 *                       workgroup scans best-first, closest first term first, before
 *                       the rest of its slice; 0 = natural order only
 *   "waves_per_workgroup" 0 = auto (most wavefronts per CU), else 4, 8 or 16
 *   "ordered_slices"    0 (default): slices in row order.  1: when a query's rows are
 *                       split over 2..4096 workgroups, dispatch the slices
 *                       best-first per query batch instead of running the sampling
 *                       pre-pass (experimental; measured slower)
 *   "seed_thresholds"   1 (default): when a query's rows are split over several
 *                       workgroups, a pre-pass over 1/64 of the rows seeds their
 *                       admission thresholds; 0: every workgroup warms up alone
 *   "early_abandon"     how a row is dropped once a partial sum exceeds the query's
 *                       current k-th best (the GPU forms of VAQ::searchEarlyAbandon,
 *                       VAQ.cpp:1694-1727); results are identical for every value:
 *                       0 never (VAQ::searchHeap as written), 1 survivors are
 *                       compacted through an LDS queue, 2 survivors finish in
 *                       place, 3 (default) 1 or 2 chosen per search            */
/*   "group_queries"     1 (default): on a streamed database (> 256 MB of codes) with several
 *                       multi-query passes, the queries are ordered by their nearest first and
 *                       second codes before they are cut into passes -- a pass can only skip a
 *                       bucket all of its queries can skip, and similar queries skip the same ones;
 *                       2: always; 0: never.  Results are written to the queries' own slots.
 *   "best_first"        1 (default): a one-query workgroup whose row slice spans many buckets (the
 *                       cache-resident databases) visits ALL of them in ascending order of their
 *                       bound, work units handed to its waves by ticket, and stops at the first
 *                       bucket out of reach ("best-first form"); 0: the
 *                       16-hot-buckets-then-natural-order form.  Results are identical.
 *   "cost_order"        1 (default): with one best-first workgroup per query and at least 1024
 *                       queries in the call, the queries are ranked by a cost key (how flat the
 *                       first lookup table is near its minimum: sum of its 16 smallest bucket
 *                       minima - 16 x the smallest) and the expensive ones are
 *                       dispatched first -- a query's cost spans 6x and a launch otherwise ends
 *                       with its few most expensive workgroups; 0: block b serves query b.
 *                       Results are written to the queries' own rows and are identical.
 *   "defer_units"       0 (default) = off; n > 0: with one best-first workgroup per query, a
 *                       query's first round takes at most n work units (64 wave steps each); the
 *                       buckets still in reach after it are scanned by a second launch, two
 *                       workgroups per query, and merged in (-1: n = 96 from 4096 queries per call
 *                       on).  Results are identical.  Measured slower than scanning on in place
 *                       (DESIGN.md section 4): the second launch has a tail of its own.
 *   "bucket_bits"       0 (default) = auto, else 1..12: width of the key the rows are bucketed by
 *                       (top bits of the first code, continued into the second); takes
 *                       effect when the codes are (re)set
 *   "bucket_skip"       1 (default); 0 visits every bucket -- for measuring the streaming
 *                       rate of the scan, results are the same                             */
/*   "exact_ties"        0 (default): among rows of exactly equal distance the smaller label wins (above).
 *                       1: the reference's own choice -- which of several rows tying at the k-th
 *                       distance survive, and the order equal distances are returned in, come from
 *                       its heap (VAQ.cpp:1750-1757, utils/Heap.hpp:115-169, 322-349).  The scan runs
 *                       with k + 1; a query whose k + 1 smallest distances are distinct is unaffected,
 *                       every other query is replayed through that heap over ALL rows in original
 *                       order (one workgroup per such query: 1M rows x 10 k queries, nine in ten of
 *                       them with ties: 0.65 -> 43 ms; about a second per tied query at 1B rows).
 *                       HEAP / EA without TI, k < 1024; labels and distances are then identical to
 *                       VAQ::search's, slot for slot.
 *   "bucket_major"      1 (default): on a streamed database (> 128 MB of byte codes) with at least 8
 *                       queries in the call, the best-first pass is cut after each query's nearest
 *                       buckets and what is left in reach is scanned bucket by bucket: a bucket's
 *                       rows are streamed once for ALL the queries that still want it, four
 *                       queries' lookup tables at a time in LDS, instead of once per query
 *                       ("Bucket-major second pass"); 0: off; 2: whenever a
 *                       kernel exists (tests).  Results are identical.
 *   "bm_candidates"     slots of a query's candidate buffer in that pass (default 4096); a query that
 *                       overflows it tries the same buckets again under the threshold the stored rows
 *                       give, and is finished by the best-first form if that overflows too
 *   "bm_units"          work units (64 wave steps) of the first pass per query; 0 = about one
 *                       average bucket
 *   "bm_boot"           2: no best-first pass at all -- every query gets a threshold from a
 *                       sample of its nearest rows, and its nearest bucket is the first bucket-major
 *                       round; 0: a capped best-first pass ("bm_units") comes first; 1 (default):
 *                       the former when buckets are large (>= 24 work units on average)
 *   "bm_round"          buckets per query of the middle round (default 6, 0 = no middle round): after
 *                       it the thresholds are near their final values, and the last round -- every
 *                       bucket still in reach -- meets far fewer rows
 *   "bm_runs"           1 (default): runs of rows sharing their first two codes are skipped when out of
 *                       every query's reach ("sub_order": the rows of a bucket are kept ordered by the
 *                       second code; set before the codes)
 *   "bm_queries_per_group", "bm_waves"   launch shape of the second pass (0 = default 4 / 16)    */
int vaqhip_set_option(vaqhip_index *ix, const char *key, int64_t value);

typedef struct {
  float project_ms, lut_ms, seed_ms, scan_ms, merge_ms; /* mean device time per search over
                                                  the searches recorded since the last read;
                                                  seed = threshold pre-pass, scan = the
                                                  full code scan kernel alone             */
  int n_searches;                              /* how many searches that mean covers   */
  int queries_per_pass;                        /* Qb actually used              */
  int slices;                                  /* row slices per query batch    */
  int workgroups;                              /* scan kernel grid size         */
  int passes;                                  /* ceil(nq / Qb)                 */
  int lds_bytes;                               /* LDS per scan workgroup        */
  int seed_slices;                             /* row slices of the pre-pass (0 = none) */
  int early_abandon;                           /* form the scan ran in: 0 none, 1 queue, 2 in place */
  int best_first;                              /* 1: the best-first form ("best_first" option) ran */
  int deferred_queries;                        /* queries of the last search cut in two ("defer_units"): handed
                                                  to the second launch; -1 = the search did not defer */
  int bucket_major;                            /* 1: the bucket-major second pass ran ("bucket_major") */
} vaqhip_timing;
int vaqhip_last_timing(vaqhip_index *ix, vaqhip_timing *out);

const char *vaqhip_last_error(void);
int vaqhip_version(void);
/* number of HIP devices visible, or a negative error code */
int vaqhip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* VAQHIP_H_ */
