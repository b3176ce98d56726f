// vaqhip_multi.cpp -- the multi-GPU form of the index behind the C ABI (include/vaqhip.h,
// "multi-device"): SURVEY.md section 8(b) rows 1-3 / 8(e).
//
// One process drives the GPUs of a node, one host thread per device.  The code rows are cut
// into contiguous shards (shard g = rows [g * ceil(N/G), (g+1) * ceil(N/G))), every shard is an
// ordinary vaqhip_index on its device with id_base = its first row, every device answers ALL
// queries on its shard, and ONE exchange step finishes the search: an all-gather of the packed
// per-shard results [2][nq][k] (labels, distance bits) over RCCL -- ncclAllGather on communicators
// made by ncclCommInitAll, i.e. xGMI between the GPUs of the node -- followed by the k-min merge
// kernel by (distance, label).  Shards are contiguous in label order and a single index orders by
// (distance, label) too, so the merged result equals the single-index result bit for bit.
// The reference's precedent for shard-and-merge: BitVecEngine.cpp:1034-1132 (merge :1114-1126).
//
// RCCL is loaded with dlopen at the first multi-device search (libvaqhip.so itself does not link
// it): inside a Python process PyTorch's own copy is already mapped and is the one that gets used.
// When the device list names one GPU several times (logical shards: how the exchange and merge
// are tested on a one-GPU box) RCCL cannot be used -- it refuses duplicate devices -- and the
// gather is done with device-to-device copies instead; same buffers, same merge.
#include "vaqhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "job_pool.h"

namespace {

int mfail(int code, const char *fmt, ...);

// ---- RCCL, resolved at run time -------------------------------------------------------------
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;  // ncclSuccess == 0
enum { NCCL_INT32 = 2 };   // ncclInt32 / ncclInt (rccl.h: ncclDataType_t)
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string where;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool load_rccl(std::string *err) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return true;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names)  // a copy that is already mapped (PyTorch's) wins
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) { g_rccl.where = std::string(n) + " (already loaded)"; break; }
  if (!h)
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) { g_rccl.where = n; break; }
  if (!h) {
    *err = std::string("RCCL not found: ") + dlerror();
    return false;
  }
  Rccl r;
  r.h = h;
  r.where = g_rccl.where;
#define VAQ_SYM(field, name)                                              \
  *reinterpret_cast<void **>(&r.field) = dlsym(h, name);                  \
  if (!r.field) { *err = std::string("RCCL symbol missing: ") + name; return false; }
  VAQ_SYM(CommInitAll, "ncclCommInitAll")
  VAQ_SYM(CommDestroy, "ncclCommDestroy")
  VAQ_SYM(AllGather, "ncclAllGather")
  VAQ_SYM(GroupStart, "ncclGroupStart")
  VAQ_SYM(GroupEnd, "ncclGroupEnd")
  VAQ_SYM(GetErrorString, "ncclGetErrorString")
#undef VAQ_SYM
  g_rccl = r;
  return true;
}

enum Exchange { EX_AUTO = 0, EX_RCCL = 1, EX_COPIES = 2 };

struct Shard {
  vaqhip_index *ix = nullptr;
  int device = 0;
  int64_t lo = 0, n = 0;  // rows [lo, lo + n) of the database
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;  // this shard's packed result is complete
  float *d_queries = nullptr;
  int32_t *d_packed = nullptr;    // [2][nq][k]: labels, distance bits
  int32_t *d_gathered = nullptr;  // [G][2][nq][k] (every device under RCCL; shard 0 with copies)
  size_t cap_q = 0, cap_p = 0, cap_g = 0;
  ncclComm_t comm = nullptr;
  std::string err;  // what the shard's last phase failed with
};

} // namespace

struct vaqhip_multi {
  int D = 0, M = 0, G = 0;
  std::vector<Shard> sh;
  bool distinct = true;  // no device named twice
  int exchange = EX_AUTO;
  bool comms_ready = false;
  int64_t N = 0, id_base = 0;
  // one call at a time (mu); every phase of it runs on the shards' worker threads (job_pool.h), and the
  // caller only goes on to the next phase -- the collective -- when every shard has succeeded
  mutable std::mutex mu;
  vaq::JobPool pool;
  // the current search
  const float *queries = nullptr;      // host pointer, or
  const float *d_queries0 = nullptr;   // device pointer on shard 0's device (vaqhip_multi_search_device)
  hipEvent_t user_ready = nullptr;     //   recorded on the caller's stream: the queries are there
  hipEvent_t consumed = nullptr;       // shard 0 has read every shard's packed result (copies) / merged
  hipEvent_t finished = nullptr;       // the result is in the caller's device buffers
  int nq = 0, k = 0, projected = 0, use_rccl = 0;
  int32_t *d_out_labels = nullptr;
  float *d_out_dist = nullptr;
  size_t cap_out = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // shard 0: start, searched, gathered, merged
  vaqhip_multi_info last = {};
};

namespace {

thread_local std::string g_merr;

int mfail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_merr = buf;
  return code;
}

#define MHIP(expr)                                                                         \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      s.err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
      return e_ == hipErrorOutOfMemory ? VAQHIP_ENOMEM : VAQHIP_EHIP;                      \
    }                                                                                      \
  } while (0)

int grow(Shard &s, void **p, size_t *cap, size_t bytes) {
  if (bytes <= *cap) return 0;
  if (*p) MHIP(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  MHIP(hipMalloc(p, bytes));
  *cap = bytes;
  return 0;
}

// one shard's part of a search; runs on that shard's worker thread with its device current
int run_shard(vaqhip_multi *mx, int g) {
  Shard &s = mx->sh[g];
  const int G = mx->G, nq = mx->nq, k = mx->k;
  const size_t plane = (size_t)nq * k;
  MHIP(hipSetDevice(s.device));
  if (int rc = grow(s, reinterpret_cast<void **>(&s.d_queries), &s.cap_q, (size_t)nq * mx->D * 4)) return rc;
  if (int rc = grow(s, reinterpret_cast<void **>(&s.d_packed), &s.cap_p, 2 * plane * 4)) return rc;
  const bool holds_all = mx->use_rccl || g == 0;
  if (G > 1 && holds_all)
    if (int rc = grow(s, reinterpret_cast<void **>(&s.d_gathered), &s.cap_g, (size_t)G * 2 * plane * 4)) return rc;
  if (g == 0) {
    if (mx->cap_out < 2 * plane * 4) {
      if (mx->d_out_labels) MHIP(hipFree(mx->d_out_labels));
      mx->d_out_labels = nullptr;
      mx->cap_out = 0;
      MHIP(hipMalloc(reinterpret_cast<void **>(&mx->d_out_labels), 2 * plane * 4));
      mx->cap_out = 2 * plane * 4;
    }
    mx->d_out_dist = reinterpret_cast<float *>(mx->d_out_labels + plane);
  }
  // (the previous search's exchange has read this shard's packed result: never recorded = no wait)
  MHIP(hipStreamWaitEvent(s.stream, mx->consumed, 0));
  if (g == 0) MHIP(hipEventRecord(mx->ev[0], s.stream));
  if (mx->d_queries0) {
    // device entry: the queries sit on shard 0's device; every shard takes its copy over the fabric
    MHIP(hipStreamWaitEvent(s.stream, mx->user_ready, 0));
    MHIP(hipMemcpyPeerAsync(s.d_queries, s.device, mx->d_queries0, mx->sh[0].device, (size_t)nq * mx->D * 4, s.stream));
  } else {
    MHIP(hipMemcpyAsync(s.d_queries, mx->queries, (size_t)nq * mx->D * 4, hipMemcpyHostToDevice, s.stream));
  }
  int32_t *labels = G == 1 ? mx->d_out_labels : s.d_packed;
  float *dist = G == 1 ? mx->d_out_dist : reinterpret_cast<float *>(s.d_packed + plane);
  const int rc = vaqhip_search_device(s.ix, s.d_queries, nq, k, mx->projected, labels, dist, s.stream);
  if (rc) {
    s.err = vaqhip_last_error();
    return rc;
  }
  if (g == 0) MHIP(hipEventRecord(mx->ev[1], s.stream));
  MHIP(hipEventRecord(s.done, s.stream));
  return 0;
}

// The exchange step, issued by the CALLING thread once every shard's search is enqueued without error:
// one ncclAllGather per device inside a group (nq * k * 8 bytes per rank over xGMI).  A shard that
// failed has returned before this point and no collective was enqueued anywhere, so nothing can be
// left waiting for a peer that never arrives.
int exchange_rccl(vaqhip_multi *mx) {
  const size_t plane = (size_t)mx->nq * mx->k;
  ncclResult_t nr = g_rccl.GroupStart();
  if (nr != 0) return mfail(VAQHIP_EHIP, "ncclGroupStart: %s", g_rccl.GetErrorString(nr));
  ncclResult_t first = 0;
  for (int g = 0; g < mx->G; g++) {
    Shard &s = mx->sh[g];
    if (hipSetDevice(s.device) != hipSuccess) { first = first ? first : -1; continue; }
    nr = g_rccl.AllGather(s.d_packed, s.d_gathered, 2 * plane, NCCL_INT32, s.comm, s.stream);
    if (nr != 0 && !first) first = nr;
  }
  nr = g_rccl.GroupEnd();
  if (first != 0) return mfail(VAQHIP_EHIP, "ncclAllGather: %s", first > 0 ? g_rccl.GetErrorString(first) : "hipSetDevice");
  if (nr != 0) return mfail(VAQHIP_EHIP, "ncclGroupEnd: %s", g_rccl.GetErrorString(nr));
  return 0;
}

// after every shard has enqueued its part: gather by copies when RCCL is not in play, merge on
// shard 0's device, bring the result to the host
int finish_on_shard0(vaqhip_multi *mx, int32_t *labels, float *distances, hipStream_t user) {
  Shard &s = mx->sh[0];
  const int G = mx->G, nq = mx->nq, k = mx->k;
  const size_t plane = (size_t)nq * k;
  MHIP(hipSetDevice(s.device));
  if (G > 1) {
    if (!mx->use_rccl) {
      for (int g = 0; g < G; g++) {
        MHIP(hipStreamWaitEvent(s.stream, mx->sh[g].done, 0));
        MHIP(hipMemcpyPeerAsync(s.d_gathered + (size_t)g * 2 * plane, s.device, mx->sh[g].d_packed, mx->sh[g].device,
                                2 * plane * 4, s.stream));
      }
    }
    MHIP(hipEventRecord(mx->ev[2], s.stream));
    const int rc = vaqhip_merge_topk_strided_device(
        s.device, reinterpret_cast<const float *>(s.d_gathered + plane), s.d_gathered, G, (int64_t)(2 * plane),
        (int64_t)k, nq, k, mx->d_out_labels, mx->d_out_dist, s.stream);
    if (rc) {
      s.err = vaqhip_last_error();
      return rc;
    }
  } else {
    MHIP(hipEventRecord(mx->ev[2], s.stream));
  }
  MHIP(hipEventRecord(mx->ev[3], s.stream));
  MHIP(hipEventRecord(mx->consumed, s.stream));
  if (mx->d_queries0) {
    // device entry: results into the caller's buffers on shard 0's device; the caller's stream waits
    // for them, the host does not
    MHIP(hipMemcpyAsync(labels, mx->d_out_labels, plane * 4, hipMemcpyDeviceToDevice, s.stream));
    MHIP(hipMemcpyAsync(distances, mx->d_out_dist, plane * 4, hipMemcpyDeviceToDevice, s.stream));
    MHIP(hipEventRecord(mx->finished, s.stream));
    MHIP(hipStreamWaitEvent(user, mx->finished, 0));
    return 0;
  }
  MHIP(hipMemcpyAsync(labels, mx->d_out_labels, plane * 4, hipMemcpyDeviceToHost, s.stream));
  MHIP(hipMemcpyAsync(distances, mx->d_out_dist, plane * 4, hipMemcpyDeviceToHost, s.stream));
  MHIP(hipStreamSynchronize(s.stream));
  for (int g = 1; g < G; g++) {  // (their collective / copies are complete before anyone reuses the buffers)
    MHIP(hipSetDevice(mx->sh[g].device));
    MHIP(hipStreamSynchronize(mx->sh[g].stream));
  }
  MHIP(hipSetDevice(s.device));
  float ms[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) MHIP(hipEventElapsedTime(&ms[i], mx->ev[i], mx->ev[i + 1]));
  mx->last.last_search_ms = ms[0];
  mx->last.last_exchange_ms = ms[1];
  mx->last.last_merge_ms = ms[2];
  return 0;
}

int ensure_comms(vaqhip_multi *mx) {
  if (mx->comms_ready) return 0;
  std::string err;
  if (!load_rccl(&err)) return mfail(VAQHIP_ENODEVICE, "%s", err.c_str());
  std::vector<ncclComm_t> comms(mx->G);
  std::vector<int> devs(mx->G);
  for (int g = 0; g < mx->G; g++) devs[g] = mx->sh[g].device;
  const ncclResult_t nr = g_rccl.CommInitAll(comms.data(), mx->G, devs.data());
  if (nr != 0) return mfail(VAQHIP_EHIP, "ncclCommInitAll(%d devices): %s", mx->G, g_rccl.GetErrorString(nr));
  for (int g = 0; g < mx->G; g++) mx->sh[g].comm = comms[g];
  mx->comms_ready = true;
  return 0;
}

} // namespace

extern "C" {

const char *vaqhip_multi_last_error(void) { return g_merr.c_str(); }

int vaqhip_multi_create(vaqhip_multi **out, int D, int M, const int *bits, const float *const *centroids,
                        const float *eig, int n_devices, const int *device_ids, unsigned flags) {
  if (!out) return mfail(VAQHIP_EINVAL, "out is null");
  *out = nullptr;
  if (n_devices < 1 || n_devices > VAQHIP_MAX_DEVICES || !device_ids)
    return mfail(VAQHIP_EINVAL, "n_devices=%d outside 1..%d (or no device list)", n_devices, VAQHIP_MAX_DEVICES);
  vaqhip_multi *mx = new (std::nothrow) vaqhip_multi();
  if (!mx) return mfail(VAQHIP_ENOMEM, "host allocation");
  mx->D = D;
  mx->M = M;
  mx->G = n_devices;
  mx->sh.resize(n_devices);
  for (int g = 0; g < n_devices; g++)
    for (int h = 0; h < g; h++)
      if (device_ids[g] == device_ids[h]) mx->distinct = false;
  for (int g = 0; g < n_devices; g++) {
    Shard &s = mx->sh[g];
    s.device = device_ids[g];
    const int rc = vaqhip_index_create_ex(&s.ix, D, M, bits, centroids, eig, s.device, flags);
    if (rc) {
      g_merr = vaqhip_last_error();
      vaqhip_multi_destroy(mx);
      return rc;
    }
    bool ok = hipSetDevice(s.device) == hipSuccess && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
    if (g == 0) {
      for (auto &e : mx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&mx->user_ready, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&mx->consumed, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&mx->finished, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
      vaqhip_multi_destroy(mx);
      return mfail(VAQHIP_EHIP, "stream / event creation on device %d failed", s.device);
    }
  }
  mx->pool.start(n_devices);
  *out = mx;
  return VAQHIP_OK;
}

void vaqhip_multi_destroy(vaqhip_multi *mx) {
  if (!mx) return;
  mx->pool.stop();
  for (auto &s : mx->sh) {
    (void)hipSetDevice(s.device);
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s.comm);
    if (s.d_queries) (void)hipFree(s.d_queries);
    if (s.d_packed) (void)hipFree(s.d_packed);
    if (s.d_gathered) (void)hipFree(s.d_gathered);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.ix) vaqhip_index_destroy(s.ix);
  }
  if (!mx->sh.empty()) (void)hipSetDevice(mx->sh[0].device);
  if (mx->d_out_labels) (void)hipFree(mx->d_out_labels);
  for (auto &e : mx->ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {mx->user_ready, mx->consumed, mx->finished})
    if (e) (void)hipEventDestroy(e);
  delete mx;
}

int vaqhip_multi_set_codes_u16(vaqhip_multi *mx, const uint16_t *codes, int64_t N, int64_t id_base) {
  if (!mx) return mfail(VAQHIP_EINVAL, "multi index is null");
  if (N < 0 || (N > 0 && !codes) || id_base < 0) return mfail(VAQHIP_EINVAL, "bad codes/N/id_base");
  std::lock_guard<std::mutex> lk(mx->mu);
  const int64_t per = (N + mx->G - 1) / mx->G;  // contiguous shards of ceil(N / G) rows
  for (int g = 0; g < mx->G; g++) {
    Shard &s = mx->sh[g];
    s.lo = std::min<int64_t>(N, (int64_t)g * per);
    s.n = std::min<int64_t>(N, (int64_t)(g + 1) * per) - s.lo;
  }
  // every shard uploads, sorts and packs its rows on its own device, all of them at once
  const int rc = mx->pool.run([&](int g) -> int {
    Shard &s = mx->sh[g];
    s.err.clear();
    const int r = vaqhip_index_set_codes_u16(s.ix, codes + s.lo * mx->M, s.n, id_base + s.lo);
    if (r) s.err = vaqhip_last_error();
    return r;
  });
  if (rc)
    for (int g = 0; g < mx->G; g++)
      if (mx->pool.rc(g)) return mfail(mx->pool.rc(g), "shard %d (device %d): %s", g, mx->sh[g].device, mx->sh[g].err.c_str());
  mx->N = N;
  mx->id_base = id_base;
  return VAQHIP_OK;
}

int vaqhip_multi_add_codes_u16(vaqhip_multi *mx, const uint16_t *codes, int64_t n_new) {
  if (!mx) return mfail(VAQHIP_EINVAL, "multi index is null");
  if (n_new < 0 || (n_new > 0 && !codes)) return mfail(VAQHIP_EINVAL, "bad codes/N");
  std::lock_guard<std::mutex> lk(mx->mu);
  // labels are global row numbers and shards are contiguous ranges of them, so new rows (which
  // continue the numbering) extend the LAST shard; set_codes re-balances
  Shard &s = mx->sh[mx->G - 1];
  const int rc = vaqhip_index_add_codes_u16(s.ix, codes, n_new);
  if (rc) {
    g_merr = vaqhip_last_error();
    return rc;
  }
  s.n += n_new;
  mx->N += n_new;
  return VAQHIP_OK;
}

int vaqhip_multi_set_ti_clusters(vaqhip_multi *mx, const float *clusters, int T, int seg_num) {
  if (!mx) return mfail(VAQHIP_EINVAL, "multi index is null");
  std::lock_guard<std::mutex> lk(mx->mu);
  for (auto &s : mx->sh) {  // every shard regroups its own rows under the same centres
    const int rc = vaqhip_index_set_ti_clusters(s.ix, clusters, T, seg_num);
    if (rc) {
      g_merr = vaqhip_last_error();
      return rc;
    }
  }
  return VAQHIP_OK;
}

int vaqhip_multi_set_method(vaqhip_multi *mx, unsigned methods, float visit) {
  if (!mx) return mfail(VAQHIP_EINVAL, "multi index is null");
  std::lock_guard<std::mutex> lk(mx->mu);
  for (auto &s : mx->sh) {
    const int rc = vaqhip_index_set_method(s.ix, methods, visit);
    if (rc) {
      g_merr = vaqhip_last_error();
      return rc;
    }
  }
  return VAQHIP_OK;
}

int vaqhip_multi_set_option(vaqhip_multi *mx, const char *key, int64_t value) {
  if (!mx || !key) return mfail(VAQHIP_EINVAL, "null pointer");
  std::lock_guard<std::mutex> lk(mx->mu);
  if (std::strcmp(key, "exchange") == 0) {
    if (value < 0 || value > 2) return mfail(VAQHIP_EINVAL, "exchange must be 0 (auto), 1 (RCCL) or 2 (copies)");
    if (value == EX_RCCL && !mx->distinct)
      return mfail(VAQHIP_EINVAL, "RCCL needs distinct devices (the list names a GPU twice)");
    mx->exchange = (int)value;
    return VAQHIP_OK;
  }
  for (auto &s : mx->sh) {
    const int rc = vaqhip_set_option(s.ix, key, value);
    if (rc) {
      g_merr = vaqhip_last_error();
      return rc;
    }
  }
  return VAQHIP_OK;
}

static int multi_search_common(vaqhip_multi *mx, const float *queries, const float *d_queries0, hipStream_t user, int nq, int k,
                               int projected, int32_t *labels, float *distances) {
  if (!mx) return mfail(VAQHIP_EINVAL, "multi index is null");
  if (nq < 0 || k <= 0) return mfail(VAQHIP_EINVAL, "nq=%d k=%d", nq, k);
  if (nq == 0) return VAQHIP_OK;
  if ((!queries && !d_queries0) || !labels || !distances) return mfail(VAQHIP_EINVAL, "null pointer");
  std::lock_guard<std::mutex> lk(mx->mu);
  // RCCL when the GPUs are distinct and there is something to exchange (or when asked for by
  // option, which also exercises it on one device); device-to-device copies otherwise
  bool rccl = mx->exchange == EX_RCCL || (mx->exchange == EX_AUTO && mx->distinct && mx->G > 1);
  if (rccl) {
    const int rc = ensure_comms(mx);
    if (rc) return rc;
  }
  // a one-shard index asked to use RCCL still goes through the collective (G == 1 skips packing)
  mx->use_rccl = rccl && mx->G > 1;
  mx->queries = queries;
  mx->d_queries0 = d_queries0;
  mx->nq = nq;
  mx->k = k;
  mx->projected = projected;
  if (d_queries0) {
    if (hipSetDevice(mx->sh[0].device) != hipSuccess || hipEventRecord(mx->user_ready, user) != hipSuccess)
      return mfail(VAQHIP_EHIP, "recording the caller's stream");
  }
  // phase 1: every shard uploads (or copies) the queries and enqueues its search
  const int rc1 = mx->pool.run([&](int g) -> int {
    mx->sh[g].err.clear();
    return run_shard(mx, g);
  });
  if (rc1) {
    // Nothing of the exchange has been enqueued: the shards that did succeed have complete, ordinary
    // work on their streams, and the index stays usable (and destroyable).
    for (int g = 0; g < mx->G; g++)
      if (mx->pool.rc(g))
        return mfail(mx->pool.rc(g), "shard %d (device %d): %s", g, mx->sh[g].device, mx->sh[g].err.c_str());
  }
  // phase 2: the exchange, only now that every shard is known to take part
  if (mx->use_rccl) {
    const int rc = exchange_rccl(mx);
    if (rc) return rc;
  }
  if (rccl && mx->G == 1) {
    // one rank: the collective degenerates to a copy; run it anyway so that a one-GPU box
    // proves the RCCL binding (communicator, stream, datatype) end to end
    Shard &s = mx->sh[0];
    const size_t plane = (size_t)nq * k;
    if (hipSetDevice(s.device) != hipSuccess) return mfail(VAQHIP_EHIP, "hipSetDevice");
    if (grow(s, reinterpret_cast<void **>(&s.d_gathered), &s.cap_g, 2 * plane * 4))
      return mfail(VAQHIP_ENOMEM, "%s", s.err.c_str());
    const ncclResult_t nr = g_rccl.AllGather(mx->d_out_labels, s.d_gathered, 2 * plane, NCCL_INT32, s.comm, s.stream);
    if (nr != 0) return mfail(VAQHIP_EHIP, "ncclAllGather: %s", g_rccl.GetErrorString(nr));
    if (hipMemcpyAsync(mx->d_out_labels, s.d_gathered, 2 * plane * 4, hipMemcpyDeviceToDevice, s.stream) != hipSuccess)
      return mfail(VAQHIP_EHIP, "copy back from the gathered buffer");
  }
  Shard &s0 = mx->sh[0];
  const int rc = finish_on_shard0(mx, labels, distances, user);
  if (rc) return mfail(rc, "exchange / merge on device %d: %s", s0.device, s0.err.c_str());
  mx->last.exchange = rccl ? EX_RCCL : (mx->G == 1 ? 0 : EX_COPIES);
  return VAQHIP_OK;
}

int vaqhip_multi_search(vaqhip_multi *mx, const float *queries, int nq, int k, int projected, int32_t *labels,
                        float *distances) {
  return multi_search_common(mx, queries, nullptr, nullptr, nq, k, projected, labels, distances);
}

int vaqhip_multi_search_device(vaqhip_multi *mx, const float *d_queries, int nq, int k, int projected, int32_t *d_labels,
                               float *d_distances, void *stream) {
  return multi_search_common(mx, nullptr, d_queries, static_cast<hipStream_t>(stream), nq, k, projected, d_labels,
                             d_distances);
}

int vaqhip_multi_get_info(const vaqhip_multi *mx, vaqhip_multi_info *out) {
  if (!mx || !out) return mfail(VAQHIP_EINVAL, "null pointer");
  std::lock_guard<std::mutex> lk(mx->mu);  // (a search in flight writes these)
  *out = mx->last;
  out->n_devices = mx->G;
  out->N = mx->N;
  out->id_base = mx->id_base;
  for (int g = 0; g < VAQHIP_MAX_DEVICES; g++) {
    out->device_ids[g] = g < mx->G ? mx->sh[g].device : -1;
    out->shard_rows[g] = g < mx->G ? mx->sh[g].n : 0;
  }
  return VAQHIP_OK;
}

vaqhip_index *vaqhip_multi_shard(vaqhip_multi *mx, int g) {
  if (!mx || g < 0 || g >= mx->G) return nullptr;
  return mx->sh[g].ix;
}

} // extern "C"
