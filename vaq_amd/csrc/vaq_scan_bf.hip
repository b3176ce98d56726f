// vaq_scan_bf.hip -- kernels of the best-first scan form (vaq_scan_bf.h) and their dispatch.
#include "vaq_scan_bf.h"

namespace vaq {

// UL0: every row of a bucket shares its first term (bucket key = the whole first code)
template <int M, bool UL0>
__global__ __launch_bounds__(SCAN_MAX_THREADS) VAQ_SCAN_SGPRS void scan_bytes_bf_kernel(ScanParams p) {
  scan_bytes_bf_body<M, UL0>(p);
}

int scan_bf_max_buckets() { return BF_MAX_BUCKETS; }

size_t scan_bf_lds_bytes(int layout, int M, int lut_entries, int k, int nwaves, int n_buckets) {
  int kp = 1;
  while (kp < k) kp <<= 1;
  if (layout == LAYOUT_BYTES) return bf_lds_bytes(M * 256, kp, n_buckets, nwaves, bf_queue_code_words(M));
  return bf_lds_bytes(lut_entries, kp, n_buckets, nwaves, 0);
}

// work units of a slice must fit the 31-bit ticket: always true (rows < 2^31)
bool scan_bf_supported(int layout, int M, int qb, int ea, int n_buckets, int seq) {
  if (layout != LAYOUT_BYTES) return false;
  return qb == 1 && ea == EA_QUEUE && !seq && n_buckets >= 2 && n_buckets <= BF_MAX_BUCKETS;
}

#define VAQ_BF_M(A)                                                                              \
  return p.bucket_shift == 0 ? launch_scan_kernel(scan_bytes_bf_kernel<A, true>, p, lds, grid, st) \
                             : launch_scan_kernel(scan_bytes_bf_kernel<A, false>, p, lds, grid, st);

hipError_t launch_scan_bf(const ScanParams &p, int grid, hipStream_t st) {
  if (!scan_bf_supported(p.layout, p.M, p.qb, p.ea, p.n_buckets, p.seq) || p.ti || p.slice_order)
    return hipErrorInvalidValue;
  const size_t lds = scan_bf_lds_bytes(p.layout, p.M, p.lut_lds_entries, p.k, p.nwaves, p.n_buckets);
  switch (p.M) {
  case 8:  VAQ_BF_M(8)
  case 16: VAQ_BF_M(16)
  case 32: VAQ_BF_M(32)
  default: return hipErrorInvalidValue;
  }
}

} // namespace vaq
