// vaq_scan_bf.hip -- kernels of the best-first scan form (vaq_scan_bf.h) and their dispatch.
#include "vaq_scan_bf.h"

namespace vaq {

// UL0: every row of a bucket shares its first term (bucket key = the whole first code)
// Residency: eight 4-wave workgroups per CU -- 64 VGPRs (waves_per_eu below), 80 SGPRs (up to 80 a CU
// admits min(8, 800 / (ceil(sgpr / 16) * 16 + 16)) = 8 blocks of 256 threads, MI355X_MICROARCH.md
// "Residency"; 96 would keep ~400 values fewer in VGPR lanes but stop at seven: C2 0.420 ms at seven,
// 0.410 at eight) and 20 336 B of LDS at C2 (vaq_scan_bf.h, VAQ_BF_POOL_MIN).
#ifndef VAQ_BF_WAVES_PER_SIMD
#define VAQ_BF_WAVES_PER_SIMD 8
#endif
#define VAQ_BF_VGPR_CAP __attribute__((amdgpu_waves_per_eu(VAQ_BF_WAVES_PER_SIMD, 8)))
#ifndef VAQ_BF_SGPR_CAP
#define VAQ_BF_SGPR_CAP 80
#endif
#define VAQ_BF_SGPRS __attribute__((amdgpu_num_sgpr(VAQ_BF_SGPR_CAP)))
template <int M, bool UL0>
__global__ __launch_bounds__(SCAN_MAX_THREADS) VAQ_BF_SGPRS VAQ_BF_VGPR_CAP void scan_bytes_bf_kernel(ScanParams p) {
  scan_bf_body<BfBytes<M>, UL0>(p);
}
// bit-packed rows of W dwords; CARRY: the row's last dword rides through the survivor queue
template <int W, bool CARRY, bool UL0>
__global__ __launch_bounds__(SCAN_MAX_THREADS) VAQ_BF_SGPRS VAQ_BF_VGPR_CAP void scan_bits_bf_kernel(ScanParams p) {
  scan_bf_body<BfBits<W, CARRY>, UL0>(p);
}

int scan_bf_queue_words(int layout, int M, int bf_carry) {
  return layout == LAYOUT_BYTES ? bf_queue_code_words(M) : (bf_carry ? 1 : 0);
}

void scan_bf_pool_range(int k, int *lo, int *hi) {
  int kp = 1;
  while (kp < k) kp <<= 1;
  *lo = bf_pool_min(kp);
  *hi = bf_pool_max(kp);
}

size_t scan_bf_lds_bytes(int layout, int M, int lut_entries, int pool, int nwaves, int n_buckets, int bf_carry) {
  return bf_lds_bytes(layout == LAYOUT_BYTES ? M * 256 : lut_entries, pool, n_buckets, nwaves,
                      scan_bf_queue_words(layout, M, bf_carry), layout == LAYOUT_BYTES ? 0 : VAQ_BF_MAX_SUBS);
}

bool scan_bf_supported(int layout, int M, int qb, int ea, int n_buckets, int seq) {
  if (layout == LAYOUT_BYTES && M != 8 && M != 16 && M != 32) return false;
  if (layout == LAYOUT_BITS && (M < 4 || M % 4 != 0)) return false;
  return qb == 1 && ea == EA_QUEUE && !seq && n_buckets >= 16 && n_buckets <= BF_MAX_BUCKETS &&
         (n_buckets & (n_buckets - 1)) == 0;
}

#define VAQ_BF_M(A)                                                                              \
  return p.bucket_shift == 0 ? launch_scan_kernel(scan_bytes_bf_kernel<A, true>, p, lds, grid, st) \
                             : launch_scan_kernel(scan_bytes_bf_kernel<A, false>, p, lds, grid, st);
#define VAQ_BF_W(A)                                                                                           \
  if (p.bf_carry)                                                                                             \
    return p.bucket_shift == 0 ? launch_scan_kernel(scan_bits_bf_kernel<A, true, true>, p, lds, grid, st)      \
                               : launch_scan_kernel(scan_bits_bf_kernel<A, true, false>, p, lds, grid, st);    \
  return p.bucket_shift == 0 ? launch_scan_kernel(scan_bits_bf_kernel<A, false, true>, p, lds, grid, st)       \
                             : launch_scan_kernel(scan_bits_bf_kernel<A, false, false>, p, lds, grid, st);

hipError_t launch_scan_bf(const ScanParams &p, int grid, hipStream_t st) {
  if (!scan_bf_supported(p.layout, p.M, p.qb, p.ea, p.n_buckets, p.seq) || p.ti || p.slice_order)
    return hipErrorInvalidValue;
  if (p.layout == LAYOUT_BITS && (p.lds_subs != p.M || p.lut_lds_entries != p.lut_floats))
    return hipErrorInvalidValue;  // every table must sit in LDS
  int pool_lo, pool_hi;
  scan_bf_pool_range(p.k, &pool_lo, &pool_hi);
  if (p.bf_pool < pool_lo || p.bf_pool > pool_hi || p.bf_pool % 64 != 0) return hipErrorInvalidValue;
  const size_t lds = scan_bf_lds_bytes(p.layout, p.M, p.lut_lds_entries, p.bf_pool, p.nwaves, p.n_buckets, p.bf_carry);
  if (p.layout == LAYOUT_BYTES) {
    switch (p.M) {
    case 8:  VAQ_BF_M(8)
    case 16: VAQ_BF_M(16)
    case 32: VAQ_BF_M(32)
    default: return hipErrorInvalidValue;
    }
  }
  switch (p.W) {
  case 1: VAQ_BF_W(1)
  case 2: VAQ_BF_W(2)
  case 3: VAQ_BF_W(3)
  case 4: VAQ_BF_W(4)
  case 5: VAQ_BF_W(5)
  case 6: VAQ_BF_W(6)
  case 7: VAQ_BF_W(7)
  case 8: VAQ_BF_W(8)
  default: return hipErrorInvalidValue;
  }
}

} // namespace vaq
