// vaq_scan_bits.hip -- the scan kernels for bit-packed codes (any 1..15-bit allocation, W = 1..8
// dwords per row): instantiations of scan_bits_body (vaq_scan.h) and their dispatch.
#include "vaq_scan.h"

namespace vaq {

template <int W, int QB, int EA>
__global__ __launch_bounds__(SCAN_MAX_THREADS) VAQ_SCAN_SGPRS void scan_bits_kernel(ScanParams p) {
  scan_bits_body<W, QB, EA, false, false>(p);
}
// one query per pass: kept within 64 VGPRs so that 8 waves fit a SIMD (the generic build of
// the early-abandon form needs 65 and loses a wave; measured on C3)
template <int W, int EA>
__global__ __launch_bounds__(SCAN_MAX_THREADS) VAQ_SCAN_SGPRS __attribute__((amdgpu_waves_per_eu(8, 8)))
void scan_bits_q1_kernel(ScanParams p) {
  scan_bits_body<W, 1, EA, false, false>(p);
}
template <int W, int QB>
__global__ __launch_bounds__(SCAN_MAX_THREADS) void scan_bits_inplace_kernel(ScanParams p) {
  scan_bits_body<W, QB, EA_INPLACE, false, false>(p);
}
// some LUT tables left in global memory (TAIL): the rarely taken allocations with
// more table entries than LDS holds
template <int W, int QB, int EA>
__global__ __launch_bounds__(SCAN_MAX_THREADS) void scan_bits_tail_kernel(ScanParams p) {
  scan_bits_body<W, QB, EA, true, false>(p);
}
template <int W>
__global__ __launch_bounds__(SCAN_MAX_THREADS) void scan_bits_ti_kernel(ScanParams p) {
  scan_bits_body<W, 1, EA_QUEUE, true, true>(p);
}


#define VAQ_DISPATCH_EA(A, Q)                                                             \
  switch (p.ea) {                                                                         \
  case EA_NONE: return launch_scan_kernel(scan_bits_kernel<A, Q, EA_NONE>, p, lds, grid, st);   \
  case EA_QUEUE: return launch_scan_kernel(scan_bits_kernel<A, Q, EA_QUEUE>, p, lds, grid, st); \
  case EA_INPLACE: return launch_scan_kernel(scan_bits_inplace_kernel<A, Q>, p, lds, grid, st); \
  default: return hipErrorInvalidValue;                                                   \
  }
#define VAQ_DISPATCH_TAIL_EA(A, Q)                                                        \
  switch (p.ea) {                                                                         \
  case EA_NONE: return launch_scan_kernel(scan_bits_tail_kernel<A, Q, EA_NONE>, p, lds, grid, st);       \
  case EA_QUEUE: return launch_scan_kernel(scan_bits_tail_kernel<A, Q, EA_QUEUE>, p, lds, grid, st);     \
  case EA_INPLACE: return launch_scan_kernel(scan_bits_tail_kernel<A, Q, EA_INPLACE>, p, lds, grid, st); \
  default: return hipErrorInvalidValue;                                                   \
  }
#define VAQ_DISPATCH_TAIL(A)                                                              \
  switch (p.qb) {                                                                         \
  case 1: VAQ_DISPATCH_TAIL_EA(A, 1)                                                      \
  case 2: VAQ_DISPATCH_TAIL_EA(A, 2)                                                      \
  default: return hipErrorInvalidValue; /* the host plans Qb <= 2 when tables spill */    \
  }
#define VAQ_DISPATCH_BITS(A)                                                              \
  if (p.qb == 1 && p.ea == EA_NONE) return launch_scan_kernel(scan_bits_q1_kernel<A, EA_NONE>, p, lds, grid, st);   \
  if (p.qb == 1 && p.ea == EA_QUEUE) return launch_scan_kernel(scan_bits_q1_kernel<A, EA_QUEUE>, p, lds, grid, st); \
  switch (p.qb) {                                                                         \
  case 1: return launch_scan_kernel(scan_bits_inplace_kernel<A, 1>, p, lds, grid, st);    \
  case 2: VAQ_DISPATCH_EA(A, 2)                                                           \
  case 4: VAQ_DISPATCH_EA(A, 4)                                                           \
  default: return hipErrorInvalidValue;                                                   \
  }
#define VAQ_FOR_W(MACRO)                                                                  \
  switch (p.W) {                                                                          \
  case 1: MACRO(1)                                                                        \
  case 2: MACRO(2)                                                                        \
  case 3: MACRO(3)                                                                        \
  case 4: MACRO(4)                                                                        \
  case 5: MACRO(5)                                                                        \
  case 6: MACRO(6)                                                                        \
  case 7: MACRO(7)                                                                        \
  case 8: MACRO(8)                                                                        \
  default: return hipErrorInvalidValue;                                                   \
  }
#define VAQ_TI_W(A) return launch_scan_kernel(scan_bits_ti_kernel<A>, p, lds, grid, st);

hipError_t launch_scan_bits(const ScanParams &p, size_t lds, int grid, hipStream_t st) {
  if (p.ti) VAQ_FOR_W(VAQ_TI_W)
  if (p.lds_subs < p.M) VAQ_FOR_W(VAQ_DISPATCH_TAIL)  // some tables stay in global memory
  VAQ_FOR_W(VAQ_DISPATCH_BITS)
}

} // namespace vaq
