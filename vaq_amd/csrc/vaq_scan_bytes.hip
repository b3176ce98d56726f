// vaq_scan_bytes.hip -- the scan kernels for one byte per subspace (every code 8 bits,
// M in {8, 16, 32}): instantiations of scan_bytes_body (vaq_scan.h) and their dispatch.
#include "vaq_scan.h"

namespace vaq {

template <int M, int QB, int EA>
__global__ __launch_bounds__(SCAN_MAX_THREADS) VAQ_SCAN_SGPRS void scan_bytes_kernel(ScanParams p) {
  scan_bytes_body<M, QB, EA, false>(p);
}
template <int M, int QB, bool STREAM>
__global__ __launch_bounds__(SCAN_MAX_THREADS) void scan_bytes_inplace_kernel(ScanParams p) {
  scan_bytes_body<M, QB, EA_INPLACE, false, STREAM>(p);
}
// triangle-inequality form (VAQ::searchTriangleInequality): one query per workgroup,
// survivors queued; the bit-packed one always allows spilled tables
template <int M>
__global__ __launch_bounds__(SCAN_MAX_THREADS) VAQ_SCAN_SGPRS void scan_bytes_ti_kernel(ScanParams p) {
  scan_bytes_body<M, 1, EA_QUEUE, true>(p);
}

#define VAQ_DISPATCH_EA(A, Q)                                                             \
  switch (p.ea) {                                                                         \
  case EA_NONE: return launch_scan_kernel(scan_bytes_kernel<A, Q, EA_NONE>, p, lds, grid, st);   \
  case EA_QUEUE: return launch_scan_kernel(scan_bytes_kernel<A, Q, EA_QUEUE>, p, lds, grid, st); \
  case EA_INPLACE:                                                                        \
    return p.no_skip ? launch_scan_kernel(scan_bytes_inplace_kernel<A, Q, true>, p, lds, grid, st) \
                     : launch_scan_kernel(scan_bytes_inplace_kernel<A, Q, false>, p, lds, grid, st); \
  default: return hipErrorInvalidValue;                                                   \
  }
#define VAQ_DISPATCH_QB(A)                                                                \
  switch (p.qb) {                                                                         \
  case 1: VAQ_DISPATCH_EA(A, 1)                                                           \
  case 2: VAQ_DISPATCH_EA(A, 2)                                                           \
  case 4: VAQ_DISPATCH_EA(A, 4)                                                           \
  default: return hipErrorInvalidValue;                                                   \
  }

hipError_t launch_scan_bytes(const ScanParams &p, size_t lds, int grid, hipStream_t st) {
  if (p.ti) {
    switch (p.M) {
    case 8:  return launch_scan_kernel(scan_bytes_ti_kernel<8>, p, lds, grid, st);
    case 16: return launch_scan_kernel(scan_bytes_ti_kernel<16>, p, lds, grid, st);
    case 32: return launch_scan_kernel(scan_bytes_ti_kernel<32>, p, lds, grid, st);
    default: return hipErrorInvalidValue;
    }
  }
  switch (p.M) {
  case 8:  VAQ_DISPATCH_QB(8)
  case 16: VAQ_DISPATCH_QB(16)
  case 32: VAQ_DISPATCH_QB(32)
  default: return hipErrorInvalidValue;
  }
}

} // namespace vaq
