// Streaming-read ceiling of the GPU: 16-byte loads, grid-stride, 4 loads in
// flight per lane, result folded into one dword per workgroup.  Calibrates the
// "achievable" line next to the 8 TB/s spec peak in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_kernel(const uint4 *__restrict__ p, size_t n, unsigned *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n; i += stride) { uint4 a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;  // never true for the fill below; keeps loads alive
}

// the scan kernels' shape: every workgroup streams ITS OWN contiguous slice, every wave a contiguous
// part of it, 2 loads in flight per lane (what scan_bytes_inplace_kernel does on a 1B-row index)
__global__ __launch_bounds__(512) void read_sliced_kernel(const uint4 *__restrict__ p, size_t n, unsigned *out) {
  const size_t per_wg = n / gridDim.x;
  const int nw = blockDim.x / 64, wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  const size_t per_wave = per_wg / nw;
  const uint4 *q = p + (size_t)blockIdx.x * per_wg + (size_t)wave * per_wave + lane;
  unsigned acc = 0;
  size_t i = 0;
  for (; i + 64 < per_wave; i += 128) {
    uint4 a = q[i], b = q[i + 64];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
// the same bytes per workgroup, but the workgroups' chunks interleaved at 64 KB (chunk c of
// workgroup g at (c * gridDim + g) * 64 KB): neighbouring workgroups read neighbouring memory
__global__ __launch_bounds__(512) void read_interleaved_kernel(const uint4 *__restrict__ p, size_t n, unsigned *out) {
  const size_t chunk = 4096;  // uint4s = 64 KB
  const size_t nchunks = n / chunk / gridDim.x;
  unsigned acc = 0;
  for (size_t c = 0; c < nchunks; c++) {
    const uint4 *q = p + (c * gridDim.x + blockIdx.x) * chunk + threadIdx.x;
    for (size_t i = 0; i + blockDim.x < chunk; i += 2 * blockDim.x) {
      uint4 a = q[i], b = q[i + blockDim.x];
      acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main() {
  const size_t bytes = (size_t)16 << 30;
  uint4 *p; unsigned *out;
  if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 1 << 20);
  hipMemset(p, 1, bytes);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    read_kernel<<<blocks, 256>>>(p, bytes / 16, out);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int it = 0; it < 5; it++) read_kernel<<<blocks, 256>>>(p, bytes / 16, out);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    printf("blocks %5d: %.0f GB/s\n", blocks, bytes * 5.0 / (ms * 1e-3) / 1e9);
  }
  for (int blocks : {2048, 4096}) {
    for (int mode = 0; mode < 2; mode++) {
      auto launch = [&]() {
        if (mode == 0) read_sliced_kernel<<<blocks, 512>>>(p, bytes / 16, out);
        else read_interleaved_kernel<<<blocks, 512>>>(p, bytes / 16, out);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(s);
      for (int it = 0; it < 5; it++) launch();
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      printf("%s, %d workgroups of 8 waves: %.0f GB/s\n", mode == 0 ? "contiguous slice per workgroup (scan shape)" : "64 KB chunks interleaved over workgroups", blocks, bytes * 5.0 / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
