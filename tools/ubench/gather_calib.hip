// FETCH_SIZE calibration for the access pattern of k_accum1: each lane reads one random 64-byte
// row (4 x dwordx4) of a 1 GiB table.  Known byte count: rows x 64 B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, unsigned rows_mask, unsigned per_thread, uint4* out) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;
  unsigned x = t * 2654435761u + 12345u;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (unsigned i = 0; i < per_thread; i++) {
    x = x * 1664525u + 1013904223u;
    const uint4* p = table + (size_t)((x >> 4) & rows_mask) * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + b.z + c.w + d.x; acc.z ^= a.z ^ b.w ^ c.x ^ d.y; acc.w += a.w + b.x + c.y + d.z;
  }
  out[t] = acc;
}
__global__ void k_stream(const uint4* __restrict__ table, size_t n16, uint4* out) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; size_t stride = (size_t)gridDim.x * 256;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = t; i < n16; i += stride) { uint4 a = table[i]; acc.x ^= a.x; acc.y += a.y; acc.z ^= a.z; acc.w += a.w; }
  out[t] = acc;
}
int main() {
  size_t rows = 1u << 24;   // 1 GiB table
  uint4 *table, *out; CK(hipMalloc(&table, rows * 64)); CK(hipMalloc(&out, (size_t)(1 << 20) * 16));
  CK(hipMemset(table, 1, rows * 64));
  hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, 0, table, (unsigned)(rows - 1), 16u, out);   // 2^24 row reads = 1 GiB
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, table, rows * 4, out);                       // 1 GiB streamed
  CK(hipDeviceSynchronize());
  printf("k_gather: 2^24 random 64-B rows = %.3f GB expected; k_stream: %.3f GB\n", rows * 64 / 1e9, rows * 64 / 1e9);
  return 0;
}
