// Streaming copy with 4-, 8- and 16-byte accesses per lane (micro-benchmark for DESIGN.md: what a kernel whose threads
// own one fp32 column each can reach).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o copy_shapes.so copy_shapes.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
template <typename T>
__global__ void __launch_bounds__(256) k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n, int rows) {
  // each workgroup streams `rows` consecutive pieces of 256 elements, like a column walk over rows
  size_t i = (size_t)blockIdx.x * 256 * rows + threadIdx.x;
  for (int r = 0; r < rows && i < n; r++, i += 256) b[i] = a[i];
}
extern "C" int copy_shape(const void* a, void* b, uint64_t bytes, int width, int rows, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (width == 4) { size_t n = bytes / 4; hipLaunchKernelGGL(k_copy<float>, dim3((n + 256 * rows - 1) / (256 * rows)), dim3(256), 0, st, (const float*)a, (float*)b, n, rows); }
  else if (width == 8) { size_t n = bytes / 8; hipLaunchKernelGGL(k_copy<float2>, dim3((n + 256 * rows - 1) / (256 * rows)), dim3(256), 0, st, (const float2*)a, (float2*)b, n, rows); }
  else { size_t n = bytes / 16; hipLaunchKernelGGL(k_copy<float4>, dim3((n + 256 * rows - 1) / (256 * rows)), dim3(256), 0, st, (const float4*)a, (float4*)b, n, rows); }
  return (int)hipGetLastError();
}
