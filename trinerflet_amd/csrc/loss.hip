// loss.hip -- background mix + MSE + its gradient in one pass over the rays (gfx950).
//
// Replaces the elementwise chain between composite forward and backward of one training step
// (reconstruction/nerf/renderer.py:317 `image + (1 - weights_sum) * bg_color`; nerf/utils.py:595,633
// `criterion(pred_rgb, gt_rgb).mean(-1)` ... `.mean()`; GradScaler.scale(loss).backward()): ten small torch kernels
// over [N,3] tensors become one launch.  68 B read + 28 B written per ray.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trinerflet_hip.h"

namespace {

__global__ void __launch_bounds__(256)
k_mse_loss(const float* __restrict__ image, const float* __restrict__ ws, const float* __restrict__ gt, float bg,
           const float* __restrict__ bg_rays, uint32_t N, float inv_norm, const float* __restrict__ scale_dev,
           float* __restrict__ pred, float* __restrict__ g_pred, float* __restrict__ g_ws, float* __restrict__ mse) {
  const uint32_t n = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (n < N) {
    const float scale = scale_dev != nullptr ? scale_dev[0] : 1.f;
    const float w = ws[n];
    float gw = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float b = bg_rays != nullptr ? bg_rays[(size_t)n * 3 + k] : bg;
      const float p = image[(size_t)n * 3 + k] + (1.f - w) * b;
      const float d = p - gt[(size_t)n * 3 + k];
      const float g = d * (2.f * inv_norm) * scale;       // d(scale * mean(d^2)) / d pred
      pred[(size_t)n * 3 + k] = p;
      g_pred[(size_t)n * 3 + k] = g;
      gw -= g * b;                                         // pred depends on weights_sum through -(bg)
      acc += d * d;
    }
    g_ws[n] = gw;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(mse, (part[0] + part[1] + part[2] + part[3]) * inv_norm);
}

// ---- mean |x| and its gradient: the wavelet L1 regulariser's term for one coefficient tensor (nerf/utils.py:639-655:
// val.abs().mean()) as one read pass forward and one read + one write pass backward, instead of torch's abs, mean, sign,
// mul (3.3 ms per step at the base configuration's 403 M coefficients through autograd).
constexpr int ABS_BLOCKS = 2048;

__global__ void __launch_bounds__(256)
k_abs_sum(const float* __restrict__ x, uint64_t n, double* __restrict__ partial) {
  const uint64_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float acc = 0.f;
  // a contiguous chunk per workgroup (fixed partition: the result does not depend on scheduling)
  const uint64_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 v = x4[i];
    acc += (fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w));
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < (n & 3)) acc += fabsf(x[n4 * 4 + threadIdx.x]);
  double d = (double)acc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ void __launch_bounds__(256)
k_abs_mean_finish(const double* __restrict__ partial, int nb, double inv_n, float* __restrict__ out) {
  double d = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) d += partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((part[0] + part[1]) + (part[2] + part[3])) * inv_n);
}

__global__ void __launch_bounds__(256)
k_abs_mean_bwd(const float* __restrict__ x, uint64_t n, const float* __restrict__ g, float inv_n, float* __restrict__ grad) {
  const float s = g[0] * inv_n;
  const uint64_t n4 = n / 4;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* o4 = reinterpret_cast<float4*>(grad);
  auto sg = [s](float v) { return v > 0.f ? s : (v < 0.f ? -s : (v == 0.f ? 0.f : v)); };   // torch.sign: sign(0) = 0, NaN stays NaN
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
    const float4 v = x4[i];
    o4[i] = make_float4(sg(v.x), sg(v.y), sg(v.z), sg(v.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) grad[n4 * 4 + threadIdx.x] = sg(x[n4 * 4 + threadIdx.x]);
}

// ---- GradScaler's inf check (torch/amp/grad_scaler.py: _amp_foreach_non_finite_check_and_unscale_ with a scale of 1)
// as a read-only pass: that kernel also writes every gradient back (8 B per element, 0.9 ms per step at the base
// configuration); this one reads 4 B per element.
__global__ void __launch_bounds__(256)
k_nonfinite_scan(const float* __restrict__ x, uint64_t n, float* __restrict__ found_inf) {
  const uint64_t n4 = n / 4;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4* x4 = reinterpret_cast<const u4*>(x);
  uint32_t bad = 0;
  const uint64_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  auto test = [](uint32_t b) { return (uint32_t)((b & 0x7f800000u) == 0x7f800000u); };   // inf or nan
  uint64_t i = lo + threadIdx.x;
  for (; i + 768 < hi; i += 1024) {
    const u4 a = __builtin_nontemporal_load(x4 + i), b = __builtin_nontemporal_load(x4 + i + 256),
             c = __builtin_nontemporal_load(x4 + i + 512), d = __builtin_nontemporal_load(x4 + i + 768);
    bad |= test(a.x) | test(a.y) | test(a.z) | test(a.w) | test(b.x) | test(b.y) | test(b.z) | test(b.w) |
           test(c.x) | test(c.y) | test(c.z) | test(c.w) | test(d.x) | test(d.y) | test(d.z) | test(d.w);
  }
  for (; i < hi; i += 256) {
    const u4 a = x4[i];
    bad |= test(a.x) | test(a.y) | test(a.z) | test(a.w);
  }
  if (blockIdx.x == 0)
    for (uint64_t k = n4 * 4 + threadIdx.x; k < n; k += 256) bad |= test(__float_as_uint(x[k]));
  if (__ballot(bad != 0) != 0ull && (threadIdx.x & 63) == 0) found_inf[0] = 1.f;   // same value from every writer
}

}  // namespace

extern "C" int tnl_nonfinite_check(const float* x, uint64_t n, float* found_inf, void* stream) {
  if (n == 0) return 0;
  if (((uintptr_t)x & 15) != 0 || found_inf == nullptr) return (int)hipErrorInvalidValue;
  const uint64_t n4 = n / 4;
  int nb = (int)((n4 + 1023) / 1024);
  nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
  hipLaunchKernelGGL(k_nonfinite_scan, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, found_inf);
  return (int)hipGetLastError();
}

extern "C" uint64_t tnl_abs_mean_workspace(void) { return (uint64_t)ABS_BLOCKS * sizeof(double); }

extern "C" int tnl_abs_mean_forward(const float* x, uint64_t n, void* workspace, float* out, void* stream) {
  if (n == 0 || ((uintptr_t)x & 15) != 0) return (int)hipErrorInvalidValue;
  const uint64_t n4 = n / 4;
  int nb = (int)((n4 + 1023) / 1024);
  nb = nb < 1 ? 1 : (nb > ABS_BLOCKS ? ABS_BLOCKS : nb);
  hipLaunchKernelGGL(k_abs_sum, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, reinterpret_cast<double*>(workspace));
  hipLaunchKernelGGL(k_abs_mean_finish, dim3(1), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const double*>(workspace), nb, 1.0 / (double)n, out);
  return (int)hipGetLastError();
}

extern "C" int tnl_abs_mean_backward(const float* x, uint64_t n, const float* grad_out, float* grad_x, void* stream) {
  if (n == 0 || ((uintptr_t)x & 15) != 0 || ((uintptr_t)grad_x & 15) != 0) return (int)hipErrorInvalidValue;
  const uint64_t n4 = n / 4;
  int nb = (int)((n4 + 255) / 256);
  nb = nb < 1 ? 1 : (nb > 8192 ? 8192 : nb);
  hipLaunchKernelGGL(k_abs_mean_bwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, grad_out, (float)(1.0 / (double)n),
                     grad_x);
  return (int)hipGetLastError();
}

extern "C" int tnl_mse_loss(const float* image, const float* weights_sum, const float* gt_rgb, float bg_color,
                            const float* bg_rays, uint32_t N, float inv_norm, const float* scale_dev, float* pred,
                            float* grad_pred, float* grad_weights_sum, float* mse_accum, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_mse_loss, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, image, weights_sum, gt_rgb,
                     bg_color, bg_rays, N, inv_norm, scale_dev, pred, grad_pred, grad_weights_sum, mse_accum);
  return (int)hipGetLastError();
}
