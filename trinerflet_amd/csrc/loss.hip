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

}  // namespace

extern "C" int tnl_mse_loss(const float* image, const float* weights_sum, const float* gt_rgb, float bg_color,
                            const float* bg_rays, uint32_t N, float inv_norm, const float* scale_dev, float* pred,
                            float* grad_pred, float* grad_weights_sum, float* mse_accum, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_mse_loss, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, image, weights_sum, gt_rgb,
                     bg_color, bg_rays, N, inv_norm, scale_dev, pred, grad_pred, grad_weights_sum, mse_accum);
  return (int)hipGetLastError();
}
