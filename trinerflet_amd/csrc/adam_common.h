// adam_common.h -- the Adam(+L1) element update shared by adam.hip and the fused adjoint-IDWT epilogue.
// Follows torch.optim.Adam's single-tensor path operation by operation:
//   m.lerp_(g, 1-b1) ; v.mul_(b2).addcmul_(g, g, 1-b2) ; denom = sqrt(v)/sqrt(bc2) + eps ; p.addcdiv_(m, denom, -lr/bc1)
#pragma once
#include <hip/hip_runtime.h>

struct AdamArgs {
  float step_size, bias2_sqrt, beta1, beta2, eps, inv_scale, l1_coef;
};

__device__ __forceinline__ float adam_sgn(float x) { return (x > 0.f) - (x < 0.f); }

__device__ __forceinline__ void adam1(float& p, float g_in, float& m, float& v, const AdamArgs& a, float& abs_acc) {
  abs_acc += fabsf(p);
  const float g = g_in * a.inv_scale + a.l1_coef * adam_sgn(p);
  m = m + (g - m) * (1.f - a.beta1);
  v = v * a.beta2 + (1.f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);
}
