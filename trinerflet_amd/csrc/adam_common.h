// adam_common.h -- the Adam(+L1) element update shared by adam.hip and the fused adjoint-IDWT epilogue.
// Follows torch.optim.Adam's single-tensor path operation by operation:
//   m.lerp_(g, 1-b1) ; v.mul_(b2).addcmul_(g, g, 1-b2) ; denom = sqrt(v)/sqrt(bc2) + eps ; p.addcdiv_(m, denom, -lr/bc1)
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct AdamArgs {
  float step_size, bias2_sqrt, beta1, beta2, eps, inv_scale, l1_coef;
  float omb1, omb2;   // 1 - beta1, 1 - beta2 as torch forms them: in double, from the decimal beta, then to fp32
  double b1d, b2d;    // the decimal betas in double, for the bias corrections 1 - beta^t (Python doubles in torch)
};

// torch.optim.Adam takes its betas as Python doubles and hands (1 - beta) to the fp32 kernels as a double scalar:
// 1 - 0.99 = 0.010000000000000009 -> 0.01f.  The C ABI carries the betas as floats, and 1.f - 0.99f = 0.0099999905 --
// 9.3e-7 off, in every coefficient's second moment.  The float is taken back to the decimal it was written as (7
// significant digits, the shortest that round-trips for any beta a user types), the subtraction is done in double.
static inline double adam_decimal(float beta) {
  char buf[32];
  snprintf(buf, sizeof(buf), "%.7g", (double)beta);
  return strtod(buf, nullptr);
}
static inline float adam_one_minus(float beta) { return (float)(1.0 - adam_decimal(beta)); }
static inline AdamArgs make_adam_args(float step_size, float bias2_sqrt, float beta1, float beta2, float eps, float inv_scale,
                                      float l1_coef) {
  return AdamArgs{step_size, bias2_sqrt, beta1, beta2, eps, inv_scale, l1_coef, adam_one_minus(beta1), adam_one_minus(beta2),
                  adam_decimal(beta1), adam_decimal(beta2)};
}

// one slot of the step ring (k_adam_record): the step's bias-corrected scalars, GradScaler's verdict, and (optim.FusedAdamL1)
// a folded L1 coefficient
struct AdamStepRec { float step_size, bias2_sqrt, skip, pad; };

__device__ __forceinline__ float adam_sgn(float x) { return (x > 0.f) - (x < 0.f); }
// c * sign(x) for c >= 0 (sign(+-0) = 0): the magnitude with x's sign bit, or zero -- three instructions
__device__ __forceinline__ float adam_signed(float c, float x) { return x == 0.f ? 0.f : copysignf(c, x); }

// x / y for finite, non-zero y: hardware reciprocal and one Newton step on the quotient (within 1 ulp of the correctly
// rounded quotient, 4 instructions instead of the 10 of the IEEE sequence).  The replay of deferred steps
// (k_adam_l1_catchup) is ALU-bound -- two divisions and a square root per coefficient and step -- and every Adam kernel
// of this library must compute the same bits, so all of them divide this way; denominators here are
// sqrt(1 - beta2^t) in (0, 1] and sqrt(v) / that + eps >= eps > 0.
__device__ __forceinline__ float adam_div(float x, float y) {
  const float r = __builtin_amdgcn_rcpf(y);
  const float q = x * r;
  return fmaf(fmaf(-y, q, x), r, q);
}

// Every operation is rounded on its own (no multiply-add contraction inside this function beyond the two explicit
// ones of adam_div), so the update does not depend on which kernel it was inlined into -- the live-rectangle pass and
// the replay of deferred steps (k_adam_l1_live, k_adam_l1_catchup) must reproduce the whole-level pass bit for bit.
// sqrt is the hardware's (1 ulp).
__device__ __forceinline__ void adam1(float& p, float g_in, float& m, float& v, const AdamArgs& a, float& abs_acc) {
#pragma clang fp contract(off)
  abs_acc += fabsf(p);
  const float g = g_in * a.inv_scale + adam_signed(a.l1_coef, p);
  m = m + (g - m) * a.omb1;
  v = v * a.beta2 + a.omb2 * g * g;
  const float denom = adam_div(__builtin_amdgcn_sqrtf(v), a.bias2_sqrt) + a.eps;
  p = p - a.step_size * adam_div(m, denom);
}
