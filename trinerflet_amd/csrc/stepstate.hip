// stepstate.hip -- the scalar bookkeeping of one optimisation step as three launches (gfx950).
//
// The reference leaves this to torch.cuda.amp.GradScaler and a handful of tensor expressions
// (reconstruction/nerf/utils.py:1158-1166 `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`,
// :641-655 the wavelet L1 term): about two dozen 5-us launches per step, each a dependent hop on the stream.
//   prologue: zero the step's accumulators and the MLP gradient, 1 / loss scale
//   probe:    found_inf of GradScaler.unscale_ for the gradients that are not checked inside other kernels
//   epilogue: optimiser-step count, GradScaler.update() (torch/amp/grad_scaler.py `_amp_update_scale_`), reg term
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/trinerflet_hip.h"

namespace {

__global__ void __launch_bounds__(256)
k_step_prologue(const float* __restrict__ scale, float* __restrict__ inv_scale, float* __restrict__ abs_sum,
                int32_t* __restrict__ nonfinite, float* __restrict__ mse, float* __restrict__ grad, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) grad[i] = 0.f;
  if (i == 0) {
    inv_scale[0] = 1.f / scale[0];
    abs_sum[0] = 0.f;
    nonfinite[0] = 0;
    mse[0] = 0.f;
  }
}

// probe = sum |g| (+inf when the flag is set): finite <=> every gradient is finite.  One workgroup, fixed order.
__global__ void __launch_bounds__(1024)
k_scaler_probe(const float* __restrict__ g0, uint32_t n0, const float* __restrict__ g1, uint32_t n1,
               const int32_t* __restrict__ flag, float* __restrict__ probe, float* __restrict__ found_inf) {
  float s = 0.f;
  for (uint32_t i = threadIdx.x; i < n0; i += 1024) s += fabsf(g0[i]);
  for (uint32_t i = threadIdx.x; i < n1; i += 1024) s += fabsf(g1[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  __shared__ float part[16];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; k++) t += part[k];
    if (flag != nullptr && flag[0] != 0) t = INFINITY;
    probe[0] = t;
    found_inf[0] = isfinite(t) ? 0.f : 1.f;
  }
}

__global__ void k_step_epilogue(const float* __restrict__ found_inf, float* __restrict__ opt_steps,
                                float* __restrict__ scale, int32_t* __restrict__ growth_tracker, float growth,
                                float backoff, int growth_interval, int update_scale,
                                const float* __restrict__ abs_sum, float l1_coef, float* __restrict__ reg) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool inf = found_inf[0] != 0.f;
  if (!inf) opt_steps[0] += 1.f;
  if (update_scale) {
    if (inf) {
      scale[0] *= backoff;
      growth_tracker[0] = 0;
    } else {
      const int ok = growth_tracker[0] + 1;
      if (ok == growth_interval) {
        const float ns = scale[0] * growth;
        if (isfinite(ns)) scale[0] = ns;
        growth_tracker[0] = 0;
      } else {
        growth_tracker[0] = ok;
      }
    }
  }
  reg[0] = abs_sum != nullptr ? abs_sum[0] * l1_coef : 0.f;
}

// Streaming copy, 16 bytes per lane, one 4-KB piece per workgroup (6.3 TB/s in tools/micro/copy_shapes.py):
// bench.py measures with it what this box's memory system gives a plain copy, to print beside the 8 TB/s spec.
typedef float v4f_probe __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
k_copy_probe(const v4f_probe* __restrict__ src, v4f_probe* __restrict__ dst, uint64_t n4, uint32_t per_block) {
  uint64_t i = (uint64_t)blockIdx.x * per_block + threadIdx.x;
  const uint64_t end = min(n4, ((uint64_t)blockIdx.x + 1) * per_block);
  for (; i < end; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}

}  // namespace

extern "C" {

int tnl_step_prologue(const float* scale, float* inv_scale, float* abs_sum, int32_t* nonfinite, float* mse,
                      float* small_grad, uint32_t n, void* stream) {
  hipLaunchKernelGGL(k_step_prologue, dim3(n / 256 + 1), dim3(256), 0, (hipStream_t)stream, scale, inv_scale, abs_sum,
                     nonfinite, mse, small_grad, n);
  return (int)hipGetLastError();
}

int tnl_scaler_probe(const float* g0, uint32_t n0, const float* g1, uint32_t n1, const int32_t* nonfinite, float* probe,
                     float* found_inf, void* stream) {
  hipLaunchKernelGGL(k_scaler_probe, dim3(1), dim3(1024), 0, (hipStream_t)stream, g0, n0, g1, n1, nonfinite, probe,
                     found_inf);
  return (int)hipGetLastError();
}

int tnl_step_epilogue(const float* found_inf, float* opt_steps, float* scale, int32_t* growth_tracker, float growth,
                      float backoff, int32_t growth_interval, int32_t update_scale, const float* abs_sum, float l1_coef,
                      float* reg, void* stream) {
  hipLaunchKernelGGL(k_step_epilogue, dim3(1), dim3(64), 0, (hipStream_t)stream, found_inf, opt_steps, scale,
                     growth_tracker, growth, backoff, (int)growth_interval, (int)update_scale, abs_sum, l1_coef, reg);
  return (int)hipGetLastError();
}

}  // extern "C"

extern "C" int tnl_copy_probe(const void* src, void* dst, uint64_t bytes, void* stream) {
  const uint64_t n4 = bytes / 16;
  if (n4 == 0) return 0;
  const uint32_t per_block = 256;                            // one 4-KB piece per workgroup: the shape that streams fastest
  const uint64_t blocks = (n4 + per_block - 1) / per_block;
  hipLaunchKernelGGL(k_copy_probe, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const v4f_probe*>(src), reinterpret_cast<v4f_probe*>(dst), n4, per_block);
  return (int)hipGetLastError();
}

