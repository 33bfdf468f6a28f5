// triplane.hip -- stand-alone triplane bilinear lookup and its scatter backward (gfx950).
//
// Replaces TriPlaneVolume.sample_from_planes_aux (reconstruction/triplaneencoder/triplane_encoder.py
// :314-332: one-hot matmul projection + F.grid_sample(bilinear, border, align_corners=True) + permute)
// for callers that want the [N,3C] features themselves (TriPlaneVolume.forward, density-grid
// updates).  The training hot path uses the fused field kernels in field.hip instead.
//
// Planes are stored texel-major [3][R][R][C] so that one texel's C channels are contiguous
// (C*e bytes: 64 B at C=32 fp16): consecutive lanes read consecutive channels of the same texel and
// the x+1 neighbour is the next C*e bytes, i.e. a sample touches 2 contiguous segments per plane
// instead of 4*C scattered 4-byte words in the reference's (3,C,R,R) layout.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trinerflet_hip.h"
#include "triplane_common.h"

namespace {

template <bool HALF>
__global__ void __launch_bounds__(256)
k_sample_fwd(const void* __restrict__ planes, const float* __restrict__ xyz, float bound, uint32_t N, int C, int R,
             float* __restrict__ feats) {
  const int F = 3 * C;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)N * F) return;
  const uint32_t i = (uint32_t)(gid / F);
  const int f = (int)(gid - (size_t)i * F);
  const int p = f / C, c = f - p * C;
  TexelTap t;
  triplane_tap(xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2], bound, R, p, t);
  const size_t pb = (size_t)p * R * R;
  const size_t i00 = (pb + (size_t)t.y0 * R + t.x0) * C + c, i01 = (pb + (size_t)t.y0 * R + t.x1) * C + c;
  const size_t i10 = (pb + (size_t)t.y1 * R + t.x0) * C + c, i11 = (pb + (size_t)t.y1 * R + t.x1) * C + c;
  float v00, v01, v10, v11;
  if (HALF) {
    const __half* h = reinterpret_cast<const __half*>(planes);
    v00 = __half2float(h[i00]); v01 = __half2float(h[i01]); v10 = __half2float(h[i10]); v11 = __half2float(h[i11]);
  } else {
    const float* g = reinterpret_cast<const float*>(planes);
    v00 = g[i00]; v01 = g[i01]; v10 = g[i10]; v11 = g[i11];
  }
  feats[gid] = v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
}

__global__ void __launch_bounds__(256)
k_sample_bwd(const float* __restrict__ grad_feats, const float* __restrict__ xyz, float bound, uint32_t N, int C,
             int R, float* __restrict__ grad_tm) {
  const int F = 3 * C;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)N * F) return;
  const uint32_t i = (uint32_t)(gid / F);
  const int f = (int)(gid - (size_t)i * F);
  const int p = f / C, c = f - p * C;
  TexelTap t;
  triplane_tap(xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2], bound, R, p, t);
  const float g = grad_feats[gid];
  const size_t pb = (size_t)p * R * R;
  // consecutive lanes = consecutive channels of one texel: contiguous 4-B atomics
  atomicAdd(grad_tm + (pb + (size_t)t.y0 * R + t.x0) * C + c, g * t.w00);
  atomicAdd(grad_tm + (pb + (size_t)t.y0 * R + t.x1) * C + c, g * t.w01);
  atomicAdd(grad_tm + (pb + (size_t)t.y1 * R + t.x0) * C + c, g * t.w10);
  atomicAdd(grad_tm + (pb + (size_t)t.y1 * R + t.x1) * C + c, g * t.w11);
}

// ---------------------------------------------------------------------------------------------
// General lookup: F.grid_sample(bilinear, border, align_corners=True) on texel-major planes for the optional
// TriPlaneVolume features whose coordinates are not the plain axis projection -- learn_rotation_axis (every channel
// its own rotated axes, triplane_encoder.py:335-362), lbound_auto_scale (per-plane zoom + clamp, :323-326), the
// nested zoom planes (:453-483).  grid: [N][3][CG][2] normalised (gx, gy) in the reference's convention (gx -> W),
// CG = 1 (one coordinate pair per plane) or C (one per channel).  Backward = grid_sampler_2d_backward: atomics into
// the texel-major plane gradient and the gradient w.r.t. the grid (zero where a coordinate was clipped).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void grid_tap(float gx, float gy, int R, TexelTap& t, float& mx, float& my, float& wx,
                                         float& wy) {
  const float rm1 = (float)(R - 1);
  float fx = ((gx + 1.f) / 2.f) * rm1, fy = ((gy + 1.f) / 2.f) * rm1;
  // clip_coordinates_set_grad: the gradient w.r.t. a clipped coordinate is zero
  mx = (fx <= 0.f || fx >= rm1) ? 0.f : rm1 * 0.5f;
  my = (fy <= 0.f || fy >= rm1) ? 0.f : rm1 * 0.5f;
  fx = fminf(rm1, fmaxf(fx, 0.f));
  fy = fminf(rm1, fmaxf(fy, 0.f));
  const float flx = floorf(fx), fly = floorf(fy);
  t.x0 = (int)flx; t.y0 = (int)fly;
  t.x1 = min(t.x0 + 1, R - 1); t.y1 = min(t.y0 + 1, R - 1);
  wx = fx - flx; wy = fy - fly;
  t.w00 = (1.f - wx) * (1.f - wy); t.w01 = wx * (1.f - wy); t.w10 = (1.f - wx) * wy; t.w11 = wx * wy;
}

template <bool HALF>
__device__ __forceinline__ void corners(const void* planes, size_t pb, const TexelTap& t, int R, int C, int c,
                                        float& v00, float& v01, float& v10, float& v11) {
  const size_t i00 = (pb + (size_t)t.y0 * R + t.x0) * C + c, i01 = (pb + (size_t)t.y0 * R + t.x1) * C + c;
  const size_t i10 = (pb + (size_t)t.y1 * R + t.x0) * C + c, i11 = (pb + (size_t)t.y1 * R + t.x1) * C + c;
  if (HALF) {
    const __half* h = reinterpret_cast<const __half*>(planes);
    v00 = __half2float(h[i00]); v01 = __half2float(h[i01]); v10 = __half2float(h[i10]); v11 = __half2float(h[i11]);
  } else {
    const float* g = reinterpret_cast<const float*>(planes);
    v00 = g[i00]; v01 = g[i01]; v10 = g[i10]; v11 = g[i11];
  }
}

template <bool HALF>
__global__ void __launch_bounds__(256)
k_grid_sample_fwd(const void* __restrict__ planes, const float* __restrict__ grid, uint32_t N, int C, int CG, int R,
                  float* __restrict__ feats) {
  const int F = 3 * C;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)N * F) return;
  const uint32_t i = (uint32_t)(gid / F);
  const int f = (int)(gid - (size_t)i * F);
  const int p = f / C, c = f - p * C;
  const float* gp = grid + (((size_t)i * 3 + p) * CG + (CG == 1 ? 0 : c)) * 2;
  TexelTap t;
  float mx, my, wx, wy;
  grid_tap(gp[0], gp[1], R, t, mx, my, wx, wy);
  float v00, v01, v10, v11;
  corners<HALF>(planes, (size_t)p * R * R, t, R, C, c, v00, v01, v10, v11);
  feats[gid] = v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
}

template <bool HALF>
__global__ void __launch_bounds__(256)
k_grid_sample_bwd(const void* __restrict__ planes, const float* __restrict__ grid, const float* __restrict__ grad_feats,
                  uint32_t N, int C, int CG, int R, float* __restrict__ grad_tm, float* __restrict__ grad_grid) {
  const int F = 3 * C;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)N * F) return;
  const uint32_t i = (uint32_t)(gid / F);
  const int f = (int)(gid - (size_t)i * F);
  const int p = f / C, c = f - p * C;
  const size_t go = (((size_t)i * 3 + p) * CG + (CG == 1 ? 0 : c)) * 2;
  TexelTap t;
  float mx, my, wx, wy;
  grid_tap(grid[go], grid[go + 1], R, t, mx, my, wx, wy);
  const float g = grad_feats[gid];
  const size_t pb = (size_t)p * R * R;
  if (grad_tm != nullptr) {
    atomicAdd(grad_tm + (pb + (size_t)t.y0 * R + t.x0) * C + c, g * t.w00);
    atomicAdd(grad_tm + (pb + (size_t)t.y0 * R + t.x1) * C + c, g * t.w01);
    atomicAdd(grad_tm + (pb + (size_t)t.y1 * R + t.x0) * C + c, g * t.w10);
    atomicAdd(grad_tm + (pb + (size_t)t.y1 * R + t.x1) * C + c, g * t.w11);
  }
  if (grad_grid != nullptr) {
    float v00, v01, v10, v11;
    corners<HALF>(planes, pb, t, R, C, c, v00, v01, v10, v11);
    const float dgx = g * ((v01 - v00) * (1.f - wy) + (v11 - v10) * wy) * mx;
    const float dgy = g * ((v10 - v00) * (1.f - wx) + (v11 - v01) * wx) * my;
    if (CG == 1) {   // one coordinate pair per plane: the C channels add up (grad_grid is zero-filled by the caller)
      atomicAdd(grad_grid + go, dgx);
      atomicAdd(grad_grid + go + 1, dgy);
    } else {
      grad_grid[go] = dgx;
      grad_grid[go + 1] = dgy;
    }
  }
}

}  // namespace

extern "C" {

int tnl_grid_sample_tm_forward(const void* planes_tm, int half_in, const float* grid, uint32_t N, uint32_t C,
                               uint32_t CG, uint32_t R, float* feats, void* stream) {
  if (N == 0 || C == 0) return 0;
  if (CG != 1 && CG != C) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)N * 3 * C;
  const dim3 g((unsigned)((total + 255) / 256));
  if (half_in)
    hipLaunchKernelGGL(k_grid_sample_fwd<true>, g, dim3(256), 0, (hipStream_t)stream, planes_tm, grid, N, (int)C,
                       (int)CG, (int)R, feats);
  else
    hipLaunchKernelGGL(k_grid_sample_fwd<false>, g, dim3(256), 0, (hipStream_t)stream, planes_tm, grid, N, (int)C,
                       (int)CG, (int)R, feats);
  return (int)hipGetLastError();
}

int tnl_grid_sample_tm_backward(const void* planes_tm, int half_in, const float* grid, const float* grad_feats,
                                uint32_t N, uint32_t C, uint32_t CG, uint32_t R, float* grad_tm, float* grad_grid,
                                void* stream) {
  if (N == 0 || C == 0) return 0;
  if (CG != 1 && CG != C) return (int)hipErrorInvalidValue;
  const size_t total = (size_t)N * 3 * C;
  const dim3 g((unsigned)((total + 255) / 256));
  if (half_in)
    hipLaunchKernelGGL(k_grid_sample_bwd<true>, g, dim3(256), 0, (hipStream_t)stream, planes_tm, grid, grad_feats, N,
                       (int)C, (int)CG, (int)R, grad_tm, grad_grid);
  else
    hipLaunchKernelGGL(k_grid_sample_bwd<false>, g, dim3(256), 0, (hipStream_t)stream, planes_tm, grid, grad_feats, N,
                       (int)C, (int)CG, (int)R, grad_tm, grad_grid);
  return (int)hipGetLastError();
}

int tnl_triplane_sample_forward(const void* planes_tm, int half_in, const float* xyz, float bound, uint32_t N,
                                uint32_t C, uint32_t R, float* feats, void* stream) {
  if (N == 0 || C == 0) return 0;
  const size_t total = (size_t)N * 3 * C;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (half_in)
    hipLaunchKernelGGL(k_sample_fwd<true>, grid, dim3(256), 0, (hipStream_t)stream, planes_tm, xyz, bound, N, (int)C,
                       (int)R, feats);
  else
    hipLaunchKernelGGL(k_sample_fwd<false>, grid, dim3(256), 0, (hipStream_t)stream, planes_tm, xyz, bound, N, (int)C,
                       (int)R, feats);
  return (int)hipGetLastError();
}

int tnl_triplane_sample_backward(const float* grad_feats, const float* xyz, float bound, uint32_t N, uint32_t C,
                                 uint32_t R, float* grad_tm, void* stream) {
  if (N == 0 || C == 0) return 0;
  const size_t total = (size_t)N * 3 * C;
  const dim3 grid((unsigned)((total + 255) / 256));
  hipLaunchKernelGGL(k_sample_bwd, grid, dim3(256), 0, (hipStream_t)stream, grad_feats, xyz, bound, N, (int)C,
                     (int)R, grad_tm);
  return (int)hipGetLastError();
}

}  // extern "C"
