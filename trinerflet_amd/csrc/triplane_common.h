// triplane_common.h -- shared device helpers: triplane projection + bilinear tap computation.
//
// Follows reconstruction/triplaneencoder/triplane_encoder.py:250-300,314-332 and torch's
// grid_sampler (align_corners=True, padding_mode='border'):
//   u = xyz / bound ; plane0 <- (u.x, u.z), plane1 <- (u.x, u.y), plane2 <- (u.y, u.z)
//   px = ((gx + 1) / 2) * (R - 1), clipped to [0, R-1] ; same for py (grid x -> W, grid y -> H)
//   corners (floor, floor+1) ; the +1 corner is clamped (its weight is 0 when it would fall outside).
#pragma once
#include <hip/hip_runtime.h>

struct TexelTap {
  int x0, y0, x1, y1;
  float w00, w01, w10, w11;  // (y0,x0) (y0,x1) (y1,x0) (y1,x1)
};

// The clipped texel coordinates of a position on one plane (the first half of triplane_tap) ...
__device__ __forceinline__ void triplane_texel(float x, float y, float z, float bound, int R, int plane, float& fx,
                                               float& fy) {
  const float ux = x / bound, uy = y / bound, uz = z / bound;
  const float gx = plane == 2 ? uy : ux;
  const float gy = plane == 1 ? uy : uz;
  const float rm1 = (float)(R - 1);
  fx = ((gx + 1.f) / 2.f) * rm1;
  fy = ((gy + 1.f) / 2.f) * rm1;
  fx = fminf(rm1, fmaxf(fx, 0.f));
  fy = fminf(rm1, fmaxf(fy, 0.f));
}

// ... and corners + weights from them (the second half): tap_from_texel(triplane_texel(...)) == triplane_tap(...), bit
// for bit -- the tile lists of scatter.hip carry (fx, fy) so that the reduction does not gather the position again.
__device__ __forceinline__ void tap_from_texel(float fx, float fy, int R, TexelTap& t) {
  const float flx = floorf(fx), fly = floorf(fy);
  t.x0 = (int)flx;
  t.y0 = (int)fly;
  t.x1 = min(t.x0 + 1, R - 1);
  t.y1 = min(t.y0 + 1, R - 1);
  const float wx = fx - flx, wy = fy - fly;
  t.w00 = (1.f - wx) * (1.f - wy);
  t.w01 = wx * (1.f - wy);
  t.w10 = (1.f - wx) * wy;
  t.w11 = wx * wy;
}

__device__ __forceinline__ void triplane_tap(float x, float y, float z, float bound, int R, int plane, TexelTap& t) {
  const float ux = x / bound, uy = y / bound, uz = z / bound;
  const float gx = plane == 2 ? uy : ux;
  const float gy = plane == 1 ? uy : uz;
  const float rm1 = (float)(R - 1);
  float fx = ((gx + 1.f) / 2.f) * rm1;
  float fy = ((gy + 1.f) / 2.f) * rm1;
  fx = fminf(rm1, fmaxf(fx, 0.f));
  fy = fminf(rm1, fmaxf(fy, 0.f));
  const float flx = floorf(fx), fly = floorf(fy);
  t.x0 = (int)flx;
  t.y0 = (int)fly;
  t.x1 = min(t.x0 + 1, R - 1);
  t.y1 = min(t.y0 + 1, R - 1);
  const float wx = fx - flx, wy = fy - fly;
  t.w00 = (1.f - wx) * (1.f - wy);
  t.w01 = wx * (1.f - wy);
  t.w10 = (1.f - wx) * wy;
  t.w11 = wx * wy;
}
