// rays.hip -- training / evaluation ray batches straight from a device-resident pixel pool (gfx950).
//
// Replaces, per step, the reference's host pipeline: get_rays over whole images (nerf/utils.py:65-149) ->
// concat_data -> shuffle_data (CPU randperm over all B*H*W pixels + a gather of every tensor, utils.py:228-236) ->
// select_batch (slice + H2D copy, utils.py:238-243) -> the background blend of train_step (utils.py:559-577).
// Here poses and images stay on the device; one launch turns N consecutive positions of the epoch's permutation
// into rays_o, rays_d and the blended ground-truth colour.  The permutation is never materialised: position g maps
// to pixel perm(g) through a keyed Feistel bijection of [0, B*H*W) (cycle-walked), a new key per epoch.
// HBM traffic: 36 B written per ray + one 3..16-B pixel read; nothing proportional to the pool size.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trinerflet_hip.h"

namespace {

constexpr int NT = 256;

__host__ __device__ __forceinline__ uint32_t feistel_f(uint32_t x, uint64_t key, uint32_t round) {
  uint32_t h = x * 0x9E3779B1u + (uint32_t)(key >> (16 * (round & 3))) + round * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

__host__ __device__ __forceinline__ uint64_t permute_index(uint64_t g, uint64_t total, uint64_t key, uint32_t half) {
  const uint32_t mask = (uint32_t)((1ull << half) - 1);
  uint64_t v = g;
  do {
    uint32_t l = (uint32_t)(v >> half) & mask, r = (uint32_t)v & mask;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t t = l ^ (feistel_f(r, key, k) & mask);
      l = r; r = t;
    }
    v = ((uint64_t)l << half) | r;
  } while (v >= total);
  return v;
}

struct RayArgs {
  float fx, fy, cx, cy;
  uint32_t H, W;
  uint64_t first, total, key;
  uint32_t half;
  int channels, u8;
  float bg;
};

__global__ void __launch_bounds__(NT)
k_ray_batch(const float* __restrict__ poses, const void* __restrict__ images, const int64_t* __restrict__ pix_in,
            const float* __restrict__ bg_rand, RayArgs a, uint32_t N, float* __restrict__ rays_o,
            float* __restrict__ rays_d, float* __restrict__ gt_rgb, int64_t* __restrict__ pix_out) {
  const uint32_t n = blockIdx.x * NT + threadIdx.x;
  if (n >= N) return;
  uint64_t pix;
  if (pix_in != nullptr) pix = (uint64_t)pix_in[n];
  else if (a.total != 0) pix = permute_index(a.first + n, a.total, a.key, a.half);
  else pix = a.first + n;
  if (pix_out != nullptr) pix_out[n] = (int64_t)pix;
  const uint64_t HW = (uint64_t)a.H * a.W;
  const uint64_t b = pix / HW;
  const uint32_t p = (uint32_t)(pix - b * HW);
  const uint32_t y = p / a.W, x = p - y * a.W;
  // utils.py:82-84,139-143: pixel centre, pinhole direction, normalise, rotate
  const float i = (float)x + 0.5f, j = (float)y + 0.5f;
  const float xs = (i - a.cx) / a.fx, ys = (j - a.cy) / a.fy;
  const float nrm = sqrtf(xs * xs + ys * ys + 1.0f);
  const float d0 = xs / nrm, d1 = ys / nrm, d2 = 1.0f / nrm;
  const float* P = poses + 16 * b;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    rays_d[(size_t)n * 3 + k] = d0 * P[4 * k] + d1 * P[4 * k + 1] + d2 * P[4 * k + 2];
    rays_o[(size_t)n * 3 + k] = P[4 * k + 3];
  }
  if (gt_rgb != nullptr && a.channels >= 3) {
    float c[4] = {0.f, 0.f, 0.f, 1.f};
    const size_t base = (size_t)pix * a.channels;
    if (a.u8) {
      const uint8_t* im = reinterpret_cast<const uint8_t*>(images);
      for (int k = 0; k < a.channels; k++) c[k] = (float)im[base + k] / 255.0f;   // provider.py: image / 255
    } else {
      const float* im = reinterpret_cast<const float*>(images);
      for (int k = 0; k < a.channels; k++) c[k] = im[base + k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float v = c[k];
      if (a.channels == 4) {   // utils.py:576: rgb * alpha + bg * (1 - alpha)
        const float bg = bg_rand != nullptr ? bg_rand[(size_t)n * 3 + k] : a.bg;
        v = c[k] * c[3] + bg * (1.0f - c[3]);
      }
      gt_rgb[(size_t)n * 3 + k] = v;
    }
  }
}

}  // namespace

extern "C" {

uint64_t tnl_permute_index(uint64_t g, uint64_t total, uint64_t key) {
  if (total == 0) return g;
  uint32_t half = 1;
  while ((1ull << (2 * half)) < total) half++;
  return permute_index(g, total, key, half);
}

int tnl_ray_batch(const float* poses, const float* intrinsics_host, uint32_t B, uint32_t H, uint32_t W,
                  const void* images, int channels, int images_u8, const int64_t* pix, uint64_t first,
                  uint64_t perm_total, uint64_t perm_key, uint32_t N, float bg_color, const float* bg_rand,
                  float* rays_o, float* rays_d, float* gt_rgb, int64_t* pix_out, void* stream) {
  if (N == 0) return 0;
  if (poses == nullptr || intrinsics_host == nullptr || H == 0 || W == 0 || B == 0) return (int)hipErrorInvalidValue;
  if (gt_rgb != nullptr && (images == nullptr || (channels != 3 && channels != 4))) return (int)hipErrorInvalidValue;
  const uint64_t pool = (uint64_t)B * H * W;
  if (pix == nullptr) {
    if (perm_total != 0 && perm_total != pool) return (int)hipErrorInvalidValue;
    if (first + N > pool) return (int)hipErrorInvalidValue;
  }
  RayArgs a;
  a.fx = intrinsics_host[0]; a.fy = intrinsics_host[1]; a.cx = intrinsics_host[2]; a.cy = intrinsics_host[3];
  a.H = H; a.W = W; a.first = first; a.total = perm_total; a.key = perm_key;
  a.half = 1;
  while ((1ull << (2 * a.half)) < perm_total) a.half++;
  a.channels = channels; a.u8 = images_u8; a.bg = bg_color;
  hipLaunchKernelGGL(k_ray_batch, dim3((N + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, poses, images, pix, bg_rand,
                     a, N, rays_o, rays_d, gt_rgb, pix_out);
  return (int)hipGetLastError();
}

}  // extern "C"
