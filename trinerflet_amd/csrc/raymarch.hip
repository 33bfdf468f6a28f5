// raymarch.hip -- occupancy-grid ray marching and alpha compositing for gfx950 (MI355X).
//
// Replaces aux_libs/raymarching/src/raymarching.cu of the reference.  Design notes:
//  * march_rays_train is a deterministic count -> block-scan -> write pipeline (no atomics):
//    rays are packed in ray-id order, one of the arrival orders the reference's atomicAdd
//    packing (raymarching.cu:405-416) can produce.
//  * composite_rays_train_{forward,backward} use ONE 64-lane wavefront per ray: lanes hold
//    consecutive samples (coalesced sigma/rgb/delta loads) and the transmittance recurrence
//    T_i = prod_{j<i}(1-alpha_j) is a wavefront product scan (DPP row shifts via __shfl_up),
//    instead of the reference's serial per-thread walk (raymarching.cu:537-567).
//  * Marching arithmetic is float32 with explicit fmaf where nvcc contracts `a + b*c`; this TU is
//    compiled with -ffp-contract=off so per-ray sample counts are bit-identical to the oracle.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <stdint.h>
// Static wave priority (s_setprio) of the prefetched march passes: on the side stream they share their SIMDs with the step's
// HBM-bound kernels and are short dependent chains -- served first they finish sooner and cost those kernels nothing
// measurable: small 1.948 -> 1.915 ms per step (the forward no longer runs beside the fill pass: 0.40 -> 0.34), base 3.748 ->
// 3.724, large equal (profiles/r06p_ab_wave_priority.txt; the reverse -- priority for the step's kernels -- speeds them by
// 0.1 ms and makes the side chain the critical path: small + 0.11 ms).  0 = none (A/B builds, tools/knob_ci.sh).
#ifndef TNL_SIDE_PRIO
#define TNL_SIDE_PRIO 3
#endif

#include "../../include/trinerflet_hip.h"
#include "bin_common.h"
#include "march_device.h"

namespace {


// ---------------------------------------------------------------------------------------------
// utils
// ---------------------------------------------------------------------------------------------
__global__ void k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                           const float* __restrict__ aabb, uint32_t N, float min_near,
                           float* __restrict__ nears, float* __restrict__ fars) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  if (n >= N) return;
  const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
  const float rdx = 1 / rays_d[n * 3], rdy = 1 / rays_d[n * 3 + 1], rdz = 1 / rays_d[n * 3 + 2];
  const float FMAX = 3.402823466e+38f;
  float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
  if (near > far) { tmp = near; near = far; far = tmp; }
  float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
  if (near_y > far_y) { tmp = near_y; near_y = far_y; far_y = tmp; }
  if (near > far_y || near_y > far) { nears[n] = fars[n] = FMAX; return; }
  if (near_y > near) near = near_y;
  if (far_y < far) far = far_y;
  float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
  if (near_z > far_z) { tmp = near_z; near_z = far_z; far_z = tmp; }
  if (near > far_z || near_z > far) { nears[n] = fars[n] = FMAX; return; }
  if (near_z > near) near = near_z;
  if (far_z < far) far = far_z;
  if (near < min_near) near = min_near;
  nears[n] = near;
  fars[n] = far;
}

__global__ void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                               float radius, uint32_t N, float* __restrict__ coords) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  if (n >= N) return;
  const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
  const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
  const float A = dx * dx + dy * dy + dz * dz;
  const float B = ox * dx + oy * dy + oz * dz;
  const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
  const float t = (-B + sqrtf(B * B - A * Cc)) / A;
  const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
  const float theta = atan2f(sqrtf(x * x + z * z), y);
  const float phi = atan2f(z, x);
  coords[n * 2] = 2 * theta * RPI - 1;
  coords[n * 2 + 1] = phi * RPI;
}

__global__ void k_morton3D(const int* __restrict__ coords, uint32_t N, int* __restrict__ indices) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  if (n >= N) return;
  indices[n] = (int)morton3D_(coords[n * 3], coords[n * 3 + 1], coords[n * 3 + 2]);
}

__global__ void k_morton3D_invert(const int* __restrict__ indices, uint32_t N, int* __restrict__ coords) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  if (n >= N) return;
  const int ind = indices[n];
  coords[n * 3 + 0] = (int)morton3D_invert_((uint32_t)(ind >> 0));
  coords[n * 3 + 1] = (int)morton3D_invert_((uint32_t)(ind >> 1));
  coords[n * 3 + 2] = (int)morton3D_invert_((uint32_t)(ind >> 2));
}

// Bounding box (cell coordinates) of the occupied cells of each cascade of a Morton-ordered bitfield:
// bounds[c] = {min x, min y, min z, max x, max y, max z}; the caller presets {H, H, H, -1, -1, -1}.
__global__ void __launch_bounds__(256)
k_occupancy_bounds(const uint8_t* __restrict__ bitfield, uint32_t bytes_per_cascade, uint32_t cascades,
                   int* __restrict__ bounds) {
  __shared__ int sb[6];
  const uint32_t c = blockIdx.y;
  if (threadIdx.x < 6) sb[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : -1;
  __syncthreads();
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  for (uint32_t n = blockIdx.x * 256 + threadIdx.x; n < bytes_per_cascade; n += gridDim.x * 256) {
    uint32_t b = bitfield[(size_t)c * bytes_per_cascade + n];
    while (b) {
      const uint32_t j = __builtin_ctz(b);
      b &= b - 1;
      const uint32_t cell = 8 * n + j;   // bit j of byte n = cell 8n + j (raymarching.cu:268-289)
      const int x = (int)morton3D_invert_(cell), y = (int)morton3D_invert_(cell >> 1), z = (int)morton3D_invert_(cell >> 2);
      lo[0] = min(lo[0], x); lo[1] = min(lo[1], y); lo[2] = min(lo[2], z);
      hi[0] = max(hi[0], x); hi[1] = max(hi[1], y); hi[2] = max(hi[2], z);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (hi[a] >= 0) { atomicMin(&sb[a], lo[a]); atomicMax(&sb[3 + a], hi[a]); }
  }
  __syncthreads();
  if (threadIdx.x < 3 && sb[3 + threadIdx.x] >= 0) {
    atomicMin(&bounds[c * 6 + threadIdx.x], sb[threadIdx.x]);
    atomicMax(&bounds[c * 6 + 3 + threadIdx.x], sb[3 + threadIdx.x]);
  }
}


// Per plane and 8-texel row group of the R x R planes: the column extent [lo, end) of the bilinear footprints of every
// position inside an occupied cell (any cascade) -- a finer description of where samples can fall than the bounding
// window (a ball fills 78 % of its window).  Same texel mapping as TrainStep._compute_roi: floor of the cell's low /
// high corner, -1 / +3 texels of slack.  ext[(p * G + g) * 2 + {0, 1}], preset by the caller to {INT_MAX, -1}.
// Plane p samples (axis xa[p] -> texel x, axis ya[p] -> texel y) with xa = (0, 0, 1), ya = (2, 1, 2).
__global__ void __launch_bounds__(256)
k_occupancy_rows(const uint8_t* __restrict__ bitfield, uint32_t bytes_per_cascade, uint32_t cascades, int H, float bound,
                 int R, int G, int* __restrict__ ext) {
  extern __shared__ int srow[];                 // [3][G][2]
  for (int i = threadIdx.x; i < 3 * G * 2; i += 256) srow[i] = (i & 1) ? -1 : 0x7fffffff;
  __syncthreads();
  const uint32_t c = blockIdx.y;
  const float sc = fminf((float)(1u << c), bound);
  for (uint32_t n = blockIdx.x * 256 + threadIdx.x; n < bytes_per_cascade; n += gridDim.x * 256) {
    uint32_t b = bitfield[(size_t)c * bytes_per_cascade + n];
    while (b) {
      const uint32_t j = __builtin_ctz(b);
      b &= b - 1;
      const uint32_t cell = 8 * n + j;
      const int cc[3] = {(int)morton3D_invert_(cell), (int)morton3D_invert_(cell >> 1), (int)morton3D_invert_(cell >> 2)};
      int t0[3], t1[3];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float w0 = ((float)cc[a] / (float)H * 2.f - 1.f) * sc, w1 = ((float)(cc[a] + 1) / (float)H * 2.f - 1.f) * sc;
        const float f0 = (fminf(fmaxf(w0 / bound, -1.f), 1.f) + 1.f) * 0.5f * (float)(R - 1);
        const float f1 = (fminf(fmaxf(w1 / bound, -1.f), 1.f) + 1.f) * 0.5f * (float)(R - 1);
        t0[a] = max((int)floorf(f0) - 1, 0);
        t1[a] = min((int)floorf(f1) + 3, R);          // exclusive
      }
#pragma unroll
      for (int p = 0; p < 3; p++) {
        const int xa = p == 2 ? 1 : 0, ya = p == 1 ? 1 : 2;
        for (int g = t0[ya] >> 3; g <= (t1[ya] - 1) >> 3; g++) {
          atomicMin(&srow[(p * G + g) * 2], t0[xa]);
          atomicMax(&srow[(p * G + g) * 2 + 1], t1[xa]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * G; i += 256) {
    if (srow[2 * i + 1] >= 0) {
      atomicMin(&ext[2 * i], srow[2 * i]);
      atomicMax(&ext[2 * i + 1], srow[2 * i + 1]);
    }
  }
}

// One thread packs 4 output bytes from 32 floats read as 8 x float4 (coalesced 128 B per lane).
__global__ void k_packbits(const float* __restrict__ grid, uint32_t N, float thresh,
                           uint8_t* __restrict__ bitfield, const float* __restrict__ thresh_dev) {
  if (thresh_dev != nullptr) thresh = fminf(thresh, thresh_dev[0]);   // min(density_thresh, mean density), renderer.py:533
  const uint32_t q = threadIdx.x + blockIdx.x * blockDim.x;  // group of 4 bytes
  const uint32_t n0 = q * 4;
  if (n0 >= N) return;
  if (n0 + 4 <= N && ((reinterpret_cast<uintptr_t>(grid) & 15) == 0) &&
      ((reinterpret_cast<uintptr_t>(bitfield) & 3) == 0)) {
    const float4* g4 = reinterpret_cast<const float4*>(grid) + (size_t)q * 8;
    uint32_t word = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const float4 a = g4[b * 2], c = g4[b * 2 + 1];
      uint32_t bits = (a.x > thresh ? 1u : 0u) | (a.y > thresh ? 2u : 0u) | (a.z > thresh ? 4u : 0u) |
                      (a.w > thresh ? 8u : 0u) | (c.x > thresh ? 16u : 0u) | (c.y > thresh ? 32u : 0u) |
                      (c.z > thresh ? 64u : 0u) | (c.w > thresh ? 128u : 0u);
      word |= bits << (8 * b);
    }
    reinterpret_cast<uint32_t*>(bitfield)[q] = word;
  } else {
    for (uint32_t n = n0; n < N && n < n0 + 4; n++) {
      uint8_t bits = 0;
      for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > thresh) ? (uint8_t)(1u << i) : 0;
      bitfield[n] = bits;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// training march: count -> (block sums) -> scan + write -> finalize
// ---------------------------------------------------------------------------------------------
constexpr int MARCH_BLOCK = 256;
#ifndef TNL_MARCH_FAST_LANE
#define TNL_MARCH_FAST_LANE 1   // 0: the per-lane count pass through the generic march_run (A/B builds)
#endif
#ifndef TNL_MARCH_WAVE
#define TNL_MARCH_WAVE 1     // 0: the per-lane count pass everywhere (A/B builds)
#endif

__device__ __forceinline__ int wave_incl_scan_add_i(int v, int lane) {
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const int u = __shfl_up(v, off);
    if (lane >= off) v += u;
  }
  return v;
}

// exclusive scan of one int per thread over a 256-thread block; returns block total in *total
__device__ __forceinline__ int block_excl_scan_256(int v, int* smem4, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int incl = wave_incl_scan_add_i(v, lane);
  if (lane == 63) smem4[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int s = smem4[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

// One bit per 64-bit word of the occupancy bitfield (64 cells in Morton order = a 4 x 4 x 4 block): set where the word is not
// zero.  2 cascades x 128^3 cells = 65 536 words = NZ_WORDS x 32 bits (8 KiB: staged in LDS by the per-lane count pass).
#ifndef TNL_MARCH_NZ_FILTER
#define TNL_MARCH_NZ_FILTER 1
#endif
constexpr int NZ_WORDS = 2048;
__global__ void __launch_bounds__(MARCH_BLOCK)
k_nonzero_words(const unsigned long long* __restrict__ grid64, uint32_t n_words, uint32_t* __restrict__ nzmap) {
  const uint32_t w = blockIdx.x * MARCH_BLOCK + threadIdx.x;        // (the launch covers NZ_WORDS * 32 words exactly)
  const unsigned long long b = __ballot(w < n_words && grid64[w] != 0ull);
  if ((threadIdx.x & 63) == 0) {
    nzmap[w >> 5] = (uint32_t)b;
    nzmap[(w >> 5) + 1] = (uint32_t)(b >> 32);
  }
}

// The serial march of one ray on one lane (march_run<false, true, true, REC>) for the fast path of the level arithmetic
// (dt_gamma = 0, <= 2 cascades, H <= 256), with the per-ray constants hoisted, the Morton code from a 256-entry LDS table
// and the exact-by-construction steps folded -- the identities listed in k_march_train_count_wave, the same values bit
// for bit.  A probe is a dependent chain (position -> cell -> word -> bit) at one wave per SIMD: its length is the
// kernel's time (round 5: 789 -> see profiles/r05_*).
struct FastProbe {
  float x, y, z, mipb;
  int nx, ny, nz;
  bool occ;
};

// (round 6, each measured against this form on one box and removed -- docs/EXPERIMENTS.md, profiles/r06k_*, r06i_*: the t record
//  written four values at a time; the empty-cell skip through chain_skip.h (7-14 adds per cell at max_steps 1024: the walk is
//  the shorter program); one probe per trip instead of two; every occupancy word in LDS.  Beside the step's kernels this pass
//  is slowed by sharing the issue ports, and every instruction added to it showed in the small step.)
template <bool REC>
__device__ __forceinline__ uint32_t march_run_fast(const MarchCtx& m, float& t, float far, uint32_t limit, float* trec,
                                                   const uint32_t* __restrict__ lut, const uint32_t* __restrict__ nz) {
#pragma clang fp contract(off)
  const float hH = 0.5f * m.Hf, rH2 = m.rH * 2, hmax = (float)(m.H - 1), dt = m.dt0;
  const int incx = __float_as_uint(m.dx) >> 31 ? 0 : 1, incy = __float_as_uint(m.dy) >> 31 ? 0 : 1,
            incz = __float_as_uint(m.dz) >> 31 ? 0 : 1;
  const uint32_t H3u = m.H * m.H * m.H;
  const unsigned long long* grid64 = reinterpret_cast<const unsigned long long*>(m.grid);
  uint32_t cached_blk = 0xffffffffu;
  unsigned long long cached_bits = 0ull;
  auto probe = [&](float tt_) {
    FastProbe q;
    q.x = clampf_(fmaf(tt_, m.dx, m.ox), -m.bound, m.bound);
    q.y = clampf_(fmaf(tt_, m.dy, m.oy), -m.bound, m.bound);
    q.z = clampf_(fmaf(tt_, m.dz, m.oz), -m.bound, m.bound);
    const bool lv1 = (m.two_levels && fmaxf(fabsf(q.x), fmaxf(fabsf(q.y), fabsf(q.z))) >= 1.0f) || m.level_dt0 > 0;
    q.mipb = lv1 ? m.mb1 : m.mb0;
    const float rb = lv1 ? m.rb1 : m.rb0;
    q.nx = (int)clampf_(fmaf(q.x, rb, 1.0f) * hH, 0.0f, hmax);
    q.ny = (int)clampf_(fmaf(q.y, rb, 1.0f) * hH, 0.0f, hmax);
    q.nz = (int)clampf_(fmaf(q.z, rb, 1.0f) * hH, 0.0f, hmax);
    const uint32_t index = (lv1 ? H3u : 0u) + (lut[q.nx] | (lut[q.ny] << 1) | (lut[q.nz] << 2));
    const uint32_t blk = index >> 6;
    if (blk != cached_blk) {
      // nz (LDS): one bit per 64-cell word of the bitfield, set where the word is not zero (k_nonzero_words) -- an empty
      // 4 x 4 x 4 block of cells costs no trip to L2 / HBM (two thirds of a ray's probes are in empty space)
      cached_bits = (nz == nullptr || ((nz[blk >> 5] >> (blk & 31u)) & 1u)) ? grid64[blk] : 0ull;
      cached_blk = blk;
    }
    q.occ = (cached_bits >> (index & 63u)) & 1ull;
    return q;
  };
  auto skip = [&](const FastProbe& q) {
    const float ex = fmaf((float)(q.nx + incx) * rH2 - 1, q.mipb, -q.x) * m.rdx;
    const float ey = fmaf((float)(q.ny + incy) * rH2 - 1, q.mipb, -q.y) * m.rdy;
    const float ez = fmaf((float)(q.nz + incz) * rH2 - 1, q.mipb, -q.z) * m.rdz;
    const float tt = t + fmaxf(0.0f, fminf(ex, fminf(ey, ez)));
    do { t += dt; } while (t < tt);
  };
  uint32_t step = 0;
  while (t < far && step < limit) {
    const FastProbe a = probe(t);
    const float t1 = t + dt;
    const FastProbe b = probe(t1);
    if (a.occ) {
      if (REC) *trec++ = t;
      t = t1;
      step++;
      if (t < far && step < limit) {
        if (b.occ) {
          if (REC) *trec++ = t;
          t = t + dt;
          step++;
        } else {
          skip(b);
        }
      }
    } else {
      skip(a);
    }
  }
  return step;
}

template <bool WIDE, bool REC>
__global__ void __launch_bounds__(MARCH_BLOCK)
k_march_train_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                    const uint8_t* __restrict__ grid, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t N, uint32_t C, uint32_t H, const float* __restrict__ nears,
                    const float* __restrict__ fars, const float* __restrict__ noises,
                    int* __restrict__ num_steps_out, int* __restrict__ block_sums, float* __restrict__ tbuf,
                    const uint32_t* __restrict__ nzmap = nullptr) {
  if (TNL_SIDE_PRIO) __builtin_amdgcn_s_setprio(TNL_SIDE_PRIO);
  __shared__ int smem4[4];
  __shared__ uint32_t s_lut[MARCH_BLOCK];
  __shared__ uint32_t s_nz[NZ_WORDS];
  const bool fast = WIDE && TNL_MARCH_FAST_LANE && dt_gamma == 0.f && C <= 2 && H <= 256;     // block-uniform
  if (fast) {
    s_lut[threadIdx.x] = expand_bits(threadIdx.x);
    if (nzmap != nullptr)
      for (int k = threadIdx.x; k < NZ_WORDS; k += MARCH_BLOCK) s_nz[k] = nzmap[k];
    __syncthreads();
  }
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  int ns = 0;
  if (n < N) {
    MarchCtx m;
    march_init(m, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid);
    float t = nears[n];
    t = fmaf(clampf_(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
    if (fast)
      ns = (int)march_run_fast<REC>(m, t, fars[n], max_steps, REC ? tbuf + (size_t)n * max_steps : nullptr, s_lut,
                                    nzmap != nullptr ? s_nz : nullptr);
    else
      ns = (int)march_run<false, WIDE, true, REC>(m, t, fars[n], max_steps, nullptr, nullptr, nullptr,
                                                  REC ? tbuf + (size_t)n * max_steps : nullptr);
    num_steps_out[n] = ns;
  }
  int total;
  block_excl_scan_256(ns, smem4, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------
// Count pass, ONE WAVEFRONT PER RAY (round 5).  k_march_train_count above walks a ray on one lane: a serial chain of
// ~60-instruction dependent probes per visited point, 789 us at base with 0.9 waves per SIMD.  With dt_gamma = 0 (every
// README configuration: MarchCtx::fast) the points a ray can visit are a fixed CHAIN t_0, t_1 = fl(t_0 + dt), t_2 = ...
// that does not depend on the occupancy: an occupied point steps to the next chain point, an empty one to the first
// chain point not below its cell's exit (march_skip: `do t += dt while (t < tt)`).  So the 64 lanes take 64 CONSECUTIVE
// chain points: every lane probes its point (position -> cell -> bit) and, where the cell is empty, finds the chain index
// the march would resume at; the visit order is then resolved by a scalar pointer chase over the wave's masks -- a whole
// run of occupied points is one hop (count trailing ones of the ballot), an empty point one v_readlane.  Which points
// are probed does not change a result: the samples are exactly the occupied chain points the serial march visits.
//
// The chain itself is a serial float accumulation.  Inside one binade [2^e, 2^(e+1)) it is an exact arithmetic
// progression: every t there is a multiple of u = 2^(e-23), so fl(t + dt) = t + qu with qu = rn(dt / u) u whenever dt / u
// is not a tie -- qu is measured with one real add (fl(T0 + dt) - T0, exact) and the tie excluded by |dt - qu| != u / 2
// (exact).  A chunk whose 64 points stay in T0's binade (and whose 63 qu is exact) is lane j: T0 + j qu; any other chunk
// (two per ray: t crosses 2 and 4) runs the 64 dependent adds on every lane.  The next chunk's base is always the real
// add t_63 + dt.  Bit-identical counts and records (tests/test_raymarching_gpu.py::test_march_60k_rays_bit_exact).
// ---------------------------------------------------------------------------------------------
template <bool REC>
__global__ void __launch_bounds__(MARCH_BLOCK)
k_march_train_count_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                         const uint8_t* __restrict__ grid, float bound, uint32_t max_steps, uint32_t N, uint32_t C,
                         uint32_t H, const float* __restrict__ nears, const float* __restrict__ fars,
                         const float* __restrict__ noises, int* __restrict__ num_steps_out, float* __restrict__ tbuf) {
#pragma clang fp contract(off)
  __shared__ float s_t[MARCH_BLOCK];        // the chunk's chain points, per wave (read by the same wave only)
  __shared__ uint32_t s_lut[256];            // expand_bits(i): the Morton code of a cell is three look-ups (H <= 256)
  s_lut[threadIdx.x] = expand_bits(threadIdx.x);
  __syncthreads();
  // the ray index is wave-uniform BY CONSTRUCTION (readfirstlane): the ray's constants are scalar loads and the whole
  // visit-order chase below compiles to scalar code (left to its divergence analysis the compiler kept the chase index in
  // a vector register and ran every hop under saved exec masks: ~60 vector instructions per hop, 683 us at base)
  const uint32_t n = blockIdx.x * (MARCH_BLOCK / WAVE) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
  const int lane = threadIdx.x % WAVE;
  if (n >= N) return;
  MarchCtx m;
  march_init(m, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, 0.f, max_steps, C, H, grid);
  const float dt = m.dt0, far = fars[n];
  float T0 = nears[n];
  T0 = fmaf(clampf_(T0 * 0.f, m.dt_min, m.dt_max), noises[n], T0);   // dt_gamma = 0 (k_march_train_count's expression)
  const int limit = (int)max_steps;
  int count = 0;
  float pend = -__builtin_inff();          // a skip that left the previous chunk: resume at the first t >= pend
  float* trec = REC ? tbuf + (size_t)n * max_steps : nullptr;
  const float flane = (float)lane;
  // march_probe / march_skip_target of the fast path (dt_gamma = 0, <= 2 cascades) with the per-ray constants hoisted and
  // the exact-by-construction steps folded -- the same values bit for bit:
  //   0.5f * a * Hf == a * (0.5f * Hf) and v * rH * 2 == v * (2 rH) (a scaling by two commutes with the rounding),
  //   (float)nx + 0.5f + 0.5f * sign(d) == (float)(nx + (d is not negative)) (small integers and halves are exact),
  //   (uint32_t)((float)level * H3) == level ? H^3 : 0
  const float hH = 0.5f * m.Hf, rH2 = m.rH * 2, hmax = (float)(m.H - 1);
  const int incx = __float_as_uint(m.dx) >> 31 ? 0 : 1, incy = __float_as_uint(m.dy) >> 31 ? 0 : 1,
            incz = __float_as_uint(m.dz) >> 31 ? 0 : 1;
  const uint32_t H3u = H * H * H;
  const bool lut = H <= 256;
  const unsigned long long* grid64 = reinterpret_cast<const unsigned long long*>(grid);
  auto uni = [](float x) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
  while (true) {
    // the ray's running state is wave-uniform; saying so (readfirstlane) keeps it in scalar registers and the loops below
    // scalar branches
    T0 = uni(T0);
    pend = uni(pend);
    count = __builtin_amdgcn_readfirstlane(count);
    if (!(T0 < far && count < limit)) break;
    // ---- this lane's chain point
    float tj;
    const uint32_t eb = __float_as_uint(T0) & 0x7f800000u;                    // T0 > 0: its exponent field
    const float T1 = T0 + dt;
    const float qu = T1 - T0;
    const float uh = __uint_as_float(eb - (24u << 23));                      // u / 2
    const float top = __uint_as_float(eb + (1u << 23));                      // 2^(e+1)
    const float p63 = 63.0f * qu;
    const float ru = __uint_as_float((277u << 23) - eb);                       // 1 / u (eb > 30 << 23 below)
    const bool closed = __builtin_amdgcn_readfirstlane((int)(eb > (30u << 23) && (__float_as_uint(T1) & 0x7f800000u) == eb && fabsf(dt - qu) != uh &&
                        63.0f * (qu * ru) < 16777216.0f && T0 + p63 < top)) != 0;
    const float rqu = __builtin_amdgcn_rcpf(qu);        // for the index guess only
    float tlast;
    if (closed) {       // wave-uniform
      tj = T0 + flane * qu;
      tlast = T0 + p63;
    } else {
      float acc = T0;
      tj = T0;
      for (int i = 1; i < WAVE; i++) {
        acc += dt;
        if (lane == i) tj = acc;
      }
      tlast = acc;
    }
    const float Tnext = tlast + dt;
    // ---- probe it
    const bool valid = tj < far;
    const float px = clampf_(fmaf(tj, m.dx, m.ox), -m.bound, m.bound);
    const float py = clampf_(fmaf(tj, m.dy, m.oy), -m.bound, m.bound);
    const float pz = clampf_(fmaf(tj, m.dz, m.oz), -m.bound, m.bound);
    const bool lv1 = (m.two_levels && fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz))) >= 1.0f) || m.level_dt0 > 0;
    const float mipb = lv1 ? m.mb1 : m.mb0, rb = lv1 ? m.rb1 : m.rb0;
    const int nx = (int)clampf_(fmaf(px, rb, 1.0f) * hH, 0.0f, hmax);
    const int ny = (int)clampf_(fmaf(py, rb, 1.0f) * hH, 0.0f, hmax);
    const int nz = (int)clampf_(fmaf(pz, rb, 1.0f) * hH, 0.0f, hmax);
    const uint32_t mort = lut ? (s_lut[nx] | (s_lut[ny] << 1) | (s_lut[nz] << 2)) : morton3D_(nx, ny, nz);
    const uint32_t index = (lv1 ? H3u : 0u) + mort;
    const bool occ = valid && ((grid64[index >> 6] >> (index & 63u)) & 1ull);
    const float ex = fmaf((float)(nx + incx) * rH2 - 1, mipb, -px) * m.rdx;
    const float ey = fmaf((float)(ny + incy) * rH2 - 1, mipb, -py) * m.rdy;
    const float ez = fmaf((float)(nz + incz) * rH2 - 1, mipb, -pz) * m.rdz;
    const float tt = tj + fmaxf(0.0f, fminf(ex, fminf(ey, ez)));
    // first chain point of this chunk not below the cell's exit, > lane (WAVE: it lies behind the chunk): index guess from
    // the progression, settled against the neighbours' actual t (the spacing changes by < 2^-15 relative across a
    // binade, so the guess is off by one at most; a plain search over the chunk's t backs it up)
    int nxt = WAVE;
    s_t[threadIdx.x] = tj;
    __builtin_amdgcn_wave_barrier();          // (one wave writes and reads its 64 entries: LDS keeps a wave's accesses in order)
    if (!(tlast < tt)) {
      const float steps = (tt - tj) * rqu;
      int g = lane + (steps > 80.0f ? 80 : (int)ceilf(steps));
      g = g < lane + 1 ? lane + 1 : (g > WAVE - 1 ? WAVE - 1 : g);
      const float* tw = s_t + (threadIdx.x - lane);
      const float ta = tw[g - 1], tb = tw[g], tc = tw[g + 1 > WAVE - 1 ? WAVE - 1 : g + 1];
      if (g - 1 > lane && ta >= tt && !(g - 2 > lane && tw[g - 2] >= tt)) nxt = g - 1;
      else if (tb >= tt && (g - 1 == lane || ta < tt)) nxt = g;
      else if (g + 1 < WAVE && tc >= tt && tb < tt) nxt = g + 1;
      else {
        for (int i = lane + 1; i < WAVE; i++)
          if (tw[i] >= tt) { nxt = i; break; }
      }
    }
    // ---- the visit order
    const unsigned long long occm = __ballot(occ), validm = __ballot(valid), gem = __ballot(tj >= pend);
    // Empty space is a chain of short hops (a cell is 5-7 chain points long): three rounds of pointer doubling on the
    // vector unit make an empty point's pointer run through up to eight empty points, up to the first point the chase has
    // to look at (occupied, behind far, or behind the chunk: `ttl` is then the exit the last hop carried).
    float ttl = tt;
    {
      const unsigned long long stopm = occm | ~validm;
#pragma unroll
      for (int round = 0; round < 3; round++) {
        const int pa = nxt < WAVE ? nxt : WAVE - 1;
        const int pn = __shfl(nxt, pa);
        const float tn = __shfl(ttl, pa);
        if (nxt < WAVE && !((stopm >> pa) & 1ull)) { nxt = pn; ttl = tn; }
      }
    }
    unsigned long long taken = 0ull;
    int v = __builtin_amdgcn_readfirstlane(gem ? (int)__builtin_ctzll(gem) : WAVE);
    if (gem) pend = -__builtin_inff();         // (a chunk that lies entirely below the exit keeps it)
    bool done = false;
    const int count0 = count;
    while (v < WAVE) {
      if (!((validm >> v) & 1ull)) { done = true; break; }
      if ((occm >> v) & 1ull) {
        const unsigned long long rest = ~(occm >> v);
        int run = rest ? (int)__builtin_ctzll(rest) : WAVE;
        run = run < WAVE - v ? run : WAVE - v;
        const int room = limit - count;
        const bool last = run >= room;
        run = last ? room : run;
        taken |= (run >= 64 ? ~0ull : ((1ull << run) - 1ull)) << v;
        count += run;
        v += run;
        if (last) { done = true; break; }
      } else {
        const int hop = __builtin_amdgcn_readlane(nxt, v);
        if (hop >= WAVE) { pend = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ttl), v)); v = WAVE; }
        else v = hop;
      }
      v = __builtin_amdgcn_readfirstlane(v);
      count = __builtin_amdgcn_readfirstlane(count);
    }
    if (REC && ((taken >> lane) & 1ull))
      trec[count0 + __builtin_popcountll(taken & ((1ull << lane) - 1ull))] = tj;
    if (done) break;
    T0 = Tnext;
  }
  if (lane == 0) num_steps_out[n] = count;
}

// block_sums of k_march_train_count for the per-wave count pass: the sample counts of 256 consecutive rays summed
__global__ void __launch_bounds__(MARCH_BLOCK)
k_march_block_sums(const int* __restrict__ num_steps, uint32_t N, int* __restrict__ block_sums) {
  __shared__ int smem4[4];
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  int total;
  block_excl_scan_256(n < N ? num_steps[n] : 0, smem4, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// MARCH = true: second march of the ray, writing its samples (the two-pass form).  MARCH = false: only the ray records
// {id, offset, count}; the samples follow from the t values the count pass recorded (k_march_train_emit).
template <bool WIDE, bool MARCH>
__global__ void __launch_bounds__(MARCH_BLOCK)
k_march_train_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                    const uint8_t* __restrict__ grid, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* __restrict__ nears,
                    const float* __restrict__ fars, const float* __restrict__ noises,
                    const int* __restrict__ num_steps_in, const int* __restrict__ block_sums,
                    const int* __restrict__ counter, float* __restrict__ xyzs, float* __restrict__ dirs,
                    float* __restrict__ deltas, int* __restrict__ rays) {
  __shared__ int smem4[4];
  __shared__ int s_base;
  // offset of this block = counter[0] + sum of block_sums[0 .. blockIdx.x)
  int part = 0;
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += MARCH_BLOCK) part += block_sums[i];
  int tot;
  block_excl_scan_256(part, smem4, &tot);
  if (threadIdx.x == 0) s_base = tot + counter[0];
  __syncthreads();
  const int base = s_base;
  const int ray_base = counter[1];
  __syncthreads();
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  const int ns = n < N ? num_steps_in[n] : 0;
  int dummy;
  const uint32_t off = (uint32_t)(base + block_excl_scan_256(ns, smem4, &dummy));
  if (n >= N) return;
  int* r = rays + ((size_t)ray_base + n) * 3;
  r[0] = (int)n; r[1] = (int)off; r[2] = ns;
  if (!MARCH) return;
  if (ns == 0) return;
  if (off + (uint32_t)ns > M) {   // dropped by the budget: zero the in-buffer tail (see k_march_train_emit)
    for (uint32_t q = off; q < M; q++) {
      xyzs[(size_t)q * 3 + 0] = 0.f; xyzs[(size_t)q * 3 + 1] = 0.f; xyzs[(size_t)q * 3 + 2] = 0.f;
      dirs[(size_t)q * 3 + 0] = 0.f; dirs[(size_t)q * 3 + 1] = 0.f; dirs[(size_t)q * 3 + 2] = 0.f;
      deltas[(size_t)q * 2 + 0] = 0.f; deltas[(size_t)q * 2 + 1] = 0.f;
    }
    return;
  }
  MarchCtx m;
  march_init(m, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid);
  float t = nears[n];
  t = fmaf(clampf_(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
  march_run<true, WIDE>(m, t, fars[n], (uint32_t)ns, xyzs + (size_t)off * 3, dirs + (size_t)off * 3,
                  deltas + (size_t)off * 2);
}

// Samples from recorded t values: one wavefront per ray, lanes = consecutive samples (coalesced reads of the record and
// writes of the 32 B per sample).  Position, dt and the t-difference are the expressions of march_run's probe() / take()
// on the same operands, so the output is bit-identical to the marching writer's -- without marching a second time.
__global__ void __launch_bounds__(MARCH_BLOCK)
k_march_train_emit(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, float dt_gamma,
                   uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                   const float* __restrict__ nears, const float* __restrict__ noises,
                   const int* __restrict__ counter, const float* __restrict__ tbuf, const int* __restrict__ rays,
                   float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas, int binR,
                   int* __restrict__ bin_counts) {
  if (TNL_SIDE_PRIO) __builtin_amdgcn_s_setprio(TNL_SIDE_PRIO);
  // bin_counts != NULL: the first pass of the plane-gradient tile sort (scatter.hip k_bin<false>) rides along -- here
  // the lanes of a wave ARE consecutive samples of one ray, the case its run aggregation is made for
  const int TNX = binR / TSX, TNY = binR / TSY;
  const int lane = threadIdx.x % WAVE;
  // (round 6, measured and not kept -- docs/EXPERIMENTS.md: the sample rows staged in LDS and written as 16-byte pieces
  //  (399 -> 443 us); the next ray's inputs requested a ray ahead (398 -> 385 us, step unchanged); the tile lists FILLED here
  //  into fixed spans sized from the period's first batch, no scan and no second pass: exact, the interference only moved)
  // (a launch of fewer workgroups than rays walks them with the grid's stride: tnl_march_side_caps)
  for (uint32_t n = blockIdx.x * (MARCH_BLOCK / WAVE) + (threadIdx.x / WAVE); n < N; n += gridDim.x * (MARCH_BLOCK / WAVE)) {
  const int* r = rays + ((size_t)counter[1] + n) * 3;
  const uint32_t off = (uint32_t)r[1];
  const int ns = r[2];
  if (ns == 0) continue;
  if (off + (uint32_t)ns > M) {
    // a ray the sample budget drops: the rows it would have started in stay in the buffer (off < M for at most one
    // such ray) and are consumed as samples of no ray -- zero them, whatever the caller's buffers held
    for (uint32_t q0 = off; q0 < M; q0 += WAVE) {   // uniform trip count: bin_sample shuffles across the wave
      const uint32_t q = q0 + lane;
      const bool live = q < M;
      if (live) {
        xyzs[(size_t)q * 3 + 0] = 0.f; xyzs[(size_t)q * 3 + 1] = 0.f; xyzs[(size_t)q * 3 + 2] = 0.f;
        dirs[(size_t)q * 3 + 0] = 0.f; dirs[(size_t)q * 3 + 1] = 0.f; dirs[(size_t)q * 3 + 2] = 0.f;
        deltas[(size_t)q * 2 + 0] = 0.f; deltas[(size_t)q * 2 + 1] = 0.f;
      }
      if (bin_counts != nullptr) bin_sample<false>(0.f, 0.f, 0.f, live, q, bound, binR, TNX, TNY, bin_counts, nullptr, lane);
    }
    continue;
  }
  MarchCtx m;
  march_init(m, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, nullptr);
  float t_start = nears[n];
  t_start = fmaf(clampf_(t_start * dt_gamma, m.dt_min, m.dt_max), noises[n], t_start);
  const float* tr = tbuf + (size_t)n * max_steps;
  auto step_dt = [&](float t) { return m.fast ? m.dt0 : clampf_(t * m.dt_gamma, m.dt_min, m.dt_max); };
  for (int k0 = 0; k0 < ns; k0 += WAVE) {   // uniform trip count (bin_sample shuffles across the wave)
    const int k = k0 + lane;
    const bool live = k < ns;
    const int kl = live ? k : ns - 1;
    const float t = tr[kl];
    const float dt = step_dt(t);
    float last_t = t_start;
    if (kl > 0) { const float tp = tr[kl - 1]; last_t = tp + step_dt(tp); }
    const float t_next = t + dt;
    const size_t o = (size_t)off + kl;
    const float px = clampf_(fmaf(t, m.dx, m.ox), -m.bound, m.bound);
    const float py = clampf_(fmaf(t, m.dy, m.oy), -m.bound, m.bound);
    const float pz = clampf_(fmaf(t, m.dz, m.oz), -m.bound, m.bound);
    if (live) {
      xyzs[o * 3 + 0] = px; xyzs[o * 3 + 1] = py; xyzs[o * 3 + 2] = pz;
      dirs[o * 3 + 0] = m.dx; dirs[o * 3 + 1] = m.dy; dirs[o * 3 + 2] = m.dz;
      deltas[o * 2 + 0] = dt;
      deltas[o * 2 + 1] = t_next - last_t;
    }
    if (bin_counts != nullptr) bin_sample<false>(px, py, pz, live, (uint32_t)o, bound, binR, TNX, TNY, bin_counts, nullptr, lane);
  }
  }
}

__global__ void k_march_train_finalize(const int* __restrict__ block_sums, uint32_t nblocks, uint32_t N,
                                       int* __restrict__ counter) {
  __shared__ int smem4[4];
  int part = 0;
  for (uint32_t i = threadIdx.x; i < nblocks; i += MARCH_BLOCK) part += block_sums[i];
  int tot;
  block_excl_scan_256(part, smem4, &tot);
  if (threadIdx.x == 0) {
    counter[0] += tot;
    counter[1] += (int)N;
  }
}

// ---------------------------------------------------------------------------------------------
// training composite: one wavefront per ray
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) {
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const float u = __shfl_up(v, off);
    if (lane >= off) v *= u;
  }
  return v;
}
__device__ __forceinline__ float wave_incl_scan_add(float v, int lane) {
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const float u = __shfl_up(v, off);
    if (lane >= off) v += u;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int COMP_BLOCK = 256;  // 4 rays per workgroup

__global__ void __launch_bounds__(COMP_BLOCK)
k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                      const float* __restrict__ deltas, const int* __restrict__ rays, uint32_t M,
                      uint32_t N, float T_thresh, float* __restrict__ weights_sum,
                      float* __restrict__ depth, float* __restrict__ image) {
  const int lane = threadIdx.x & 63;
  const uint32_t n = blockIdx.x * (COMP_BLOCK / WAVE) + (threadIdx.x >> 6);
  if (n >= N) return;
  const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                 num_steps = (uint32_t)rays[n * 3 + 2];
  if (num_steps == 0 || offset + num_steps > M) {
    if (lane == 0) {
      weights_sum[index] = 0; depth[index] = 0;
      image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
    }
    return;
  }
  float T_carry = 1.0f, t_carry = 0.0f;
  float ar = 0, ag = 0, ab = 0, aw = 0, ad = 0;
  for (uint32_t base = 0; base < num_steps; base += WAVE) {
    const uint32_t i = base + lane;
    const bool valid = i < num_steps;
    const size_t s = (size_t)offset + (valid ? i : 0);
    const float sg = valid ? sigmas[s] : 0.f;
    const float2 dl = valid ? reinterpret_cast<const float2*>(deltas)[s] : make_float2(0.f, 0.f);
    const float alpha = 1.0f - __expf(-sg * dl.x);
    const float om = valid ? 1.0f - alpha : 1.0f;
    const float p_incl = wave_incl_scan_mul(om, lane);
    float p_excl = __shfl_up(p_incl, 1);
    if (lane == 0) p_excl = 1.0f;
    const float T_before = T_carry * p_excl;
    const float t_incl = t_carry + wave_incl_scan_add(dl.y, lane);
    // the reference stops after the first sample whose outgoing T drops below T_thresh
    // (raymarching.cu:553); T is non-increasing, so sample i contributes iff T_before >= T_thresh
    const bool active = valid && (T_before >= T_thresh);
    if (active) {
      const float w = alpha * T_before;
      ar = fmaf(w, rgbs[s * 3], ar);
      ag = fmaf(w, rgbs[s * 3 + 1], ag);
      ab = fmaf(w, rgbs[s * 3 + 2], ab);
      ad = fmaf(w, t_incl, ad);
      aw += w;
    }
    T_carry *= __shfl(p_incl, 63);
    t_carry = __shfl(t_incl, 63);
    if (T_carry < T_thresh) break;
  }
  ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); aw = wave_sum(aw); ad = wave_sum(ad);
  if (lane == 0) {
    weights_sum[index] = aw; depth[index] = ad;
    image[index * 3] = ar; image[index * 3 + 1] = ag; image[index * 3 + 2] = ab;
  }
}

__global__ void __launch_bounds__(COMP_BLOCK)
k_composite_train_bwd(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_image,
                      const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                      const float* __restrict__ deltas, const int* __restrict__ rays,
                      const float* __restrict__ weights_sum, const float* __restrict__ image, uint32_t M,
                      uint32_t N, float T_thresh, float* __restrict__ grad_sigmas,
                      float* __restrict__ grad_rgbs) {
  const int lane = threadIdx.x & 63;
  const uint32_t n = blockIdx.x * (COMP_BLOCK / WAVE) + (threadIdx.x >> 6);
  if (n >= N) return;
  const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                 num_steps = (uint32_t)rays[n * 3 + 2];
  if (num_steps == 0 || offset + num_steps > M) {
    // a ray dropped by the sample budget (raymarching.cu:422) may still own the tail [offset, M) of the buffers: its
    // rows get no gradient.  Writing the zeros here lets a caller skip the zero fill of grad_sigmas / grad_rgbs
    // (raymarching.py:283-284) when everything behind the sample count is ignored anyway.
    for (uint32_t s = offset + lane; s < min(offset + num_steps, M); s += WAVE) {
      grad_sigmas[s] = 0.f;
      grad_rgbs[(size_t)s * 3] = 0.f; grad_rgbs[(size_t)s * 3 + 1] = 0.f; grad_rgbs[(size_t)s * 3 + 2] = 0.f;
    }
    return;
  }
  const float gws = grad_weights_sum[index];
  const float gr = grad_image[index * 3], gg = grad_image[index * 3 + 1], gb = grad_image[index * 3 + 2];
  const float ws_final = weights_sum[index];
  const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
  const float ws_term = gws * (1 - ws_final);
  float T_carry = 1.0f, r_carry = 0, g_carry = 0, b_carry = 0;
  bool done = false;
  for (uint32_t base = 0; base < num_steps; base += WAVE) {
    const uint32_t i = base + lane;
    const bool valid = i < num_steps;
    const size_t s = (size_t)offset + (valid ? i : 0);
    if (done) {  // past the early stop: the reference leaves the caller's zero fill in place
      if (valid) {
        grad_sigmas[s] = 0.f;
        grad_rgbs[s * 3] = 0.f; grad_rgbs[s * 3 + 1] = 0.f; grad_rgbs[s * 3 + 2] = 0.f;
      }
      continue;
    }
    const float sg = valid ? sigmas[s] : 0.f;
    const float d0 = valid ? deltas[s * 2] : 0.f;
    const float cr = valid ? rgbs[s * 3] : 0.f, cg = valid ? rgbs[s * 3 + 1] : 0.f,
                cb = valid ? rgbs[s * 3 + 2] : 0.f;
    const float alpha = 1.0f - __expf(-sg * d0);
    const float om = valid ? 1.0f - alpha : 1.0f;
    const float p_incl = wave_incl_scan_mul(om, lane);
    float p_excl = __shfl_up(p_incl, 1);
    if (lane == 0) p_excl = 1.0f;
    const float T_before = T_carry * p_excl;
    const float T_after = T_carry * p_incl;
    const bool active = valid && (T_before >= T_thresh);
    const float w = active ? alpha * T_before : 0.f;
    const float r_incl = r_carry + wave_incl_scan_add(w * cr, lane);
    const float g_incl = g_carry + wave_incl_scan_add(w * cg, lane);
    const float b_incl = b_carry + wave_incl_scan_add(w * cb, lane);
    if (valid) {
      float gs = 0.f;
      if (active)
        gs = d0 * (gr * (T_after * cr - (r_final - r_incl)) + gg * (T_after * cg - (g_final - g_incl)) +
                   gb * (T_after * cb - (b_final - b_incl)) + ws_term);
      grad_sigmas[s] = gs;
      grad_rgbs[s * 3] = gr * w; grad_rgbs[s * 3 + 1] = gg * w; grad_rgbs[s * 3 + 2] = gb * w;
    }
    T_carry *= __shfl(p_incl, 63);
    r_carry = __shfl(r_incl, 63); g_carry = __shfl(g_incl, 63); b_carry = __shfl(b_incl, 63);
    if (T_carry < T_thresh) done = true;
  }
}

// ---------------------------------------------------------------------------------------------
// inference
// ---------------------------------------------------------------------------------------------
template <bool WIDE>
__global__ void k_march_rays(uint32_t n_alive, uint32_t n_step, const int* __restrict__ rays_alive,
                             const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                             const float* __restrict__ rays_d, float bound, float dt_gamma,
                             uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                             const float* __restrict__ fars, float* __restrict__ xyzs,
                             float* __restrict__ dirs, float* __restrict__ deltas,
                             const float* __restrict__ noises) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  if (n >= n_alive) return;
  const int index = rays_alive[n];
  MarchCtx m;
  march_init(m, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
  float t = rays_t[index];
  t = fmaf(clampf_(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
  march_run<true, WIDE, false>(m, t, fars[index], n_step, xyzs + (size_t)n * n_step * 3, dirs + (size_t)n * n_step * 3,
                  deltas + (size_t)n * n_step * 2);
}

__global__ void k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int* __restrict__ rays_alive,
                                 float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                 const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                 float* __restrict__ weights_sum, float* __restrict__ depth,
                                 float* __restrict__ image) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  if (n >= n_alive) return;
  const int index = rays_alive[n];
  sigmas += (size_t)n * n_step;
  rgbs += (size_t)n * n_step * 3;
  deltas += (size_t)n * n_step * 2;
  float t = rays_t[index];
  float weight_sum = weights_sum[index], d = depth[index];
  float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
  uint32_t step = 0;
  while (step < n_step) {
    if (deltas[0] == 0) break;
    const float alpha = 1.0f - __expf(-sigmas[0] * deltas[0]);
    const float T = 1 - weight_sum;
    const float weight = alpha * T;
    weight_sum += weight;
    t += deltas[1];
    d += weight * t;
    r += weight * rgbs[0]; g += weight * rgbs[1]; b += weight * rgbs[2];
    if (T < T_thresh) break;
    sigmas++; rgbs += 3; deltas += 2; step++;
  }
  if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
  weights_sum[index] = weight_sum; depth[index] = d;
  image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// ---- inference loop driven from the device -----------------------------------------------------
// The reference's loop (renderer.py:338-372) reads the survivor count back on the host every iteration to size
// the next launches.  Here the loop state lives on the device: state = {n_alive, n_step, step, rows}; every
// iteration's kernels are launched with the worst-case grid (N rays) and look the live sizes up themselves, so the
// host only polls the state occasionally to stop.  Same arithmetic, same ray order, same n_step rule.
struct InferState {
  int n_alive, n_step, step, rows;
};

__global__ void k_infer_plan(InferState* __restrict__ st, uint32_t N, uint32_t max_steps, int min_step) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n_alive = st->n_alive;
  if (st->step >= (int)max_steps) n_alive = 0;                       // `while step < max_steps`
  int n_step = 1;
  // renderer.py:349 is min_step = 1.  A larger floor regroups the same per-ray sample sequence into fewer, wider
  // iterations: every ray still composites its samples in order and stops at the same one, so a ray that ends before
  // the max_steps cap gets the identical colour; only rays still alive at the cap can see up to 7 samples more or
  // fewer than under the reference's schedule (whose own last iteration overshoots the cap by up to 7 as well).
  if (n_alive > 0) n_step = max(min((int)(N / (uint32_t)n_alive), 8), max(min_step, 1));
  st->n_alive = n_alive;
  st->n_step = n_step;
  st->rows = n_alive * n_step;
}

template <bool WIDE>
__global__ void k_march_rays_dev(const InferState* __restrict__ st, const int* __restrict__ rays_alive,
                                 const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                 const float* __restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                 uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                 const float* __restrict__ fars, float* __restrict__ xyzs, float* __restrict__ dirs,
                                 float* __restrict__ deltas, const float* __restrict__ noises) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  const uint32_t n_alive = (uint32_t)st->n_alive, n_step = (uint32_t)st->n_step;
  if (n >= n_alive) return;
  const int index = rays_alive[n];
  float* xo = xyzs + (size_t)n * n_step * 3;
  float* dro = dirs + (size_t)n * n_step * 3;
  float* dlo = deltas + (size_t)n * n_step * 2;
  // the reference zero-fills the sample buffers every iteration (raymarching.py:337-339): rows a ray does not
  // reach must read deltas == 0 in composite_rays
  for (uint32_t k = 0; k < n_step; k++) {
    xo[3 * k] = 0.f; xo[3 * k + 1] = 0.f; xo[3 * k + 2] = 0.f;
    dro[3 * k] = 0.f; dro[3 * k + 1] = 0.f; dro[3 * k + 2] = 0.f;
    dlo[2 * k] = 0.f; dlo[2 * k + 1] = 0.f;
  }
  MarchCtx m;
  march_init(m, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
  float t = rays_t[index];
  const float nz = noises != nullptr ? noises[n] : 0.f;
  t = fmaf(clampf_(t * dt_gamma, m.dt_min, m.dt_max), nz, t);
  march_run<true, WIDE, false>(m, t, fars[index], n_step, xo, dro, dlo);
}

// The same iteration with line-shaped stores: k_march_rays_dev writes every sample from the lane that marched it, 8 + 8
// scattered 4-byte stores per sample and lane (zero fill + values; each store instruction scatters 64 pieces, each its own
// fabric write) -- 60 % of an 800 x 800 render at max_steps 4096.  Here the march only records the sample's t (staged
// through LDS, written as whole rows of the block), and k_emit_rays_dev computes the samples with one thread per ROW,
// consecutive threads = consecutive rows: every store instruction covers contiguous memory.  The expressions are
// march_run's probe() / take() on the same operands (as in k_march_train_emit): bit-identical rows.
template <bool WIDE>
__global__ void __launch_bounds__(128)
k_march_rays_rec_dev(const InferState* __restrict__ st, const int* __restrict__ rays_alive,
                     const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                     const float* __restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                     uint32_t H, const uint8_t* __restrict__ grid, const float* __restrict__ fars,
                     const float* __restrict__ noises, float* __restrict__ tscr) {
  __shared__ float sh_t[128][9];
  __shared__ int sh_c[128];
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  const uint32_t n_alive = (uint32_t)st->n_alive, n_step = (uint32_t)st->n_step;
  if (blockIdx.x * 128u >= n_alive) return;
  int cnt = 0;
  if (n < n_alive) {
    const int index = rays_alive[n];
    MarchCtx m;
    march_init(m, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
    float t = rays_t[index];
    const float nz = noises != nullptr ? noises[n] : 0.f;
    t = fmaf(clampf_(t * dt_gamma, m.dt_min, m.dt_max), nz, t);
    cnt = (int)march_run<false, WIDE, false, true>(m, t, fars[index], n_step, nullptr, nullptr, nullptr,
                                                   &sh_t[threadIdx.x][0]);
  }
  sh_c[threadIdx.x] = cnt;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * 128 * n_step;
  const uint32_t rows_here = min(128u, n_alive - blockIdx.x * 128u) * n_step;
  for (uint32_t i = threadIdx.x; i < rows_here; i += 128) {
    const uint32_t nl = i / n_step, k = i - nl * n_step;
    tscr[base + i] = (int)k < sh_c[nl] ? sh_t[nl][k] : -1.f;     // t > 0 always (min_near): -1 = the ray has no k-th sample
  }
}

__global__ void __launch_bounds__(256)
k_emit_rays_dev(const InferState* __restrict__ st, const int* __restrict__ rays_alive, const float* __restrict__ rays_t,
                const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound, float dt_gamma,
                uint32_t max_steps, uint32_t C, uint32_t H, const float* __restrict__ noises,
                const float* __restrict__ tscr, float* __restrict__ xyzs, float* __restrict__ dirs,
                float* __restrict__ deltas) {
  const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
  const uint32_t n_step = (uint32_t)st->n_step;
  if (i >= (uint32_t)st->rows) return;
  const float t = tscr[i];
  float px = 0.f, py = 0.f, pz = 0.f, ddx = 0.f, ddy = 0.f, ddz = 0.f, d0 = 0.f, d1 = 0.f;
  if (t >= 0.f) {     // rows a ray does not reach stay zero (raymarching.py:337-339): composite_rays stops at deltas == 0
    const uint32_t n = i / n_step, k = i - n * n_step;
    const int index = rays_alive[n];
    MarchCtx m;
    march_init(m, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, nullptr);
    auto step_dt = [&](float tt) { return m.fast ? m.dt0 : clampf_(tt * m.dt_gamma, m.dt_min, m.dt_max); };
    const float dt = step_dt(t);
    float last_t;
    if (k == 0) {
      last_t = rays_t[index];
      const float nz = noises != nullptr ? noises[n] : 0.f;
      last_t = fmaf(clampf_(last_t * dt_gamma, m.dt_min, m.dt_max), nz, last_t);
    } else {
      const float tp = tscr[i - 1];
      last_t = tp + step_dt(tp);
    }
    px = clampf_(fmaf(t, m.dx, m.ox), -m.bound, m.bound);
    py = clampf_(fmaf(t, m.dy, m.oy), -m.bound, m.bound);
    pz = clampf_(fmaf(t, m.dz, m.oz), -m.bound, m.bound);
    ddx = m.dx; ddy = m.dy; ddz = m.dz;
    d0 = dt;
    d1 = (t + dt) - last_t;
  }
  xyzs[(size_t)i * 3 + 0] = px; xyzs[(size_t)i * 3 + 1] = py; xyzs[(size_t)i * 3 + 2] = pz;
  dirs[(size_t)i * 3 + 0] = ddx; dirs[(size_t)i * 3 + 1] = ddy; dirs[(size_t)i * 3 + 2] = ddz;
  deltas[(size_t)i * 2 + 0] = d0; deltas[(size_t)i * 2 + 1] = d1;
}

__global__ void k_composite_rays_dev(const InferState* __restrict__ st, float T_thresh, int* __restrict__ rays_alive,
                                     float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                     const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                     float* __restrict__ weights_sum, float* __restrict__ depth,
                                     float* __restrict__ image) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  const uint32_t n_alive = (uint32_t)st->n_alive, n_step = (uint32_t)st->n_step;
  if (n >= n_alive) return;
  const int index = rays_alive[n];
  sigmas += (size_t)n * n_step;
  rgbs += (size_t)n * n_step * 3;
  deltas += (size_t)n * n_step * 2;
  float t = rays_t[index];
  float weight_sum = weights_sum[index], d = depth[index];
  float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
  uint32_t step = 0;
  while (step < n_step) {
    if (deltas[0] == 0) break;
    const float alpha = 1.0f - __expf(-sigmas[0] * deltas[0]);
    const float T = 1 - weight_sum;
    const float weight = alpha * T;
    weight_sum += weight;
    t += deltas[1];
    d += weight * t;
    r += weight * rgbs[0]; g += weight * rgbs[1]; b += weight * rgbs[2];
    if (T < T_thresh) break;
    sigmas++; rgbs += 3; deltas += 2; step++;
  }
  if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
  weights_sum[index] = weight_sum; depth[index] = d;
  image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

__global__ void __launch_bounds__(256)
k_compact_count_dev(const InferState* __restrict__ st, const int* __restrict__ rays_alive,
                    int* __restrict__ block_counts) {
  __shared__ int smem4[4];
  const uint32_t n = (uint32_t)st->n_alive;
  const uint32_t i = threadIdx.x + blockIdx.x * 256;
  const int keep = (i < n && rays_alive[i] >= 0) ? 1 : 0;
  int total;
  block_excl_scan_256(keep, smem4, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256)
k_compact_write_dev(const InferState* __restrict__ st, const int* __restrict__ rays_alive,
                    const int* __restrict__ block_counts, int* __restrict__ out, int* __restrict__ next_n) {
  __shared__ int smem4[4];
  __shared__ int s_base;
  const uint32_t n = (uint32_t)st->n_alive;
  int part = 0;
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256) part += block_counts[i];
  int tot;
  block_excl_scan_256(part, smem4, &tot);
  if (threadIdx.x == 0) s_base = tot;
  __syncthreads();
  const int base = s_base;
  __syncthreads();
  const uint32_t i = threadIdx.x + blockIdx.x * 256;
  const int v = i < n ? rays_alive[i] : -1;
  const int keep = v >= 0 ? 1 : 0;
  int total;
  const int off = base + block_excl_scan_256(keep, smem4, &total);
  if (keep) out[off] = v;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *next_n = base + total;   // all blocks beyond n contribute 0
}
// closes the iteration: step += n_step, n_alive = survivors (kept apart from the count so that the kernels of the
// iteration all see one consistent state)
__global__ void k_infer_advance(InferState* __restrict__ st, const int* __restrict__ next_n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (st->n_alive > 0) {
    st->step += st->n_step;
    st->n_alive = *next_n;
  }
}

// order-preserving compaction: per-block survivor counts, then offsets + scatter
__global__ void __launch_bounds__(256)
k_compact_count(const int* __restrict__ rays_alive, uint32_t n, int* __restrict__ block_counts) {
  __shared__ int smem4[4];
  const uint32_t i = threadIdx.x + blockIdx.x * 256;
  const int keep = (i < n && rays_alive[i] >= 0) ? 1 : 0;
  int total;
  block_excl_scan_256(keep, smem4, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256)
k_compact_write(const int* __restrict__ rays_alive, uint32_t n, const int* __restrict__ block_counts,
                int* __restrict__ out, int* __restrict__ n_out) {
  __shared__ int smem4[4];
  __shared__ int s_base;
  int part = 0;
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256) part += block_counts[i];
  int tot;
  block_excl_scan_256(part, smem4, &tot);
  if (threadIdx.x == 0) s_base = tot;
  __syncthreads();
  const int base = s_base;
  __syncthreads();
  const uint32_t i = threadIdx.x + blockIdx.x * 256;
  const int v = i < n ? rays_alive[i] : -1;
  const int keep = v >= 0 ? 1 : 0;
  int total;
  const int off = base + block_excl_scan_256(keep, smem4, &total);
  if (keep) out[off] = v;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = base + total;
}

// ---------------------------------------------------------------------------------------------
// spherical harmonics (aux_libs/shencoder/src/shencoder.cu:28-355).  Degrees <= 4 (the hot path's 16 values) are the
// reference's polynomials term by term; bands l = 4..7 (degrees 5..8) are generated instead of tabulated:
//   (c_m + i s_m) = (x + i y)^m,   Q_l^m(z) = P_l^m(z) / (1 - z^2)^(m/2) by the three-term recurrence,
//   Y_l^m = N_l^m Q_l^m c_m (m > 0),  Y_l^-m = N_l^m Q_l^m s_m,  Y_l^0 = N_l^0 Q_l^0,
//   N_l^m = (-1)^m sqrt(2) K_l^m, N_l^0 = K_l^0, K_l^m = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!)
// which expands to exactly the reference's polynomials (pinned by tests/golden/sh_reference.npz: the reference's own
// statements evaluated in fp32); the optional derivatives come out of the same recurrences carried on
// (value, d/dx, d/dy, d/dz) quadruples.
// ---------------------------------------------------------------------------------------------
struct D4 { float v, x, y, z; };
__device__ __forceinline__ D4 d4mul(const D4& a, const D4& b) {
  return D4{a.v * b.v, a.x * b.v + a.v * b.x, a.y * b.v + a.v * b.y, a.z * b.v + a.v * b.z};
}
__device__ __forceinline__ D4 d4sub(const D4& a, const D4& b) { return D4{a.v - b.v, a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D4 d4add(const D4& a, const D4& b) { return D4{a.v + b.v, a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D4 d4scale(const D4& a, float s) { return D4{a.v * s, a.x * s, a.y * s, a.z * s}; }

// writes entries [16, C*C) of o (and of gx, gy, gz when WITH_GRAD)
template <bool WITH_GRAD>
__device__ __forceinline__ void sh_high_bands(float x, float y, float z, uint32_t C, float* o, float* gx, float* gy,
                                              float* gz) {
  static constexpr float NRM[4][8] = {
    {0.84628437532163447f, -0.26761861742291571f, 0.063078313050504001f, -0.016858388283618388f, 0.0059603403376112026f, 0.f, 0.f, 0.f},   // l = 4
    {0.9356025796273888f, -0.24157154730437169f, 0.045652731285460234f, -0.0093188247511476283f, 0.0021964680580751762f, -0.00069458418713245519f, 0.f, 0.f},   // l = 5
    {1.0171072362820548f, -0.22195099524523101f, 0.03509353369580661f, -0.0058489222826344353f, 0.0010678622237644956f, -0.00022766899107568562f, 6.5722376641838803e-05f, 0.f},   // l = 6
    {1.0925484305920792f, -0.20647224590289676f, 0.028097313806030647f, -0.0039735602250741348f, 0.00059903674311141165f, -9.9839457185235285e-05f, 1.9580128477462541e-05f, -5.233009453691466e-06f},   // l = 7
  };
  const D4 X{x, 1.f, 0.f, 0.f}, Y{y, 0.f, 1.f, 0.f}, Z{z, 0.f, 0.f, 1.f};
  D4 c[8], s[8];
  c[0] = D4{1.f, 0.f, 0.f, 0.f};
  s[0] = D4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 1; m < 8; m++) {
    c[m] = d4sub(d4mul(X, c[m - 1]), d4mul(Y, s[m - 1]));
    s[m] = d4add(d4mul(X, s[m - 1]), d4mul(Y, c[m - 1]));
  }
#pragma unroll
  for (int m = 0; m < 8; m++) {
    // Q_m^m = (2m-1)!!, Q_{m+1}^m = (2m+1) z Q_m^m, Q_l^m = ((2l-1) z Q_{l-1}^m - (l+m-1) Q_{l-2}^m) / (l-m)
    float dfac = 1.f;
    for (int k = 1; k <= m; k++) dfac *= (float)(2 * k - 1);
    D4 q2{dfac, 0.f, 0.f, 0.f};                       // Q_m^m
    D4 q1 = d4scale(Z, (float)(2 * m + 1) * dfac);    // Q_{m+1}^m
#pragma unroll
    for (int l = m; l < 8; l++) {
      D4 q;
      if (l == m) q = q2;
      else if (l == m + 1) q = q1;
      else {
        q = d4scale(d4sub(d4scale(d4mul(Z, q1), (float)(2 * l - 1)), d4scale(q2, (float)(l + m - 1))), 1.f / (float)(l - m));
        q2 = q1;
        q1 = q;
      }
      if (l < 4 || (uint32_t)l >= C) continue;
      const float nrm = NRM[l - 4][m];
      const int base = l * l + l;
      const D4 yp = d4scale(d4mul(q, c[m]), nrm);
      o[base + m] = yp.v;
      if (WITH_GRAD) { gx[base + m] = yp.x; gy[base + m] = yp.y; gz[base + m] = yp.z; }
      if (m > 0) {
        const D4 yn = d4scale(d4mul(q, s[m]), nrm);
        o[base - m] = yn.v;
        if (WITH_GRAD) { gx[base - m] = yn.x; gy[base - m] = yn.y; gz[base - m] = yn.z; }
      }
    }
  }
}

__global__ void k_sh(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, uint32_t C,
                     float* __restrict__ dy_dx) {
  const uint32_t b = threadIdx.x + blockIdx.x * blockDim.x;
  if (b >= B) return;
  const uint32_t C2 = C * C;
  const float x = inputs[b * 3], y = inputs[b * 3 + 1], z = inputs[b * 3 + 2];
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  float o[64];
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  float gx[64], gy[64], gz[64];
  if (C > 4) {   // wave-uniform
    if (dy_dx) sh_high_bands<true>(x, y, z, C, o, gx, gy, gz);
    else sh_high_bands<false>(x, y, z, C, o, gx, gy, gz);
  }
  for (uint32_t i = 0; i < C2; i++) outputs[(size_t)b * C2 + i] = o[i];
  if (dy_dx) {
    // shencoder.cu:125-354, degree <= 4 term by term: d/dx, d/dy, d/dz of each basis function
    gx[0] = 0; gy[0] = 0; gz[0] = 0;
    gx[1] = 0; gy[1] = -0.48860251190291987f; gz[1] = 0;
    gx[2] = 0; gy[2] = 0; gz[2] = 0.48860251190291987f;
    gx[3] = -0.48860251190291987f; gy[3] = 0; gz[3] = 0;
    gx[4] = 1.0925484305920792f * y; gy[4] = 1.0925484305920792f * x; gz[4] = 0;
    gx[5] = 0; gy[5] = -1.0925484305920792f * z; gz[5] = -1.0925484305920792f * y;
    gx[6] = 0; gy[6] = 0; gz[6] = 1.8923493915151202f * z;
    gx[7] = -1.0925484305920792f * z; gy[7] = 0; gz[7] = -1.0925484305920792f * x;
    gx[8] = 1.0925484305920792f * x; gy[8] = -1.0925484305920792f * y; gz[8] = 0;
    gx[9] = -3.5402615395598609f * xy; gy[9] = -1.7701307697799304f * x2 + 1.7701307697799304f * y2; gz[9] = 0;
    gx[10] = 2.8906114426405538f * yz; gy[10] = 2.8906114426405538f * xz; gz[10] = 2.8906114426405538f * xy;
    gx[11] = 0; gy[11] = 0.45704579946446572f - 2.2852289973223288f * z2; gz[11] = -4.5704579946446566f * yz;
    gx[12] = 0; gy[12] = 0; gz[12] = 5.597644988851731f * z2 - 1.1195289977703462f;
    gx[13] = 0.45704579946446572f - 2.2852289973223288f * z2; gy[13] = 0; gz[13] = -4.5704579946446566f * xz;
    gx[14] = 2.8906114426405538f * xz; gy[14] = -2.8906114426405538f * yz; gz[14] = 1.4453057213202769f * x2 - 1.4453057213202769f * y2;
    gx[15] = -1.7701307697799304f * x2 + 1.7701307697799304f * y2; gy[15] = 3.5402615395598609f * xy; gz[15] = 0;
    float* d = dy_dx + (size_t)b * 3 * C2;
    for (uint32_t i = 0; i < C2; i++) { d[i] = gx[i]; d[C2 + i] = gy[i]; d[2 * C2 + i] = gz[i]; }
  }
}

// shencoder.cu:359-382: grad_inputs[b,d] = sum_k grad[b,k] * dy_dx[b,d,k]
__global__ void k_sh_backward(const float* __restrict__ grad, uint32_t B, uint32_t C,
                              const float* __restrict__ dy_dx, float* __restrict__ grad_inputs) {
  const uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
  const uint32_t b = t / 3;
  if (b >= B) return;
  const uint32_t d = t - b * 3, C2 = C * C;
  float r = 0;
  for (uint32_t k = 0; k < C2; k++) r += grad[(size_t)b * C2 + k] * dy_dx[(size_t)b * 3 * C2 + d * C2 + k];
  grad_inputs[(size_t)b * 3 + d] = r;
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline int launch_status() { return (int)hipGetLastError(); }

}  // namespace

// the 64-bit cached lookups need an 8-byte aligned bitfield whose bit count is a multiple of 64
static inline bool wide_bitfield(const uint8_t* grid, uint32_t C, uint32_t H) {
  return (reinterpret_cast<uintptr_t>(grid) & 7) == 0 && ((uint64_t)C * H * H * H) % 64 == 0;
}

extern "C" {

int tnl_abi_version(void) { return 1; }

int tnl_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                           float min_near, float* nears, float* fars, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_near_far, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, aabb, N,
                     min_near, nears, fars);
  return launch_status();
}

int tnl_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                     void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_sph_from_ray, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, radius,
                     N, coords);
  return launch_status();
}

int tnl_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_morton3D, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, coords, N, indices);
  return launch_status();
}

int tnl_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_morton3D_invert, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, indices, N, coords);
  return launch_status();
}

int tnl_occupancy_bounds(const uint8_t* bitfield, uint32_t bytes_per_cascade, uint32_t cascades, int32_t* bounds,
                         void* stream) {
  if (bytes_per_cascade == 0 || cascades == 0) return 0;
  hipLaunchKernelGGL(k_occupancy_bounds, dim3(min(cdiv(bytes_per_cascade, 256u), 256u), cascades), dim3(256), 0,
                     (hipStream_t)stream, bitfield, bytes_per_cascade, cascades, bounds);
  return (int)hipGetLastError();
}

int tnl_occupancy_row_extents(const uint8_t* bitfield, uint32_t bytes_per_cascade, uint32_t cascades, uint32_t H,
                              float bound, uint32_t R, int32_t* ext, void* stream) {
  if (bytes_per_cascade == 0 || cascades == 0) return 0;
  if (R % 8 != 0 || ext == nullptr || H == 0) return (int)hipErrorInvalidValue;
  const int G = (int)(R / 8);
  hipLaunchKernelGGL(k_occupancy_rows, dim3(min(cdiv(bytes_per_cascade, 256u), 128u), cascades), dim3(256),
                     (size_t)3 * G * 2 * sizeof(int), (hipStream_t)stream, bitfield, bytes_per_cascade, cascades, (int)H,
                     bound, (int)R, G, ext);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// No sample can lie outside the box of the occupied cells: a ray that has left it is done, and the march need not probe
// its way through the empty cells between the object and `far` (one dependent bitfield load chain per cell: the slowest
// lane of nearly every wavefront of every inference iteration, and the tail of every training ray).  `far` only enters the
// march's loop conditions, so the callers simply pass min(far, exit of the box): the samples are the same to the bit.
// (The ENTRY into the box cannot be used the same way: where the march lands after a skip depends, in the last bit, on
// the cell it skipped from.)
//   k_occupied_box: world-space box of the occupied cells of all cascades from tnl_occupancy_bounds' cell bounds, grown
//   by one cell; a face within one cell of the volume's boundary is opened to infinity (positions are clamped to the
//   volume there, raymarching.cu:367-369, so a ray never really leaves through it).  No occupied cell: an empty box.
//   k_clip_fars: fars_out[n] = min(fars[n], slab exit of ray n), or -FLT_MAX if the ray misses the box.
// ---------------------------------------------------------------------------------------------
__global__ void k_init_bounds(int* __restrict__ bounds, int cascades, int H) {   // {H + 1 x3, -1 x3} per cascade
  const int i = threadIdx.x;
  if (i < cascades * 6) bounds[i] = (i % 6) < 3 ? H + 1 : -1;
}

__global__ void k_occupied_box(const int* __restrict__ bounds, int cascades, int H, float bound, float* __restrict__ box) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float INF = __builtin_inff();
  float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
  for (int k = 0; k < cascades; k++) {
    if (bounds[k * 6 + 3] < 0) continue;
    const float sk = fminf(exp2f((float)k), bound), cell = 2.f * sk / (float)H;
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], ((float)bounds[k * 6 + a] / (float)H * 2.f - 1.f) * sk - cell);
      hi[a] = fmaxf(hi[a], ((float)(bounds[k * 6 + 3 + a] + 1) / (float)H * 2.f - 1.f) * sk + cell);
    }
  }
  const float edge = 2.f * bound / (float)H;
  for (int a = 0; a < 3; a++) {
    if (lo[a] <= hi[a]) {
      if (lo[a] <= -bound + edge) lo[a] = -INF;
      if (hi[a] >= bound - edge) hi[a] = INF;
    }
    box[a] = lo[a];
    box[3 + a] = hi[a];
  }
}

__global__ void k_clip_fars(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                            const float* __restrict__ fars, const float* __restrict__ box, uint32_t N,
                            float* __restrict__ out) {
  const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
  if (n >= N) return;
  const float FMAX = 3.402823466e+38f;
  float t_in = -FMAX, t_out = FMAX;
  bool miss = false;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float o = rays_o[(size_t)n * 3 + a], d = rays_d[(size_t)n * 3 + a], lo = box[a], hi = box[3 + a];
    if (lo > hi) miss = true;                       // empty box
    if (d == 0.f) {
      if (o < lo || o > hi) miss = true;
    } else {
      const float t1 = (lo - o) / d, t2 = (hi - o) / d;
      t_in = fmaxf(t_in, fminf(t1, t2));
      t_out = fminf(t_out, fmaxf(t1, t2));
    }
  }
  if (t_in > t_out) miss = true;
  out[n] = miss ? -FMAX : fminf(fars[n], t_out);
}

int tnl_occupied_box(const uint8_t* bitfield, uint32_t bytes_per_cascade, uint32_t cascades, uint32_t H, float bound,
                     int32_t* bounds_scratch, float* box, void* stream) {
  if (bitfield == nullptr || bounds_scratch == nullptr || box == nullptr || cascades == 0 || cascades > 16 || H == 0)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(128), 0, st, bounds_scratch, (int)cascades, (int)H);
  hipLaunchKernelGGL(k_occupancy_bounds, dim3(min(cdiv(bytes_per_cascade, 256u), 256u), cascades), dim3(256), 0, st,
                     bitfield, bytes_per_cascade, cascades, bounds_scratch);
  hipLaunchKernelGGL(k_occupied_box, dim3(1), dim3(64), 0, st, bounds_scratch, (int)cascades, (int)H, bound, box);
  return (int)hipGetLastError();
}

int tnl_clip_fars(const float* rays_o, const float* rays_d, const float* fars, const float* box, uint32_t N,
                  float* fars_out, void* stream) {
  if (N == 0) return 0;
  if (box == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_clip_fars, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, fars, box, N,
                     fars_out);
  return (int)hipGetLastError();
}

int tnl_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_packbits, dim3(cdiv(cdiv(N, 4), 256)), dim3(256), 0, (hipStream_t)stream, grid, N,
                     density_thresh, bitfield, (const float*)nullptr);
  return launch_status();
}

int tnl_packbits_dev(const float* grid, uint32_t N, float density_thresh, const float* mean_density_dev, uint8_t* bitfield,
                     void* stream) {
  if (N == 0) return 0;
  if (mean_density_dev == nullptr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_packbits, dim3(cdiv(cdiv(N, 4), 256)), dim3(256), 0, (hipStream_t)stream, grid, N,
                     density_thresh, bitfield, mean_density_dev);
  return launch_status();
}

uint32_t tnl_march_rays_train_workspace(uint32_t N) { return N + cdiv(N, MARCH_BLOCK) + 1; }
// with room for the count pass's record of every sample's t (N * max_steps floats): the samples are then written
// from the record instead of by a second march; 0 if that does not fit 32 bits
uint32_t tnl_march_rays_train_workspace_rec(uint32_t N, uint32_t max_steps) {
  const uint64_t w = (uint64_t)tnl_march_rays_train_workspace(N) + (uint64_t)N * max_steps + 4 + NZ_WORDS;   // + k_nonzero_words' map
  return w > 0xffffffffull ? 0u : (uint32_t)w;
}

// tnl_march_count_form: 0 = the per-wavefront count pass wherever it applies, 1 = always one ray per lane
// (thread-local, like the emit / fill caps: a `with raymarching.count_form(...)` / `side_caps(...)` block of one host thread
//  does not change the launches another thread makes meanwhile -- ADVICE r05)
static thread_local int g_count_form = 0;
// tnl_march_emit_cap: most workgroups the emit pass is launched with (0 = one wavefront per ray all at once)
static thread_local int g_emit_cap = 0;

static int march_rays_train_impl(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                 float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                 const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                 int32_t* rays, int32_t* counter, const float* noises, int32_t* workspace,
                                 uint32_t workspace_words, uint32_t binR, void* sort_workspace, void* stream) {
  if (sort_workspace != nullptr) {   // the tile sort's bin counts (first nb + 1 ints of its workspace) start from zero
    if (binR == 0 || binR % TSX != 0) return (int)hipErrorInvalidValue;
    const size_t nbins = 3ull * (binR / TSX) * (binR / TSY) * BIN_SUBS;
    hipError_t e = hipMemsetAsync(sort_workspace, 0, (nbins + 1) * sizeof(int), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  if (N == 0) return 0;
  const uint32_t nb = cdiv(N, MARCH_BLOCK);
  int* num_steps = workspace;
  int* block_sums = workspace + N;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t need_rec = tnl_march_rays_train_workspace_rec(N, max_steps);
  if (need_rec != 0 && workspace_words >= need_rec) {
    // one march: the count pass records each sample's t, the samples are written from the record
    float* tbuf = reinterpret_cast<float*>(workspace + ((tnl_march_rays_train_workspace(N) + 3) & ~3u));
    if (wide_bitfield(grid, C, H) && dt_gamma == 0.f && C <= 2 && TNL_MARCH_WAVE && g_count_form != 1) {
      // one wavefront per ray over 64 consecutive chain points (see k_march_train_count_wave)
      hipLaunchKernelGGL((k_march_train_count_wave<true>), dim3(cdiv(N, MARCH_BLOCK / WAVE)), dim3(MARCH_BLOCK), 0, st, rays_o,
                         rays_d, grid, bound, max_steps, N, C, H, nears, fars, noises, num_steps, tbuf);
      hipLaunchKernelGGL(k_march_block_sums, dim3(nb), dim3(MARCH_BLOCK), 0, st, num_steps, N, block_sums);
    } else if (wide_bitfield(grid, C, H)) {
      // the per-lane count pass skips the loads of all-empty 64-cell words through a one-bit-per-word map (behind the record)
      uint32_t* nzmap = nullptr;
      const uint64_t n_words = (uint64_t)C * H * H * H / 64;
      if (TNL_MARCH_NZ_FILTER && TNL_MARCH_FAST_LANE && dt_gamma == 0.f && C <= 2 && H <= 256 && n_words <= (uint64_t)NZ_WORDS * 32) {
        nzmap = reinterpret_cast<uint32_t*>(tbuf + (size_t)N * max_steps);   // (the 4 spare words cover tbuf's alignment)
        hipLaunchKernelGGL(k_nonzero_words, dim3(NZ_WORDS * 32 / MARCH_BLOCK), dim3(MARCH_BLOCK), 0, st,
                           reinterpret_cast<const unsigned long long*>(grid), (uint32_t)n_words, nzmap);
      }
      hipLaunchKernelGGL((k_march_train_count<true, true>), dim3(nb), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d, grid,
                         bound, dt_gamma, max_steps, N, C, H, nears, fars, noises, num_steps, block_sums, tbuf, nzmap);
    } else
      hipLaunchKernelGGL((k_march_train_count<false, true>), dim3(nb), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d, grid,
                         bound, dt_gamma, max_steps, N, C, H, nears, fars, noises, num_steps, block_sums, tbuf);
    hipLaunchKernelGGL((k_march_train_write<true, false>), dim3(nb), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d, grid, bound,
                       dt_gamma, max_steps, N, C, H, M, nears, fars, noises, num_steps, block_sums, counter, xyzs,
                       dirs, deltas, rays);
    // (tnl_march_emit_cap: a march enqueued beside other kernels keeps to a few waves per SIMD)
    const uint32_t emit_blocks = g_emit_cap > 0 ? std::min<uint32_t>(cdiv(N, MARCH_BLOCK / WAVE), (uint32_t)g_emit_cap) : cdiv(N, MARCH_BLOCK / WAVE);
    hipLaunchKernelGGL(k_march_train_emit, dim3(emit_blocks), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d,
                       bound, dt_gamma, max_steps, N, C, H, M, nears, noises, counter, tbuf, rays, xyzs, dirs, deltas,
                       (int)binR, reinterpret_cast<int*>(sort_workspace));
  } else if (sort_workspace != nullptr) {
    return (int)hipErrorInvalidValue;   // the fused count needs the record path (workspace_rec words of scratch)
  } else if (wide_bitfield(grid, C, H)) {
    hipLaunchKernelGGL((k_march_train_count<true, false>), dim3(nb), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d, grid, bound,
                       dt_gamma, max_steps, N, C, H, nears, fars, noises, num_steps, block_sums, nullptr);
    hipLaunchKernelGGL((k_march_train_write<true, true>), dim3(nb), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d, grid, bound,
                       dt_gamma, max_steps, N, C, H, M, nears, fars, noises, num_steps, block_sums, counter, xyzs,
                       dirs, deltas, rays);
  } else {
    hipLaunchKernelGGL((k_march_train_count<false, false>), dim3(nb), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d, grid, bound,
                       dt_gamma, max_steps, N, C, H, nears, fars, noises, num_steps, block_sums, nullptr);
    hipLaunchKernelGGL((k_march_train_write<false, true>), dim3(nb), dim3(MARCH_BLOCK), 0, st, rays_o, rays_d, grid, bound,
                       dt_gamma, max_steps, N, C, H, M, nears, fars, noises, num_steps, block_sums, counter, xyzs,
                       dirs, deltas, rays);
  }
  hipLaunchKernelGGL(k_march_train_finalize, dim3(1), dim3(MARCH_BLOCK), 0, st, block_sums, nb, N, counter);
  return launch_status();
}

int tnl_march_emit_cap(int blocks) {
  const int prev = g_emit_cap;
  if (blocks >= 0) g_emit_cap = blocks;
  return prev;
}

int tnl_march_count_form(int form) {
  const int prev = g_count_form;
  if (form == 0 || form == 1) g_count_form = form;
  return prev;
}

int tnl_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                         float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                         int32_t* rays, int32_t* counter, const float* noises, int32_t* workspace,
                         uint32_t workspace_words, void* stream) {
  return march_rays_train_impl(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs,
                               deltas, rays, counter, noises, workspace, workspace_words, 0, nullptr, stream);
}

int tnl_march_rays_train_binned(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                                int32_t* rays, int32_t* counter, const float* noises, int32_t* workspace,
                                uint32_t workspace_words, uint32_t R, void* sort_workspace, void* stream) {
  if (sort_workspace == nullptr) return (int)hipErrorInvalidValue;
  return march_rays_train_impl(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs,
                               deltas, rays, counter, noises, workspace, workspace_words, R, sort_workspace, stream);
}

int tnl_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                     const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                     float* weights_sum, float* depth, float* image, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_composite_train_fwd, dim3(cdiv(N, COMP_BLOCK / WAVE)), dim3(COMP_BLOCK), 0,
                     (hipStream_t)stream, sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image);
  return launch_status();
}

int tnl_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                      const float* rgbs, const float* deltas, const int32_t* rays,
                                      const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                      float T_thresh, float* grad_sigmas, float* grad_rgbs, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_composite_train_bwd, dim3(cdiv(N, COMP_BLOCK / WAVE)), dim3(COMP_BLOCK), 0,
                     (hipStream_t)stream, grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum,
                     image, M, N, T_thresh, grad_sigmas, grad_rgbs);
  return launch_status();
}

int tnl_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                   float* xyzs, float* dirs, float* deltas, const float* noises, void* stream) {
  (void)nears;
  if (n_alive == 0) return 0;
  if (wide_bitfield(grid, C, H))
    hipLaunchKernelGGL(k_march_rays<true>, dim3(cdiv(n_alive, 128)), dim3(128), 0, (hipStream_t)stream, n_alive,
                       n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs,
                       dirs, deltas, noises);
  else
    hipLaunchKernelGGL(k_march_rays<false>, dim3(cdiv(n_alive, 128)), dim3(128), 0, (hipStream_t)stream, n_alive,
                       n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs,
                       dirs, deltas, noises);
  return launch_status();
}

int tnl_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                       const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                       float* depth, float* image, void* stream) {
  if (n_alive == 0) return 0;
  hipLaunchKernelGGL(k_composite_rays, dim3(cdiv(n_alive, 128)), dim3(128), 0, (hipStream_t)stream, n_alive, n_step,
                     T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
  return launch_status();
}

int tnl_infer_plan(int32_t* state, uint32_t N, uint32_t max_steps, uint32_t min_step, void* stream) {
  // n_step never exceeds 8 (the record kernel's LDS rows hold 8 samples + pad): a larger min_step is a caller error
  if (min_step > 8) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_infer_plan, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<InferState*>(state), N,
                     max_steps, (int)min_step);
  return launch_status();
}

int tnl_march_rays_dev(const int32_t* state, uint32_t N, const int32_t* rays_alive, const float* rays_t,
                       const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                       uint32_t C, uint32_t H, const uint8_t* grid, const float* fars, float* xyzs, float* dirs,
                       float* deltas, const float* noises, float* t_scratch, uint32_t rows_cap, void* stream) {
  if (N == 0) return 0;
  const InferState* st = reinterpret_cast<const InferState*>(state);
  hipStream_t s_ = (hipStream_t)stream;
  if (t_scratch != nullptr) {
    // record + emit: line-shaped stores (see k_march_rays_rec_dev); rows <= 8 N
    if (wide_bitfield(grid, C, H))
      hipLaunchKernelGGL(k_march_rays_rec_dev<true>, dim3(cdiv(N, 128)), dim3(128), 0, s_, st, rays_alive, rays_t, rays_o,
                         rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, noises, t_scratch);
    else
      hipLaunchKernelGGL(k_march_rays_rec_dev<false>, dim3(cdiv(N, 128)), dim3(128), 0, s_, st, rays_alive, rays_t, rays_o,
                         rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, noises, t_scratch);
    hipLaunchKernelGGL(k_emit_rays_dev, dim3(cdiv(rows_cap, 256)), dim3(256), 0, s_, st, rays_alive, rays_t, rays_o, rays_d,
                       bound, dt_gamma, max_steps, C, H, noises, t_scratch, xyzs, dirs, deltas);
    return launch_status();
  }
  if (wide_bitfield(grid, C, H))
    hipLaunchKernelGGL(k_march_rays_dev<true>, dim3(cdiv(N, 128)), dim3(128), 0, s_, st, rays_alive,
                       rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises);
  else
    hipLaunchKernelGGL(k_march_rays_dev<false>, dim3(cdiv(N, 128)), dim3(128), 0, s_, st, rays_alive,
                       rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, noises);
  return launch_status();
}

int tnl_composite_rays_dev(const int32_t* state, uint32_t N, float T_thresh, int32_t* rays_alive, float* rays_t,
                           const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                           float* depth, float* image, void* stream) {
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_composite_rays_dev, dim3(cdiv(N, 128)), dim3(128), 0, (hipStream_t)stream,
                     reinterpret_cast<const InferState*>(state), T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas,
                     weights_sum, depth, image);
  return launch_status();
}

int tnl_compact_rays_dev(int32_t* state, uint32_t N, const int32_t* rays_alive, int32_t* rays_alive_out,
                         int32_t* workspace, void* stream) {
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t nb = cdiv(N, 256);
  InferState* s = reinterpret_cast<InferState*>(state);
  int* next_n = workspace + nb;
  hipLaunchKernelGGL(k_compact_count_dev, dim3(nb), dim3(256), 0, st, s, rays_alive, workspace);
  hipLaunchKernelGGL(k_compact_write_dev, dim3(nb), dim3(256), 0, st, s, rays_alive, workspace, rays_alive_out, next_n);
  hipLaunchKernelGGL(k_infer_advance, dim3(1), dim3(64), 0, st, s, next_n);
  return launch_status();
}

int tnl_compact_rays(const int32_t* rays_alive, uint32_t n_alive, int32_t* rays_alive_out, int32_t* n_out,
                     int32_t* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n_alive == 0) return (int)hipMemsetAsync(n_out, 0, sizeof(int32_t), st);
  const uint32_t nb = cdiv(n_alive, 256);
  hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(256), 0, st, rays_alive, n_alive, workspace);
  hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(256), 0, st, rays_alive, n_alive, workspace, rays_alive_out,
                     n_out);
  return launch_status();
}

int tnl_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx,
                          void* stream) {
  if (D != 3 || C < 1 || C > 8) return (int)hipErrorInvalidValue;
  if (B == 0) return 0;
  hipLaunchKernelGGL(k_sh, dim3(cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, inputs, outputs, B, C, dy_dx);
  return launch_status();
}

int tnl_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                           const float* dy_dx, float* grad_inputs, void* stream) {
  (void)inputs;
  if (D != 3 || C < 1 || C > 8) return (int)hipErrorInvalidValue;
  if (B == 0) return 0;
  hipLaunchKernelGGL(k_sh_backward, dim3(cdiv(B * 3, 256)), dim3(256), 0, (hipStream_t)stream, grad, B, C, dy_dx,
                     grad_inputs);
  return launch_status();
}

}  // extern "C"
