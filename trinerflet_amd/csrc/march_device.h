// march_device.h -- the marching state machine (raymarching.cu:358-398, 430-479, 749-805) as device code shared by
// raymarch.hip (training march, inference loop kernels) and render.hip (the fused per-ray render kernel).
//
// Floating point: the expressions below write the FMAs nvcc emits for `a + b * c` as explicit fmaf and must not be
// contracted any further (bit-identical sample counts against the oracle): every function carries
// `#pragma clang fp contract(off)`, so the header can be included in a translation unit that is otherwise compiled with
// the default contraction (render.hip: its field part must match field.hip's).
#pragma once
#ifndef TNL_CHAIN_JUMP
#define TNL_CHAIN_JUMP 1     // 0: the empty-cell skip walks its chain of adds (the form of rounds 1-5), for A/B
#endif
#include "chain_skip.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr float SQRT3 = 1.7320508075688772f;
constexpr float RPI = 0.3183098861837907f;
constexpr int WAVE = 64;

__device__ __forceinline__ float signf_(float x) { return copysignf(1.0f, x); }
__device__ __forceinline__ float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// raymarching.cu:42-54
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
  const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  int e;
  frexpf(mx, &e);
  return (int)fminf(max_cascade - 1, fmaxf(0.f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
  const float mx = dt * H * 0.5f;  // power-of-two scaling: exact in float and double alike
  int e;
  frexpf(mx, &e);
  return (int)fminf(max_cascade - 1, fmaxf(0.f, (float)e));
}
// raymarching.cu:56-81
__host__ __device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
__host__ __device__ __forceinline__ uint32_t morton3D_(uint32_t x, uint32_t y, uint32_t z) {
  return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__host__ __device__ __forceinline__ uint32_t morton3D_invert_(uint32_t x) {
  x = x & 0x49249249;
  x = (x | (x >> 2)) & 0xc30c30c3;
  x = (x | (x >> 4)) & 0x0f00f00f;
  x = (x | (x >> 8)) & 0xff0000ff;
  x = (x | (x >> 16)) & 0x0000ffff;
  return x;
}

// ---------------------------------------------------------------------------------------------
// The marching state machine shared by the training and inference kernels
// (raymarching.cu:358-398, 430-479, 749-805).  WRITE=false only counts occupied steps.
// ---------------------------------------------------------------------------------------------
struct MarchCtx {
  float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, bound, dt_gamma, dt_min, dt_max, Hf, Cf;
  uint32_t H;
  const uint8_t* grid;
  // fast path of the per-step level arithmetic (every README configuration: dt_gamma = 0, at most two cascades): dt and
  // its level are constants, the position's level is 0 or 1, and the cascade scale and its IEEE reciprocal are two
  // precomputed pairs -- the same numbers the general expressions produce, without frexp / scalbn / a division per step
  bool fast, two_levels;
  float dt0, mb0, mb1, rb0, rb1;
  int level_dt0;
};

__device__ __forceinline__ void march_init(MarchCtx& m, const float* o, const float* d, float bound,
                                           float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                           const uint8_t* grid) {
#pragma clang fp contract(off)
  m.ox = o[0]; m.oy = o[1]; m.oz = o[2];
  m.dx = d[0]; m.dy = d[1]; m.dz = d[2];
  m.rdx = 1 / m.dx; m.rdy = 1 / m.dy; m.rdz = 1 / m.dz;
  m.rH = 1 / (float)H;
  m.H3 = (float)(H * H * H);
  m.bound = bound; m.dt_gamma = dt_gamma;
  m.dt_min = 2 * SQRT3 / max_steps;
  m.dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / H;
  m.Hf = (float)H; m.Cf = (float)C; m.H = H; m.grid = grid;
  m.fast = dt_gamma == 0.f && C <= 2;
  m.two_levels = C == 2;
  m.dt0 = clampf_(0.f, m.dt_min, m.dt_max);
  m.level_dt0 = mip_from_dt(m.dt0, m.Hf, m.Cf);
  m.mb0 = fminf(1.0f, bound); m.mb1 = fminf(2.0f, bound);
  m.rb0 = 1 / m.mb0; m.rb1 = 1 / m.mb1;
}

// Occupancy lookups go through a one-entry register cache of 64 consecutive bits: in Morton order that is an aligned
// 4x4x4 block of cells, so successive tests of a ray (samples dt apart inside a cell, neighbouring cells while
// skipping) mostly hit the cached word.  The loop is a chain of dependent global loads otherwise (one wave per SIMD:
// nothing hides their latency); with the cache only every ~10th test loads.  Needs the bitfield 8-byte aligned
// (WIDE); the byte-wise path is kept for arbitrary pointers.  Results are identical.
struct MarchProbe {   // everything one step of raymarching.cu:365-398 derives from t
  float x, y, z, dt, mip_bound;
  int nx, ny, nz;
  bool occ;
};

// cached_blk / cached_bits: the one-entry register cache of 64 consecutive occupancy bits (see march_run); the caller
// keeps them across probes of one ray.
template <bool WIDE>
__device__ __forceinline__ MarchProbe march_probe(const MarchCtx& m, float tt_, uint32_t& cached_blk,
                                                  unsigned long long& cached_bits) {
#pragma clang fp contract(off)
  MarchProbe q;
  q.x = clampf_(fmaf(tt_, m.dx, m.ox), -m.bound, m.bound);
  q.y = clampf_(fmaf(tt_, m.dy, m.oy), -m.bound, m.bound);
  q.z = clampf_(fmaf(tt_, m.dz, m.oz), -m.bound, m.bound);
  float mip_rbound;
  int level;
  if (m.fast) {   // wave-uniform
    q.dt = m.dt0;
    // mip_from_pos for two cascades: frexp exponent >= 1  <=>  max|x| >= 1
    const bool l1 = m.two_levels && fmaxf(fabsf(q.x), fmaxf(fabsf(q.y), fabsf(q.z))) >= 1.0f;
    level = max(l1 ? 1 : 0, m.level_dt0);
    q.mip_bound = level ? m.mb1 : m.mb0;
    mip_rbound = level ? m.rb1 : m.rb0;
  } else {
    q.dt = clampf_(tt_ * m.dt_gamma, m.dt_min, m.dt_max);
    level = max(mip_from_pos(q.x, q.y, q.z, m.Cf), mip_from_dt(q.dt, m.Hf, m.Cf));
    q.mip_bound = fminf(scalbnf(1.0f, level), m.bound);
    mip_rbound = 1 / q.mip_bound;
  }
  q.nx = (int)clampf_(0.5f * fmaf(q.x, mip_rbound, 1.0f) * m.Hf, 0.0f, (float)(m.H - 1));
  q.ny = (int)clampf_(0.5f * fmaf(q.y, mip_rbound, 1.0f) * m.Hf, 0.0f, (float)(m.H - 1));
  q.nz = (int)clampf_(0.5f * fmaf(q.z, mip_rbound, 1.0f) * m.Hf, 0.0f, (float)(m.H - 1));
  const uint32_t index = (uint32_t)((float)level * m.H3) + morton3D_(q.nx, q.ny, q.nz);
  if (WIDE) {
    const uint32_t blk = index >> 6;
    if (blk != cached_blk) {
      cached_bits = reinterpret_cast<const unsigned long long*>(m.grid)[blk];
      cached_blk = blk;
    }
    q.occ = (cached_bits >> (index & 63u)) & 1ull;
  } else {
    q.occ = m.grid[index >> 3] & (1u << (index & 7u));
  }
  return q;
}

// an empty cell probed at t: the t behind its exit (raymarching.cu:386-392) -- the march resumes at the first chain
// point that is not below it
__device__ __forceinline__ float march_skip_target(const MarchCtx& m, const MarchProbe& q, float t) {
#pragma clang fp contract(off)
  const float tx = fmaf(((float)q.nx + 0.5f + 0.5f * signf_(m.dx)) * m.rH * 2 - 1, q.mip_bound, -q.x) * m.rdx;
  const float ty = fmaf(((float)q.ny + 0.5f + 0.5f * signf_(m.dy)) * m.rH * 2 - 1, q.mip_bound, -q.y) * m.rdy;
  const float tz = fmaf(((float)q.nz + 0.5f + 0.5f * signf_(m.dz)) * m.rH * 2 - 1, q.mip_bound, -q.z) * m.rdz;
  return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
}

// an empty cell: jump to the first chain point behind its exit
__device__ __forceinline__ void march_skip(const MarchCtx& m, const MarchProbe& q, float& t) {
#pragma clang fp contract(off)
  const float tt = march_skip_target(m, q, t);
  if (m.fast) {
#if TNL_CHAIN_JUMP
    t = chain_skip_or_walk(t, m.dt0, tt);   // the loop below, long chains without walking them (chain_skip.h)
#else
    do { t += m.dt0; } while (t < tt);
#endif
  } else {
    do {
      t += clampf_(t * m.dt_gamma, m.dt_min, m.dt_max);
    } while (t < tt);
  }
}

// The next sample of a ray, or false when the ray leaves [t, far) without one: exactly what march_run<SPEC = false>
// does per step (the inference march), with the state (t, last_t, the occupancy cache) kept by the caller -- so a
// caller that takes one sample at a time (render.hip) sees the sample sequence of the alive-ray loop.
// tdiff = t_next - last_t is the composite's depth increment (deltas[1] of raymarching.cu:790).
template <bool WIDE>
__device__ __forceinline__ bool march_one(const MarchCtx& m, float& t, float& last_t, float far, uint32_t& cached_blk,
                                          unsigned long long& cached_bits, MarchProbe& out, float& tdiff) {
#pragma clang fp contract(off)
  while (t < far) {
    const MarchProbe a = march_probe<WIDE>(m, t, cached_blk, cached_bits);
    if (a.occ) {
      const float t_next = t + a.dt;
      tdiff = t_next - last_t;
      t = t_next;
      last_t = t;
      out = a;
      return true;
    }
    march_skip(m, a, t);
  }
  return false;
}

template <bool WRITE, bool WIDE, bool SPEC = true, bool REC = false>
__device__ __forceinline__ uint32_t march_run(const MarchCtx& m, float& t, float far, uint32_t limit,
                                              float* xyzs, float* dirs, float* deltas, float* trec = nullptr) {
#pragma clang fp contract(off)
  float last_t = t;
  uint32_t step = 0;
  uint32_t cached_blk = 0xffffffffu;
  unsigned long long cached_bits = 0ull;

  typedef MarchProbe Probe;
  auto probe = [&](float tt_) { return march_probe<WIDE>(m, tt_, cached_blk, cached_bits); };
  auto take = [&](const Probe& q, float t_next) {   // an occupied step: emit the sample, advance
    if (REC) *trec++ = t;   // the sample's t: everything k_march_train_emit writes follows from it
    if (WRITE) {
      xyzs[0] = q.x; xyzs[1] = q.y; xyzs[2] = q.z;
      dirs[0] = m.dx; dirs[1] = m.dy; dirs[2] = m.dz;
      deltas[0] = q.dt;
      deltas[1] = t_next - last_t;
      xyzs += 3; dirs += 3; deltas += 2;
    }
    t = t_next;
    last_t = t;
    step++;
  };
  auto skip = [&](const Probe& q) { march_skip(m, q, t); };
  // Two steps per trip: the step at t and, speculatively, the one at t + dt -- the successor whenever the first is
  // occupied, which is the common case inside an object.  The two dependent chains (position -> cell -> bit) are
  // independent of each other, so the in-order wave interleaves them; the serial result is unchanged (the second
  // probe is simply dropped when the first cell is empty).  SPEC = false (the inference loop's calls, 1-8 steps each,
  // where the extra probe is mostly wasted: 24.5 vs 27.2 ms per 800x800 image) walks one step per trip.
  if (!SPEC) {
    while (t < far && step < limit) {
      const Probe a = probe(t);
      if (a.occ) take(a, t + a.dt);
      else skip(a);
    }
    return step;
  }
  while (t < far && step < limit) {
    const Probe a = probe(t);
    const float t1 = t + a.dt;
    const Probe b = probe(t1);
    if (a.occ) {
      take(a, t1);
      if (t < far && step < limit) {
        if (b.occ) take(b, t + b.dt);
        else skip(b);
      }
    } else {
      skip(a);
    }
  }
  return step;
}

}  // namespace
