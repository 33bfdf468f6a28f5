// adam.hip -- fused optimiser pass over the wavelet coefficients (gfx950).
//
// Replaces, for the 402 M coefficient parameters of the base configuration, the chain
//   GradScaler.unscale_ -> wavelet L1 regulariser forward (abs, mean) and backward (sign, scale, add into .grad)
//   -> torch.optim.Adam (multi-tensor: ~6 elementwise kernels) -> optimizer.zero_grad
// (reconstruction/nerf/utils.py:639-655,1166-1173; reconstruction/main_nerf.py:119) with ONE streaming pass:
// 16 B/lane loads of p, g, m, v; 16 B/lane stores of p, m, v (28 B per parameter, the HBM floor for Adam).
// Arithmetic follows torch.optim.Adam's single-tensor path operation by operation:
//   m.lerp_(g, 1-b1) ; v.mul_(b2).addcmul_(g, g, 1-b2) ; denom = sqrt(v)/sqrt(bc2) + eps ; p.addcdiv_(m, denom, -lr/bc1)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#ifndef TNL_MAIN_PRIO
#define TNL_MAIN_PRIO 0   // A/B builds: static wave priority (s_setprio) of the step's kernels that run beside the side chain
#endif
#define TNL_SET_MAIN_PRIO() do { if (TNL_MAIN_PRIO) __builtin_amdgcn_s_setprio(TNL_MAIN_PRIO); } while (0)

#include "../../include/trinerflet_hip.h"
#include "adam_common.h"

#ifndef TNL_ADAM_PIECE
#define TNL_ADAM_PIECE 0
#endif
#ifndef TNL_ADAM_UNROLL
#define TNL_ADAM_UNROLL 2
#endif
#ifndef TNL_ADAM_ORDER
#define TNL_ADAM_ORDER 0   // A/B knob: 0 = p loaded ahead of the found_inf branch, 1 = with g, m, v, 2 = v, m, g, p
#endif
#ifndef TNL_ADAM_STORE_ORDER
#define TNL_ADAM_STORE_ORDER 0
#endif
#ifndef TNL_ADAM_ZERO_SKIP
#define TNL_ADAM_ZERO_SKIP 1   // wavefronts at the p = m = v = g = 0 fixed point skip arithmetic and stores (A/B knob)
#endif
#ifndef TNL_ADAM_BLOCKS
#define TNL_ADAM_BLOCKS 4096
#endif

namespace {

// Gradient support of one wavelet level laid out [S][bands][n][n] (n a power of two): per plane a rectangle of the
// n x n grid outside which the gradient is identically zero and is NOT stored -- the adjoint IDWT skipped those tiles
// (tnl_idwt_level_backward_win).  Elements outside read g = 0 instead of the (stale) buffer: 24 B instead of 28.
struct AdamRect {
  int rx[3], ry[3], rw, rh;
  int log2n, bands, spp, s0;
};

__device__ __forceinline__ float4 ld_nt(const float4* p) {
  float4 r;
  r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y);
  r.z = __builtin_nontemporal_load(&p->z); r.w = __builtin_nontemporal_load(&p->w);
  return r;
}
__device__ __forceinline__ void st_nt(float4* p, const float4& v) {
  __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}


template <bool RECT, bool NTMP = false>
__global__ void __launch_bounds__(256)
k_adam_l1(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, uint64_t n,
          AdamArgs a, const float* __restrict__ inv_scale_dev, const float* __restrict__ found_inf,
          float* __restrict__ abs_sum, int zero_grad, const float* __restrict__ opt_step_dev, AdamRect rc,
          const float* __restrict__ l1_dev, const AdamStepRec* __restrict__ rec) {
  if (rec != nullptr) {
    // the step's scalars as k_adam_record wrote them (the same double-precision expressions as below, evaluated once):
    // nothing of this launch depends on a host value that changes from step to step -- a captured graph can replay it
    a.step_size = rec->step_size;
    a.bias2_sqrt = rec->bias2_sqrt;
  } else if (opt_step_dev != nullptr) {
    // a.step_size carries the learning rate; the bias corrections come from the DEVICE count of optimiser steps
    // actually taken (torch.optim.Adam's per-parameter `step`, which GradScaler.step does not advance on a skipped
    // iteration) -- so the host never has to read found_inf back
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
      const double t = (double)opt_step_dev[0] + 1.0;
      bc[0] = (float)((double)a.step_size / (1.0 - pow(a.b1d, t)));
      bc[1] = (float)sqrt(1.0 - pow(a.b2d, t));
    }
    __syncthreads();
    a.step_size = bc[0];
    a.bias2_sqrt = bc[1];
  }
  if (inv_scale_dev != nullptr) a.inv_scale *= inv_scale_dev[0];
  bool skip = found_inf != nullptr && found_inf[0] != 0.f;
  if (l1_dev != nullptr) {
    // an L1 term whose (loss-scaled) coefficient only exists on the device: d/dp of s * sum|p| is s * sign(p); it is
    // unscaled like the data gradient it would have been added to.  A non-finite coefficient skips the update.
    const float s = l1_dev[0] * a.inv_scale;
    a.l1_coef += s;
    skip = skip || !(fabsf(s) <= 3.0e38f);
  }
  float acc = 0.f;
  const uint64_t n4 = n / 4;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  // Walk: each workgroup streams ONE contiguous chunk of the arrays (a pure grid-stride walk, every trip 16 MB
  // further, measured 8 % slower: 2.255 vs 2.45 ms at 402 M parameters).  TNL_ADAM_PIECE > 0 selects block-cyclic
  // pieces of that many float4 instead (workgroup b takes pieces b, b + G, ...; the resident workgroups then stay within
  // G pieces of each other): on ONE 402 M-element array with 2048-float4 pieces and 4 float4 per thread it is 4-5 %
  // faster in both placement regimes of the pass (1.82-1.87 vs 1.87-1.97 ms, 2.03-2.08 vs 2.07-2.23 ms over 24 sets
  // of arrays), but inside the step (one rectangle-aware launch per level) it measured 2.06 vs 1.98-2.06 ms, step
  // 6.35-6.43 vs 6.28-6.39 ms -- not the default.
  const uint64_t chunk = TNL_ADAM_PIECE ? (uint64_t)TNL_ADAM_PIECE : (n4 + gridDim.x - 1) / gridDim.x;
  struct Quad { float4 pp, gg, mm, vv; };
  auto load = [&](uint64_t i, Quad& q) {
#if TNL_ADAM_ORDER == 0
    q.pp = NTMP ? ld_nt(p4 + i) : p4[i];
#endif
    q.gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!skip) {
      bool inside = true;
      if (RECT) {
        const uint64_t e = i * 4;
        const int nm = (1 << rc.log2n) - 1;
        const int c = (int)(e & nm), r = (int)((e >> rc.log2n) & nm);
        const uint32_t sb = (uint32_t)(e >> (2 * rc.log2n));                      // slice * bands + band (< 2^15)
        const int sl = (int)(rc.bands == 3 ? (sb * 0xAAABu) >> 17 : sb) + rc.s0;  // exact sb / 3 for sb < 98304
        const int pl = sl >= 2 * rc.spp ? 2 : (sl >= rc.spp ? 1 : 0);
        inside = c >= rc.rx[pl] && c < rc.rx[pl] + rc.rw && r >= rc.ry[pl] && r < rc.ry[pl] + rc.rh;
      }
#if TNL_ADAM_ORDER == 1
      q.pp = NTMP ? ld_nt(p4 + i) : p4[i];
#endif
#if TNL_ADAM_ORDER == 2
      q.vv = NTMP ? ld_nt(v4 + i) : v4[i];
      q.mm = NTMP ? ld_nt(m4 + i) : m4[i];
      if (inside) q.gg = NTMP ? ld_nt(g4 + i) : g4[i];
      q.pp = NTMP ? ld_nt(p4 + i) : p4[i];
#else
      if (inside) q.gg = NTMP ? ld_nt(g4 + i) : g4[i];
      q.mm = NTMP ? ld_nt(m4 + i) : m4[i];
      q.vv = NTMP ? ld_nt(v4 + i) : v4[i];
#endif
    }
#if TNL_ADAM_ORDER != 0
    else { q.pp = NTMP ? ld_nt(p4 + i) : p4[i]; }
#endif
  };
  auto finish = [&](uint64_t i, Quad& q) {
    if (!skip) {
      // (p = m = v = g = 0 stays there -- see k_adam_l1_live: the stores are skipped, the gradient is already the zero a
      //  zero_grad would write)
      auto bits = [](const float4& t) { return __float_as_uint(t.x) | __float_as_uint(t.y) | __float_as_uint(t.z) | __float_as_uint(t.w); };
      const uint32_t any = (bits(q.pp) | bits(q.gg) | bits(q.mm) | bits(q.vv)) & 0x7fffffffu;
      const bool moved = !TNL_ADAM_ZERO_SKIP || (zero_grad & 2) || __ballot(any != 0u) != 0ull;   // bit 1: always store
      adam1(q.pp.x, q.gg.x, q.mm.x, q.vv.x, a, acc);
      adam1(q.pp.y, q.gg.y, q.mm.y, q.vv.y, a, acc);
      adam1(q.pp.z, q.gg.z, q.mm.z, q.vv.z, a, acc);
      adam1(q.pp.w, q.gg.w, q.mm.w, q.vv.w, a, acc);
      if (!moved) return;
#if TNL_ADAM_STORE_ORDER == 0
      if (NTMP) { st_nt(p4 + i, q.pp); st_nt(m4 + i, q.mm); st_nt(v4 + i, q.vv); }
      else { p4[i] = q.pp; m4[i] = q.mm; v4[i] = q.vv; }
#else
      if (NTMP) { st_nt(v4 + i, q.vv); st_nt(m4 + i, q.mm); st_nt(p4 + i, q.pp); }
      else { v4[i] = q.vv; m4[i] = q.mm; p4[i] = q.pp; }
#endif
    } else {
      acc += fabsf(q.pp.x) + fabsf(q.pp.y) + fabsf(q.pp.z) + fabsf(q.pp.w);
    }
    if (zero_grad & 1) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // TNL_ADAM_UNROLL float4 per thread and trip, all loads first (bytes in flight per wave)
  constexpr int U = TNL_ADAM_UNROLL;
  // TNL_ADAM_PIECE > 0: block-cyclic pieces of that many float4 (workgroup b takes pieces b, b + G, ...): each piece is
  // still a contiguous stream, but the resident workgroups stay within G pieces of each other instead of being spread
  // over the whole array
  for (uint64_t c0 = (uint64_t)blockIdx.x * chunk; c0 < n4; c0 += TNL_ADAM_PIECE ? (uint64_t)gridDim.x * chunk : n4) {
  const uint64_t c1 = min(c0 + chunk, n4);
  uint64_t i = c0 + threadIdx.x;
  if (U > 1) {
    for (; i + (uint64_t)(U - 1) * blockDim.x < c1; i += (uint64_t)U * blockDim.x) {
      Quad q[U];
#pragma unroll
      for (int k = 0; k < U; k++) load(i + (uint64_t)k * blockDim.x, q[k]);
#pragma unroll
      for (int k = 0; k < U; k++) finish(i + (uint64_t)k * blockDim.x, q[k]);
    }
  }
  for (; i < c1; i += blockDim.x) {
    Quad q;
    load(i, q);
    finish(i, q);
  }
  }
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // ragged tail
  for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pp = p[i];
    if (!skip) {
      float mm = m[i], vv = v[i];
      adam1(pp, g[i], mm, vv, a, acc);
      p[i] = pp; m[i] = mm; v[i] = vv;
    } else {
      acc += fabsf(pp);
    }
    if (zero_grad & 1) g[i] = 0.f;
  }
  if (abs_sum != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(abs_sum, part[0] + part[1] + part[2] + part[3]);
  }
}


// ---------------------------------------------------------------------------------------------
// Live / deferred split of one level's pass (TrainStep between two density-grid refreshes).
// Outside the occupancy window's footprint a coefficient is neither read by the windowed plane rebuild nor reached by
// a data gradient: its update p, m, v <- adam(p, l1 * sign(p), m, v) is a closed recurrence in its own three
// numbers and the step's two scalars.  k_adam_l1_live therefore updates only the live rectangle (28 B per coefficient
// of the window instead of 24-28 B per coefficient of the level), k_adam_record keeps each step's scalars
// (step size, bias correction, GradScaler skip) in a device ring, and k_adam_l1_catchup later replays the pending
// steps for everything outside the rectangle in registers: one 24-byte pass per flush instead of one per step, the
// same operations in the same order -- bit-identical p, m, v.
// ---------------------------------------------------------------------------------------------
constexpr int ADAM_REPLAY_MAX = 16;

__global__ void k_adam_record(AdamStepRec* __restrict__ ring, int slot, float lr, const float* __restrict__ opt_step_dev,
                              double beta1, double beta2, const float* __restrict__ found_inf,
                              const float* __restrict__ lr_dev, const float* __restrict__ l1_scaled_dev = nullptr,
                              const float* __restrict__ inv_scale_dev = nullptr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (lr_dev != nullptr) lr = lr_dev[0];
  const double t = (double)opt_step_dev[0] + 1.0;          // k_adam_l1's expressions
  AdamStepRec r;
  r.step_size = (float)((double)lr / (1.0 - pow(beta1, t)));
  r.bias2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
  r.skip = (found_inf != nullptr && found_inf[0] != 0.f) ? 1.f : 0.f;
  r.pad = 0.f;
  if (l1_scaled_dev != nullptr) {
    // optim.FusedAdamL1's folded regulariser (k_adam_l1's l1_dev branch: the same product, the same skip rule): the step's
    // L1 coefficient in true units rides in the record's fourth float, for the live pass and for the replay
    float inv = 1.0f;
    if (inv_scale_dev != nullptr) inv *= inv_scale_dev[0];
    const float s = l1_scaled_dev[0] * inv;
    r.pad = s;
    if (!(fabsf(s) <= 3.0e38f)) r.skip = 1.f;
  }
  ring[slot] = r;
}

__device__ __forceinline__ int rect_plane(const AdamRect& rc, uint32_t sb) {
  const int sl = (int)(rc.bands == 3 ? (sb * 0xAAABu) >> 17 : sb) + rc.s0;   // exact sb / 3 for sb < 98304
  return sl >= 2 * rc.spp ? 2 : (sl >= rc.spp ? 1 : 0);
}

// One launch for all wavelet levels of the step (they share the flat p / g / m / v arrays): per level an iteration
// domain `live` (per plane origin, common size; the whole level where nothing is deferred) and the rectangle `gr`
// where the gradient is stored (0 elsewhere).  The small levels no longer pay a launch tail each, and the step's
// scalars come from the record k_adam_record wrote (same double-precision expressions, evaluated once per step
// instead of by every workgroup).
constexpr int ADAM_MAX_SEGS = 8;
// Band table of a live rectangle (device, int32): the rectangle's rows in groups of 8 ("bands", nb = rh / 8), each with its
// own column piece -- bt[0 .. nb] prefix sums of the bands' float4 counts (8 * w_b / 4; bt[nb] = float4s per slice),
// bt[nb + 1 + b] = w_b / 4, bt[2 nb + 1 + pl * nb + b] = first column of band b on plane pl.  The live set is then the
// union of the pieces (what the occupied cells' projection can reach, level by level) instead of the whole rectangle.
constexpr int ADAM_MAX_BANDS = 128;
struct LiveSeg {
  AdamRect live, gr;
  uint64_t off;            // element offset of the level in the flat arrays
  uint32_t rows, blocks0;  // S * bands * live.rh ; first workgroup of the segment
  float l1_coef;
  const int* bt;           // band table or nullptr (the whole rectangle)
  uint32_t nb, quads;      // bands ; bt[nb] as the host knows it
};
struct LiveSegs {
  LiveSeg s[ADAM_MAX_SEGS];
  int n;
};

__global__ void __launch_bounds__(256)
k_adam_l1_live(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
               AdamArgs a, const float* __restrict__ inv_scale_dev, const float* __restrict__ found_inf,
               float* __restrict__ abs_sum, const float* __restrict__ opt_step_dev,
               const AdamStepRec* __restrict__ rec, LiveSegs segs) {
  TNL_SET_MAIN_PRIO();
  if (rec != nullptr) {
    a.step_size = rec->step_size;
    a.bias2_sqrt = rec->bias2_sqrt;
  } else {
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
      const double t = (double)opt_step_dev[0] + 1.0;
      bc[0] = (float)((double)a.step_size / (1.0 - pow(a.b1d, t)));
      bc[1] = (float)sqrt(1.0 - pow(a.b2d, t));
    }
    __syncthreads();
    a.step_size = bc[0];
    a.bias2_sqrt = bc[1];
  }
  if (inv_scale_dev != nullptr) a.inv_scale *= inv_scale_dev[0];
  // (rec->skip is the same flag unless the record carries a folded L1 coefficient that is not finite; rec->pad is that
  //  coefficient, 0 for TrainStep's records: l1 + 0 = l1)
  const bool skip = (found_inf != nullptr && found_inf[0] != 0.f) || (rec != nullptr && rec->skip != 0.f);
  const float l1_rec = rec != nullptr ? rec->pad : 0.f;
  int si = 0;
#pragma unroll
  for (int k = 1; k < ADAM_MAX_SEGS; k++)
    if (k < segs.n && blockIdx.x >= segs.s[k].blocks0) si = k;
  const LiveSeg& sg = segs.s[si];
  const AdamRect& live = sg.live;
  const AdamRect& gr = sg.gr;
  a.l1_coef = sg.l1_coef + l1_rec;
  const uint32_t nblk = (si + 1 < segs.n ? segs.s[si + 1].blocks0 : gridDim.x) - sg.blocks0;
  float acc = 0.f;
  __shared__ int s_bt[5 * ADAM_MAX_BANDS + 1];
  const uint32_t w4 = (uint32_t)live.rw / 4;
  const uint32_t total = sg.bt == nullptr ? sg.rows * w4 : sg.rows / (uint32_t)live.rh * sg.quads;
  const uint32_t chunk = (total + nblk - 1) / nblk;
  const uint32_t c0 = (blockIdx.x - sg.blocks0) * chunk, c1 = min(c0 + chunk, total);
  float* pb = p + sg.off;
  const float* gb = g + sg.off;
  float* mb = m + sg.off;
  float* vb = v + sg.off;
  struct Quad { float4 pp, gg, mm, vv; uint64_t e; };
  auto fetch = [&](uint32_t sb, int pl, int x, int y, Quad& q) {
    q.e = ((uint64_t)sb << (2 * live.log2n)) + ((uint64_t)y << live.log2n) + (uint64_t)x;
    q.pp = ld_nt(reinterpret_cast<const float4*>(pb + q.e));
    q.gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!skip) {
      const bool inside = x >= gr.rx[pl] && x < gr.rx[pl] + gr.rw && y >= gr.ry[pl] && y < gr.ry[pl] + gr.rh;
      if (inside) q.gg = ld_nt(reinterpret_cast<const float4*>(gb + q.e));
      q.mm = ld_nt(reinterpret_cast<const float4*>(mb + q.e));
      q.vv = ld_nt(reinterpret_cast<const float4*>(vb + q.e));
    }
  };
  auto finish = [&](Quad& q) {
    if (!skip) {
      // p = m = v = g = 0 is a fixed point of the update (the L1 term is l1 * sign(0) = 0): the reference initialises the
      // wavelet levels to zero and a coefficient no sample's gradient has reached yet is still there.  A wavefront whose
      // coefficients are all at it skips the 12 bytes per coefficient of stores -- the same bits.  Only the stores: the
      // arithmetic stays unconditional (zeros in, zeros out); with the whole update behind the branch the pass lost
      // 0.05-0.1 ms at the base configuration, whose SURVEY 8(d) field starts from non-zero coefficients and never skips.
      // A real trajectory spends its first hundreds of steps with most of the fine levels at zero.
      auto bits = [](const float4& t) { return __float_as_uint(t.x) | __float_as_uint(t.y) | __float_as_uint(t.z) | __float_as_uint(t.w); };
      const uint32_t any = (bits(q.pp) | bits(q.gg) | bits(q.mm) | bits(q.vv)) & 0x7fffffffu;
      const bool moved = !TNL_ADAM_ZERO_SKIP || __ballot(any != 0u) != 0ull;
      adam1(q.pp.x, q.gg.x, q.mm.x, q.vv.x, a, acc);
      adam1(q.pp.y, q.gg.y, q.mm.y, q.vv.y, a, acc);
      adam1(q.pp.z, q.gg.z, q.mm.z, q.vv.z, a, acc);
      adam1(q.pp.w, q.gg.w, q.mm.w, q.vv.w, a, acc);
      if (!moved) return;
      st_nt(reinterpret_cast<float4*>(pb + q.e), q.pp);
      st_nt(reinterpret_cast<float4*>(mb + q.e), q.mm);
      st_nt(reinterpret_cast<float4*>(vb + q.e), q.vv);
    } else {
      acc += fabsf(q.pp.x) + fabsf(q.pp.y) + fabsf(q.pp.z) + fabsf(q.pp.w);
    }
  };
  auto load = [&](uint32_t i, Quad& q) {
    const uint32_t row = i / w4, c4 = i - row * w4;
    const uint32_t sb = row / (uint32_t)live.rh, r = row - sb * (uint32_t)live.rh;
    const int pl = rect_plane(live, sb);
    fetch(sb, pl, live.rx[pl] + 4 * (int)c4, live.ry[pl] + (int)r, q);
  };
  if (sg.bt == nullptr) {
    uint32_t i = c0 + threadIdx.x;
    for (; i + 256 < c1; i += 512) {
      Quad q0, q1;
      load(i, q0); load(i + 256, q1);
      finish(q0); finish(q1);
    }
    for (; i < c1; i += 256) { Quad q; load(i, q); finish(q); }
  } else {
    // banded: a thread's float4s advance by 256 along (slice, band, row, column), so it keeps a cursor (slice, offset
    // in the slice, band) and only the first position is searched for
    const int nb = (int)sg.nb;
    const uint32_t Q = sg.quads;
    for (int k = threadIdx.x; k < 5 * nb + 1; k += 256) s_bt[k] = sg.bt[k];
    __syncthreads();
    const int* s_w4 = s_bt + nb + 1;
    const int* s_x0 = s_bt + 2 * nb + 1;
    struct Cur { uint32_t sb, rem; int b; };
    auto advance = [&](Cur& c, uint32_t d) {
      c.rem += d;
      while (c.rem >= Q) { c.rem -= Q; c.sb++; c.b = 0; }
      while (c.rem >= (uint32_t)s_bt[c.b + 1]) c.b++;
    };
    auto load_at = [&](const Cur& c, Quad& q) {
      const uint32_t off = c.rem - (uint32_t)s_bt[c.b], bw4 = (uint32_t)s_w4[c.b];
      const uint32_t r = off / bw4, c4 = off - r * bw4;
      const int pl = rect_plane(live, c.sb);
      fetch(c.sb, pl, s_x0[pl * nb + c.b] + 4 * (int)c4, live.ry[pl] + 8 * c.b + (int)r, q);
    };
    uint32_t i = c0 + threadIdx.x;
    Cur c;
    c.sb = i / Q; c.rem = i - c.sb * Q;
    int lo = 0, hi = nb;                                     // s_bt[lo] <= rem < s_bt[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((uint32_t)s_bt[mid] <= c.rem) lo = mid; else hi = mid; }
    c.b = lo;
    for (; i + 256 < c1; i += 512) {
      Quad q0, q1;
      Cur d = c;
      advance(d, 256);
      load_at(c, q0); load_at(d, q1);
      finish(q0); finish(q1);
      c = d;
      advance(c, 256);
    }
    for (; i < c1; i += 256) { Quad q; load_at(c, q); finish(q); advance(c, 256); }
  }
  if (abs_sum != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(abs_sum, part[0] + part[1] + part[2] + part[3]);
  }
}

// Replays `count` (<= ADAM_REPLAY_MAX) recorded steps, oldest first, for every coefficient of the level OUTSIDE
// the live rectangle; abs_sums[r] += sum |p| as step r saw it (the L1 value's deferred share).  ALU-bound (count x two
// divisions and a square root per coefficient against 24 bytes): two float4 per thread are in flight, the record loop
// is a real loop (one scalar load per record and 8 coefficients), the per-record |p| sums live in per-thread LDS slots.
__global__ void __launch_bounds__(256)
k_adam_l1_catchup(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, uint64_t n, AdamArgs a,
                  const AdamStepRec* __restrict__ ring, int count, float* __restrict__ abs_sums, AdamRect live,
                  const int* __restrict__ bt, int nb) {
  __shared__ float s_acc[ADAM_REPLAY_MAX][256];
  __shared__ int s_bt[5 * ADAM_MAX_BANDS + 1];
  if (bt != nullptr) {
    for (int k = threadIdx.x; k < 5 * nb + 1; k += 256) s_bt[k] = bt[k];
    __syncthreads();
  }
  const bool sums = abs_sums != nullptr;
  if (sums)
    for (int r = 0; r < ADAM_REPLAY_MAX; r++) s_acc[r][threadIdx.x] = 0.f;
  const uint64_t n4 = n / 4;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  const uint64_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
  const uint64_t c0 = (uint64_t)blockIdx.x * chunk, c1 = min(c0 + chunk, n4);
  const int nm = (1 << live.log2n) - 1;
  auto outside = [&](uint64_t i) {
    const uint64_t e = i * 4;
    const int c = (int)(e & nm), r0 = (int)((e >> live.log2n) & nm);
    const int pl = rect_plane(live, (uint32_t)(e >> (2 * live.log2n)));
    if (!(c >= live.rx[pl] && c < live.rx[pl] + live.rw && r0 >= live.ry[pl] && r0 < live.ry[pl] + live.rh)) return true;
    if (bt == nullptr) return false;
    const int b = (r0 - live.ry[pl]) >> 3, x0 = s_bt[2 * nb + 1 + pl * nb + b];
    return !(c >= x0 && c < x0 + 4 * s_bt[nb + 1 + b]);
  };
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float l1_base = a.l1_coef;
  for (uint64_t i = c0 + threadIdx.x; i < c1; i += 512) {
    const uint64_t j = i + 256;
    const bool oi = outside(i), oj = j < c1 && outside(j);
    if (!oi && !oj) continue;
    // a slot that lies inside the rectangle (or past the chunk) is carried along as zeros: p = m = v = 0 stays 0
    float4 pi = zero, mi = zero, vi = zero, pj = zero, mj = zero, vj = zero;
    if (oi) { pi = ld_nt(p4 + i); mi = ld_nt(m4 + i); vi = ld_nt(v4 + i); }
    if (oj) { pj = ld_nt(p4 + j); mj = ld_nt(m4 + j); vj = ld_nt(v4 + j); }
    // A coefficient with p = m = v = 0 is a fixed point of the replay (g = l1 sign(0) = 0): the wavelet levels start at
    // zero, and outside the live pieces nothing but this recurrence ever touches them.  A wavefront whose 512 coefficients
    // are all at that fixed point skips the (ALU-bound) record loop and the stores: the same bits, half the bytes.
    {
      auto bits = [](const float4& q) { return __float_as_uint(q.x) | __float_as_uint(q.y) | __float_as_uint(q.z) | __float_as_uint(q.w); };
      const uint32_t any = (bits(pi) | bits(mi) | bits(vi) | bits(pj) | bits(mj) | bits(vj)) & 0x7fffffffu;
      if (__ballot(any != 0u) == 0ull) continue;
    }
    for (int r = 0; r < count; r++) {
      const AdamStepRec rec = ring[r];                      // uniform: a scalar load
      if (sums)
        s_acc[r][threadIdx.x] += fabsf(pi.x) + fabsf(pi.y) + fabsf(pi.z) + fabsf(pi.w) +
                                 fabsf(pj.x) + fabsf(pj.y) + fabsf(pj.z) + fabsf(pj.w);
      if (rec.skip != 0.f) continue;                        // GradScaler skipped this step: nothing moves
      a.step_size = rec.step_size;
      a.bias2_sqrt = rec.bias2_sqrt;
      a.l1_coef = l1_base + rec.pad;                        // (pad: a folded L1 coefficient of that step, else 0)
      // (round 6, measured and not kept: the update without its dead gradient terms -- g = l1 sign(p) directly and
      //  omb2 g^2 as a per-record constant, 4 instructions of ~21 fewer, the same bits -- made the replay 8 % SLOWER,
      //  1.51 -> 1.64 ms at base: profiles/r06g_ab_replay.txt)
      float unused = 0.f;
      adam1(pi.x, 0.f, mi.x, vi.x, a, unused); adam1(pi.y, 0.f, mi.y, vi.y, a, unused);
      adam1(pi.z, 0.f, mi.z, vi.z, a, unused); adam1(pi.w, 0.f, mi.w, vi.w, a, unused);
      adam1(pj.x, 0.f, mj.x, vj.x, a, unused); adam1(pj.y, 0.f, mj.y, vj.y, a, unused);
      adam1(pj.z, 0.f, mj.z, vj.z, a, unused); adam1(pj.w, 0.f, mj.w, vj.w, a, unused);
    }
    if (oi) { st_nt(p4 + i, pi); st_nt(m4 + i, mi); st_nt(v4 + i, vi); }
    if (oj) { st_nt(p4 + j, pj); st_nt(m4 + j, mj); st_nt(v4 + j, vj); }
  }
  if (sums) {
    __shared__ float part[ADAM_REPLAY_MAX][4];
    for (int r = 0; r < count; r++) {
      float x = s_acc[r][threadIdx.x];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
      if ((threadIdx.x & 63) == 0) part[r][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if ((int)threadIdx.x < count) atomicAdd(abs_sums + threadIdx.x, part[threadIdx.x][0] + part[threadIdx.x][1] +
                                                                     part[threadIdx.x][2] + part[threadIdx.x][3]);
  }
}

}  // namespace

static int adam_launch(float* p, float* grad, float* m, float* v, uint64_t n, float step_size, float bias2_sqrt,
                       float beta1, float beta2, float eps, float inv_scale, const float* inv_scale_dev,
                       float l1_coef, const float* found_inf, float* abs_sum, int zero_grad,
                       const float* opt_step_dev, void* stream, const AdamRect* rect = nullptr,
                       const float* l1_dev = nullptr, const AdamStepRec* rec = nullptr) {
  if (n == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
       reinterpret_cast<uintptr_t>(v)) & 15)
    return (int)hipErrorInvalidValue;
  AdamArgs a = make_adam_args(step_size, bias2_sqrt, beta1, beta2, eps, inv_scale, l1_coef);
  uint64_t blocks = (n / 4 + 255) / 256;
  if (blocks > TNL_ADAM_BLOCKS) blocks = TNL_ADAM_BLOCKS;
  if (blocks == 0) blocks = 1;
  // Non-temporal loads/stores: every byte is touched exactly once per step and the arrays are ~40x the Infinity Cache;
  // measured 1.99 -> 1.85 ms per step at base (A/B in one session).  TNL_ADAM_TEMPORAL=1 restores default caching.
  static const bool use_nt = getenv("TNL_ADAM_TEMPORAL") == nullptr;
  if (rect == nullptr && use_nt)
    hipLaunchKernelGGL((k_adam_l1<false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v,
                       n, a, inv_scale_dev, found_inf, abs_sum, zero_grad, opt_step_dev, AdamRect{}, l1_dev, rec);
  else if (rect != nullptr && use_nt)
    hipLaunchKernelGGL((k_adam_l1<true, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, n,
                       a, inv_scale_dev, found_inf, abs_sum, zero_grad, opt_step_dev, *rect, l1_dev, rec);
  else if (rect != nullptr)
    hipLaunchKernelGGL(k_adam_l1<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, n, a,
                       inv_scale_dev, found_inf, abs_sum, zero_grad, opt_step_dev, *rect, l1_dev, rec);
  else
    hipLaunchKernelGGL(k_adam_l1<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, n, a,
                       inv_scale_dev, found_inf, abs_sum, zero_grad, opt_step_dev, AdamRect{}, l1_dev, rec);
  return (int)hipGetLastError();
}

extern "C" int tnl_adam_l1_step(float* p, float* grad, float* m, float* v, uint64_t n, float step_size,
                                float bias2_sqrt, float beta1, float beta2, float eps, float inv_scale,
                                const float* inv_scale_dev, float l1_coef, const float* found_inf,
                                float* abs_sum, int zero_grad, void* stream) {
  return adam_launch(p, grad, m, v, n, step_size, bias2_sqrt, beta1, beta2, eps, inv_scale, inv_scale_dev, l1_coef,
                     found_inf, abs_sum, zero_grad, nullptr, stream);
}

extern "C" int tnl_adam_l1_step_dev(float* p, float* grad, float* m, float* v, uint64_t n, float lr,
                                    const float* opt_step_dev, float beta1, float beta2, float eps,
                                    float inv_scale, const float* inv_scale_dev, float l1_coef,
                                    const float* found_inf, float* abs_sum, int zero_grad, void* stream) {
  if (opt_step_dev == nullptr) return (int)hipErrorInvalidValue;
  return adam_launch(p, grad, m, v, n, lr, 1.0f, beta1, beta2, eps, inv_scale, inv_scale_dev, l1_coef, found_inf,
                     abs_sum, zero_grad, opt_step_dev, stream);
}

extern "C" int tnl_adam_l1_step_sink(float* p, float* grad, float* m, float* v, uint64_t n, float lr,
                                     const float* opt_step_dev, float beta1, float beta2, float eps,
                                     const float* inv_scale_dev, float l1_coef, const float* l1_scaled_dev,
                                     const float* found_inf, void* stream) {
  if (opt_step_dev == nullptr) return (int)hipErrorInvalidValue;
  return adam_launch(p, grad, m, v, n, lr, 1.0f, beta1, beta2, eps, 1.0f, inv_scale_dev, l1_coef, found_inf, nullptr, 0,
                     opt_step_dev, stream, nullptr, l1_scaled_dev);
}

// One wavelet level [S][bands][n][n] whose gradient is stored only inside a per-plane rectangle (rect_host: ox[3],
// oy[3], w, h in the level's own n x n coordinates, as returned by tnl_idwt_level_backward_win; multiples of 4).
static int adam_rect_launch(float* p, float* grad, float* m, float* v, uint32_t S, uint32_t bands, uint32_t n, uint32_t spp,
                            uint32_t s0, const int32_t* rect_host, float lr, const float* opt_step_dev, const AdamStepRec* rec,
                            float beta1, float beta2, float eps, float inv_scale, const float* inv_scale_dev, float l1_coef,
                            const float* found_inf, float* abs_sum, void* stream) {
  if ((opt_step_dev == nullptr && rec == nullptr) || rect_host == nullptr || n == 0 || (n & (n - 1)) != 0 || n % 4 != 0 ||
      bands == 0 || spp == 0)
    return (int)hipErrorInvalidValue;
  AdamRect rc;
  for (int k = 0; k < 3; k++) { rc.rx[k] = rect_host[k]; rc.ry[k] = rect_host[3 + k]; }
  rc.rw = rect_host[6]; rc.rh = rect_host[7];
  if (rc.rw % 4 != 0) return (int)hipErrorInvalidValue;
  for (int k = 0; k < 3; k++)
    if (rc.rx[k] % 4 != 0 || rc.rx[k] < 0 || rc.ry[k] < 0 || rc.rx[k] + rc.rw > (int)n || rc.ry[k] + rc.rh > (int)n)
      return (int)hipErrorInvalidValue;
  rc.log2n = 0;
  while ((1u << rc.log2n) < n) rc.log2n++;
  rc.bands = (int)bands; rc.spp = (int)spp; rc.s0 = (int)s0;
  return adam_launch(p, grad, m, v, (uint64_t)S * bands * n * n, lr, 1.0f, beta1, beta2, eps, inv_scale, inv_scale_dev,
                     l1_coef, found_inf, abs_sum, 0, opt_step_dev, stream, &rc, nullptr, rec);
}

extern "C" int tnl_adam_l1_step_rect(float* p, float* grad, float* m, float* v, uint32_t S, uint32_t bands, uint32_t n,
                                     uint32_t spp, uint32_t s0, const int32_t* rect_host, float lr,
                                     const float* opt_step_dev, float beta1, float beta2, float eps, float inv_scale,
                                     const float* inv_scale_dev, float l1_coef, const float* found_inf,
                                     float* abs_sum, void* stream) {
  return adam_rect_launch(p, grad, m, v, S, bands, n, spp, s0, rect_host, lr, opt_step_dev, nullptr, beta1, beta2, eps,
                          inv_scale, inv_scale_dev, l1_coef, found_inf, abs_sum, stream);
}

// The same two passes with the step's scalars read from a slot of the step ring (tnl_adam_record_step[_dev]) instead of
// being derived from a learning rate passed by value: no argument changes from step to step (TrainStep's captured graphs).
extern "C" int tnl_adam_l1_step_rec(float* p, float* grad, float* m, float* v, uint64_t n, const float* step_rec, float beta1,
                                    float beta2, float eps, const float* inv_scale_dev, float l1_coef,
                                    const float* found_inf, float* abs_sum, void* stream) {
  if (step_rec == nullptr) return (int)hipErrorInvalidValue;
  return adam_launch(p, grad, m, v, n, 0.f, 1.0f, beta1, beta2, eps, 1.0f, inv_scale_dev, l1_coef, found_inf, abs_sum, 0,
                     nullptr, stream, nullptr, nullptr, reinterpret_cast<const AdamStepRec*>(step_rec));
}

extern "C" int tnl_adam_l1_step_rect_rec(float* p, float* grad, float* m, float* v, uint32_t S, uint32_t bands, uint32_t n,
                                         uint32_t spp, uint32_t s0, const int32_t* rect_host, const float* step_rec,
                                         float beta1, float beta2, float eps, const float* inv_scale_dev, float l1_coef,
                                         const float* found_inf, float* abs_sum, void* stream) {
  if (step_rec == nullptr) return (int)hipErrorInvalidValue;
  return adam_rect_launch(p, grad, m, v, S, bands, n, spp, s0, rect_host, 0.f, nullptr,
                          reinterpret_cast<const AdamStepRec*>(step_rec), beta1, beta2, eps, 1.0f, inv_scale_dev, l1_coef,
                          found_inf, abs_sum, stream);
}

static int fill_rect(AdamRect& rc, const int32_t* h, uint32_t n, uint32_t bands, uint32_t spp, uint32_t s0) {
  for (int k = 0; k < 3; k++) { rc.rx[k] = h[k]; rc.ry[k] = h[3 + k]; }
  rc.rw = h[6]; rc.rh = h[7];
  if (rc.rw % 4 != 0 || rc.rw <= 0 || rc.rh <= 0) return 1;
  for (int k = 0; k < 3; k++)
    if (rc.rx[k] % 4 != 0 || rc.rx[k] < 0 || rc.ry[k] < 0 || rc.rx[k] + rc.rw > (int)n || rc.ry[k] + rc.rh > (int)n)
      return 1;
  rc.log2n = 0;
  while ((1u << rc.log2n) < n) rc.log2n++;
  rc.bands = (int)bands; rc.spp = (int)spp; rc.s0 = (int)s0;
  return 0;
}

extern "C" int tnl_adam_record_step(float* ring, int32_t slot, float lr, const float* opt_step_dev, float beta1,
                                    float beta2, const float* found_inf, void* stream) {
  if (ring == nullptr || opt_step_dev == nullptr || slot < 0 || slot >= ADAM_REPLAY_MAX) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_adam_record, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<AdamStepRec*>(ring),
                     (int)slot, lr, opt_step_dev, adam_decimal(beta1), adam_decimal(beta2), found_inf, (const float*)nullptr);
  return (int)hipGetLastError();
}

extern "C" int tnl_adam_record_step_l1(float* ring, int32_t slot, float lr, const float* opt_step_dev, float beta1,
                                       float beta2, const float* found_inf, const float* l1_scaled_dev,
                                       const float* inv_scale_dev, void* stream) {
  if (ring == nullptr || opt_step_dev == nullptr || slot < 0 || slot >= ADAM_REPLAY_MAX) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_adam_record, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<AdamStepRec*>(ring),
                     (int)slot, lr, opt_step_dev, adam_decimal(beta1), adam_decimal(beta2), found_inf, (const float*)nullptr,
                     l1_scaled_dev, inv_scale_dev);
  return (int)hipGetLastError();
}

extern "C" int tnl_adam_record_step_dev(float* ring, int32_t slot, const float* lr_dev, const float* opt_step_dev, float beta1,
                                        float beta2, const float* found_inf, void* stream) {
  if (ring == nullptr || opt_step_dev == nullptr || lr_dev == nullptr || slot < 0 || slot >= ADAM_REPLAY_MAX)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_adam_record, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<AdamStepRec*>(ring),
                     (int)slot, 0.f, opt_step_dev, adam_decimal(beta1), adam_decimal(beta2), found_inf, lr_dev);
  return (int)hipGetLastError();
}

extern "C" int tnl_adam_l1_step_live_bands(float* p, float* grad, float* m, float* v, uint32_t S, uint32_t spp, uint32_t s0,
                                           uint32_t n_levels, const uint64_t* offsets, const uint32_t* sizes,
                                           const uint32_t* bands, const int32_t* live_host, const int32_t* grad_rect_host,
                                           const int32_t* const* band_tables, const uint32_t* band_quads,
                                           const float* l1_coefs, float lr, const float* opt_step_dev, const float* step_rec,
                                           float beta1, float beta2, float eps, float inv_scale, const float* inv_scale_dev,
                                           const float* found_inf, float* abs_sum, void* stream) {
  if ((opt_step_dev == nullptr && step_rec == nullptr) || live_host == nullptr || grad_rect_host == nullptr ||
      offsets == nullptr || sizes == nullptr || bands == nullptr || l1_coefs == nullptr || n_levels == 0 ||
      n_levels > ADAM_MAX_SEGS || spp == 0 || S == 0)
    return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
       reinterpret_cast<uintptr_t>(v)) & 15)
    return (int)hipErrorInvalidValue;
  LiveSegs segs;
  segs.n = (int)n_levels;
  uint64_t items[ADAM_MAX_SEGS], all = 0;
  for (uint32_t k = 0; k < n_levels; k++) {
    const uint32_t n = sizes[k];
    if (n == 0 || (n & (n - 1)) != 0 || n % 4 != 0 || bands[k] == 0 || offsets[k] % 4 != 0) return (int)hipErrorInvalidValue;
    LiveSeg& sg = segs.s[k];
    if (fill_rect(sg.live, live_host + 8 * k, n, bands[k], spp, s0) ||
        fill_rect(sg.gr, grad_rect_host + 8 * k, n, bands[k], spp, s0))
      return (int)hipErrorInvalidValue;
    const uint64_t rows = (uint64_t)S * bands[k] * sg.live.rh;
    sg.bt = band_tables != nullptr ? band_tables[k] : nullptr;
    sg.nb = 0; sg.quads = 0;
    if (sg.bt != nullptr) {
      if (band_quads == nullptr || band_quads[k] == 0 || sg.live.rh % 8 != 0 || sg.live.rh / 8 > ADAM_MAX_BANDS ||
          band_quads[k] > (uint32_t)(sg.live.rh * (sg.live.rw / 4)))
        return (int)hipErrorInvalidValue;
      sg.nb = (uint32_t)sg.live.rh / 8;
      sg.quads = band_quads[k];
    }
    items[k] = sg.bt != nullptr ? (uint64_t)S * bands[k] * sg.quads : rows * (sg.live.rw / 4);
    if (items[k] >= (1ull << 32)) return (int)hipErrorInvalidValue;
    sg.off = offsets[k];
    sg.rows = (uint32_t)rows;
    sg.l1_coef = l1_coefs[k];
    all += items[k];
  }
  // workgroups dealt in proportion to the levels' float4 counts (512 per workgroup and trip), at least one each
  static const uint64_t max_blocks = getenv("TNL_ADAM_LIVE_BLOCKS") ? strtoull(getenv("TNL_ADAM_LIVE_BLOCKS"), nullptr, 10)
                                                                    : TNL_ADAM_BLOCKS;
  uint64_t want = (all + 511) / 512;
  if (want > max_blocks) want = max_blocks;
  if (want < n_levels) want = n_levels;
  uint32_t next = 0;
  for (uint32_t k = 0; k < n_levels; k++) {
    uint64_t b = (items[k] * want + all - 1) / all;
    if (b == 0) b = 1;
    segs.s[k].blocks0 = next;
    next += (uint32_t)b;
  }
  AdamArgs a = make_adam_args(lr, 1.0f, beta1, beta2, eps, inv_scale, 0.f);
  hipLaunchKernelGGL(k_adam_l1_live, dim3(next), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, a,
                     inv_scale_dev, found_inf, abs_sum, opt_step_dev, reinterpret_cast<const AdamStepRec*>(step_rec), segs);
  return (int)hipGetLastError();
}

extern "C" int tnl_adam_l1_step_live(float* p, float* grad, float* m, float* v, uint32_t S, uint32_t spp, uint32_t s0,
                                     uint32_t n_levels, const uint64_t* offsets, const uint32_t* sizes,
                                     const uint32_t* bands, const int32_t* live_host, const int32_t* grad_rect_host,
                                     const float* l1_coefs, float lr, const float* opt_step_dev, const float* step_rec,
                                     float beta1, float beta2, float eps, float inv_scale, const float* inv_scale_dev,
                                     const float* found_inf, float* abs_sum, void* stream) {
  return tnl_adam_l1_step_live_bands(p, grad, m, v, S, spp, s0, n_levels, offsets, sizes, bands, live_host, grad_rect_host,
                                     nullptr, nullptr, l1_coefs, lr, opt_step_dev, step_rec, beta1, beta2, eps, inv_scale,
                                     inv_scale_dev, found_inf, abs_sum, stream);
}

extern "C" int tnl_adam_l1_catchup_bands(float* p, float* m, float* v, uint32_t S, uint32_t bands, uint32_t n, uint32_t spp,
                                         uint32_t s0, const int32_t* live_host, const int32_t* band_table,
                                         const float* ring, int32_t count, float beta1, float beta2, float eps,
                                         float l1_coef, float* abs_sums, void* stream) {
  if (count == 0) return 0;
  if (ring == nullptr || live_host == nullptr || count < 0 || count > ADAM_REPLAY_MAX || n == 0 || (n & (n - 1)) != 0 ||
      n % 4 != 0 || bands == 0 || spp == 0 || S == 0)
    return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
    return (int)hipErrorInvalidValue;
  AdamRect live;
  if (fill_rect(live, live_host, n, bands, spp, s0)) return (int)hipErrorInvalidValue;
  if (band_table != nullptr && (live.rh % 8 != 0 || live.rh / 8 > ADAM_MAX_BANDS)) return (int)hipErrorInvalidValue;
  const uint64_t total = (uint64_t)S * bands * n * n;
  AdamArgs a = make_adam_args(0.f, 1.0f, beta1, beta2, eps, 1.0f, l1_coef);
  uint64_t blocks = (total / 4 + 255) / 256;
  if (blocks > 4 * TNL_ADAM_BLOCKS) blocks = 4 * TNL_ADAM_BLOCKS;
  hipLaunchKernelGGL(k_adam_l1_catchup, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, m, v, total, a,
                     reinterpret_cast<const AdamStepRec*>(ring), (int)count, abs_sums, live, band_table,
                     band_table != nullptr ? live.rh / 8 : 0);
  return (int)hipGetLastError();
}

extern "C" int tnl_adam_l1_catchup(float* p, float* m, float* v, uint32_t S, uint32_t bands, uint32_t n, uint32_t spp,
                                   uint32_t s0, const int32_t* live_host, const float* ring, int32_t count, float beta1,
                                   float beta2, float eps, float l1_coef, float* abs_sums, void* stream) {
  return tnl_adam_l1_catchup_bands(p, m, v, S, bands, n, spp, s0, live_host, nullptr, ring, count, beta1, beta2, eps,
                                   l1_coef, abs_sums, stream);
}
