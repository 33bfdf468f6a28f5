// render.hip -- the inference render as ONE persistent kernel (gfx950): march + fused field + compositing per ray.
//
// Replaces, for `NeRFRenderer.run_cuda`'s eval branch (reconstruction/nerf/renderer.py:324-374), the alive-ray loop of
// march_rays / network forward / composite_rays / compaction -- several hundred iterations of ~9 launches over a
// shrinking ray list, every sample's position, direction, step, sigma and colour written to and read back from HBM.
// Here a wavefront holds 32 rays (lane (r, h): ray slot r, k-half h of the field's MFMA operands; the two halves of a
// slot carry identical ray state) and per trip
//   1. refills the slots whose ray has ended from a global ray queue (one wave-aggregated atomic),
//   2. advances every ray to its NEXT sample with the marching state machine of the loop kernels (march_device.h:
//      the same probes, skips and step sizes -- the sample sequence of raymarching.cu:749-805),
//   3. evaluates the field on the 32 samples exactly as k_field_fwd does on a 32-sample tile (field_device.h: texel
//      gather straight into the layer-0 MFMA operand, five layers in registers),
//   4. composites the sample into the ray's accumulators with the arithmetic of raymarching.cu:853-904
//      (alpha = 1 - __expf(-sigma dt), T = 1 - weight_sum, stop after the sample at which T < T_thresh).
// No sample ever touches memory; the only traffic is the plane texels, one ray record in and 20 bytes per ray out.
//
// Equality with the loop: every ray sees the loop's samples in the loop's order, so weights / colours / depths agree
// to the last bit for a ray that ends before the max_steps cap, except where the loop's hand-over of t between two of
// its iterations (rays_t = t0 + fl(t1 - t0), raymarching.cu:893) rounds differently from the march's own t1 -- a tie
// case of one ulp in t.  A ray still alive after max_steps samples stops there (the loop's own cap is a schedule-
// dependent max_steps ... max_steps + 7).  tests/test_render_fused_gpu.py.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trinerflet_hip.h"
#include "field_device.h"
#include "march_device.h"

namespace {

// threads per workgroup: four waves = 128 ray slots.  Hidden 128 (92 KB of fragments: one workgroup per CU, 295 registers)
// with eight waves (two per SIMD, 256 registers + 54 spilled): 800 x 800 at max_steps 1024 18.1 -> 15.3 ms, at 4096
// 56.3 -> 57.9 ms -- the BASELINE render is the 4096 one, so four.
#ifndef TNL_RENDER_RT128
#define TNL_RENDER_RT128 256
#endif
template <int H>
constexpr int render_threads() { return H > 64 ? TNL_RENDER_RT128 : 256; }

// march_one (march_device.h) in bounded pieces (round 6).  A ray that enters the volume walks ~30 empty cells before its
// first sample, and at max_steps = 4096 (dt = 2 sqrt(3) / 4096) every empty cell is a probe plus 28-55 DEPENDENT adds of
// march_skip's `do t += dt while (t < tt)`: ~7 us of one lane's work, for which the other 31 rays of the wave -- and their
// field evaluation, ~5 us per trip -- used to wait whenever a slot had just been refilled.  On a trained field (23 samples
// per ray) nearly every trip holds a fresh ray: 0.23 of the HBM roof against 0.62 on a field whose rays last 285 samples.
// Now a trip spends at most TNL_RENDER_WORK units per ray (a probe counts 8, a skip add 1) and a ray that has not reached
// its next sample rides along without one and resumes next trip, its skip target kept in the ray state.  The same
// probes, adds and comparisons in the same order: the sample sequence is march_one's to the bit.
#ifndef TNL_RENDER_CHAIN_WALK
#define TNL_RENDER_CHAIN_WALK 16     // skips of up to this many steps walk, longer ones jump (chain_skip.h)
#endif
#ifndef TNL_RENDER_WORK
#define TNL_RENDER_WORK 96
#endif
enum { MARCH_SAMPLE = 0, MARCH_DONE = 1, MARCH_PAUSED = 2 };
template <bool WIDE>
__device__ __forceinline__ int march_one_bounded(const MarchCtx& m, float& t, float& last_t, float far, uint32_t& cached_blk,
                                                 unsigned long long& cached_bits, MarchProbe& out, float& tdiff, float& tt,
                                                 bool& skipping, int work) {
#pragma clang fp contract(off)
  while (true) {
    if (skipping) {            // the rest of a `do t += dt while (t < tt)` (its first add was taken when the skip began)
      while (t < tt && work > 0) {
        t += m.fast ? m.dt0 : clampf_(t * m.dt_gamma, m.dt_min, m.dt_max);
        work--;
      }
      if (t < tt) return MARCH_PAUSED;
      skipping = false;
    }
    if (!(t < far)) return MARCH_DONE;
    if (work <= 0) return MARCH_PAUSED;
    const MarchProbe a = march_probe<WIDE>(m, t, cached_blk, cached_bits);
    work -= 8;
    if (a.occ) {
      const float t_next = t + a.dt;
      tdiff = t_next - last_t;
      t = t_next;
      last_t = t;
      out = a;
      return MARCH_SAMPLE;
    }
    tt = march_skip_target(m, a, t);
#if TNL_CHAIN_JUMP
    if (m.fast) {            // constant step: the chain point behind the cell in O(1) (chain_skip.h), nothing to pause in
      t = chain_skip_or_walk_n(t, m.dt0, tt, (float)TNL_RENDER_CHAIN_WALK);
      work -= 4;
      continue;
    }
#endif
    t += m.fast ? m.dt0 : clampf_(t * m.dt_gamma, m.dt_min, m.dt_max);
    skipping = true;
  }
}

template <int C, int H, bool HALFP, bool WIDE>
__global__ void __launch_bounds__(render_threads<H>())
k_render_rays(const void* __restrict__ planes, int R, const half8* __restrict__ packed,
              const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ nears,
              const float* __restrict__ fars, uint32_t N, const uint8_t* __restrict__ grid, float bound, float dt_gamma,
              uint32_t max_steps, uint32_t Cas, uint32_t Hg, float T_thresh, float density_scale,
              const float* __restrict__ noises, int* __restrict__ queue, float* __restrict__ weights_sum,
              float* __restrict__ depth, float* __restrict__ image, int work) {
  using G = FieldGeom<C, H>;
  constexpr int RT = render_threads<H>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half8* w = reinterpret_cast<half8*>(smem);
  for (int i = threadIdx.x; i < G::NF * 64; i += RT) w[i] = packed[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  // ray state, identical in lanes r and r + 32
  int idx = -1;
  MarchCtx m = {};     // a slot the queue never filled still rides through the MFMA chain: defined (zero) operands
  float t = 0.f, last_t = 0.f, far = 0.f, tc = 0.f, skip_to = 0.f;
  bool skipping = false;
  uint32_t cblk = 0xffffffffu;
  unsigned long long cbits = 0ull;
  float ws = 0.f, dd = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
  uint32_t ns = 0;
  bool exhausted = false;   // wave-uniform: the queue has no ray left

  while (true) {
    // ---- 1. refill
    const bool dead = idx < 0;
    const unsigned long long bm = __ballot(dead) & 0xffffffffull;
    if (bm != 0ull && !exhausted) {
      const int cnt = __popcll(bm);
      int base = 0;
      if (lane == 0) base = atomicAdd(queue, cnt);
      base = __shfl(base, 0);
      if (dead) {
        const int id = base + __popcll(bm & ((1ull << r) - 1ull));
        if ((uint32_t)id < N) {
          idx = id;
          march_init(m, rays_o + (size_t)id * 3, rays_d + (size_t)id * 3, bound, dt_gamma, max_steps, Cas, Hg, grid);
          t = nears[id];
          // the perturbation of the first iteration (raymarching.cu:744-746); later iterations add zero
          t = fmaf(clampf_(t * dt_gamma, m.dt_min, m.dt_max), noises != nullptr ? noises[id] : 0.f, t);
          // the composite's own t starts from the UNperturbed near (rays_t, raymarching.cu:866), see below
          tc = nears[id];
          last_t = t; far = fars[id];
          skipping = false;
          cblk = 0xffffffffu;
          ws = dd = cr = cg = cb = 0.f;
          ns = 0;
        }
      }
      if ((uint32_t)(base + cnt) >= N) exhausted = true;
    }
    if (__ballot(idx >= 0) == 0ull) break;

    // ---- 2. next sample of every live ray
    MarchProbe q;
    q.x = q.y = q.z = 0.f; q.dt = 0.f;
    float tdiff = 0.f;
    bool have = false;
    int mst = MARCH_PAUSED;
    if (idx >= 0) {
      mst = march_one_bounded<WIDE>(m, t, last_t, far, cblk, cbits, q, tdiff, skip_to, skipping, work);
      have = mst == MARCH_SAMPLE;
    }
    bool fin = idx >= 0 && mst == MARCH_DONE;      // left the volume without another sample
    // The loop's first iteration takes exactly ONE sample per ray (n_step = N / n_alive = 1) and hands rays_t = near +
    // (t_next - t_start) to the second one (raymarching.cu:893): the perturbation of the start is dropped there, and the
    // long first difference (entry skip) is where that sum can round away from t_next.  Same hand-over here.
    const bool first = have && ns == 0;

    // ---- 3. the field on the 32 samples (the body of k_field_fwd for one tile; rays without a sample ride along)
    if (__ballot(have) != 0ull) {
      constexpr int PG = H > 64 ? 1 : 3;   // planes per group (hidden 128 is register-bound, see field.hip)
      f32x16 acc0[G::OB];
#pragma unroll
      for (int ob = 0; ob < G::OB; ob++) acc0[ob] = zero16();
#pragma unroll
      for (int p0 = 0; p0 < 3; p0 += PG) {
        half8 fk[PG * (C / 16)];
#pragma unroll
        for (int p = p0; p < p0 + PG; p++) {
          TexelTap tp;
          triplane_tap(q.x, q.y, q.z, bound, R, p, tp);
#pragma unroll
          for (int kk = 0; kk < C / 16; kk++) fk[(p - p0) * (C / 16) + kk] = gather_frag<C, HALFP>(planes, R, p, kk, h, tp);
        }
#pragma unroll
        for (int qq = 0; qq < PG * (C / 16); qq++) {
          const int ks = p0 * (C / 16) + qq;
#pragma unroll
          for (int ob = 0; ob < G::OB; ob++) acc0[ob] = MFMA32(w[(G::F0 + ob * G::KS0 + ks) * 64 + lane], fk[qq], acc0[ob]);
        }
        if (PG < 3) __builtin_amdgcn_sched_barrier(0);
      }
      Chain<C, H> ch;
      chain_tail<C, H, false>(w, w, lane, h, acc0, m.dx, m.dy, m.dz, ch);
      // ---- 4. composite (raymarching.cu:853-904); sigma / rgb live in the lanes h == 0, the decision is mirrored
      int stop = 0;
      if (have && h == 0) {
        float sigma = expf(ch.o8[0]);                      // trunc_exp forward (activation.py:9-10)
        if (density_scale != 1.f) sigma = density_scale * sigma;
        const float c0 = 1.f / (1.f + expf(-ch.rgbl[0])), c1 = 1.f / (1.f + expf(-ch.rgbl[1])),
                    c2 = 1.f / (1.f + expf(-ch.rgbl[2]));
        const float alpha = 1.0f - __expf(-sigma * q.dt);
        const float T = 1 - ws;
        const float weight = alpha * T;
        ws += weight;
        tc += tdiff;
        dd += weight * tc;
        cr += weight * c0; cg += weight * c1; cb += weight * c2;
        ns++;
        if (T < T_thresh || ns >= max_steps) stop = 1;
      }
      stop = __shfl(stop, r);
      if (stop) fin = true;
      if (first) {
        const float tc_all = __shfl(tc, r);      // the h == 0 lane's composite t
        tc = tc_all; t = tc_all; last_t = tc_all;
        ns = 1;                                  // (the h == 1 mirror does not count otherwise)
      }
    }

    // ---- 5. finished rays leave their 20 bytes
    if (fin) {
      if (h == 0) {
        weights_sum[idx] = ws;
        depth[idx] = dd;
        image[(size_t)idx * 3] = cr; image[(size_t)idx * 3 + 1] = cg; image[(size_t)idx * 3 + 2] = cb;
      }
      idx = -1;
    }
  }
}

int g_render_work = TNL_RENDER_WORK;   // tnl_render_work: marching work units per ray and trip (<= 0: unbounded)

template <int C, int H>
int launch_render(const void* planes, int half_in, uint32_t R, const void* packed, const float* rays_o, const float* rays_d,
                  const float* nears, const float* fars, uint32_t N, const uint8_t* grid, float bound, float dt_gamma,
                  uint32_t max_steps, uint32_t Cas, uint32_t Hg, float T_thresh, float density_scale, const float* noises,
                  int* queue, float* weights_sum, float* depth, float* image, hipStream_t st) {
  using G = FieldGeom<C, H>;
  const size_t lds = (size_t)G::NF * 1024;
  // persistent workgroups: a few per CU; the queue balances the load
  constexpr int RT = render_threads<H>();
  uint32_t blocks = (N + RT / 2 - 1) / (RT / 2);
  if (blocks > 1024) blocks = 1024;
  const half8* pk = reinterpret_cast<const half8*>(packed);
  const bool wide = (reinterpret_cast<uintptr_t>(grid) & 7u) == 0 && ((size_t)Cas * Hg * Hg * Hg) % 64 == 0;
#define TNL_RENDER(HP, WD)                                                                                               \
  do {                                                                                                                   \
    static bool attr_set[64] = {};          /* per device: the attribute belongs to the device's code object */          \
    int dev_ = 0;                                                                                                        \
    (void)hipGetDevice(&dev_);                                                                                           \
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_rays<C, H, HP, WD>),                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
      if (e != hipSuccess) return (int)e;                                                                                \
      if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;                                                                 \
    }                                                                                                                    \
    hipLaunchKernelGGL((k_render_rays<C, H, HP, WD>), dim3(blocks), dim3(RT), lds, st, planes, (int)R, pk, rays_o, rays_d, \
                       nears, fars, N, grid, bound, dt_gamma, max_steps, Cas, Hg, T_thresh, density_scale, noises, queue,  \
                       weights_sum, depth, image, g_render_work > 0 ? g_render_work : 0x3fffffff);                        \
  } while (0)
  if (half_in) { if (wide) TNL_RENDER(true, true); else TNL_RENDER(true, false); }
  else { if (wide) TNL_RENDER(false, true); else TNL_RENDER(false, false); }
#undef TNL_RENDER
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int tnl_render_work(int units) {
  const int prev = g_render_work;
  if (units != -1) g_render_work = units;
  return prev;
}

int tnl_render_rays(const void* planes_tm, int half_in, uint32_t C, uint32_t R, uint32_t Hd, uint32_t Hc, const void* packed,
                    const float* rays_o, const float* rays_d, const float* nears, const float* fars, uint32_t N,
                    const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t cascades, uint32_t H,
                    float T_thresh, float density_scale, const float* noises, int32_t* queue, float* weights_sum,
                    float* depth, float* image, void* stream) {
  if (N == 0) return 0;
  if (Hd != Hc || queue == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(queue, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
#define TNL_GO(CC, HH)                                                                                                  \
  return launch_render<CC, HH>(planes_tm, half_in, R, packed, rays_o, rays_d, nears, fars, N, grid, bound, dt_gamma,   \
                               max_steps, cascades, H, T_thresh, density_scale, noises, queue, weights_sum, depth, image, st)
  if (C == 16 && Hd == 64) TNL_GO(16, 64);
  if (C == 32 && Hd == 64) TNL_GO(32, 64);
  if (C == 48 && Hd == 128) TNL_GO(48, 128);
#undef TNL_GO
  return (int)hipErrorInvalidValue;
}

}  // extern "C"
