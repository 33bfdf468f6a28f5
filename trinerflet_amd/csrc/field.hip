// field.hip -- fused NeRF field for gfx950: triplane lookup + sigma MLP + SH-4 + colour MLP.
//
// Replaces NeRFNetwork.forward / density of reconstruction/nerf/network.py:118-166 (five cuBLAS
// GEMMs + grid_sample + SH kernel + cat + elementwise kernels, every [M,64] activation round-tripping
// HBM) with ONE kernel in which nothing but the inputs (xyz, dirs), the texels and the outputs
// (sigma, rgb, optional fp16 feature copy for the backward pass) touches memory.
//
// One 64-lane wavefront owns 32 samples: lane (r, h) = (sample r, k-half h).
//  * gather: for every plane and 16-channel k-step the lane loads the 8 channels [16ks+8h, +8) of the
//    four bilinear corners as one 16-byte word each (texel-major fp16 planes: a texel's C channels are
//    contiguous, the x+1 corner is the next C*2 bytes), blends them in fp32 and packs 8 halfs: that IS the
//    MFMA B fragment of layer 0 (see field_common.h) -- no shuffle, no LDS.
//  * the five layers run as v_mfma_f32_32x32x16_f16 chains in registers (field_common.h "chain layout");
//    weights are pre-packed fragments staged once per workgroup in LDS and read with ds_read_b128.
//  * trunc_exp (activation.py:5-17) and sigmoid are applied on the accumulator registers that hold
//    row 0 (sigma) / rows 0..2 (rgb): lanes with h == 0.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trinerflet_hip.h"
#include "field_device.h"

namespace {

// ---------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------
// (Occupancy is not the lever: capping the registers at 256 for two waves per SIMD, `__launch_bounds__(FWD_THREADS, 2)`,
// ran slower at base (1.03 vs 0.95 ms), and an 8-wave workgroup for hidden 128 -- two waves per SIMD sharing the 92 KB
// of weights -- measured 1.61 vs 1.62 ms.  With all of a tile's loads in flight the kernel sits at the ~4 TB/s that
// 64-byte texel gathers reach here: 3.6 GB in 0.81 ms at base, 5.4 GB in 1.5 ms at C = 48.)
#ifndef TNL_FWD_PREFETCH
#define TNL_FWD_PREFETCH 1     // hidden 128: weight fragments a group ahead of their MFMAs (0: A/B builds)
#endif
#ifndef TNL_FWD_LDSW
#define TNL_FWD_LDSW 0
#endif
#ifndef TNL_FWD_MINWAVES
#define TNL_FWD_MINWAVES 1     // A/B builds: minimum workgroups per CU the register allocation must allow
#endif
template <int C, int H, bool HALFP, bool DENSITY_ONLY, bool SAVE>
__global__ void __launch_bounds__(FWD_THREADS, TNL_FWD_MINWAVES)
k_field_fwd(const void* __restrict__ planes, const float* __restrict__ xyz, const float* __restrict__ dirs,
            float bound, uint32_t M, int R, const half8* __restrict__ packed, float* __restrict__ sigma,
            float* __restrict__ rgb, _Float16* __restrict__ feats_save, _Float16* __restrict__ geo_save,
            const int32_t* __restrict__ m_actual) {
  using G = FieldGeom<C, H>;
  if (m_actual != nullptr) M = min(M, (uint32_t)max(*m_actual, 0));
  if (M == 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half8* w = reinterpret_cast<half8*>(smem);
  constexpr int NFR = DENSITY_ONLY ? G::F2 : G::NF;
  constexpr int FT = FWD_THREADS;
  for (int i = threadIdx.x; i < NFR * 64; i += FT) w[i] = packed[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const uint32_t waves = gridDim.x * (FT / 64);
  const uint32_t ntiles = (M + 31) / 32;
  // positions / directions of the NEXT tile are requested one trip ahead: the gather addresses depend on them, and with
  // two waves per SIMD a dependent pair of global latencies per tile is not hidden otherwise
  const uint32_t tile0 = blockIdx.x * (FT / 64) + (threadIdx.x >> 6);
  float npx = 0.f, npy = 0.f, npz = 0.f, ndx = 0.f, ndy = 0.f, ndz = 0.f;
  auto fetch_pos = [&](uint32_t t) {
    const uint32_t i_ = t * 32 + r;
    const uint32_t il_ = i_ < M ? i_ : M - 1;
    npx = xyz[(size_t)il_ * 3]; npy = xyz[(size_t)il_ * 3 + 1]; npz = xyz[(size_t)il_ * 3 + 2];
    if (!DENSITY_ONLY) { ndx = dirs[(size_t)il_ * 3]; ndy = dirs[(size_t)il_ * 3 + 1]; ndz = dirs[(size_t)il_ * 3 + 2]; }
  };
  if (tile0 < ntiles) fetch_pos(tile0);
  for (uint32_t tile = tile0; tile < ntiles; tile += waves) {
#if TNL_FWD_LDSW
    asm volatile("" : "+s"(w));     // A/B: the weight fragments stay in LDS (no hoisting into 128 registers): two waves per SIMD
#endif
    const uint32_t i = tile * 32 + r;
    const bool valid = i < M;
    const float px = npx, py = npy, pz = npz;
    const float cdx = ndx, cdy = ndy, cdz = ndz;
    if (tile + waves < ntiles) fetch_pos(tile + waves);   // (A/B on one box: 1.02 -> 0.96..1.02 ms, kept)
    // All 12 * C / 16 texel loads of the tile are requested before anything consumes them: the feature store used to
    // sit behind `if (feats_save && valid)` inside the gather loop, and that branch kept the scheduler from moving
    // the next group of four loads above it -- six dependent round trips to memory per tile with one wave per SIMD.
    // SAVE is a template parameter and rows past M store into row M - 1 what row M - 1 stores itself (their clamped
    // position is that sample's), so the store needs no predicate.
    // (hidden 128 is register-bound: there the planes are taken one at a time, 12 loads in flight instead of 36)
#ifndef TNL_FWD_PG128
#define TNL_FWD_PG128 1
#endif
    constexpr int PG = H > 64 ? TNL_FWD_PG128 : 3;   // planes per group
    const uint32_t il = valid ? i : M - 1;
    f32x16 acc0[G::OB];
#pragma unroll
    for (int ob = 0; ob < G::OB; ob++) acc0[ob] = zero16();
#pragma unroll
    for (int p0 = 0; p0 < 3; p0 += PG) {
      half8 fk[PG * (C / 16)];
#pragma unroll
      for (int p = p0; p < p0 + PG; p++) {
        TexelTap t;
        triplane_tap(px, py, pz, bound, R, p, t);
#pragma unroll
        for (int kk = 0; kk < C / 16; kk++) fk[(p - p0) * (C / 16) + kk] = gather_frag<C, HALFP>(planes, R, p, kk, h, t);
      }
      if (TNL_FWD_PREFETCH && H > 64) {
        // hidden 128 runs one wave per SIMD: the weight fragments a group ahead of their MFMAs (with_weights), as in the
        // split backward -- left alone every MFMA waits out the LDS read of its own operand
#pragma unroll
        for (int q = 0; q < PG * (C / 16); q++)
          if (SAVE) *reinterpret_cast<half8*>(feats_save + feat_slot<G::KS0>(il, p0 * (C / 16) + q, h)) = fk[q];
        with_weights<PG * (C / 16) * G::OB, G::OB>(
            [&](int i) { return w[(G::F0 + (i % G::OB) * G::KS0 + p0 * (C / 16) + i / G::OB) * 64 + lane]; },
            [&](int i, const half8& f) { acc0[i % G::OB] = MFMA32(f, fk[i / G::OB], acc0[i % G::OB]); });
      } else {
#pragma unroll
        for (int q = 0; q < PG * (C / 16); q++) {
          const int ks = p0 * (C / 16) + q;
          // (a non-temporal store here was measured SLOWER: field_fwd 1.05 -> 1.14 ms)
          if (SAVE) *reinterpret_cast<half8*>(feats_save + feat_slot<G::KS0>(il, ks, h)) = fk[q];
#pragma unroll
          for (int ob = 0; ob < G::OB; ob++) acc0[ob] = MFMA32(w[(G::F0 + ob * G::KS0 + ks) * 64 + lane], fk[q], acc0[ob]);
        }
      }
      if (PG < 3) __builtin_amdgcn_sched_barrier(0);
    }
    const float dx = cdx, dy = cdy, dz = cdz;
    Chain<C, H> ch;
    chain_tail<C, H, DENSITY_ONLY, (TNL_FWD_PREFETCH && H > 64)>(w, w, lane, h, acc0, dx, dy, dz, ch);
    if (!DENSITY_ONLY && geo_save != nullptr && valid) {
      // hidden 128: the colour half of the split backward starts from the 16 sigma-net outputs instead of recomputing them
      half8 g;
#pragma unroll
      for (int j = 0; j < 8; j++) g[j] = (_Float16)ch.o8[j];
      *reinterpret_cast<half8*>(geo_save + (size_t)i * 16 + 8 * h) = g;
    }
    if (valid && h == 0) {
      sigma[i] = expf(ch.o8[0]);  // trunc_exp forward (activation.py:9-10)
      if (!DENSITY_ONLY) {
        rgb[(size_t)i * 3 + 0] = 1.f / (1.f + expf(-ch.rgbl[0]));
        rgb[(size_t)i * 3 + 1] = 1.f / (1.f + expf(-ch.rgbl[1]));
        rgb[(size_t)i * 3 + 2] = 1.f / (1.f + expf(-ch.rgbl[2]));
      }
    }
    if (DENSITY_ONLY && rgb != nullptr && valid) {
      // geo features (rows 1..15 of the layer-1 tile) -> rgb reused as a [M,15] fp32 buffer by density()
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const int row = acc_row(g, h);
        if (row >= 1) rgb[(size_t)i * 15 + row - 1] = ch.o8[g];
      }
    }
  }
}

template <int C, int H>
__global__ void k_field_pack(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2,
                             const float* __restrict__ W3, const float* __restrict__ W4,
                             _Float16* __restrict__ packed) {
  using G = FieldGeom<C, H>;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G::NTOT * 512) return;
  const int src = field_pack_source<C, H>(idx);
  float v = 0.f;
  if (src >= 0) {
    if (src < G::OFF1) v = W0[src - G::OFF0];
    else if (src < G::OFF2) v = W1[src - G::OFF1];
    else if (src < G::OFF3) v = W2[src - G::OFF2];
    else if (src < G::OFF4) v = W3[src - G::OFF3];
    else v = W4[src - G::OFF4];
  }
  packed[idx] = (_Float16)v;
}

template <int C, int H>
int launch_fwd(const void* planes, int half_in, const float* xyz, const float* dirs, float bound, uint32_t M,
               uint32_t R, const void* packed, float* sigma, float* rgb, void* feats_save, bool density_only,
               const int32_t* m_actual, hipStream_t st) {
  using G = FieldGeom<C, H>;
  const uint32_t ntiles = (M + 31) / 32;
  constexpr uint32_t WPB = FWD_THREADS / 64;   // tiles (waves) per workgroup
  uint32_t blocks = (ntiles + WPB - 1) / WPB;
  if (blocks > 2048) blocks = 2048;
  const half8* pk = reinterpret_cast<const half8*>(packed);
  _Float16* fs = reinterpret_cast<_Float16*>(feats_save);
  _Float16* gs = (split_backward<H>() && fs != nullptr) ? fs + (size_t)((M + 31) / 32 * 32) * G::F : nullptr;   // see tnl_field_feats_save_bytes
#define TNL_LAUNCH(HP, DO, SV)                                                                                        \
  hipLaunchKernelGGL((k_field_fwd<C, H, HP, DO, SV>), dim3(blocks), dim3(FWD_THREADS), (DO ? G::F2 : G::NF) * 1024, st, \
                     planes, xyz, dirs, bound, M, (int)R, pk, sigma, rgb, fs, gs, m_actual)
#define TNL_LAUNCH_SV(HP, DO) do { if (fs != nullptr) TNL_LAUNCH(HP, DO, true); else TNL_LAUNCH(HP, DO, false); } while (0)
  if (density_only) {
    if (half_in) TNL_LAUNCH_SV(true, true); else TNL_LAUNCH_SV(false, true);
  } else {
    if (half_in) TNL_LAUNCH_SV(true, false); else TNL_LAUNCH_SV(false, false);
  }
#undef TNL_LAUNCH_SV
#undef TNL_LAUNCH
  return (int)hipGetLastError();
}

}  // namespace

// The hidden-128 instantiation is compiled as its own object, field_h128.hip (this file with TNL_FIELD_H128_ONLY), WITHOUT the
// SLP vectoriser: packing the blend's fp32 multiply-adds into v_pk_*_f32 costs that kernel 10 % (large: forward 1.41 -> 1.27 ms)
// and gains the hidden-64 kernels 8 % (base 0.78 vs 0.85 ms): profiles/r06o_ab_no_slp.txt.
int tnl_field_forward_h128(const void* planes, int half_in, const float* xyz, const float* dirs, float bound, uint32_t M,
                           uint32_t R, const void* packed, float* sigma, float* rgb, void* feats_save, bool density_only,
                           const int32_t* m_actual, hipStream_t st);
#ifdef TNL_FIELD_H128_ONLY
int tnl_field_forward_h128(const void* planes, int half_in, const float* xyz, const float* dirs, float bound, uint32_t M,
                           uint32_t R, const void* packed, float* sigma, float* rgb, void* feats_save, bool density_only,
                           const int32_t* m_actual, hipStream_t st) {
  return launch_fwd<48, 128>(planes, half_in, xyz, dirs, bound, M, R, packed, sigma, rgb, feats_save, density_only, m_actual, st);
}
#else

// backward kernels live in field_bwd.hip; these helpers are shared through this header-less pair
extern "C" {

uint32_t tnl_field_packed_bytes(uint32_t C, uint32_t Hd, uint32_t Hc) {
  if (Hd != Hc) return 0;
  if (C == 16 && Hd == 64) return FieldGeom<16, 64>::NTOT * 1024;
  if (C == 32 && Hd == 64) return FieldGeom<32, 64>::NTOT * 1024;
  if (C == 48 && Hd == 128) return FieldGeom<48, 128>::NTOT * 1024;
  return 0;
}

uint64_t tnl_field_feats_save_bytes(uint32_t M, uint32_t C, uint32_t Hd) {
  // [ceil(M/32)*32][3C] blocked by 32-sample tiles (field_common.h feat_slot) + [M][16] sigma-net outputs at hidden 128;
  // 5 GB at the 26 M samples of an untrained grid: 64-bit
  return ((uint64_t)((M + 31) / 32) * 32 * 3 * C + (uint64_t)M * (Hd > 64 ? 16 : 0)) * 2;
}

int tnl_field_pack(const float* W0, const float* W1, const float* W2, const float* W3, const float* W4, uint32_t C,
                   uint32_t Hd, uint32_t Hc, void* packed, void* stream) {
  if (Hd != Hc) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  _Float16* pk = reinterpret_cast<_Float16*>(packed);
#define TNL_PACK(CC, HH)                                                                                      \
  hipLaunchKernelGGL((k_field_pack<CC, HH>), dim3((FieldGeom<CC, HH>::NTOT * 512 + 255) / 256), dim3(256), 0, st, \
                     W0, W1, W2, W3, W4, pk)
  if (C == 16 && Hd == 64) TNL_PACK(16, 64);
  else if (C == 32 && Hd == 64) TNL_PACK(32, 64);
  else if (C == 48 && Hd == 128) TNL_PACK(48, 128);
  else return (int)hipErrorInvalidValue;
#undef TNL_PACK
  return (int)hipGetLastError();
}

// rgb == NULL, or dirs == NULL: density only (NeRFNetwork.density, network.py:149-166); if dirs == NULL and
// rgb != NULL, rgb receives the 15 geo features per sample ([M,15] fp32).
int tnl_field_forward(const void* planes_tm, int half_in, const float* xyz, const float* dirs, float bound,
                      uint32_t M, uint32_t C, uint32_t R, uint32_t Hd, uint32_t Hc, const void* packed,
                      float* sigma, float* rgb, void* feats_save, const int32_t* m_actual, void* stream) {
  if (M == 0) return 0;
  if (Hd != Hc) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const bool density_only = (dirs == nullptr) || (rgb == nullptr);
  if (C == 16 && Hd == 64)
    return launch_fwd<16, 64>(planes_tm, half_in, xyz, dirs, bound, M, R, packed, sigma, rgb, feats_save, density_only, m_actual, st);
  if (C == 32 && Hd == 64)
    return launch_fwd<32, 64>(planes_tm, half_in, xyz, dirs, bound, M, R, packed, sigma, rgb, feats_save, density_only, m_actual, st);
  if (C == 48 && Hd == 128)
    return tnl_field_forward_h128(planes_tm, half_in, xyz, dirs, bound, M, R, packed, sigma, rgb, feats_save, density_only, m_actual, st);
  return (int)hipErrorInvalidValue;
}

}  // extern "C"
#endif  // TNL_FIELD_H128_ONLY
