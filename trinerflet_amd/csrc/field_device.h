// field_device.h -- device code shared by field.hip (forward) and field_bwd.hip (backward).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field_common.h"
#include "triplane_common.h"

namespace {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int FWD_THREADS = 256;

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): `done` is the call site's own static table (one
// per template instantiation); the attribute belongs to the device's code object, not to the process
template <typename K>
inline hipError_t ensure_dynamic_lds(K kernel, int bytes, bool (&done)[64]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && dev >= 0 && dev < 64) done[dev] = true;
  return e;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; i++) z[i] = 0.f;
  return z;
}

// registers 8s..8s+7 of an accumulator tile -> fp16 fragment (optionally through ReLU).  The ReLU runs on the packed
// halfs (v_cvt_pk_f16_f32 + v_pk_max_f16: 1 instruction per element instead of 2.5): rounding to fp16 is monotone and
// keeps the sign, so max(round(v), 0) == round(max(v, 0)).
template <bool RELU>
__device__ __forceinline__ half8 acc_to_frag(const f32x16& a, int s) {
  half8 f;
#pragma unroll
  for (int j = 0; j < 8; j++) f[j] = (_Float16)(s == 0 ? a[j] : a[8 + j]);
  if (RELU) {
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    f = __builtin_elementwise_max(f, z);
  }
  return f;
}

// ReLU backward on fragments: d where the (ReLU'd, hence >= +0) activation x is positive, else 0.  Per packed pair:
// (x + 0x7fff) has its sign bit set iff x != 0; an arithmetic shift by 15 turns that into 0xffff / 0x0000.  Pinned with
// asm because the compiler rewrites the C form into 16-bit compares and selects (3 instructions per element).
__device__ __forceinline__ half8 relu_mask_frag(half8 d, half8 x) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 db = __builtin_bit_cast(u32x4, d);
  const u32x4 xb = __builtin_bit_cast(u32x4, x);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t t, m;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(xb[q]), "s"(0x7fff7fffu));
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(m) : "s"(0x000f000fu), "v"(t));
    db[q] &= m;
  }
  return __builtin_bit_cast(half8, db);
}

// SH degree 4 (shencoder.cu:50-68): the 8 values [8h, 8h+8) of the 16, as a fragment
__device__ __forceinline__ half8 sh_frag(float x, float y, float z, int h) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  half8 f;
  if (h == 0) {
    f[0] = (_Float16)0.28209479177387814f;
    f[1] = (_Float16)(-0.48860251190291987f * y);
    f[2] = (_Float16)(0.48860251190291987f * z);
    f[3] = (_Float16)(-0.48860251190291987f * x);
    f[4] = (_Float16)(1.0925484305920792f * xy);
    f[5] = (_Float16)(-1.0925484305920792f * yz);
    f[6] = (_Float16)(0.94617469575755997f * z2 - 0.31539156525251999f);
    f[7] = (_Float16)(-1.0925484305920792f * xz);
  } else {
    f[0] = (_Float16)(0.54627421529603959f * x2 - 0.54627421529603959f * y2);
    f[1] = (_Float16)(0.59004358992664352f * y * (-3.0f * x2 + y2));
    f[2] = (_Float16)(2.8906114426405538f * xy * z);
    f[3] = (_Float16)(0.45704579946446572f * y * (1.0f - 5.0f * z2));
    f[4] = (_Float16)(0.3731763325901154f * z * (5.0f * z2 - 3.0f));
    f[5] = (_Float16)(0.45704579946446572f * x * (1.0f - 5.0f * z2));
    f[6] = (_Float16)(1.4453057213202769f * z * (x2 - y2));
    f[7] = (_Float16)(0.59004358992664352f * x * (-x2 + 3.0f * y2));
  }
  return f;
}

// 8 channels [c0, c0+8) of one texel as floats
template <bool HALFP>
__device__ __forceinline__ void load8(const void* planes, size_t elem, float (&v)[8]) {
  if (HALFP) {
    const half8 t = *reinterpret_cast<const half8*>(reinterpret_cast<const _Float16*>(planes) + elem);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (float)t[j];
  } else {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(planes) + elem);
    const float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}

// bilinear blend of 8 channels of plane p at k-step kk for this lane's sample -> layer-0 B fragment
template <int C, bool HALFP>
__device__ __forceinline__ half8 gather_frag(const void* planes, int R, int p, int kk, int h, const TexelTap& t) {
  const int c0 = 16 * kk + 8 * h;
  const size_t pb = (size_t)p * R * R;
  float v00[8], v01[8], v10[8], v11[8];
  load8<HALFP>(planes, (pb + (size_t)t.y0 * R + t.x0) * C + c0, v00);
  load8<HALFP>(planes, (pb + (size_t)t.y0 * R + t.x1) * C + c0, v01);
  load8<HALFP>(planes, (pb + (size_t)t.y1 * R + t.x0) * C + c0, v10);
  load8<HALFP>(planes, (pb + (size_t)t.y1 * R + t.x1) * C + c0, v11);
  half8 f;
#pragma unroll
  for (int j = 0; j < 8; j++)
    f[j] = (_Float16)(v00[j] * t.w00 + v01[j] * t.w01 + v10[j] * t.w10 + v11[j] * t.w11);
  return f;
}

// N weight fragments wsrc(0..N-1), consumed in order by body(i, fragment): the fragments are requested GROUP at a time,
// ONE GROUP AHEAD of the MFMAs that take them, with a scheduling fence per group.  One wave per SIMD issues in order and
// nothing else covers an LDS read (ds_read_b128: ~100+ cycles): left alone the compiler places every read right in front
// of its MFMA (`wait, MFMA, read` per 32-cycle MFMA -- profiles of the hidden-128 backward, round 5); here a group's reads
// are in flight while the previous group's MFMAs execute.  Costs 2 * GROUP * 4 registers.
template <int N, int GROUP, class WSrc, class Body>
__device__ __forceinline__ void with_weights(WSrc wsrc, Body body) {
  constexpr int NG = (N + GROUP - 1) / GROUP;
  half8 buf[2][GROUP];
#pragma unroll
  for (int j = 0; j < GROUP; j++)
    if (j < N) buf[0][j] = wsrc(j);
#pragma unroll
  for (int g = 0; g < NG; g++) {
#pragma unroll
    for (int j = 0; j < GROUP; j++)
      if (g + 1 < NG && (g + 1) * GROUP + j < N) buf[(g + 1) & 1][j] = wsrc((g + 1) * GROUP + j);
#pragma unroll
    for (int j = 0; j < GROUP; j++)
      if (g * GROUP + j < N) body(g * GROUP + j, buf[g & 1][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------
// the MLP chain on one 32-sample tile, shared by forward and backward-recompute
// ---------------------------------------------------------------------------------------------
template <int C, int H>
struct Chain {
  using G = FieldGeom<C, H>;
  half8 h1[G::KH];   // relu(H1) fragments
  float o8[8];       // layer-1 output tile, registers 0..7: rows {0..3, 8..11} + 4h (row 0 = sigma logit, 1..15 geo)
  half8 h3[G::KH];
  half8 h4[G::KH];
  float rgbl[3];     // layer-4 output rows 0..2 (rgb logits; meaningful on lanes h == 0)
};

template <int C, int H>
__device__ __forceinline__ const half8& wfrag(const half8* w, int f, int lane) { return w[f * 64 + lane]; }

template <int C, int H, bool FENCE = false>
__device__ __forceinline__ void chain_colour(const half8* w, const half8* wH, int lane, int h, const half8 geo,
                                             float dx, float dy, float dz, Chain<C, H>& ch);

// layers 1..4 given acc0 = W0 * F^T ; DENSITY_ONLY stops after layer 1.  `w` serves the fragments of layers 1 and 2,
// `wH` those of layers 3 and 4 (the same table unless a kernel keeps only part of it in LDS).
template <int C, int H, bool DENSITY_ONLY, bool FENCE = false>
__device__ __forceinline__ void chain_tail(const half8* w, const half8* wH, int lane, int h, f32x16 (&acc0)[H / 32],
                                           float dx, float dy, float dz, Chain<C, H>& ch) {
  using G = FieldGeom<C, H>;
#pragma unroll
  for (int ks = 0; ks < G::KH; ks++) {
    ch.h1[ks] = (ks & 1) ? acc_to_frag<true>(acc0[ks >> 1], 1) : acc_to_frag<true>(acc0[ks >> 1], 0);
  }
  f32x16 o = zero16();
  if (FENCE) {
    with_weights<G::KH, 4>([&](int i) { return w[(G::F1 + i) * 64 + lane]; },
                           [&](int i, const half8& f) { o = MFMA32(f, ch.h1[i], o); });
  } else {
#pragma unroll
    for (int ks = 0; ks < G::KH; ks++) o = MFMA32(w[(G::F1 + ks) * 64 + lane], ch.h1[ks], o);
  }
#pragma unroll
  for (int g = 0; g < 8; g++) ch.o8[g] = o[g];
  if (DENSITY_ONLY) return;
  chain_colour<C, H, FENCE>(w, wH, lane, h, acc_to_frag<false>(o, 0), dx, dy, dz, ch);
}

// layers 2..4 from the 16 sigma-net outputs as a fragment (slot 0 = the logit, multiplied by zero weights)
template <int C, int H, bool FENCE>
__device__ __forceinline__ void chain_colour(const half8* w, const half8* wH, int lane, int h, const half8 geo,
                                             float dx, float dy, float dz, Chain<C, H>& ch) {
  using G = FieldGeom<C, H>;
  const half8 shf = sh_frag(dx, dy, dz, h);
  f32x16 acc2[G::OB];
#pragma unroll
  for (int ob = 0; ob < G::OB; ob++) {
    acc2[ob] = MFMA32(w[(G::F2 + 2 * ob) * 64 + lane], shf, zero16());
    acc2[ob] = MFMA32(w[(G::F2 + 2 * ob + 1) * 64 + lane], geo, acc2[ob]);
  }
#pragma unroll
  for (int ks = 0; ks < G::KH; ks++)
    ch.h3[ks] = (ks & 1) ? acc_to_frag<true>(acc2[ks >> 1], 1) : acc_to_frag<true>(acc2[ks >> 1], 0);
  f32x16 out = zero16();
  if (FENCE) {     // the split backward of hidden 128 (one wave per SIMD): weight fragments a group ahead (with_weights)
    // (tile ob's conversion is issued behind the first group of tile ob + 1's MFMAs: two accumulators alternate)
    f32x16 t3a = zero16(), t3b = zero16();
    auto post3 = [&](int ob, const f32x16& t) { ch.h4[2 * ob] = acc_to_frag<true>(t, 0); ch.h4[2 * ob + 1] = acc_to_frag<true>(t, 1); };
    with_weights<G::OB * G::KH, 4>([&](int i) { return wH[(G::F3 + i) * 64 + lane]; }, [&](int i, const half8& f) {
      const int ob = i / G::KH, ks = i % G::KH;
      f32x16& t3 = (ob & 1) ? t3b : t3a;
      if (ks == 0) t3 = zero16();
      t3 = MFMA32(f, ch.h3[ks], t3);
      if (ks == 3 && ob > 0) post3(ob - 1, (ob & 1) ? t3a : t3b);
      if (i == G::OB * G::KH - 1) post3(ob, t3);
    });
    with_weights<G::KH, 4>([&](int i) { return wH[(G::F4 + i) * 64 + lane]; },
                           [&](int i, const half8& f) { out = MFMA32(f, ch.h4[i], out); });
  } else {
    f32x16 acc3[G::OB];
#pragma unroll
    for (int ob = 0; ob < G::OB; ob++) {
      acc3[ob] = zero16();
#pragma unroll
      for (int ks = 0; ks < G::KH; ks++) acc3[ob] = MFMA32(wH[(G::F3 + ob * G::KH + ks) * 64 + lane], ch.h3[ks], acc3[ob]);
    }
#pragma unroll
    for (int ks = 0; ks < G::KH; ks++)
      ch.h4[ks] = (ks & 1) ? acc_to_frag<true>(acc3[ks >> 1], 1) : acc_to_frag<true>(acc3[ks >> 1], 0);
#pragma unroll
    for (int ks = 0; ks < G::KH; ks++) out = MFMA32(wH[(G::F4 + ks) * 64 + lane], ch.h4[ks], out);
  }
  ch.rgbl[0] = out[0]; ch.rgbl[1] = out[1]; ch.rgbl[2] = out[2];
}


}  // namespace
