// field_bwd_rows.hip -- the hidden-64 binned backward of the fused field in its register-only ("rows") form.
// Own translation unit: it is compiled with its own scheduler settings (build.py PER_FILE).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "field_bwd_rows.h"
#include "field_device.h"

#ifndef TNL_ROWS_STAMP
#define TNL_ROWS_STAMP 0   // 1: wave 0 of workgroup 0 records s_memtime after every slot of its first tiles (tools/rows_stamps.py)
#endif
#if TNL_ROWS_STAMP
__device__ unsigned long long g_rows_stamps[64 * 64];
extern "C" __attribute__((visibility("default"))) int tnl_debug_rows_stamps(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rows_stamps), sizeof(g_rows_stamps));
}
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the hidden-64 binned backward WITHOUT stage images, barriers or transposing LDS reads ("rows" form).
//
// The weight gradient dW_l = dY_l^T X_l needs both operands with the SAMPLES on the k axis, while the chain keeps every
// activation with the samples on the lanes.  The shared-stage forms above turn one into the other through LDS (ds_write_b64
// of the chain fragments, a workgroup barrier, ds_read_b64_tr_b16) and spend more than half of their wave cycles parked
// on that round trip.  Here the turn is an MFMA: a chain fragment f (A operand: A[row = sample][k = feature slot]) times
// a constant 0/1 fragment I (B[k = feature slot][col = feature]) gives D[sample][feature] -- the same numbers (exact:
// one product per output, fp32 accumulate) in the accumulator layout, whose lanes are FEATURES and whose registers are
// SAMPLES; converted back to fp16, registers 8s .. 8s+7 are the k-step-s operand of the weight-gradient MFMA (A and B
// take the same lane map, and both operands carry the same slot -> sample permutation, so the sum over k is the sum over
// the tile's 32 samples).  Per 32-sample tile: 60 chain MFMAs as before + 34 turning MFMAs + 32 weight-gradient MFMAs
// (K = 32: two per tile), 144 more v_cvt_pk -- and no LDS traffic besides the weight fragments, no barrier, no wave ever
// waits for another.  Every wave holds all 16 (C = 16: 14) weight-gradient tiles (256 accumulator registers: one wave per
// SIMD, four independent waves per workgroup sharing the fragment table); the four waves' tiles are summed through LDS
// once, at the end of the kernel, into the workgroup's slab.
struct IdFrags {
  half8 pe, po;   // permuted slot order (fragments made from accumulator tiles), even / odd k-step of a 32-feature block
  half8 ne, no;   // natural slot order (saved features, SH)
};

__device__ __forceinline__ IdFrags make_identity(int r, int h) {
  IdFrags I;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int cp = kslot_feature(0, h, j), cn = 8 * h + j;   // column (feature within the block) slot (h, j) feeds, even k-step
    I.pe[j] = (_Float16)(r == cp ? 1.f : 0.f);
    I.po[j] = (_Float16)(r == 16 + cp ? 1.f : 0.f);
    I.ne[j] = (_Float16)(r == cn ? 1.f : 0.f);
    I.no[j] = (_Float16)(r == 16 + cn ? 1.f : 0.f);
  }
  return I;
}

// One weight-gradient tile += dY'^T X' over the tile's 32 samples (two k-steps).  Written as inline assembly with the
// accumulator constrained to AGPRs: the 16 tiles (256 registers) are read by nothing but these MFMAs until the kernel's
// epilogue, and with them pinned to the accumulator half of the register file the 256 architectural VGPRs are left to the
// chain (the file's other MFMAs are built in VGPR form, -amdgpu-mfma-vgpr-form=1, so that the VALU converts their results
// without v_accvgpr_read).  The compiler's hazard recogniser does not look inside inline assembly: the s_nop in front
// covers the VALU-write -> MFMA-read distance of the operands, the one between the two MFMAs the dependent accumulate.
__device__ __forceinline__ void dw_rows(const half8 (&y)[2], const half8 (&x)[2], f32x16& acc) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %3, %4, %0"
               : "+a"(acc) : "v"(y[0]), "v"(x[0]), "v"(y[1]), "v"(x[1]));
}

// slab_tile into / through the workgroup's LDS reduction buffer: OP 0 store, 1 add in place, 2 global = LDS + tile
template <int MODE, int OP>
__device__ __forceinline__ void red_tile(float* lds, float* slab, int off, int out_dim, int in_dim, int ob, int ib,
                                         const f32x16& a, int r, int h, int off4 = 0) {
  int in = 32 * ib + r;
  bool in_ok = in < in_dim;
  if (MODE == 2) { in_ok = r != 16; in = r < 16 ? r : r - 1; }
#pragma unroll
  for (int g = 0; g < 16; g++) {
    int out = 32 * ob + acc_row(g, h);
    bool out_ok = out < out_dim;
    int base = off;
    if (MODE == 1) {     // rows 0..15: W1 in chain-slot order; rows 16..18: W4 (off4)
      out_ok = out < 19;
      if (out >= 16) { out -= 16; base = off4; }
      else out = out == 15 ? 0 : out + 1;
    }
    if (out_ok && in_ok) {
      const int idx = base + out * in_dim + in;
      if (OP == 0) lds[idx] = a[g];
      else if (OP == 1) lds[idx] += a[g];
      else slab[idx] = lds[idx] + a[g];
    }
  }
}

template <int C>
struct RowsGeom {
  static constexpr int H = 64;
  using G = FieldGeom<C, H>;
  static constexpr int NWV = 4, ST = 32 * NWV;
  static constexpr int NT0 = G::OB * G::IB0, NT1 = G::OB, NT2 = G::OB, NT3 = G::OB * G::OB, NT4 = G::OB;
  // layer 4 (3 outputs) shares layer 1's tiles (16 outputs): its dY is turned into columns 16.. of the block, so its
  // weight gradient lands in rows 16..18 of the same accumulators (two tiles = 32 AGPRs less)
  static constexpr int B3 = 0, B2 = B3 + NT3, B1 = B2 + NT2, B0 = B1 + NT1, NTILES = B0 + NT0, B4 = B1;
  static constexpr size_t W_BYTES = (size_t)G::NTOT * 1024;
  static constexpr size_t RED_BYTES = (size_t)G::NW * 4;      // aliases the fragment table after the sample loop
  // per wave two buffers of one tile's saved features (KS0 pieces of 1 KiB = 32 samples x 16 halfs x 2 lane halves),
  // filled by LDS-DMA one tile ahead (no registers held) and read twice per tile (layer 0 forward, layer 0 weight gradient)
  static constexpr size_t FBUF_BYTES = (size_t)G::KS0 * 1024;
  static constexpr size_t F_BYTES = (size_t)NWV * 2 * FBUF_BYTES;
  static constexpr size_t LDS_BYTES = (W_BYTES > RED_BYTES ? W_BYTES : RED_BYTES) + F_BYTES;
};

template <int C>
__global__ void __launch_bounds__(256, 1)
k_field_bwd_rows(const float* __restrict__ gsig, const float* __restrict__ grgb, const _Float16* __restrict__ feats,
                 const float* __restrict__ dirs, uint32_t M, const half8* __restrict__ packed, float* __restrict__ slabs,
                 const int32_t* __restrict__ m_actual, _Float16* __restrict__ dfeat) {
  constexpr int H = 64;
  using G = FieldGeom<C, H>;
  using B = RowsGeom<C>;
  constexpr int ST = B::ST;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t Mcap = M;   // row capacity: the plane stride of the plane-major dfeat output
  if (m_actual != nullptr) M = min(M, (uint32_t)max(*m_actual, 0));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half8* const w = reinterpret_cast<half8*>(smem);
  for (int i = threadIdx.x; i < G::NTOT * 64; i += 256) w[i] = packed[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int col = 32 * wv + r;
  const IdFrags I = make_identity(r, h);

  f32x16 dw[B::NTILES];
#pragma unroll
  for (int k = 0; k < B::NTILES; k++) dw[k] = zero16();

  struct Inputs {
    float dx, dy, dz, g_s, g_c0, g_c1, g_c2;
    half8 fk[G::KS0];
  };
  // Weight fragments come from LDS (ds_read_b128, ~100+ cycles).  With one wave per SIMD nothing hides that latency, and
  // left to itself the compiler issues each read right in front of the MFMA that takes it (read, wait, MFMA: ~100 cycles
  // per 32-cycle MFMA, measured 0.86 ms against the shared-stage kernel's 0.76).  So the loop is written as stages: the
  // fragments of stage k+1 are requested at the top of stage k, and scheduling fences (sched_barrier) keep the requests
  // above the stage's arithmetic and the stages in order; inside a stage the compiler is free to interleave the
  // independent MFMA chains (data path, turns, weight gradients) and the conversions between them.
  auto ldw = [&](half8* dst, int base, int n) {
#pragma unroll
    for (int k = 0; k < 16; k++)
      if (k < n) dst[k] = w[(base + k) * 64 + lane];
  };
#define ROWS_STAGE __builtin_amdgcn_sched_barrier(0)
  constexpr int OB = G::OB, KH = G::KH, KS0 = G::KS0, IB0 = G::IB0;
  const uint32_t nst = M == 0 ? 0 : (M + ST - 1) / ST;
  Inputs nxt;
  if (blockIdx.x < nst) {
    const uint32_t i_ = blockIdx.x * ST + col, il_ = i_ < M ? i_ : M - 1;
    nxt.dx = dirs[(size_t)il_ * 3]; nxt.dy = dirs[(size_t)il_ * 3 + 1]; nxt.dz = dirs[(size_t)il_ * 3 + 2];
    nxt.g_s = i_ < M ? gsig[i_] : 0.f;
    nxt.g_c0 = i_ < M ? grgb[(size_t)i_ * 3] : 0.f; nxt.g_c1 = i_ < M ? grgb[(size_t)i_ * 3 + 1] : 0.f;
    nxt.g_c2 = i_ < M ? grgb[(size_t)i_ * 3 + 2] : 0.f;
  }
  half8 wA[OB * KS0];                     // layer 0 forward, [ob][ks]
  ldw(wA, G::F0, OB * KS0);
  // One wave per SIMD issues in order, so an MFMA batch followed by the
  // conversion of its own result leaves the matrix pipe idle during the conversion and the vector ALU idle during the
  // MFMAs (the staged form above: 0.89 ms).  Here the tile's work is a fixed list of BATCHES -- a few MFMAs into one or
  // two accumulator tiles, then a POST part on the vector ALU (convert / ReLU / mask / store) -- issued as
  //     M(b1) | M(b2) post(b1) | M(b3) post(b2) | ...
  // with a scheduling fence at every bar: the post of a batch runs while the NEXT batch's MFMAs execute.  Batch k+1 must
  // therefore not read what post(k) produces: links of the dependent chain (layer l -> l+1) alternate with independent
  // fillers -- the turning MFMAs and the weight-gradient MFMAs, each placed where its inputs die anyway (the turned copy
  // replaces the chain fragments in the register file).  Weight fragments are requested (ldw) several batches ahead.
#if TNL_ROWS_STAMP
#define STAMP() if (stamp_on) { if (stamp_k < 64) g_rows_stamps[stamp_t * 64 + stamp_k] = __builtin_readcyclecounter(); stamp_k++; }
#else
#define STAMP()
#endif
// (no fence between a slot's two parts: the compiler may place the previous batch's post between this batch's MFMAs)
#define SLOT(MCODE, ...) { MCODE; } { __VA_ARGS__; } ROWS_STAGE; STAMP() if (TNL_ROWS_STAMP) ROWS_STAGE;
#define TURN2(T, FE, FO, IE, IO) T = MFMA32(FE, IE, zero16()); T = MFMA32(FO, IO, T)
#define TURN1(T, FE, IE) T = MFMA32(FE, IE, zero16())
#define POST_T(T, X) X[0] = acc_to_frag<false>(T, 0); X[1] = acc_to_frag<false>(T, 1)
#define POST_RELU(T, F0_, F1_) F0_ = acc_to_frag<true>(T, 0); F1_ = acc_to_frag<true>(T, 1)
#define POST_MASK(T, F0_, F1_, H0_, H1_) F0_ = relu_mask_frag(acc_to_frag<false>(T, 0), H0_); F1_ = relu_mask_frag(acc_to_frag<false>(T, 1), H1_)
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
#if TNL_ROWS_STAMP
  int stamp_t = 0;
#endif
  // The saved features reach the wave through LDS: global -> LDS by DMA (global_load_lds_dwordx4, 1 KiB per instruction,
  // lane-linear: lane l's 16 bytes land at piece + 16 l, and lane l reads them back from there), issued a whole tile ahead
  // -- a global load takes 2-2.5 us under this kernel's load (measured with s_memtime stamps: 3.6 k cycles parked on a
  // re-read issued 2.5 k cycles earlier), and with one wave per SIMD nothing covers it, while holding the next tile's
  // 24 registers beside this tile's would spill.  Only the issuing wave reads its pieces: its own vmcnt orders them.
  char* const fbuf = smem + (B::LDS_BYTES - B::F_BYTES) + (size_t)wv * 2 * B::FBUF_BYTES;
  auto dma_feats = [&](uint32_t st_, int b_) {
    const uint32_t i_ = st_ * ST + col;
    const uint32_t il_ = i_ < M ? i_ : M - 1;
#pragma unroll
    for (int ks = 0; ks < KS0; ks++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(feats + feat_slot<KS0>(il_, ks, h)),
                                       (__attribute__((address_space(3))) void*)(fbuf + ((size_t)b_ * KS0 + ks) * 1024), 16, 0, 0);
  };
  auto read_feats = [&](half8* dst, int b_, bool valid_) {
#pragma unroll
    for (int ks = 0; ks < KS0; ks++) {
      dst[ks] = *reinterpret_cast<const half8*>(fbuf + ((size_t)b_ * KS0 + ks) * 1024 + lane * 16);
      if (!valid_) {
#pragma unroll
        for (int j = 0; j < 8; j++) dst[ks][j] = (_Float16)0.f;
      }
    }
  };
  auto load_scalars = [&](uint32_t st_, Inputs& in) {
    const uint32_t i_ = st_ * ST + col;
    const bool v_ = i_ < M;
    const uint32_t il_ = v_ ? i_ : M - 1;
    in.dx = dirs[(size_t)il_ * 3]; in.dy = dirs[(size_t)il_ * 3 + 1]; in.dz = dirs[(size_t)il_ * 3 + 2];
    in.g_s = v_ ? gsig[i_] : 0.f;
    in.g_c0 = v_ ? grgb[(size_t)i_ * 3] : 0.f; in.g_c1 = v_ ? grgb[(size_t)i_ * 3 + 1] : 0.f;
    in.g_c2 = v_ ? grgb[(size_t)i_ * 3 + 2] : 0.f;
  };
  int cur = 0;
  if (blockIdx.x < nst) dma_feats(blockIdx.x, 0);
  for (uint32_t st = blockIdx.x; st < nst; st += gridDim.x) {
#if TNL_ROWS_STAMP
    const bool stamp_on = blockIdx.x == 0 && threadIdx.x == 0 && stamp_t < 63;
    int stamp_k = 0;
    STAMP()
#endif
    const uint32_t i = st * ST + col;
    const bool valid = i < M;
    Inputs in = nxt;
    const float dx = in.dx, dy = in.dy, dz = in.dz;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's pieces have landed
    ROWS_STAGE;
    read_feats(in.fk, cur, valid);
    if (st + gridDim.x < nst) { dma_feats(st + gridDim.x, cur ^ 1); load_scalars(st + gridDim.x, nxt); }
    ROWS_STAGE;
    half8 h1[KH], h3[KH], h4[KH], d4f[KH], d3f[KH], d1f[KH], geo, dof, dz4f;
    float logit0;
    f32x16 tA0, tA1, tO, tL2a, tL2b, tL3a, tL3b, tOut, tD4a, tD4b, tD3a, tD3b, tDzz, tD1a, tD1b, tDF[3];
    f32x16 tXF[3], tX4a, tX4b, tY4, tX3a, tX3b, tY3a, tY3b, tXz, tY2a, tY2b, tX1a, tX1b, tYo, tY0a, tY0b;
    half8 xF[3][2], x4[2][2], y4[2], x3[2][2], y3[2][2], xz[2], y2[2][2], x1[2][2], yo[2], y0[2][2];
    half8 wB[KH + 2 * OB], wC[OB * KH], wD[KH + OB], wE[OB * KH], wF[KH + OB], wG[IB0 * KH];
    static_assert(G::NF == G::T4 && G::T4 + OB == G::T3 && G::T2 + KH == G::T1 && G::T1 + OB == G::T0, "fragment table order");
    static_assert(OB == 2 && KH == 4, "hidden 64");
    half8 shf;
    half8 fk2[KS0];   // the features again (L2-hot), for the layer-0 weight gradient at the end of the tile
    auto df_store = [&](int ib) {     // one 32-feature block of dF: fp16, plane-major [3][M][C] (see the shared-stage kernel)
      const f32x16& df = tDF[ib];
      u2 pc[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        half4 v;
        v[0] = (_Float16)df[4 * q]; v[1] = (_Float16)df[4 * q + 1];
        v[2] = (_Float16)df[4 * q + 2]; v[3] = (_Float16)df[4 * q + 3];
        pc[q] = __builtin_bit_cast(u2, v);
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {     // lower lane: pc[k + 2] <- partner's pc[k]; upper lane: pc[k] <- partner's pc[k + 2]
#pragma unroll
        for (int d = 0; d < 2; d++) {
          const auto sw = __builtin_amdgcn_permlane32_swap(pc[k][d], pc[k + 2][d], false, false);
          pc[k][d] = sw[0];
          pc[k + 2][d] = sw[1];
        }
      }
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int f0 = 32 * ib + 16 * h + 8 * k;
        if (f0 < G::F && valid) {
          const u4 v = {pc[k][0], pc[k][1], pc[k + 2][0], pc[k + 2][1]};
          const int pl = f0 / C, fc = f0 - pl * C;   // 8 consecutive features never straddle planes (C % 8 == 0)
          *reinterpret_cast<u4*>(dfeat + ((size_t)pl * Mcap + i) * C + fc) = v;
        }
      }
    };
    auto xf_turn = [&](int ib) {      // block ib of the saved features turned (natural slot order)
      constexpr int KL = KS0 - 1;
      if (2 * ib + 1 < KS0) { TURN2(tXF[ib], fk2[2 * ib], fk2[2 * ib + 1 < KL ? 2 * ib + 1 : KL], I.ne, I.no); }
      else { TURN1(tXF[ib], fk2[2 * ib], I.ne); }
    };
    // ---- forward (recompute from the saved fp16 features)
    SLOT(tA0 = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KS0; ks++) tA0 = MFMA32(wA[ks], in.fk[ks], tA0),
         ldw(wA + KS0, G::F0 + KS0, KS0); ldw(wB, G::F1, KH + 2 * OB))
    SLOT(tA1 = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KS0; ks++) tA1 = MFMA32(wA[KS0 + ks], in.fk[ks], tA1),
         POST_RELU(tA0, h1[0], h1[1]); shf = sh_frag(dx, dy, dz, h))
    SLOT(, POST_RELU(tA1, h1[2], h1[3]); ldw(wC, G::F3, OB * KH))
    SLOT(tO = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tO = MFMA32(wB[ks], h1[ks], tO), )
    SLOT(, geo = acc_to_frag<false>(tO, 0); logit0 = tO[0])
    SLOT(tL2a = MFMA32(wB[KH], shf, zero16()); tL2a = MFMA32(wB[KH + 1], geo, tL2a);
         tL2b = MFMA32(wB[KH + 2], shf, zero16()); tL2b = MFMA32(wB[KH + 3], geo, tL2b),
         ldw(wD, G::F4, KH + OB))
    SLOT(, POST_RELU(tL2a, h3[0], h3[1]); POST_RELU(tL2b, h3[2], h3[3]))
    SLOT(tL3a = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tL3a = MFMA32(wC[ks], h3[ks], tL3a), )
    SLOT(tL3b = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tL3b = MFMA32(wC[KH + ks], h3[ks], tL3b),
         POST_RELU(tL3a, h4[0], h4[1]); ldw(wE, G::T3, OB * KH))
    SLOT(, POST_RELU(tL3b, h4[2], h4[3]))
    SLOT(tOut = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tOut = MFMA32(wD[ks], h4[ks], tOut), )
    // ---- layer 4: dZ4 = drgb * rgb * (1 - rgb) on rows 0..2 (lanes h == 0)
    SLOT(, f32x16 dz4 = zero16();
           if (h == 0) {
             // fast exp / reciprocal (v_exp_f32, v_rcp_f32: ~1e-7 relative): the products below are rounded to fp16 anyway
             const float c0 = __frcp_rn(1.f + __expf(-tOut[0])), c1 = __frcp_rn(1.f + __expf(-tOut[1])), c2 = __frcp_rn(1.f + __expf(-tOut[2]));
             dz4[0] = in.g_c0 * c0 * (1.f - c0);
             dz4[1] = in.g_c1 * c1 * (1.f - c1);
             dz4[2] = in.g_c2 * c2 * (1.f - c2);
           }
           dz4f = acc_to_frag<false>(dz4, 0))
    SLOT(tD4a = MFMA32(wD[KH], dz4f, zero16()); tD4b = MFMA32(wD[KH + 1], dz4f, zero16()), ldw(wF, G::T2, KH + OB))
    SLOT(TURN1(tY4, dz4f, I.po), POST_MASK(tD4a, d4f[0], d4f[1], h4[0], h4[1]); POST_MASK(tD4b, d4f[2], d4f[3], h4[2], h4[3]))
    SLOT(TURN2(tX4a, h4[0], h4[1], I.pe, I.po), POST_T(tY4, y4))
    SLOT(TURN2(tX4b, h4[2], h4[3], I.pe, I.po), POST_T(tX4a, x4[0]))
    // ---- layer 3 backward
    SLOT(tD3a = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tD3a = MFMA32(wE[ks], d4f[ks], tD3a),
         POST_T(tX4b, x4[1]))
    SLOT(tD3b = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tD3b = MFMA32(wE[KH + ks], d4f[ks], tD3b),
         POST_MASK(tD3a, d3f[0], d3f[1], h3[0], h3[1]))
    SLOT(dw_rows(y4, x4[0], dw[B::B4]); dw_rows(y4, x4[1], dw[B::B4 + 1]), POST_MASK(tD3b, d3f[2], d3f[3], h3[2], h3[3]))
    SLOT(TURN2(tY3a, d4f[0], d4f[1], I.pe, I.po), )
    // ---- layer 2 backward.  X = [SH(16), natural order | the 16 chain slots of the sigma net's outputs] (slot 0 = the
    // logit, no input of the colour net: red_tile<2> drops that column and shifts the geo features back by one)
    SLOT(tDzz = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tDzz = MFMA32(wF[ks], d3f[ks], tDzz),
         POST_T(tY3a, y3[0]))
    SLOT(TURN2(tY3b, d4f[2], d4f[3], I.pe, I.po),
         // dO fragment: slots rho = 0..14 <- d geo (rows 16..30 of dz = regs 8..15); slot rho = 15 <- d logit
         // trunc_exp backward (activation.py:14-17): g * exp(clamp(logit, -15, 15))
         dof = acc_to_frag<false>(tDzz, 1);
         {
           const float logit = __shfl(logit0, r);  // row 0 lives in lanes h == 0
           const float dlogit = in.g_s * __expf(fminf(fmaxf(logit, -15.f), 15.f));
           if (h == 1) dof[7] = (_Float16)dlogit;
         })
    SLOT(TURN2(tX3a, h3[0], h3[1], I.pe, I.po), POST_T(tY3b, y3[1]))
    // ---- layer 1 backward: dY = dO in chain-slot order (slot 15 = the logit's gradient; red_tile<1> maps the rows back)
    SLOT(tD1a = MFMA32(wF[KH], dof, zero16()); tD1b = MFMA32(wF[KH + 1], dof, zero16()), POST_T(tX3a, x3[0]))
    SLOT(TURN2(tX3b, h3[2], h3[3], I.pe, I.po), POST_MASK(tD1a, d1f[0], d1f[1], h1[0], h1[1]); POST_MASK(tD1b, d1f[2], d1f[3], h1[2], h1[3]); ldw(wG, G::T0, KH))
    SLOT(TURN2(tXz, shf, geo, I.ne, I.po), POST_T(tX3b, x3[1]))
    // ---- layer 0 backward: feature gradient dF^T = W0^T dH1^T, weight gradients
    SLOT(tDF[0] = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tDF[0] = MFMA32(wG[ks], d1f[ks], tDF[0]),
         POST_T(tXz, xz); ldw(wG + KH, G::T0 + KH, KH))
    SLOT(dw_rows(y3[0], x3[0], dw[B::B3]); dw_rows(y3[0], x3[1], dw[B::B3 + 1]);
         dw_rows(y3[1], x3[0], dw[B::B3 + 2]); dw_rows(y3[1], x3[1], dw[B::B3 + 3]),
         df_store(0))
    SLOT(tDF[1] = zero16();
         _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tDF[1] = MFMA32(wG[KH + ks], d1f[ks], tDF[1]),
         if (IB0 > 2) ldw(wG + 2 * KH, G::T0 + 2 * KH, KH))
    SLOT(TURN2(tY2a, d3f[0], d3f[1], I.pe, I.po), df_store(1))
    SLOT(if (IB0 > 2) { tDF[2] = zero16();
           _Pragma("unroll") for (int ks = 0; ks < KH; ks++) tDF[2] = MFMA32(wG[2 * KH + ks], d1f[ks], tDF[2]); },
         POST_T(tY2a, y2[0]))
    SLOT(TURN2(tY2b, d3f[2], d3f[3], I.pe, I.po), if (IB0 > 2) df_store(2))
    SLOT(TURN2(tX1a, h1[0], h1[1], I.pe, I.po), POST_T(tY2b, y2[1]))
    SLOT(dw_rows(y2[0], xz, dw[B::B2]); dw_rows(y2[1], xz, dw[B::B2 + 1]), POST_T(tX1a, x1[0]))
    SLOT(TURN2(tX1b, h1[2], h1[3], I.pe, I.po), )
    SLOT(TURN1(tYo, dof, I.pe), POST_T(tX1b, x1[1]))
    SLOT(TURN2(tY0a, d1f[0], d1f[1], I.pe, I.po), POST_T(tYo, yo); read_feats(fk2, cur, valid))
    SLOT(TURN2(tY0b, d1f[2], d1f[3], I.pe, I.po), POST_T(tY0a, y0[0]))
    SLOT(dw_rows(yo, x1[0], dw[B::B1]); dw_rows(yo, x1[1], dw[B::B1 + 1]), POST_T(tY0b, y0[1]))
    SLOT(xf_turn(0), )
    SLOT(xf_turn(1), POST_T(tXF[0], xF[0]))
    SLOT(if (IB0 > 2) xf_turn(2), POST_T(tXF[1], xF[1]))
    SLOT(dw_rows(y0[0], xF[0], dw[B::B0]); dw_rows(y0[1], xF[0], dw[B::B0 + IB0]), if (IB0 > 2) { POST_T(tXF[2], xF[2]); })
    SLOT(dw_rows(y0[0], xF[1], dw[B::B0 + 1]); dw_rows(y0[1], xF[1], dw[B::B0 + IB0 + 1]), ldw(wA, G::F0, KS0))
    SLOT(if (IB0 > 2) { dw_rows(y0[0], xF[2], dw[B::B0 + 2]); dw_rows(y0[1], xF[2], dw[B::B0 + IB0 + 2]); }, )
    cur ^= 1;
#if TNL_ROWS_STAMP
    if (stamp_on) { g_rows_stamps[63 * 64 + stamp_t] = __builtin_amdgcn_s_memrealtime(); stamp_t++; }
#endif
  }
  // ---- epilogue: the four waves' tiles summed through LDS (over the fragment table, no longer needed) into the slab
  float* red = reinterpret_cast<float*>(smem);
  float* slab = slabs + (size_t)blockIdx.x * G::NW;
  auto all_tiles = [&](auto op) {
    constexpr int OP = decltype(op)::value;
#pragma unroll
    for (int T = 0; T < B::NTILES; T++) {
      if (T < B::B2) red_tile<0, OP>(red, slab, G::OFF3, H, H, (T - B::B3) / G::OB, (T - B::B3) % G::OB, dw[T], r, h);
      else if (T < B::B1) red_tile<2, OP>(red, slab, G::OFF2, H, 31, T - B::B2, 0, dw[T], r, h);
      else if (T < B::B0) red_tile<1, OP>(red, slab, G::OFF1, 16, H, 0, T - B::B1, dw[T], r, h, G::OFF4);
      else red_tile<0, OP>(red, slab, G::OFF0, H, G::F, (T - B::B0) / G::IB0, (T - B::B0) % G::IB0, dw[T], r, h);
    }
  };
  __syncthreads();            // every wave is done with the fragment table
  if (wv == 0) all_tiles(std::integral_constant<int, 0>{});
  __syncthreads();
  if (wv == 1) all_tiles(std::integral_constant<int, 1>{});
  __syncthreads();
  if (wv == 2) all_tiles(std::integral_constant<int, 1>{});
  __syncthreads();
  if (wv == 3) all_tiles(std::integral_constant<int, 2>{});
}

template <int C>
int launch_rows(const float* gsig, const float* grgb, const void* feats, const float* dirs, uint32_t M, const void* packed,
                void* workspace, const int32_t* m_actual, void* dfeat, hipStream_t st, uint32_t* nslab) {
  using B = RowsGeom<C>;
  const uint32_t nst = (M + B::ST - 1) / B::ST;
  const uint32_t blocks = nst < 256 ? nst : 256;
  static bool attr_set[64] = {};
  const hipError_t ea = ensure_dynamic_lds(&k_field_bwd_rows<C>, (int)B::LDS_BYTES, attr_set);
  if (ea != hipSuccess) return (int)ea;
  float* slabs = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL((k_field_bwd_rows<C>), dim3(blocks), dim3(256), B::LDS_BYTES, st, gsig, grgb,
                     reinterpret_cast<const _Float16*>(feats), dirs, M, reinterpret_cast<const half8*>(packed), slabs,
                     m_actual, reinterpret_cast<_Float16*>(dfeat));
  *nslab = blocks;
  return (int)hipGetLastError();
}


}  // namespace

int tnl_bwd_rows_launch(int C, const float* gsig, const float* grgb, const void* feats, const float* dirs, uint32_t M,
                        const void* packed, void* workspace, const int32_t* m_actual, void* dfeat, hipStream_t st,
                        uint32_t* nslab) {
  if (C == 16) return launch_rows<16>(gsig, grgb, feats, dirs, M, packed, workspace, m_actual, dfeat, st, nslab);
  if (C == 32) return launch_rows<32>(gsig, grgb, feats, dirs, M, packed, workspace, m_actual, dfeat, st, nslab);
  return (int)hipErrorInvalidValue;
}
