// field_common.h -- fragment geometry shared by the fused-field kernels (field.hip).
//
// MFMA used: v_mfma_f32_32x32x16_f16 (D[32x32] += A[32x16] * B[16x32], fp32 accumulate).
// gfx950 lane maps (cdna_hip_programming.md section 3), lane l: r = l & 31, h = l >> 5:
//   A fragment (8 halfs): A[row r][k = 8h + j]      B fragment (8 halfs): B[k = 8h + j][col r]
//   C/D (16 floats):      D[row (g&3) + 8(g>>2) + 4h][col r],  g = register 0..15
//
// "Chain layout": every activation lives transposed, X^T[feature][sample], samples on the lanes.
// A layer is Y^T = W * X^T with the weight as the A operand.  The 32x32 result tile already is the
// B operand of the next layer: registers 8s..8s+7 (s = 0,1), converted to fp16, form the k-step-s
// fragment whose k-slot (h, j) carries row  16s + 8(j>>2) + 4h + (j&3)  of the tile.  The weights are
// pre-packed in that permuted k order (tnl_field_pack), so no activation ever leaves the registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The binned backward of the hidden-128 network runs as two launches split by layer (field_bwd.hip PART); the forward
// then saves the 16 sigma-net outputs per sample behind the features for the colour half.
template <int H>
constexpr bool split_backward() { return H > 64; }

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// feature carried by k-slot (h, j) of the k-step `ks` fragment built from accumulator tiles
__host__ __device__ constexpr int kslot_feature(int ks, int h, int j) {
  return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
}
// row of a 32x32 accumulator tile held in register g by lane-half h
__host__ __device__ constexpr int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// Fragment table for a (C channels/plane, H hidden) network.  Every fragment is 64 lanes x 8 halfs.
template <int C, int H>
struct FieldGeom {
  static constexpr int F = 3 * C;           // input features
  static constexpr int KS0 = F / 16;        // k-steps of layer 0
  static constexpr int OB = H / 32;         // 32-row blocks of a hidden layer
  static constexpr int KH = H / 16;         // k-steps over a hidden layer
  static constexpr int IB0 = (F + 31) / 32; // 32-row blocks of the feature gradient
  // forward fragments (A operand = W_l)
  static constexpr int F0 = 0;              // [ob][ks]  W0[32ob+r][16ks+8h+j]
  static constexpr int F1 = F0 + OB * KS0;  // [ks]      W1[r][kslot(ks)]            (rows >= 16 zero)
  static constexpr int F2 = F1 + KH;        // [ob][2]   ks0: W2[.][8h+j] (SH) ; ks1: W2[.][16+rho-1], rho=0 -> 0
  static constexpr int F3 = F2 + 2 * OB;    // [ob][ks]  W3[32ob+r][kslot(ks)]
  static constexpr int F4 = F3 + OB * KH;   // [ks]      W4[r][kslot(ks)]            (rows >= 3 zero)
  static constexpr int NF = F4 + KH;
  // transposed fragments (A operand = W_l^T) for the backward data path
  static constexpr int T4 = NF;             // [ib]      W4[rho][32ib+r], rho < 3
  static constexpr int T3 = T4 + OB;        // [ib][ks]  W3[kslot(ks)][32ib+r]
  static constexpr int T2 = T3 + OB * KH;   // [ks]      W2[kslot(ks)][r]            (r = 31 zero)
  static constexpr int T1 = T2 + KH;        // [ib]      W1[rho==15 ? 0 : rho+1][32ib+r]
  static constexpr int T0 = T1 + OB;        // [ib][ks]  W0[kslot(ks)][32ib+r]       (32ib+r >= F zero)
  static constexpr int NTOT = T0 + IB0 * KH;
  // offsets of W0..W4 inside the concatenated fp32 weight / gradient vector (nn.Linear layout [out][in])
  static constexpr int OFF0 = 0;
  static constexpr int OFF1 = OFF0 + H * F;
  static constexpr int OFF2 = OFF1 + 16 * H;
  static constexpr int OFF3 = OFF2 + H * 31;
  static constexpr int OFF4 = OFF3 + H * H;
  static constexpr int NW = OFF4 + 3 * H;
};

// Saved-feature layout shared by k_field_fwd (writer) and k_field_bwd (reader): blocked by the 32-sample tile a
// wavefront owns, [tile][k-step][sample in tile][16 halfs], so that ONE store instruction of the forward (all lanes,
// one k-step) covers 1 KB of contiguous memory -- eight whole 128-byte lines.  With the row-major [M][F] layout the
// six k-steps of a row were six 32-byte pieces per sample, each instruction touching 32 lines partially; every store
// instruction's bytes leave L2 as that instruction's own fabric writes, so the shape of one instruction is what counts
// (field forward 0.83 -> 0.755 ms).  The buffer holds ceil(M / 32) * 32 rows.
template <int KS0>
__device__ __forceinline__ size_t feat_slot(uint32_t i, int ks, int h) {
  return ((((size_t)(i >> 5) * KS0 + ks) * 32 + (i & 31)) * 16) + 8 * h;
}


// Source element of packed half `idx` (= (frag*64 + lane)*8 + j): returns the index into the
// concatenated fp32 weights, or -1 for a structural zero.
template <int C, int H>
__host__ __device__ inline int field_pack_source(int idx) {
  using G = FieldGeom<C, H>;
  const int j = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
  const int r = lane & 31, h = lane >> 5;
  const int rho = kslot_feature(0, h, j);
  if (f < G::F1) {  // layer 0
    const int ob = f / G::KS0, ks = f % G::KS0;
    return G::OFF0 + (32 * ob + r) * G::F + 16 * ks + 8 * h + j;
  } else if (f < G::F2) {
    const int ks = f - G::F1;
    return r < 16 ? G::OFF1 + r * H + kslot_feature(ks, h, j) : -1;
  } else if (f < G::F3) {
    const int ob = (f - G::F2) / 2, ks = (f - G::F2) % 2;
    if (ks == 0) return G::OFF2 + (32 * ob + r) * 31 + 8 * h + j;
    return rho == 0 ? -1 : G::OFF2 + (32 * ob + r) * 31 + 16 + rho - 1;
  } else if (f < G::F4) {
    const int ob = (f - G::F3) / G::KH, ks = (f - G::F3) % G::KH;
    return G::OFF3 + (32 * ob + r) * H + kslot_feature(ks, h, j);
  } else if (f < G::NF) {
    const int ks = f - G::F4;
    return r < 3 ? G::OFF4 + r * H + kslot_feature(ks, h, j) : -1;
  } else if (f < G::T3) {
    const int ib = f - G::T4;
    return rho < 3 ? G::OFF4 + rho * H + 32 * ib + r : -1;
  } else if (f < G::T2) {
    const int ib = (f - G::T3) / G::KH, ks = (f - G::T3) % G::KH;
    return G::OFF3 + kslot_feature(ks, h, j) * H + 32 * ib + r;
  } else if (f < G::T1) {
    const int ks = f - G::T2;
    return r < 31 ? G::OFF2 + kslot_feature(ks, h, j) * 31 + r : -1;
  } else if (f < G::T0) {
    const int ib = f - G::T1;
    const int orow = rho == 15 ? 0 : rho + 1;
    return G::OFF1 + orow * H + 32 * ib + r;
  } else {
    const int ib = (f - G::T0) / G::KH, ks = (f - G::T0) % G::KH;
    const int in = 32 * ib + r;
    return in < G::F ? G::OFF0 + kslot_feature(ks, h, j) * G::F + in : -1;
  }
}
