// wavelet.hip -- multiscale inverse DWT of the triplane coefficients and its adjoint (gfx950).
//
// Replaces the pytorch_wavelets.DWTInverse + F.pad + 2* chain of
// reconstruction/triplaneencoder/triplane_encoder.py:364-405 (three grouped conv_transpose2d pairs
// per level, each materialising a full intermediate) with ONE LDS-tiled kernel per level:
//   * a workgroup owns a 64x64 output tile of one (plane, channel) slice; the four coefficient
//     bands of the 32x32 input tile (+ halo L/4) are staged in LDS once,
//   * the separable synthesis runs column pass -> LDS -> row pass entirely on chip, register-blocked
//     (each thread produces 8 outputs from a 12-sample window: ~6 FMA per LDS read),
//   * polyphase form: even outputs use even taps, odd outputs odd taps; the taps are compile-time
//     constants per wavelet so zero taps (bior6.8 rec_lo has 7 of 18) cost nothing,
//   * the zero-padding of mode='zero' and of F.pad is the bounds check of the tile load.
// The adjoint (autograd of SFB2D: analysis with the same rec_* taps, then crop and 2*) mirrors it:
// an 80x80 fine tile -> row pass -> column pass -> four 32x32 coarse bands.
//
// Closed form (verified against pywt.idwt2(mode='zero')): K = (L-2)/2,
//   out[o] = sum_j lo[j]*g0[o-2j+K] + hi[j]*g1[o-2j+K].
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef TNL_MAIN_PRIO
#define TNL_MAIN_PRIO 0   // A/B builds: static wave priority (s_setprio) of the step's kernels that run beside the side chain
#endif
#ifndef TNL_PRIO_FWD
#define TNL_PRIO_FWD TNL_MAIN_PRIO
#endif
#ifndef TNL_PRIO_BWD
#define TNL_PRIO_BWD TNL_MAIN_PRIO
#endif
#define TNL_SET_PRIO(P) do { if (P) __builtin_amdgcn_s_setprio(P); } while (0)

#include "../../include/trinerflet_hip.h"
#include "adam_common.h"
#include "roi_common.h"

// The adjoint's coefficient-gradient stores are non-temporal (read once, by Adam, after 0.6 GB more have been
// written): adjoint 1.154 -> 1.128 ms per step in an A/B on one box.  Non-temporal LOADS of the coefficients in the
// forward kernels were slower (0.66 -> 0.85 ms) and are not used.
#ifndef TNL_IDWT_BWD_NT
#define TNL_IDWT_BWD_NT 1
#endif
__device__ __forceinline__ void stg(float* p, float v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, p);
  else *p = v;
}

namespace {

struct WTaps {
  int L;
  float g0[18];
  float g1[18];
};

// pywt.Wavelet(name).rec_lo / rec_hi rounded to float32 (pytorch_wavelets keeps float32 buffers)
__host__ __device__ constexpr WTaps wtaps(int w) {
  switch (w) {
    case 0:
      return WTaps{2, {0.7071067811865476f, 0.7071067811865476f}, {0.7071067811865476f, -0.7071067811865476f}};
    case 1:
      return WTaps{6,
                   {0.f, 0.3535533905932738f, 0.7071067811865476f, 0.3535533905932738f, 0.f, 0.f},
                   {0.f, 0.1767766952966369f, 0.3535533905932738f, -1.0606601717798212f, 0.3535533905932738f,
                    0.1767766952966369f}};
    case 2:
      return WTaps{10,
                   {0.f, -0.06453888262869706f, -0.04068941760916406f, 0.41809227322161724f, 0.7884856164055829f,
                    0.41809227322161724f, -0.04068941760916406f, -0.06453888262869706f, 0.f, 0.f},
                   {0.f, -0.03782845550726404f, -0.023849465019556843f, 0.11062440441843718f, 0.37740285561283066f,
                    -0.8526986790088938f, 0.37740285561283066f, 0.11062440441843718f, -0.023849465019556843f,
                    -0.03782845550726404f}};
    case 3:
      return WTaps{14,
                   {0.f, 0.f, 0.f, 0.f, 0.f, 0.3535533905932738f, 0.7071067811865476f, 0.3535533905932738f, 0.f, 0.f,
                    0.f, 0.f, 0.f, 0.f},
                   {0.f, 0.006905339660024878f, 0.013810679320049757f, -0.04695630968816917f, -0.1077232986963881f,
                    0.16987135563661201f, 0.4474660099696121f, -0.966747552403483f, 0.4474660099696121f,
                    0.16987135563661201f, -0.1077232986963881f, -0.04695630968816917f, 0.013810679320049757f,
                    0.006905339660024878f}};
    default:
      return WTaps{18,
                   {0.f, 0.f, 0.f, 0.014426282505624435f, 0.014467504896790148f, -0.07872200106262882f,
                    -0.04036797903033992f, 0.41784910915027457f, 0.7589077294536541f, 0.41784910915027457f,
                    -0.04036797903033992f, -0.07872200106262882f, 0.014467504896790148f, 0.014426282505624435f, 0.f,
                    0.f, 0.f, 0.f},
                   {0.f, -0.0019088317364812906f, -0.0019142861290887667f, 0.016990639867602342f,
                    0.01193456527972926f, -0.04973290349094079f, -0.07726317316720414f, 0.09405920349573646f,
                    0.4207962846098268f, -0.8259229974584023f, 0.4207962846098268f, 0.09405920349573646f,
                    -0.07726317316720414f, -0.04973290349094079f, 0.01193456527972926f, 0.016990639867602342f,
                    -0.0019142861290887667f, -0.0019088317364812906f}};
  }
}

constexpr int TI = 32;   // coarse tile edge
constexpr int RM = 4;    // coarse samples per thread-run
constexpr int NT = 256;  // threads per workgroup
constexpr int TX = 64;   // texels per layout-change tile

// ---------------------------------------------------------------------------------------------
// forward: x:[S][n][n], yh:[S][3][n][n] -> out:[S][2n][2n]
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(NT)
k_idwt_fwd(const float* __restrict__ x, const float* __restrict__ yh, int n, float* __restrict__ out) {
  constexpr WTaps T = wtaps(W);
  constexpr int L = T.L, K = (L - 2) / 2, HW = L / 4;
  constexpr int TIH = TI + 2 * HW;  // staged input edge
  constexpr int LS = TIH + 1;       // LDS row stride (odd: conflict-free column walks)
  constexpr int WIN = RM + 2 * HW;
  __shared__ float band[4][TIH][LS];
  __shared__ float mid[2][2 * TI][LS];

  const int s = blockIdx.z;
  const int a_r = blockIdx.y * TI, a_c = blockIdx.x * TI;  // coarse tile origin
  const size_t nn = (size_t)n * n;
  const float* ll = x + (size_t)s * nn;
  const float* hb = yh + (size_t)s * 3 * nn;

  // stage the four bands (zero outside [0,n): the zero-mode padding); ll carries the 2* of :379
  for (int idx = threadIdx.x; idx < 4 * TIH * TIH; idx += NT) {
    const int b = idx / (TIH * TIH), rem = idx - b * TIH * TIH;
    const int r = rem / TIH, c = rem - r * TIH;
    const int gr = a_r - HW + r, gc = a_c - HW + c;
    float v = 0.f;
    if (gr >= 0 && gr < n && gc >= 0 && gc < n) {
      const size_t off = (size_t)gr * n + gc;
      v = b == 0 ? 2.0f * ll[off] : hb[(size_t)(b - 1) * nn + off];
    }
    band[b][r][c] = v;
  }
  __syncthreads();

  // column pass (synthesis along H): lo = syn(ll, lh), hi = syn(hl, hh)
  for (int u = threadIdx.x; u < TIH * (TI / RM); u += NT) {
    const int c = u % TIH, m0 = (u / TIH) * RM;
    float w0[WIN], w1[WIN], w2[WIN], w3[WIN];
#pragma unroll
    for (int i = 0; i < WIN; i++) {
      w0[i] = band[0][m0 + i][c]; w1[i] = band[1][m0 + i][c];
      w2[i] = band[2][m0 + i][c]; w3[i] = band[3][m0 + i][c];
    }
#pragma unroll
    for (int m = 0; m < RM; m++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int d = -HW; d <= HW; d++) {
          constexpr int dummy = 0; (void)dummy;
          const int k = e + K - 2 * d;
          if (k >= 0 && k < L) {
            const float t0 = T.g0[k], t1 = T.g1[k];
            if (t0 != 0.f) { lo = fmaf(w0[m + d + HW], t0, lo); hi = fmaf(w2[m + d + HW], t0, hi); }
            if (t1 != 0.f) { lo = fmaf(w1[m + d + HW], t1, lo); hi = fmaf(w3[m + d + HW], t1, hi); }
          }
        }
        mid[0][2 * (m0 + m) + e][c] = lo;
        mid[1][2 * (m0 + m) + e][c] = hi;
      }
    }
  }
  __syncthreads();

  // row pass (synthesis along W) + store; a thread owns 2*RM consecutive outputs of one row
  const int m2 = 2 * n;
  float* dst = out + (size_t)s * m2 * m2;
  for (int u = threadIdx.x; u < 2 * TI * (TI / RM); u += NT) {
    const int run = u % (TI / RM), r = u / (TI / RM);
    const int m0 = run * RM;
    float wl[WIN], wh[WIN];
#pragma unroll
    for (int i = 0; i < WIN; i++) { wl[i] = mid[0][r][m0 + i]; wh[i] = mid[1][r][m0 + i]; }
    float o[2 * RM];
#pragma unroll
    for (int m = 0; m < RM; m++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        float acc = 0.f;
#pragma unroll
        for (int d = -HW; d <= HW; d++) {
          const int k = e + K - 2 * d;
          if (k >= 0 && k < L) {
            const float t0 = T.g0[k], t1 = T.g1[k];
            if (t0 != 0.f) acc = fmaf(wl[m + d + HW], t0, acc);
            if (t1 != 0.f) acc = fmaf(wh[m + d + HW], t1, acc);
          }
        }
        o[2 * m + e] = acc;
      }
    }
    const int gr = 2 * a_r + r, gc = 2 * (a_c + m0);
    if (gr < m2) {
      float* p = dst + (size_t)gr * m2 + gc;
      if (gc + 2 * RM <= m2 && (m2 & 3) == 0) {
        reinterpret_cast<float4*>(p)[0] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4*>(p)[1] = make_float4(o[4], o[5], o[6], o[7]);
      } else {
#pragma unroll
        for (int i = 0; i < 2 * RM; i++)
          if (gc + i < m2) p[i] = o[i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// adjoint: dout:[S][2n][2n] -> dx:[S][n][n] (includes the 2*), dyh:[S][3][n][n]
//   d_lo[j] = sum_k dout[2j-K+k]*g0[k],  d_hi[j] = sum_k dout[2j-K+k]*g1[k]
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(NT)
k_idwt_bwd(const float* __restrict__ dout, int n, float* __restrict__ dx, float* __restrict__ dyh) {
  constexpr WTaps T = wtaps(W);
  constexpr int L = T.L, K = (L - 2) / 2;
  constexpr int FT = 2 * TI + L - 2;  // fine tile edge
  constexpr int FS = FT + 1;
  constexpr int LS = TI + 1;
  constexpr int WIN = 2 * RM + L - 2;
  __shared__ float fine[FT][FS];
  __shared__ float mid[2][FT][LS];

  const int s = blockIdx.z;
  const int a_r = blockIdx.y * TI, a_c = blockIdx.x * TI;
  const int m2 = 2 * n;
  const float* src = dout + (size_t)s * m2 * m2;
  for (int idx = threadIdx.x; idx < FT * FT; idx += NT) {
    const int r = idx / FT, c = idx - r * FT;
    const int gr = 2 * a_r - K + r, gc = 2 * a_c - K + c;
    fine[r][c] = (gr >= 0 && gr < m2 && gc >= 0 && gc < m2) ? src[(size_t)gr * m2 + gc] : 0.f;
  }
  __syncthreads();

  // row pass (analysis along W)
  for (int u = threadIdx.x; u < FT * (TI / RM); u += NT) {
    const int r = u % FT, j0 = (u / FT) * RM;
    float w[WIN];
#pragma unroll
    for (int i = 0; i < WIN; i++) w[i] = fine[r][2 * j0 + i];
#pragma unroll
    for (int j = 0; j < RM; j++) {
      float lo = 0.f, hi = 0.f;
#pragma unroll
      for (int k = 0; k < L; k++) {
        const float t0 = T.g0[k], t1 = T.g1[k];
        if (t0 != 0.f) lo = fmaf(w[2 * j + k], t0, lo);
        if (t1 != 0.f) hi = fmaf(w[2 * j + k], t1, hi);
      }
      mid[0][r][j0 + j] = lo;
      mid[1][r][j0 + j] = hi;
    }
  }
  __syncthreads();

  // column pass (analysis along H) + store of the four coarse bands
  const size_t nn = (size_t)n * n;
  float* o_ll = dx + (size_t)s * nn;
  float* o_h = dyh + (size_t)s * 3 * nn;
  for (int u = threadIdx.x; u < TI * (TI / RM); u += NT) {
    const int c = u % TI, j0 = (u / TI) * RM;
    float wl[WIN], wh[WIN];
#pragma unroll
    for (int i = 0; i < WIN; i++) { wl[i] = mid[0][2 * j0 + i][c]; wh[i] = mid[1][2 * j0 + i][c]; }
    const int gc = a_c + c;
#pragma unroll
    for (int j = 0; j < RM; j++) {
      float a = 0.f, b = 0.f, cc = 0.f, d = 0.f;
#pragma unroll
      for (int k = 0; k < L; k++) {
        const float t0 = T.g0[k], t1 = T.g1[k];
        if (t0 != 0.f) { a = fmaf(wl[2 * j + k], t0, a); cc = fmaf(wh[2 * j + k], t0, cc); }
        if (t1 != 0.f) { b = fmaf(wl[2 * j + k], t1, b); d = fmaf(wh[2 * j + k], t1, d); }
      }
      const int gr = a_r + j0 + j;
      if (gr < n && gc < n) {
        const size_t off = (size_t)gr * n + gc;
        o_ll[off] = 2.0f * a;
        o_h[off] = b;
        o_h[nn + off] = cc;
        o_h[2 * nn + off] = d;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Pipelined variants (n % 4 == 0): a workgroup walks TPW consecutive tiles along x.  The four bands of
// tile t+1 are fetched with 16-byte loads into registers while tile t is being computed (the loads stay
// in flight across the two barriers of the tile), so HBM latency is hidden by the tile's own ~1500 VALU
// instructions instead of by occupancy.  HALF_OUT writes the level as fp16 (used for the finest level:
// the sampler's planes are fp16, so the fp32 copy is never materialised).
// ---------------------------------------------------------------------------------------------
constexpr int TPW_MAX = 8;  // tiles per workgroup walk (fewer on small levels, where the serial walk of one workgroup,
                            // not bandwidth, sets the kernel's time: 3 x 75 us of the base step's adjoint were levels <= 256)


template <int W, bool HALF_OUT>
__global__ void __launch_bounds__(NT)
k_idwt_fwd_pipe(const float* __restrict__ x, const float* __restrict__ yh, int n, void* __restrict__ out, Roi roi,
                int TPW) {
  TNL_SET_PRIO(TNL_PRIO_FWD);
  constexpr WTaps T = wtaps(W);
  constexpr int L = T.L, K = (L - 2) / 2, HW = L / 4;
  constexpr int HWA = HW ? 4 : 0;            // staged halo, 16-byte aligned
  constexpr int SH = HWA - HW;               // shift between staged and used halo
  constexpr int TIH = TI + 2 * HWA;
  constexpr int LSB = TIH + 4;               // band row stride (floats): 16-B aligned rows for b128 writes
  constexpr int LSM = TIH + 1;               // mid row stride: odd, conflict-free row walks
  constexpr int NQ = 4 * TIH * (TIH / 4);    // float4 words per tile
  constexpr int KQ = (NQ + NT - 1) / NT;
  constexpr int WIN = RM + 2 * HW;
  __shared__ __attribute__((aligned(16))) float band[4][TIH][LSB];
  __shared__ float mid[2][2 * TI][LSM];

  const int s = blockIdx.z;
  const int m2 = 2 * n;
  const int pl = roi.rw ? (s + roi.s0) / roi.spp : 0;
  const int fox = roi.rw ? roi.ox[pl] : 0, foy = roi.rw ? roi.oy[pl] : 0;   // fine-grid origin of the tile walk
  const bool compact = roi.rw && !roi.strided;   // strided: the window of a full-size output array is written
  const int orow = compact ? roi.rw : m2;                                     // output row stride
  const size_t oplane = compact ? (size_t)roi.rh * roi.rw : (size_t)m2 * m2;  // output slice stride
  const int sox = compact ? fox : 0, soy = compact ? foy : 0;                 // array origin in fine coordinates
  const int a_r = foy / 2 + blockIdx.y * TI;
  const int ntx = roi.rw ? roi.rw / (2 * TI) : (n + TI - 1) / TI;
  const int tx0 = blockIdx.x * TPW, tx1 = min(tx0 + TPW, ntx);
  const size_t nn = (size_t)n * n;
  const float* ll = x + (size_t)s * nn;
  const float* hb = yh + (size_t)s * 3 * nn;

  // tile-independent part of the staging map
  int qb[KQ], qr[KQ], qc[KQ];
  bool qv[KQ];
#pragma unroll
  for (int k = 0; k < KQ; k++) {
    const int q = threadIdx.x + NT * k;
    qv[k] = q < NQ;
    const int b = q / (TIH * (TIH / 4)), rem = q - b * (TIH * (TIH / 4));
    qb[k] = b; qr[k] = rem / (TIH / 4); qc[k] = 4 * (rem - (rem / (TIH / 4)) * (TIH / 4));
  }
  float4 pre[KQ];
  auto prefetch = [&](int tx, float4* dst) {
    const int a_c = fox / 2 + tx * TI;
#pragma unroll
    for (int k = 0; k < KQ; k++) {
      const int gr = a_r - HWA + qr[k], gc = a_c - HWA + qc[k];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qv[k] && gr >= 0 && gr < n && gc >= 0 && gc < n) {
        const float* src = qb[k] == 0 ? ll : hb + (size_t)(qb[k] - 1) * nn;
        v = *reinterpret_cast<const float4*>(src + (size_t)gr * n + gc);
      }
      dst[k] = v;   // (the 2x of the LL band is applied when the value is consumed: touching it here would make
                    //  the prefetch wait for its own loads)
    }
  };
  prefetch(tx0, pre);
  for (int tx = tx0; tx < tx1; tx++) {
#pragma unroll
    for (int k = 0; k < KQ; k++)
      if (qv[k]) {
        float4 v = pre[k];
        if (qb[k] == 0) { v.x *= 2.f; v.y *= 2.f; v.z *= 2.f; v.w *= 2.f; }
        *reinterpret_cast<float4*>(&band[qb[k]][qr[k]][qc[k]]) = v;
      }
    __syncthreads();
    // (issuing these loads a whole tile earlier -- before this tile's data is staged -- changed nothing: 819 vs 826 us)
    if (tx + 1 < tx1) prefetch(tx + 1, pre);

    for (int u = threadIdx.x; u < TIH * (TI / RM); u += NT) {
      const int c = u % TIH, m0 = (u / TIH) * RM;
      float w0[WIN], w1[WIN], w2[WIN], w3[WIN];
#pragma unroll
      for (int i = 0; i < WIN; i++) {
        w0[i] = band[0][m0 + i + SH][c]; w1[i] = band[1][m0 + i + SH][c];
        w2[i] = band[2][m0 + i + SH][c]; w3[i] = band[3][m0 + i + SH][c];
      }
#pragma unroll
      for (int m = 0; m < RM; m++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          float lo = 0.f, hi = 0.f;
#pragma unroll
          for (int d = -HW; d <= HW; d++) {
            const int k = e + K - 2 * d;
            if (k >= 0 && k < L) {
              const float t0 = T.g0[k], t1 = T.g1[k];
              if (t0 != 0.f) { lo = fmaf(w0[m + d + HW], t0, lo); hi = fmaf(w2[m + d + HW], t0, hi); }
              if (t1 != 0.f) { lo = fmaf(w1[m + d + HW], t1, lo); hi = fmaf(w3[m + d + HW], t1, hi); }
            }
          }
          mid[0][2 * (m0 + m) + e][c] = lo;
          mid[1][2 * (m0 + m) + e][c] = hi;
        }
      }
    }
    __syncthreads();

    const int a_c = fox / 2 + tx * TI;
    // (unrolling these two trips like the adjoint's row pass changes nothing here: 269 us either way)
    for (int u = threadIdx.x; u < 2 * TI * (TI / RM); u += NT) {
      const int run = u % (TI / RM), r = u / (TI / RM);
      const int m0 = run * RM;
      float wl[WIN], wh[WIN];
#pragma unroll
      for (int i = 0; i < WIN; i++) { wl[i] = mid[0][r][m0 + i + SH]; wh[i] = mid[1][r][m0 + i + SH]; }
      float o[2 * RM];
#pragma unroll
      for (int m = 0; m < RM; m++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          float acc = 0.f;
#pragma unroll
          for (int d = -HW; d <= HW; d++) {
            const int k = e + K - 2 * d;
            if (k >= 0 && k < L) {
              const float t0 = T.g0[k], t1 = T.g1[k];
              if (t0 != 0.f) acc = fmaf(wl[m + d + HW], t0, acc);
              if (t1 != 0.f) acc = fmaf(wh[m + d + HW], t1, acc);
            }
          }
          o[2 * m + e] = acc;
        }
      }
      const int gr = 2 * a_r + r, gc = 2 * (a_c + m0);
      if (gr < m2 && gc < m2) {  // m2 % 8 == 0 and gc % 8 == 0: the 8 outputs are in range together
        const size_t off = (size_t)s * oplane + (size_t)(gr - soy) * orow + (gc - sox);
        if (HALF_OUT) {
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          h8 hv;
#pragma unroll
          for (int i = 0; i < 8; i++) hv[i] = (_Float16)o[i];
          *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(out) + off) = hv;
        } else {
          float* p = reinterpret_cast<float*>(out) + off;
          reinterpret_cast<float4*>(p)[0] = make_float4(o[0], o[1], o[2], o[3]);
          reinterpret_cast<float4*>(p)[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
    }
    // the next iteration's band writes are safe (every thread is past the column pass); its column pass is
    // separated from this row pass by the barrier that follows those writes
  }
}

// Optional fused optimiser epilogue: the detail-band gradients of a level are consumed where they are produced
// (Adam + wavelet-L1 on the level's coefficients, and on the LL plane at the coarsest level) instead of being
// written and re-read by a separate pass: 8 of 36 bytes per coefficient saved across adjoint + optimiser.
struct FuseAdam {
  float *p, *m, *v;       // [S][3][n][n] coefficients of this level and their moments
  float *pl, *ml, *vl;    // [S][n][n] LL parameter and moments (coarsest level only, else NULL)
  AdamArgs a;
  const float* inv_scale_dev;
  const float* found_inf;
  float* abs_sum;
};

template <int W, bool FUSE>
__global__ void __launch_bounds__(NT)
k_idwt_bwd_pipe(const float* __restrict__ dout, int n, float* __restrict__ dx, float* __restrict__ dyh, FuseAdam fa,
                Roi roi, Roi orect, int TPW) {
  TNL_SET_PRIO(TNL_PRIO_BWD);
  constexpr WTaps T = wtaps(W);
  constexpr int L = T.L, K = (L - 2) / 2;
  constexpr int KA = (K + 3) / 4 * 4;                 // aligned left halo of the fine tile
  constexpr int SH = KA - K;
  constexpr int FT = 2 * TI + L - 2;                  // fine rows
  constexpr int FTA = (2 * TI + L - 2 + SH + 3) / 4 * 4;  // fine cols staged (16-B aligned start and length)
  constexpr int FS = FTA + 1;
  constexpr int LS = TI + 1;
  constexpr int NQ = FT * (FTA / 4);
  constexpr int KQ = (NQ + NT - 1) / NT;
  constexpr int WIN = 2 * RM + L - 2;
  __shared__ float fine[FT][FS];
  __shared__ float mid[2][FT][LS];

  const int s = blockIdx.z;
  const int a_r = blockIdx.y * TI;
  const int ntx = (n + TI - 1) / TI;
  const int tx0 = blockIdx.x * TPW, tx1 = min(tx0 + TPW, ntx);
  const int m2 = 2 * n;
  const int pl = roi.rw ? (s + roi.s0) / roi.spp : 0;
  // fine-side input: compact ROI window [oy, oy+rh) x [ox, ox+rw) (zero outside), or the whole plane
  const int fox = roi.rw ? roi.ox[pl] : 0, foy = roi.rw ? roi.oy[pl] : 0;
  const int fw = roi.rw ? roi.rw : m2, fh = roi.rw ? roi.rh : m2;
  const int sw = (roi.rw && !roi.strided) ? roi.rw : m2;                 // row stride of the input array
  const int sox = roi.strided ? 0 : fox, soy = roi.strided ? 0 : foy;   // array origin in fine coordinates
  const float* src = dout + (size_t)s * (roi.strided ? (size_t)m2 * m2 : (size_t)fh * fw);
  // output rectangle (coarse coordinates): tiles outside it are not written at all -- the caller knows they are zero
  const bool has_or = orect.rw != 0;
  const int orx = has_or ? orect.ox[pl] : 0, ory = has_or ? orect.oy[pl] : 0;
  const size_t nn = (size_t)n * n;
  float* o_ll = dx + (size_t)s * nn;
  float* o_h = dyh + (size_t)s * 3 * nn;
  bool skip = false;
  float absacc = 0.f, dummy_abs = 0.f;
  if (FUSE) {
    if (fa.inv_scale_dev != nullptr) fa.a.inv_scale *= fa.inv_scale_dev[0];
    skip = fa.found_inf != nullptr && fa.found_inf[0] != 0.f;
  }

  int qr[KQ], qc[KQ];
  bool qv[KQ];
#pragma unroll
  for (int k = 0; k < KQ; k++) {
    const int q = threadIdx.x + NT * k;
    qv[k] = q < NQ;
    qr[k] = q / (FTA / 4);
    qc[k] = 4 * (q - qr[k] * (FTA / 4));
  }
  float4 pre[KQ];
  auto prefetch = [&](int tx) {
    const int a_c = tx * TI;
#pragma unroll
    for (int k = 0; k < KQ; k++) {
      const int ar = 2 * a_r - K + qr[k], ac = 2 * a_c - KA + qc[k];      // absolute fine coordinates
      const int gr = ar - foy, gc = ac - fox;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qv[k] && gr >= 0 && gr < fh && gc >= 0 && gc < fw)
        v = *reinterpret_cast<const float4*>(src + (size_t)(ar - soy) * sw + (ac - sox));
      pre[k] = v;
    }
  };
  // does the fine support of tile tx intersect the input window?  (uniform over the workgroup)
  auto hits = [&](int tx) {
    const int r0 = 2 * a_r - K, c0 = 2 * tx * TI - KA;
    return r0 < foy + fh && r0 + FT > foy && c0 < fox + fw && c0 + FTA > fox;
  };
  if (has_or && (a_r < ory || a_r >= ory + orect.rh)) return;            // whole tile row outside the rectangle
  if (hits(tx0)) prefetch(tx0);
  for (int tx = tx0; tx < tx1; tx++) {
    if (!hits(tx)) {
      // nothing of the gradient reaches this tile: its four coarse outputs are exactly zero
      if (tx + 1 < tx1 && hits(tx + 1)) prefetch(tx + 1);
      if (has_or && (tx * TI < orx || tx * TI >= orx + orect.rw)) continue;   // outside: not even stored
      if (!FUSE) {
        for (int u = threadIdx.x; u < TI * TI; u += NT) {
          const int gr = a_r + u / TI, gc = tx * TI + (u % TI);
          if (gr < n && gc < n) {
            const size_t off = (size_t)gr * n + gc;
            o_ll[off] = 0.f; o_h[off] = 0.f; o_h[nn + off] = 0.f; o_h[2 * nn + off] = 0.f;
          }
        }
        continue;
      }
    }
#pragma unroll
    for (int k = 0; k < KQ; k++) {
      if (qv[k]) {
        float* d = &fine[qr[k]][qc[k]];
        d[0] = pre[k].x; d[1] = pre[k].y; d[2] = pre[k].z; d[3] = pre[k].w;
      }
    }
    __syncthreads();
    if (tx + 1 < tx1 && hits(tx + 1)) prefetch(tx + 1);

    // Row pass: FT * TI / RM = 640 runs over 256 threads, as KT1 = 3 unrolled trips with the run index clamped (the
    // surplus threads of the last trip recompute its last run and store the same values again) instead of a
    // `for (u = tid; u < 640; u += NT)` loop, so that the scheduler may overlap one trip's LDS window reads with the
    // previous trip's FMAs: 679 -> 671 us in an interleaved A/B (1 %: this kernel is not bound by that latency either).
    constexpr int RUNS1 = FT * (TI / RM), KT1 = (RUNS1 + NT - 1) / NT;
#pragma unroll
    for (int kt = 0; kt < KT1; kt++) {
      const int u = min((int)threadIdx.x + kt * NT, RUNS1 - 1);
      const int r = u % FT, j0 = (u / FT) * RM;
      float w[WIN];
#pragma unroll
      for (int i = 0; i < WIN; i++) w[i] = fine[r][2 * j0 + i + SH];
#pragma unroll
      for (int j = 0; j < RM; j++) {
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int k = 0; k < L; k++) {
          const float t0 = T.g0[k], t1 = T.g1[k];
          if (t0 != 0.f) lo = fmaf(w[2 * j + k], t0, lo);
          if (t1 != 0.f) hi = fmaf(w[2 * j + k], t1, hi);
        }
        mid[0][r][j0 + j] = lo;
        mid[1][r][j0 + j] = hi;
      }
    }
    __syncthreads();

    const int a_c = tx * TI;
    constexpr int RUNS2 = TI * (TI / RM), KT2 = (RUNS2 + NT - 1) / NT;   // 256 runs: one trip
#pragma unroll
    for (int kt = 0; kt < KT2; kt++) {
      const int u = min((int)threadIdx.x + kt * NT, RUNS2 - 1);
      const int c = u % TI, j0 = (u / TI) * RM;
      float wl[WIN], wh[WIN];
#pragma unroll
      for (int i = 0; i < WIN; i++) { wl[i] = mid[0][2 * j0 + i][c]; wh[i] = mid[1][2 * j0 + i][c]; }
      const int gc = a_c + c;
#pragma unroll
      for (int j = 0; j < RM; j++) {
        float a = 0.f, b = 0.f, cc = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < L; k++) {
          const float t0 = T.g0[k], t1 = T.g1[k];
          if (t0 != 0.f) { a = fmaf(wl[2 * j + k], t0, a); cc = fmaf(wh[2 * j + k], t0, cc); }
          if (t1 != 0.f) { b = fmaf(wl[2 * j + k], t1, b); d = fmaf(wh[2 * j + k], t1, d); }
        }
        const int gr = a_r + j0 + j;
        if (gr < n && gc < n) {
          const size_t off = (size_t)gr * n + gc;
          if (!FUSE) {
            stg(o_ll + off, 2.0f * a, TNL_IDWT_BWD_NT);
            stg(o_h + off, b, TNL_IDWT_BWD_NT);
            stg(o_h + nn + off, cc, TNL_IDWT_BWD_NT);
            stg(o_h + 2 * nn + off, d, TNL_IDWT_BWD_NT);
          } else {
            const size_t hb = (size_t)s * 3 * nn + off;
            const float gband[3] = {b, cc, d};
#pragma unroll
            for (int q = 0; q < 3; q++) {
              float pp = fa.p[hb + q * nn];
              if (!skip) {
                float mm = fa.m[hb + q * nn], vv = fa.v[hb + q * nn];
                adam1(pp, gband[q], mm, vv, fa.a, absacc);
                fa.p[hb + q * nn] = pp; fa.m[hb + q * nn] = mm; fa.v[hb + q * nn] = vv;
              } else {
                absacc += fabsf(pp);
              }
            }
            if (fa.pl != nullptr) {  // coarsest level: the low-pass gradient is the LL parameter's (not regularised)
              if (!skip) {
                AdamArgs al = fa.a;
                al.l1_coef = 0.f;
                const size_t lo = (size_t)s * nn + off;
                float pp = fa.pl[lo], mm = fa.ml[lo], vv = fa.vl[lo];
                adam1(pp, 2.0f * a, mm, vv, al, dummy_abs);
                fa.pl[lo] = pp; fa.ml[lo] = mm; fa.vl[lo] = vv;
              }
            } else {
              o_ll[off] = 2.0f * a;
            }
          }
        }
      }
    }
  }
  if (FUSE && fa.abs_sum != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) absacc += __shfl_xor(absacc, off);
    __shared__ float part[NT / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = absacc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(fa.abs_sum, part[0] + part[1] + part[2] + part[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// Walk kernels (large levels, n >= walk_min_n): no staged input tile at all.
//
// The vertical pass of a separable filter needs no data from another lane, so a thread OWNS ONE COLUMN and walks down
// the plane with a rolling register window: every coefficient row is loaded from HBM exactly once per tile (coalesced
// 256-byte wave rows, the next WB rows already in flight in registers while the current WB are computed) -- the
// (1 + 8/32)^2 = 1.56x halo over-fetch and the band staging through LDS of the tile kernels above are gone; only the
// vertical pass's results cross lanes, through an LDS image of WB coarse rows, for the horizontal pass (16-byte LDS
// reads of a thread's 12 / 24-sample window).  Over-fetch: 8 of 128 columns per tile plus 8 rows per row segment.
// Same FMA order per output as the tile kernels in the forward direction (bit-identical planes).
// ---------------------------------------------------------------------------------------------
constexpr int WB = 8;          // coarse rows per phase (adjoint)
#ifndef TNL_FWD_FB
#define TNL_FWD_FB 4
#endif
constexpr int WT = 128;        // forward: threads = staged coarse columns of a tile
constexpr int WV = WT - 8;     // forward: coarse columns a tile produces (4-column halo each side)
constexpr int AT = 256;        // adjoint: threads = staged fine columns of a tile
constexpr int AVC = (AT - 16) / 2;   // adjoint: coarse columns a tile produces (= WV)
typedef float v4f __attribute__((ext_vector_type(4)));

// Logical (x, y, z) block of a walk kernel.  lg.w == 0: the launch grid is the logical grid.  Otherwise the launch is
// 1-D, padded to a multiple of 8, and block b is given logical index (b % 8) * (blocks / 8) + b / 8: workgroups are dealt
// round-robin over the 8 XCDs (MI355X_MICROARCH.md "Workgroup dispatch"), so consecutive logical blocks -- x-neighbour
// tiles, which share 8 halo columns and the 128-byte lines their unaligned edges straddle -- meet in one XCD's L2.
struct LGrid { int x, y, z, w; };
__device__ __forceinline__ bool logical_block(const LGrid& lg, int& bx, int& by, int& bz) {
  if (lg.w == 0) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return true; }
  const unsigned chunk = gridDim.x >> 3;
  const unsigned id = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
  if (id >= (unsigned)lg.x * lg.y * lg.z) return false;
  bx = id % lg.x; by = (id / lg.x) % lg.y; bz = id / (lg.x * lg.y);
  return true;
}

// Needed-column table of a level ("spans"; device int32 [3][n / 8][2], nullptr = everything): for plane pl and coarse rows
// 8g .. 8g+7 the coarse columns [lo, end) whose results anything reads.  A walk workgroup (columns [vx0, vx1), rows
// [ry0, ry1), both multiples of 8 at the low end) narrows its rows to the row groups whose piece meets its columns;
// false: none does.  TrainStep builds the tables from the occupied cells' projection (tnl_occupancy_row_extents).
__device__ __forceinline__ bool narrow_rows(const int* __restrict__ spans, int n, int pl, int vx0, int vx1, int& ry0,
                                            int& ry1, int* s_span) {
  if (threadIdx.x == 0) { s_span[0] = 0x7fffffff; s_span[1] = -1; }
  __syncthreads();
  const int g0 = ry0 >> 3, g1 = (ry1 - 1) >> 3;
  for (int g = g0 + (int)threadIdx.x; g <= g1; g += (int)blockDim.x) {
    const int lo = spans[(pl * (n >> 3) + g) * 2], hi = spans[(pl * (n >> 3) + g) * 2 + 1];
    if (lo < vx1 && hi > vx0) { atomicMin(&s_span[0], g); atomicMax(&s_span[1], g); }
  }
  __syncthreads();
  const int first = s_span[0], last = s_span[1];
  if (last < 0) return false;
  ry0 = max(ry0, 8 * first);
  ry1 = min(ry1, 8 * last + 8);
  return true;
}

template <int W, bool HALF_OUT, int FB>
__global__ void __launch_bounds__(WT)
k_idwt_fwd_walk(const float* __restrict__ x, const float* __restrict__ yh, int n, void* __restrict__ out, Roi roi,
                int seg, LGrid lg, const int* __restrict__ spans) {
  TNL_SET_PRIO(TNL_PRIO_FWD);
  // FB = coarse rows per phase (4 or 8): the rolling window holds FB + 8 rows of the four bands, FB more are in flight
  constexpr WTaps T = wtaps(W);
  constexpr int L = T.L, K = (L - 2) / 2, HW = L / 4;
  static_assert(HW <= 4, "staged halo is 4 coarse samples");
  constexpr int WW = FB + 8;                           // window rows: R0-4 .. R0+FB+3
  constexpr int LSM = WT + 4;                          // mid row stride: 16-byte aligned rows
  __shared__ __attribute__((aligned(16))) float mid[2][2 * FB][LSM];

  int bx, by, s;
  if (!logical_block(lg, bx, by, s)) return;
  const int tid = threadIdx.x, m2 = 2 * n;
  const int pl = roi.rw ? (s + roi.s0) / roi.spp : 0;
  const int fox = roi.rw ? roi.ox[pl] : 0, foy = roi.rw ? roi.oy[pl] : 0;
  const bool compact = roi.rw && !roi.strided;
  const int orow = compact ? roi.rw : m2;
  const size_t oplane = compact ? (size_t)roi.rh * roi.rw : (size_t)m2 * m2;
  const int sox = compact ? fox : 0, soy = compact ? foy : 0;
  const int cx0 = fox / 2, cy0 = foy / 2, cw = roi.rw ? roi.rw / 2 : n, ch = roi.rw ? roi.rh / 2 : n;
  const int vx0 = cx0 + bx * WV, vx1 = min(vx0 + WV, cx0 + cw);        // coarse columns this tile produces
  int ry0 = cy0 + by * seg, ry1 = min(ry0 + seg, cy0 + ch);            // coarse rows of this segment
  if (vx0 >= vx1 || ry0 >= ry1) return;
  __shared__ int s_span[2];
  if (spans != nullptr && !narrow_rows(spans, n, pl, vx0, vx1, ry0, ry1, s_span)) return;   // nothing here is read
  const int c = vx0 - 4 + tid;                                                 // this thread's coarse column
  const bool col_ok = c >= 0 && c < n && c < vx1 + 4;
  const size_t nn = (size_t)n * n;
  const float* b0 = x + (size_t)s * nn + (col_ok ? c : 0);
  const float* b1 = yh + (size_t)s * 3 * nn + (col_ok ? c : 0);
  auto ld4 = [&](int r, float& v0, float& v1, float& v2, float& v3) {
    v0 = v1 = v2 = v3 = 0.f;
    if (col_ok && r >= 0 && r < n) {
      const size_t o = (size_t)r * n;
      v0 = b0[o]; v1 = b1[o]; v2 = b1[nn + o]; v3 = b1[2 * nn + o];
    }
  };
  float w0[WW], w1[WW], w2[WW], w3[WW];                      // rows R0-4 .. R0+FB+3 of the four bands
  float n0[FB], n1[FB], n2[FB], n3[FB];                      // rows R0+FB+4 .. R0+2FB+3 (next phase), in flight
#pragma unroll
  for (int i = 0; i < WW; i++) ld4(ry0 - 4 + i, w0[i], w1[i], w2[i], w3[i]);
#pragma unroll
  for (int i = 0; i < WW; i++) w0[i] *= 2.0f;                // the 2* of triplane_encoder.py:379
  char* const obase = reinterpret_cast<char*>(out);
  for (int R0 = ry0; R0 < ry1; R0 += FB) {
    const bool more = R0 + FB < ry1;
    if (more) {
#pragma unroll
      for (int i = 0; i < FB; i++) ld4(R0 + FB + 4 + i, n0[i], n1[i], n2[i], n3[i]);
    }
    // vertical synthesis of this thread's column: coarse rows R0..R0+FB-1 -> fine rows 2*R0 .. of lo and hi
#pragma unroll
    for (int m = 0; m < FB; m++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int d = -HW; d <= HW; d++) {
          const int k = e + K - 2 * d;
          if (k >= 0 && k < L) {
            const float t0 = T.g0[k], t1 = T.g1[k];
            if (t0 != 0.f) { lo = fmaf(w0[m + d + 4], t0, lo); hi = fmaf(w2[m + d + 4], t0, hi); }
            if (t1 != 0.f) { lo = fmaf(w1[m + d + 4], t1, lo); hi = fmaf(w3[m + d + 4], t1, hi); }
          }
        }
        mid[0][2 * m + e][tid] = lo;
        mid[1][2 * m + e][tid] = hi;
      }
    }
    __syncthreads();
    // horizontal synthesis + store: 2*FB fine rows x 30 runs of 4 coarse (8 fine) columns; 32 lanes per row
#pragma unroll
    for (int trip = 0; trip < (2 * FB * 32) / WT; trip++) {
      const int row = trip * (WT / 32) + (tid >> 5), run = tid & 31;
      const int v = 4 * run, gcol = vx0 + v;
      if (run < WV / 4 && gcol < vx1) {
        float wl[12], wh[12];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const v4f a = *reinterpret_cast<const v4f*>(&mid[0][row][v + 4 * q]);
          const v4f b = *reinterpret_cast<const v4f*>(&mid[1][row][v + 4 * q]);
#pragma unroll
          for (int i = 0; i < 4; i++) { wl[4 * q + i] = a[i]; wh[4 * q + i] = b[i]; }
        }
        float o[8];
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
          for (int e = 0; e < 2; e++) {
            float acc = 0.f;
#pragma unroll
            for (int d = -HW; d <= HW; d++) {
              const int k = e + K - 2 * d;
              if (k >= 0 && k < L) {
                const float t0 = T.g0[k], t1 = T.g1[k];
                if (t0 != 0.f) acc = fmaf(wl[m + d + 4], t0, acc);
                if (t1 != 0.f) acc = fmaf(wh[m + d + 4], t1, acc);
              }
            }
            o[2 * m + e] = acc;
          }
        }
        const int gr = 2 * R0 + row, gc = 2 * gcol;
        const size_t off = (size_t)s * oplane + (size_t)(gr - soy) * orow + (gc - sox);
        if (HALF_OUT) {
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          h8 hv;
#pragma unroll
          for (int i = 0; i < 8; i++) hv[i] = (_Float16)o[i];
          *reinterpret_cast<h8*>(obase + off * 2) = hv;
        } else {
          float* p = reinterpret_cast<float*>(obase) + off;
          reinterpret_cast<v4f*>(p)[0] = v4f{o[0], o[1], o[2], o[3]};
          reinterpret_cast<v4f*>(p)[1] = v4f{o[4], o[5], o[6], o[7]};
        }
      }
    }
    __syncthreads();
    // slide the window down by FB rows
#pragma unroll
    for (int i = 0; i < 8; i++) { w0[i] = w0[i + FB]; w1[i] = w1[i + FB]; w2[i] = w2[i + FB]; w3[i] = w3[i + FB]; }
#pragma unroll
    for (int i = 0; i < FB; i++) { w0[8 + i] = 2.0f * n0[i]; w1[8 + i] = n1[i]; w2[8 + i] = n2[i]; w3[8 + i] = n3[i]; }
  }
}

// Forward walk, pair form: ONE WAVE per workgroup, a thread owns TWO adjacent coarse columns (8-byte loads: a wave row is
// 512 contiguous bytes per band, half the load and LDS-store instructions per coefficient) and the vertical -> horizontal
// hand-over through LDS needs no workgroup barrier (the wave is the workgroup).  The 2* of the LL band is folded into its
// vertical taps (fma(2a, t, c) == fma(a, 2t, c) bit for bit), the window slides with 64-bit moves.  Same tile (128
// staged / 120 produced coarse columns), same FMA order per output as k_idwt_fwd_walk: bit-identical planes.
// Needs even coarse column origins and widths (the launcher checks; otherwise the one-column form runs).
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int PT = 64;         // pair form: threads (= WT / 2 column pairs)

template <int W, bool HALF_OUT, int FB>
__global__ void __launch_bounds__(PT)
k_idwt_fwd_walk2(const float* __restrict__ x, const float* __restrict__ yh, int n, void* __restrict__ out, Roi roi,
                 int seg, LGrid lg, const int* __restrict__ spans) {
  constexpr WTaps T = wtaps(W);
  constexpr int L = T.L, K = (L - 2) / 2, HW = L / 4;
  static_assert(HW <= 4, "staged halo is 4 coarse samples");
  constexpr int WW = FB + 8;                           // window rows: R0-4 .. R0+FB+3
  constexpr int LSM = WT + 4;                          // mid row stride: 16-byte aligned rows
  __shared__ __attribute__((aligned(16))) float mid[2][2 * FB][LSM];

  int bx, by, s;
  if (!logical_block(lg, bx, by, s)) return;
  const int tid = threadIdx.x, m2 = 2 * n;
  const int pl = roi.rw ? (s + roi.s0) / roi.spp : 0;
  const int fox = roi.rw ? roi.ox[pl] : 0, foy = roi.rw ? roi.oy[pl] : 0;
  const bool compact = roi.rw && !roi.strided;
  const int orow = compact ? roi.rw : m2;
  const size_t oplane = compact ? (size_t)roi.rh * roi.rw : (size_t)m2 * m2;
  const int sox = compact ? fox : 0, soy = compact ? foy : 0;
  const int cx0 = fox / 2, cy0 = foy / 2, cw = roi.rw ? roi.rw / 2 : n, ch = roi.rw ? roi.rh / 2 : n;
  const int vx0 = cx0 + bx * WV, vx1 = min(vx0 + WV, cx0 + cw);        // coarse columns this tile produces
  int ry0 = cy0 + by * seg, ry1 = min(ry0 + seg, cy0 + ch);            // coarse rows of this segment
  if (vx0 >= vx1 || ry0 >= ry1) return;
  __shared__ int s_span[2];
  if (spans != nullptr && !narrow_rows(spans, n, pl, vx0, vx1, ry0, ry1, s_span)) return;   // nothing here is read
  const int c = vx0 - 4 + 2 * tid;                                             // this thread's coarse columns c, c + 1
  const bool col_ok = c >= 0 && c < n && c < vx1 + 4;                          // (all three even: true of both or neither)
  const size_t nn = (size_t)n * n;
  const float* b0 = x + (size_t)s * nn + (col_ok ? c : 0);
  const float* b1 = yh + (size_t)s * 3 * nn + (col_ok ? c : 0);
  auto ld4 = [&](int r, v2f& v0, v2f& v1, v2f& v2, v2f& v3) {
    v0 = v1 = v2 = v3 = v2f{0.f, 0.f};
    if (col_ok && r >= 0 && r < n) {
      const size_t o = (size_t)r * n;
      v0 = *reinterpret_cast<const v2f*>(b0 + o); v1 = *reinterpret_cast<const v2f*>(b1 + o);
      v2 = *reinterpret_cast<const v2f*>(b1 + nn + o); v3 = *reinterpret_cast<const v2f*>(b1 + 2 * nn + o);
    }
  };
  v2f w0[WW], w1[WW], w2[WW], w3[WW];                        // rows R0-4 .. R0+FB+3 of the four bands
  v2f n0[FB], n1[FB], n2[FB], n3[FB];                        // rows R0+FB+4 .. R0+2FB+3 (next phase), in flight
#pragma unroll
  for (int i = 0; i < WW; i++) ld4(ry0 - 4 + i, w0[i], w1[i], w2[i], w3[i]);
  char* const obase = reinterpret_cast<char*>(out);
  for (int R0 = ry0; R0 < ry1; R0 += FB) {
    const bool more = R0 + FB < ry1;
    if (more) {
#pragma unroll
      for (int i = 0; i < FB; i++) ld4(R0 + FB + 4 + i, n0[i], n1[i], n2[i], n3[i]);
    }
    // vertical synthesis of this thread's two columns: coarse rows R0..R0+FB-1 -> fine rows 2*R0 .. of lo and hi
#pragma unroll
    for (int m = 0; m < FB; m++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        v2f lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
        for (int d = -HW; d <= HW; d++) {
          const int k = e + K - 2 * d;
          if (k >= 0 && k < L) {
            const float t0 = T.g0[k], t1 = T.g1[k];
            const float t0x2 = 2.0f * t0;                                // the 2* of triplane_encoder.py:379 (LL band)
            if (t0 != 0.f) {
              lo = __builtin_elementwise_fma(w0[m + d + 4], v2f{t0x2, t0x2}, lo);
              hi = __builtin_elementwise_fma(w2[m + d + 4], v2f{t0, t0}, hi);
            }
            if (t1 != 0.f) {
              lo = __builtin_elementwise_fma(w1[m + d + 4], v2f{t1, t1}, lo);
              hi = __builtin_elementwise_fma(w3[m + d + 4], v2f{t1, t1}, hi);
            }
          }
        }
        *reinterpret_cast<v2f*>(&mid[0][2 * m + e][2 * tid]) = lo;
        *reinterpret_cast<v2f*>(&mid[1][2 * m + e][2 * tid]) = hi;
      }
    }
    __syncthreads();      // (one wave: no s_barrier, the LDS counter wait only)
    // horizontal synthesis + store: 2*FB fine rows x 30 runs of 4 coarse (8 fine) columns; 32 lanes per row
#pragma unroll
    for (int trip = 0; trip < (2 * FB * 32) / PT; trip++) {
      const int row = trip * (PT / 32) + (tid >> 5), run = tid & 31;
      const int v = 4 * run, gcol = vx0 + v;
      if (run < WV / 4 && gcol < vx1) {
        float wl[12], wh[12];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const v4f a = *reinterpret_cast<const v4f*>(&mid[0][row][v + 4 * q]);
          const v4f b = *reinterpret_cast<const v4f*>(&mid[1][row][v + 4 * q]);
#pragma unroll
          for (int i = 0; i < 4; i++) { wl[4 * q + i] = a[i]; wh[4 * q + i] = b[i]; }
        }
        float o[8];
#pragma unroll
        for (int m = 0; m < 4; m++) {
#pragma unroll
          for (int e = 0; e < 2; e++) {
            float acc = 0.f;
#pragma unroll
            for (int d = -HW; d <= HW; d++) {
              const int k = e + K - 2 * d;
              if (k >= 0 && k < L) {
                const float t0 = T.g0[k], t1 = T.g1[k];
                if (t0 != 0.f) acc = fmaf(wl[m + d + 4], t0, acc);
                if (t1 != 0.f) acc = fmaf(wh[m + d + 4], t1, acc);
              }
            }
            o[2 * m + e] = acc;
          }
        }
        const int gr = 2 * R0 + row, gc = 2 * gcol;
        const size_t off = (size_t)s * oplane + (size_t)(gr - soy) * orow + (gc - sox);
        if (HALF_OUT) {
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          h8 hv;
#pragma unroll
          for (int i = 0; i < 8; i++) hv[i] = (_Float16)o[i];
          *reinterpret_cast<h8*>(obase + off * 2) = hv;
        } else {
          float* p = reinterpret_cast<float*>(obase) + off;
          reinterpret_cast<v4f*>(p)[0] = v4f{o[0], o[1], o[2], o[3]};
          reinterpret_cast<v4f*>(p)[1] = v4f{o[4], o[5], o[6], o[7]};
        }
      }
    }
    __syncthreads();
    // slide the window down by FB rows
#pragma unroll
    for (int i = 0; i < 8; i++) { w0[i] = w0[i + FB]; w1[i] = w1[i + FB]; w2[i] = w2[i + FB]; w3[i] = w3[i + FB]; }
#pragma unroll
    for (int i = 0; i < FB; i++) { w0[8 + i] = n0[i]; w1[8 + i] = n1[i]; w2[8 + i] = n2[i]; w3[8 + i] = n3[i]; }
  }
}

// Adjoint walk: a thread owns one FINE column of the input gradient and walks down with an 18-row rolling window
// (vertical analysis: v_lo = g0 . column, v_hi = g1 . column per coarse row), the horizontal analysis of WB coarse
// rows at a time goes through LDS.  The output region is the gradient-support rectangle `orect` (or the whole plane):
// every element of it is written -- computed from the input window `roi` (zero outside it) -- nothing outside is.
//
// FUSE: the level's band gradients are consumed where they are produced -- the Adam(+L1) update of the LIVE pieces of the
// level (adam.hip k_adam_l1_live's arithmetic, the step's scalars from its ring record) runs in this kernel's epilogue and
// dyh is never written: 24 instead of 32 bytes per live coefficient over the adjoint + optimiser pair.  `orect` is then the
// level's LIVE rectangle (a superset of the gradient's support: outside the support the sums below are exact zeros, the
// g = 0 the unfused pass takes there), `spans` the live band pieces as a span table and wa.bt the same pieces as the
// optimiser's band table (or nullptr: the whole rectangle is live).  p, m, v of a thread's three quads are requested at
// the top of the phase that produces their gradients.
struct WalkAdam {
  float *p, *m, *v;              // [S][3][n][n] of this level (the slices the launch covers)
  AdamArgs a;
  const AdamStepRec* rec;
  const float* inv_scale_dev;
  const float* found_inf;
  float* abs_sum;
  const int* bt;
  int nb;
};

#ifndef TNL_BWD_WALK_WAVES
#define TNL_BWD_WALK_WAVES 1
#endif
template <int W, bool FUSE>
__global__ void __launch_bounds__(AT) __attribute__((amdgpu_waves_per_eu(TNL_BWD_WALK_WAVES)))
k_idwt_bwd_walk(const float* __restrict__ dout, int n, float* __restrict__ dx, float* __restrict__ dyh, Roi roi,
                Roi orect, int seg, LGrid lg, const int* __restrict__ spans, WalkAdam wa) {
  TNL_SET_PRIO(TNL_PRIO_BWD);
  constexpr WTaps T = wtaps(W);
  constexpr int L = T.L, K = (L - 2) / 2;
  constexpr int KA = (K + 3) / 4 * 4, SH = KA - K;       // staged left halo (fine samples), 16-byte aligned
  static_assert(2 * (AVC - 1) + SH + L <= AT, "tile too narrow for this filter");
  constexpr int LSA = AT + 4;
  __shared__ __attribute__((aligned(16))) float mid[2][WB][LSA];

  int bx, by, s;
  if (!logical_block(lg, bx, by, s)) return;
  const int tid = threadIdx.x, m2 = 2 * n;
  const int pl = roi.rw ? (s + roi.s0) / roi.spp : (orect.rw ? (s + orect.s0) / orect.spp : 0);
  const int fox = roi.rw ? roi.ox[pl] : 0, foy = roi.rw ? roi.oy[pl] : 0;
  const int fw = roi.rw ? roi.rw : m2, fh = roi.rw ? roi.rh : m2;
  const int sw = (roi.rw && !roi.strided) ? roi.rw : m2;
  const int sox = roi.strided ? 0 : fox, soy = roi.strided ? 0 : foy;
  const float* src = dout + (size_t)s * (roi.strided ? (size_t)m2 * m2 : (size_t)fh * fw);
  const bool has_or = orect.rw != 0;
  const int ox0 = has_or ? orect.ox[pl] : 0, oy0 = has_or ? orect.oy[pl] : 0;
  const int ow = has_or ? orect.rw : n, oh = has_or ? orect.rh : n;
  const int vx0 = ox0 + bx * AVC, vx1 = min(vx0 + AVC, ox0 + ow);      // coarse columns this tile produces
  int ry0 = oy0 + by * seg, ry1 = min(ry0 + seg, oy0 + oh);            // coarse rows of this segment
  if (vx0 >= vx1 || ry0 >= ry1) return;
  if (spans != nullptr) {
    // outside the spans the band gradients are read by nobody and stay unwritten; the LL gradient is the next level's
    // input, which reads the whole rectangle: it gets its (exact) zeros
    __shared__ int s_span[2];
    const int fy0 = ry0, fy1 = ry1;
    const bool any = narrow_rows(spans, n, pl, vx0, vx1, ry0, ry1, s_span);
    if (!any) ry0 = ry1 = fy1;
    float* zl = dx + (size_t)s * n * n;
    const int zj = tid >> 5, zc = vx0 + 4 * (tid & 31);
    if ((tid & 31) < AVC / 4 && zc < vx1) {
      for (int r = fy0 + zj; r < ry0; r += AT / 32) *reinterpret_cast<v4f*>(zl + (size_t)r * n + zc) = v4f{0.f, 0.f, 0.f, 0.f};
      for (int r = ry1 + zj; r < fy1; r += AT / 32) *reinterpret_cast<v4f*>(zl + (size_t)r * n + zc) = v4f{0.f, 0.f, 0.f, 0.f};
    }
    if (!any) return;
  }
  const int fc = 2 * vx0 - KA + tid;                                            // this thread's fine column
  const bool col_ok = fc - fox >= 0 && fc - fox < fw && fc < 2 * vx1 + L;       // inside the input window
  const float* colp = src + (col_ok ? (fc - sox) : 0);
  auto ld = [&](int r) -> float {        // fine row r (absolute)
    const int gr = r - foy;
    return (col_ok && gr >= 0 && gr < fh) ? colp[(size_t)(r - soy) * sw] : 0.f;
  };
  float w[4 * WB];        // fine rows 2*R0-K .. 2*R0-K+31
  float nx[2 * WB];       // the next 16 rows, in flight
#pragma unroll
  for (int i = 0; i < 4 * WB; i++) w[i] = ld(2 * ry0 - K + i);
  const size_t nn = (size_t)n * n;
  float* o_ll = dx + (size_t)s * nn;
  float* o_h = FUSE ? nullptr : dyh + (size_t)s * 3 * nn;
  // FUSE: the step's scalars, and this thread's output quad of every phase (row R0 + qj, columns qcol .. qcol + 3)
  AdamArgs aa = wa.a;
  bool skip = false;
  float abs_acc = 0.f;
  const int qj = tid >> 5, qcol = vx0 + 4 * (tid & 31);
  const bool qcol_ok = (tid & 31) < AVC / 4 && qcol < vx1;
  if (FUSE) {
    aa.step_size = wa.rec->step_size;
    aa.bias2_sqrt = wa.rec->bias2_sqrt;
    if (wa.inv_scale_dev != nullptr) aa.inv_scale *= wa.inv_scale_dev[0];
    skip = (wa.found_inf != nullptr && wa.found_inf[0] != 0.f) || wa.rec->skip != 0.f;
    aa.l1_coef = wa.a.l1_coef + wa.rec->pad;
  }
  float* const pq = FUSE ? wa.p + (size_t)s * 3 * nn : nullptr;
  float* const mq = FUSE ? wa.m + (size_t)s * 3 * nn : nullptr;
  float* const vq = FUSE ? wa.v + (size_t)s * 3 * nn : nullptr;
  v4f Pq[3], Mq[3], Vq[3];
  // the live column piece [first, first + width) of each 8-row band of this segment (seg <= 256 rows)
  __shared__ int s_piece[2][32];
  if (FUSE) {
    for (int k = tid; k < (ry1 - ry0 + 7) / 8; k += AT) {
      const int b = ((ry0 - oy0) >> 3) + k;
      s_piece[0][k] = wa.bt != nullptr ? wa.bt[2 * wa.nb + 1 + pl * wa.nb + b] : ox0;
      s_piece[1][k] = wa.bt != nullptr ? 4 * wa.bt[wa.nb + 1 + b] : ow;
    }
    __syncthreads();
  }
  const size_t off_idle = (size_t)ry0 * n + vx0;      // what a thread without a live quad loads (and never stores)
  for (int R0 = ry0; R0 < ry1; R0 += WB) {
    const bool more = R0 + WB < ry1;
    uint32_t nx_ok = 0;       // FUSE: which of the rows in flight exist (masked when the window slides, not at the load)
    if (FUSE) {
      // unconditional loads from clamped rows: a value selected at the load is a wait at the load, and the waits of this
      // loop must leave the optimiser's stores of the previous phase in flight
#pragma unroll
      for (int i = 0; i < 2 * WB; i++) {
        const int r = 2 * R0 - K + 4 * WB + i, gr = r - foy;
        if (col_ok && gr >= 0 && gr < fh) nx_ok |= 1u << i;
        const int rc = min(max(gr, 0), fh - 1) + foy;
        nx[i] = colp[(size_t)(rc - soy) * sw];
      }
    } else if (more) {
#pragma unroll
      for (int i = 0; i < 2 * WB; i++) nx[i] = ld(2 * R0 - K + 4 * WB + i);
    }
    bool member = false;
    if (FUSE) {
      // unconditional loads (a thread without a live quad reads one that exists): no merge of loaded and unloaded
      // values for the compiler to wait on -- the nine quads stay in flight until the epilogue
      const int kb = (R0 - ry0) >> 3, x0 = s_piece[0][kb], wq = s_piece[1][kb];
      member = qcol_ok && R0 + qj < ry1 && qcol >= x0 && qcol < x0 + wq;
      const size_t off = member ? (size_t)(R0 + qj) * n + qcol : off_idle;
#pragma unroll
      for (int b = 0; b < 3; b++) {
        Pq[b] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(pq + b * nn + off));
        Mq[b] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(mq + b * nn + off));
        Vq[b] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(vq + b * nn + off));
      }
    }
#pragma unroll
    for (int j = 0; j < WB; j++) {
      float lo = 0.f, hi = 0.f;
#pragma unroll
      for (int k = 0; k < L; k++) {
        const float t0 = T.g0[k], t1 = T.g1[k];
        if (t0 != 0.f) lo = fmaf(w[2 * j + k], t0, lo);
        if (t1 != 0.f) hi = fmaf(w[2 * j + k], t1, hi);
      }
      mid[0][j][tid] = lo;
      mid[1][j][tid] = hi;
    }
    if (FUSE) {
      // the window slides HERE (nothing below reads it): the wait for the rows in flight then falls a whole vertical pass
      // behind the previous phase's stores instead of right behind this phase's
#pragma unroll
      for (int i = 0; i < 2 * WB; i++) { w[i] = w[i + 2 * WB]; w[i + 2 * WB] = ((nx_ok >> i) & 1u) ? nx[i] : 0.f; }
#pragma unroll
      for (int i = 0; i < 2 * WB; i++) asm volatile("" : "+v"(w[i + 2 * WB]));      // (here, not sunk behind the stores)
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    {
      const int j = tid >> 5, run = tid & 31;            // 8 coarse rows x 30 runs of 4 coarse columns
      const int v = 4 * run, gcol = vx0 + v, gr = R0 + j;
      if (run < AVC / 4 && gcol < vx1 && gr < ry1) {
        v4f r0, r1, r2, r3;
        if (FUSE) {
          // the same four chains per output, the vertical-low half first, then the vertical-high half: half the window
          // registers at a time (the optimiser's p, m, v quads are in flight beside them)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            float wx[24];
#pragma unroll
            for (int q = 0; q < 6; q++) {
              const v4f a = *reinterpret_cast<const v4f*>(&mid[h][j][2 * v + 4 * q]);
#pragma unroll
              for (int i = 0; i < 4; i++) wx[4 * q + i] = a[i];
            }
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
              float a = 0.f, cc = 0.f;
#pragma unroll
              for (int k = 0; k < L; k++) {
                const float t0 = T.g0[k], t1 = T.g1[k];
                if (t0 != 0.f) a = fmaf(wx[SH + 2 * jj + k], t0, a);
                if (t1 != 0.f) cc = fmaf(wx[SH + 2 * jj + k], t1, cc);
              }
              if (h == 0) { r0[jj] = 2.0f * a; r2[jj] = cc; } else { r1[jj] = a; r3[jj] = cc; }
            }
          }
        } else {
        float wl[24], wh[24];
#pragma unroll
        for (int q = 0; q < 6; q++) {
          const v4f a = *reinterpret_cast<const v4f*>(&mid[0][j][2 * v + 4 * q]);
          const v4f b = *reinterpret_cast<const v4f*>(&mid[1][j][2 * v + 4 * q]);
#pragma unroll
          for (int i = 0; i < 4; i++) { wl[4 * q + i] = a[i]; wh[4 * q + i] = b[i]; }
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
          float a = 0.f, b = 0.f, cc = 0.f, d = 0.f;
#pragma unroll
          for (int k = 0; k < L; k++) {
            const float t0 = T.g0[k], t1 = T.g1[k];
            // ll = (W lo, H lo), yh[0] = (W lo, H hi), yh[1] = (W hi, H lo), yh[2] = (W hi, H hi)
            if (t0 != 0.f) { a = fmaf(wl[SH + 2 * jj + k], t0, a); b = fmaf(wh[SH + 2 * jj + k], t0, b); }
            if (t1 != 0.f) { cc = fmaf(wl[SH + 2 * jj + k], t1, cc); d = fmaf(wh[SH + 2 * jj + k], t1, d); }
          }
          r0[jj] = 2.0f * a; r1[jj] = b; r2[jj] = cc; r3[jj] = d;
        }
        }
        const size_t off = (size_t)gr * n + gcol;
        if (FUSE) {
          *reinterpret_cast<v4f*>(o_ll + off) = r0;
          if (member) {
            const v4f gq[3] = {r1, r2, r3};
#pragma unroll
            for (int b = 0; b < 3; b++) {
              if (!skip) {
                // (a wavefront whose coefficients all sit at the p = m = v = g = 0 fixed point stores nothing: k_adam_l1_live)
                auto bits = [](const v4f& t) { return __float_as_uint(t[0]) | __float_as_uint(t[1]) | __float_as_uint(t[2]) | __float_as_uint(t[3]); };
                const uint32_t any = (bits(Pq[b]) | bits(gq[b]) | bits(Mq[b]) | bits(Vq[b])) & 0x7fffffffu;
                const bool moved = __ballot(any != 0u) != 0ull;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                  float pp = Pq[b][e], mm = Mq[b][e], vv = Vq[b][e];
                  adam1(pp, gq[b][e], mm, vv, aa, abs_acc);
                  Pq[b][e] = pp; Mq[b][e] = mm; Vq[b][e] = vv;
                }
                if (moved) {
                  __builtin_nontemporal_store(Pq[b], reinterpret_cast<v4f*>(pq + b * nn + off));
                  __builtin_nontemporal_store(Mq[b], reinterpret_cast<v4f*>(mq + b * nn + off));
                  __builtin_nontemporal_store(Vq[b], reinterpret_cast<v4f*>(vq + b * nn + off));
                }
              } else {
                abs_acc += fabsf(Pq[b][0]) + fabsf(Pq[b][1]) + fabsf(Pq[b][2]) + fabsf(Pq[b][3]);
              }
            }
          }
        } else if (TNL_IDWT_BWD_NT) {
          __builtin_nontemporal_store(r0, reinterpret_cast<v4f*>(o_ll + off));
          __builtin_nontemporal_store(r1, reinterpret_cast<v4f*>(o_h + off));
          __builtin_nontemporal_store(r2, reinterpret_cast<v4f*>(o_h + nn + off));
          __builtin_nontemporal_store(r3, reinterpret_cast<v4f*>(o_h + 2 * nn + off));
        } else {
          *reinterpret_cast<v4f*>(o_ll + off) = r0;
          *reinterpret_cast<v4f*>(o_h + off) = r1;
          *reinterpret_cast<v4f*>(o_h + nn + off) = r2;
          *reinterpret_cast<v4f*>(o_h + 2 * nn + off) = r3;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2 * WB; i++) if (!FUSE) { w[i] = w[i + 2 * WB]; w[i + 2 * WB] = nx[i]; }
  }
  if (FUSE && wa.abs_sum != nullptr) {       // sum |p| as this step saw it (the L1 value), one atomic per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) abs_acc += __shfl_xor(abs_acc, off);
    __shared__ float part[AT / 64];
    if ((tid & 63) == 0) part[tid >> 6] = abs_acc;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < AT / 64; k++) t += part[k];
      atomicAdd(wa.abs_sum, t);
    }
  }
}

// fp16 (3,C,R,R) -> fp16 [3,R,R,C]: 16-byte loads along x, 16-byte stores along the channels.
// TY rows of a 64-texel segment per workgroup (round 6; the rows' loads are all issued before the barrier): with one row
// a workgroup holds 2-4 KB in flight and the pass -- 62 000 workgroups at base -- was bound by the load latency of its ~30
// rounds per CU (82 us for 254 MB = 3.1 TB/s); TNL_LAYOUT_ROWS = 1 restores that form.
#ifndef TNL_LAYOUT_ROWS
#define TNL_LAYOUT_ROWS 4
#endif
template <int TY>
__global__ void __launch_bounds__(NT)
k_to_texel_major_h(const _Float16* __restrict__ cm, int C, int R, _Float16* __restrict__ tm, Roi roi,
                   const int* __restrict__ spans) {
  extern __shared__ __attribute__((aligned(16))) _Float16 tileh_all[];  // TY x ([C][TX + 8] + 8 halfs of shift per 8 channels)
  constexpr int LD = TX + 8;
  const int row_halfs = C * LD + (C / 8) * 8;
  // row of channel c starts at c*LD + (c/8)*8: the extra 16 bytes per channel group put the four groups that one
  // transposed read touches on different banks (8*LD halfs = 288 dwords = 0 mod 32 otherwise: measured 67 % of
  // the LDS cycles were bank conflicts)
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  // source: (3,C,R,R), or the compact ROI window [3*C][rh][rw]; destination: always the full [3,R,R,C] array
  const int p = blockIdx.z;
  const int sw = roi.rw ? roi.rw : R, shh = roi.rw ? roi.rh : R;
  const int ys0 = blockIdx.y * TY, xs0 = blockIdx.x * TX;            // source coordinates (TY rows: one 8-row group of the spans)
  const int y0 = ys0 + (roi.rw ? roi.oy[p] : 0), x0 = xs0 + (roi.rw ? roi.ox[p] : 0);
  if (spans != nullptr) {     // [3][R / 8][2]: texel columns of this row group any sample can read
    const int lo = spans[(p * (R >> 3) + (y0 >> 3)) * 2], hi = spans[(p * (R >> 3) + (y0 >> 3)) * 2 + 1];
    if (!(lo < x0 + TX && hi > x0)) return;
  }
#pragma unroll
  for (int ry = 0; ry < TY; ry++) {
    _Float16* tileh = tileh_all + (size_t)ry * row_halfs;
    const int ys = ys0 + ry;
    for (int idx = threadIdx.x; idx < C * (TX / 8); idx += NT) {
      const int c = idx / (TX / 8), x8 = (idx - c * (TX / 8)) * 8;
      h8 v;
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = (_Float16)0.f;
      if (xs0 + x8 < sw && ys < shh) v = *reinterpret_cast<const h8*>(cm + (((size_t)p * C + c) * shh + ys) * sw + xs0 + x8);
      *reinterpret_cast<h8*>(&tileh[c * LD + (c >> 3) * 8 + x8]) = v;
    }
  }
  __syncthreads();
  const int CG = C / 8;
#pragma unroll
  for (int ry = 0; ry < TY; ry++) {
    const _Float16* tileh = tileh_all + (size_t)ry * row_halfs;
    const int y = y0 + ry;
    if (ys0 + ry >= shh) break;
    const size_t base = (((size_t)p * R + y) * R + x0) * C;
    for (int idx = threadIdx.x; idx < TX * CG; idx += NT) {
      const int xx = idx / CG, cg = idx - xx * CG;
      if (x0 + xx < R) {
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = tileh[(cg * 8 + j) * LD + cg * 8 + xx];
        *reinterpret_cast<h8*>(tm + base + (size_t)xx * C + cg * 8) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// layout changes: (3,C,R,R) fp32 <-> [3,R,R,C] (texel-major, fp16 or fp32)
// One workgroup moves a 64-texel row segment for all channels through LDS, so both the read of
// each channel row (256 B) and the write of the texel block (64*C*e B) are contiguous.
// ---------------------------------------------------------------------------------------------

template <bool HALF>
__global__ void __launch_bounds__(NT)
k_to_texel_major(const float* __restrict__ cm, int C, int R, void* __restrict__ tm, Roi roi) {
  // roi.rw != 0: only the window's row segments are launched; source and destination keep their whole-plane strides
  // (the texels outside the window are left as they are).  16-byte loads along x, 16-byte stores along the channels.
  extern __shared__ float tile[];  // [C][TX+1]
  const int p = blockIdx.z;
  const int y = blockIdx.y + (roi.rw ? roi.oy[p] : 0), x0 = blockIdx.x * TX + (roi.rw ? roi.ox[p] : 0);
  if ((R & 3) == 0) {
    for (int idx = threadIdx.x; idx < C * (TX / 4); idx += NT) {
      const int c = idx / (TX / 4), xx = (idx - c * (TX / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 + xx < R) v = *reinterpret_cast<const float4*>(cm + (((size_t)p * C + c) * R + y) * R + x0 + xx);
      float* t = tile + c * (TX + 1) + xx;
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  } else {
    for (int idx = threadIdx.x; idx < C * TX; idx += NT) {
      const int c = idx / TX, xx = idx - c * TX;
      tile[c * (TX + 1) + xx] = (x0 + xx < R) ? cm[(((size_t)p * C + c) * R + y) * R + x0 + xx] : 0.f;
    }
  }
  __syncthreads();
  const size_t base = (((size_t)p * R + y) * R + x0) * C;
  if (HALF && (C & 7) == 0) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int CG = C / 8;
    for (int idx = threadIdx.x; idx < TX * CG; idx += NT) {
      const int xx = idx / CG, cg = idx - xx * CG;
      if (x0 + xx < R) {
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (_Float16)__float2half(tile[(cg * 8 + j) * (TX + 1) + xx]);
        *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(tm) + base + (size_t)xx * C + cg * 8) = v;
      }
    }
  } else if (!HALF && (C & 3) == 0) {
    const int CG = C / 4;
    for (int idx = threadIdx.x; idx < TX * CG; idx += NT) {
      const int xx = idx / CG, cg = idx - xx * CG;
      if (x0 + xx < R) {
        const float* t = tile + (cg * 4) * (TX + 1) + xx;
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(tm) + base + (size_t)xx * C + cg * 4) =
            make_float4(t[0], t[TX + 1], t[2 * (TX + 1)], t[3 * (TX + 1)]);
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < C * TX; idx += NT) {
      const int xx = idx / C, c = idx - xx * C;
      if (x0 + xx < R) {
        const float v = tile[c * (TX + 1) + xx];
        if (HALF) reinterpret_cast<__half*>(tm)[base + idx] = __float2half(v);
        else reinterpret_cast<float*>(tm)[base + idx] = v;
      }
    }
  }
}

__global__ void __launch_bounds__(NT)
k_to_channel_major(const float* __restrict__ tm, int C, int R, float* __restrict__ cm) {
  extern __shared__ float tile[];  // [C][TX+1]
  const int p = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * TX;
  const size_t base = (((size_t)p * R + y) * R + x0) * C;
  for (int idx = threadIdx.x; idx < C * TX; idx += NT) {
    const int xx = idx / C, c = idx - xx * C;
    tile[c * (TX + 1) + xx] = (x0 + xx < R) ? tm[base + idx] : 0.f;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < C * TX; idx += NT) {
    const int c = idx / TX, xx = idx - c * TX;
    if (x0 + xx < R) cm[(((size_t)p * C + c) * R + y) * R + x0 + xx] = tile[c * (TX + 1) + xx];
  }
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

int g_walk_min_n = 512;   // levels with n >= this (and n % 8 == 0) run the walk kernels; tnl_idwt_set_walk_min_n
int g_fwd_fb = TNL_FWD_FB;   // tnl_idwt_set_tuning(1, 4 | 8): coarse rows per phase of the forward walk kernel
int g_xcd = 1;               // tnl_idwt_set_tuning(2, 0 | 1): XCD-aware block order of the walk kernels
int g_seg = 0;               // tnl_idwt_set_tuning(3, rows): rows per workgroup (0 = pick_seg)
#ifndef TNL_FWD_PAIR
#define TNL_FWD_PAIR 0
#endif
int g_fwd_pair = TNL_FWD_PAIR;   // tnl_idwt_set_tuning(4, 0 | 2 | 4): pair form of the forward walk (rows per phase), 0 = one column

// launch geometry of a walk kernel: the logical grid, or its XCD-ordered 1-D form
inline void walk_grid(uint32_t gx, uint32_t gy, uint32_t gz, dim3& grid, LGrid& lg) {
  if (g_xcd) {
    const uint64_t tot = (uint64_t)gx * gy * gz;
    grid = dim3((uint32_t)((tot + 7) / 8 * 8));
    lg = LGrid{(int)gx, (int)gy, (int)gz, 1};
  } else {
    grid = dim3(gx, gy, gz);
    lg = LGrid{(int)gx, (int)gy, (int)gz, 0};
  }
}

// tiles a workgroup of the pipelined tile kernels walks: as many as keep >= ~3000 workgroups in the launch
inline int pick_tpw(uint32_t ntx, uint32_t nty, uint32_t S) {
  int tpw = TPW_MAX;
  while (tpw > 1 && (uint64_t)cdiv(ntx, tpw) * nty * S < 3000) tpw >>= 1;
  return tpw;
}
// coarse rows per workgroup of the walk kernels (multiple of WB)
inline int pick_seg(uint32_t tiles, uint32_t rows, uint32_t S) {
  // measured at the base size (n = 1024, 96 slices): 64 rows per workgroup beat 128 and 256 by 5-8 % in both
  // directions (13.8 k workgroups instead of 3.5 k: shorter tails, more row streams in flight)
  int seg = 256;
  while (seg > 32 && (uint64_t)tiles * cdiv(rows, seg) * S < 12000) seg >>= 1;
  return seg;
}

template <int W>
int launch_fwd(const float* x, const float* yh, uint32_t S, uint32_t n, void* out, int half_out, hipStream_t st,
               Roi roi = Roi{}, const int* spans = nullptr) {
  if (spans != nullptr && !(roi.rw && n % 8 == 0 && (int)n >= g_walk_min_n)) spans = nullptr;   // walk kernels only
  if (n % 8 == 0 && (int)n >= g_walk_min_n) {
    const uint32_t cw = roi.rw ? roi.rw / 2 : n, ch = roi.rw ? roi.rh / 2 : n;
    const uint32_t tiles = cdiv(cw, WV);
    const int seg = g_seg ? g_seg : pick_seg(tiles, ch, S);
    dim3 grid;
    LGrid lg;
    walk_grid(tiles, cdiv(ch, seg), S, grid, lg);
#define TNL_FWD_WALK(HALF, FB) \
  hipLaunchKernelGGL((k_idwt_fwd_walk<W, HALF, FB>), grid, dim3(WT), 0, st, x, yh, (int)n, out, roi, seg, lg, spans)
#define TNL_FWD_WALK2(HALF, FB) \
  hipLaunchKernelGGL((k_idwt_fwd_walk2<W, HALF, FB>), grid, dim3(PT), 0, st, x, yh, (int)n, out, roi, seg, lg, spans)
    // pair form: even coarse origins and widths (true of every window TrainStep builds: 64-texel alignment)
    bool pair = g_fwd_pair != 0 && cw % 2 == 0;
    if (roi.rw) for (int p = 0; p < 3; p++) pair = pair && roi.ox[p] % 4 == 0;
    if (pair) {
      if (half_out) { if (g_fwd_pair == 2) TNL_FWD_WALK2(true, 2); else TNL_FWD_WALK2(true, 4); }
      else { if (g_fwd_pair == 2) TNL_FWD_WALK2(false, 2); else TNL_FWD_WALK2(false, 4); }
    } else if (half_out) { if (g_fwd_fb == 4) TNL_FWD_WALK(true, 4); else TNL_FWD_WALK(true, 8); }
    else { if (g_fwd_fb == 4) TNL_FWD_WALK(false, 4); else TNL_FWD_WALK(false, 8); }
#undef TNL_FWD_WALK
#undef TNL_FWD_WALK2
  } else if (n % 4 == 0) {
    const uint32_t ntx = roi.rw ? roi.rw / (2 * TI) : cdiv(n, TI), nty = roi.rw ? roi.rh / (2 * TI) : cdiv(n, TI);
    const int tpw = pick_tpw(ntx, nty, S);
    const dim3 grid(cdiv(ntx, tpw), nty, S);
    if (half_out)
      hipLaunchKernelGGL((k_idwt_fwd_pipe<W, true>), grid, dim3(NT), 0, st, x, yh, (int)n, out, roi, tpw);
    else
      hipLaunchKernelGGL((k_idwt_fwd_pipe<W, false>), grid, dim3(NT), 0, st, x, yh, (int)n, out, roi, tpw);
  } else {
    if (roi.rw) return (int)hipErrorInvalidValue;
    if (half_out) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_idwt_fwd<W>, dim3(cdiv(n, TI), cdiv(n, TI), S), dim3(NT), 0, st, x, yh, (int)n,
                       reinterpret_cast<float*>(out));
  }
  return (int)hipGetLastError();
}
template <int W>
int launch_bwd(const float* dout, uint32_t S, uint32_t n, float* dx, float* dyh, hipStream_t st, Roi roi = Roi{},
               int32_t* out_rect = nullptr, const int* spans = nullptr) {
  Roi orect{};
  if (out_rect != nullptr) {
    // Tiles of coarse outputs the input window reaches (same test as the tile kernel's hits()), per plane, then grown
    // to a common size: inside the rectangle every element is written (computed or zero), outside nothing is.
    // The column-walk kernel writes any rectangle, so its levels use 8-wide granules instead of the tile kernel's
    // TI-wide tiles (a tighter rectangle: fewer zeros stored, a smaller live rectangle for the optimiser pass).
    const bool walk = n % 8 == 0 && (int)n >= g_walk_min_n;
    const int G = walk ? 8 : TI;
    if (n % 2 != 0 || n % G != 0) return (int)hipErrorInvalidValue;
    constexpr WTaps T = wtaps(W);
    constexpr int L = T.L, K = (L - 2) / 2, KA = (K + 3) / 4 * 4, SH = KA - K;
    const int FT = 2 * G + L - 2, FTA = (2 * G + L - 2 + SH + 3) / 4 * 4;
    const int nt = (int)n / G;
    int lo[2][3], hi[2][3];
    for (int p = 0; p < 3; p++) {
      const int wo[2] = {roi.rw ? roi.ox[p] : 0, roi.rw ? roi.oy[p] : 0};
      const int we[2] = {roi.rw ? roi.rw : 2 * (int)n, roi.rw ? roi.rh : 2 * (int)n};
      const int ka[2] = {KA, K}, ft[2] = {FTA, FT};
      for (int d = 0; d < 2; d++) {
        lo[d][p] = nt; hi[d][p] = -1;
        for (int t = 0; t < nt; t++) {
          const int c0 = 2 * t * G - ka[d];
          if (c0 < wo[d] + we[d] && c0 + ft[d] > wo[d]) { lo[d][p] = t < lo[d][p] ? t : lo[d][p]; hi[d][p] = t; }
        }
        if (hi[d][p] < 0) { lo[d][p] = 0; hi[d][p] = 0; }
      }
    }
    int ext[2] = {0, 0};
    for (int d = 0; d < 2; d++)
      for (int p = 0; p < 3; p++) ext[d] = (hi[d][p] - lo[d][p] + 1) > ext[d] ? (hi[d][p] - lo[d][p] + 1) : ext[d];
    for (int p = 0; p < 3; p++) {
      orect.ox[p] = G * (lo[0][p] + ext[0] > nt ? nt - ext[0] : lo[0][p]);
      orect.oy[p] = G * (lo[1][p] + ext[1] > nt ? nt - ext[1] : lo[1][p]);
      out_rect[p] = orect.ox[p]; out_rect[3 + p] = orect.oy[p];
    }
    orect.rw = G * ext[0]; orect.rh = G * ext[1];
    out_rect[6] = orect.rw; out_rect[7] = orect.rh;
    orect.spp = roi.rw ? roi.spp : (int)(S / 3);
    orect.s0 = roi.s0;
  }
  if (n % 8 == 0 && (int)n >= g_walk_min_n) {
    const uint32_t ow = orect.rw ? orect.rw : n, oh = orect.rw ? orect.rh : n;
    const uint32_t tiles = cdiv(ow, AVC);
    const int seg = g_seg ? g_seg : pick_seg(tiles, oh, S);
    dim3 grid;
    LGrid lg;
    walk_grid(tiles, cdiv(oh, seg), S, grid, lg);
    if (spans != nullptr && (out_rect == nullptr || dx == nullptr)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((k_idwt_bwd_walk<W, false>), grid, dim3(AT), 0, st, dout, (int)n, dx, dyh, roi, orect, seg, lg, spans,
                       WalkAdam{});
  } else if (n % 2 == 0) {
    const int tpw = pick_tpw(cdiv(n, TI), cdiv(n, TI), S);
    hipLaunchKernelGGL((k_idwt_bwd_pipe<W, false>), dim3(cdiv(cdiv(n, TI), tpw), cdiv(n, TI), S), dim3(NT), 0, st, dout,
                       (int)n, dx, dyh, FuseAdam{}, roi, orect, tpw);
  } else if (roi.rw)
    return (int)hipErrorInvalidValue;
  else
    hipLaunchKernelGGL(k_idwt_bwd<W>, dim3(cdiv(n, TI), cdiv(n, TI), S), dim3(NT), 0, st, dout, (int)n, dx, dyh);
  return (int)hipGetLastError();
}

// the walk kernel with the optimiser in its epilogue (k_idwt_bwd_walk<W, true>): the launch covers the live rectangle
template <int W>
int launch_bwd_fused(const float* dout, uint32_t S, uint32_t n, float* dx, hipStream_t st, Roi roi, const int32_t* live,
                     const int* spans, const WalkAdam& wa) {
  Roi orect{};
  for (int p = 0; p < 3; p++) { orect.ox[p] = live[p]; orect.oy[p] = live[3 + p]; }
  orect.rw = live[6]; orect.rh = live[7];
  orect.spp = roi.spp; orect.s0 = roi.s0;
  const uint32_t tiles = cdiv(orect.rw, AVC);
  int seg = g_seg ? g_seg : pick_seg(tiles, orect.rh, S);
  if (seg > 256) seg = 256;        // the kernel stages one live piece per 8-row band of a segment: 32 of them
  dim3 grid;
  LGrid lg;
  walk_grid(tiles, cdiv(orect.rh, seg), S, grid, lg);
  hipLaunchKernelGGL((k_idwt_bwd_walk<W, true>), grid, dim3(AT), 0, st, dout, (int)n, dx, (float*)nullptr, roi, orect, seg, lg,
                     spans, wa);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int tnl_idwt_get_walk_min_n(void) { return g_walk_min_n; }

int tnl_idwt_set_walk_min_n(uint32_t n) {
  g_walk_min_n = n == 0 ? 512 : (int)n;
  return 0;
}

int tnl_idwt_set_tuning(int key, int value) {
  switch (key) {
    case 1: if (value != 4 && value != 8) return (int)hipErrorInvalidValue; g_fwd_fb = value; return 0;
    case 2: g_xcd = value != 0; return 0;
    case 3: if (value < 0 || value % 8) return (int)hipErrorInvalidValue; g_seg = value; return 0;
    case 4:   // (-1: the build's default)
      if (value == -1) value = TNL_FWD_PAIR;
      if (value != 0 && value != 2 && value != 4) return (int)hipErrorInvalidValue;
      g_fwd_pair = value; return 0;
    default: return (int)hipErrorInvalidValue;
  }
}

static int idwt_forward_any(const float* x, const float* yh, uint32_t S, uint32_t n, int wave, void* out,
                            int half_out, void* stream, const int32_t* roi_host = nullptr, int strided = 0,
                            const int32_t* spans = nullptr) {
  Roi roi;
  if (!make_roi(roi_host, S, 2 * n, roi)) return (int)hipErrorInvalidValue;
  roi.strided = (roi.rw && strided) ? 1 : 0;
  if (S == 0 || n == 0) return 0;
  if (S > 65535) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  switch (wave) {
    case 0: return launch_fwd<0>(x, yh, S, n, out, half_out, st, roi, spans);
    case 1: return launch_fwd<1>(x, yh, S, n, out, half_out, st, roi, spans);
    case 2: return launch_fwd<2>(x, yh, S, n, out, half_out, st, roi, spans);
    case 3: return launch_fwd<3>(x, yh, S, n, out, half_out, st, roi, spans);
    case 4: return launch_fwd<4>(x, yh, S, n, out, half_out, st, roi, spans);
    default: return (int)hipErrorInvalidValue;
  }
}

int tnl_idwt_level_forward(const float* x, const float* yh, uint32_t S, uint32_t n, int wave, float* out,
                           void* stream) {
  return idwt_forward_any(x, yh, S, n, wave, out, 0, stream);
}

int tnl_idwt_level_forward_half(const float* x, const float* yh, uint32_t S, uint32_t n, int wave, void* out_half,
                                void* stream) {
  if (n % 4 != 0) return (int)hipErrorInvalidValue;
  return idwt_forward_any(x, yh, S, n, wave, out_half, 1, stream);
}

int tnl_idwt_level_forward_half_roi(const float* x, const float* yh, uint32_t S, uint32_t n, int wave, void* out_half,
                                    const int32_t* roi, void* stream) {
  if (n % 4 != 0) return (int)hipErrorInvalidValue;
  return idwt_forward_any(x, yh, S, n, wave, out_half, 1, stream, roi);
}

int tnl_idwt_level_forward_win(const float* x, const float* yh, uint32_t S, uint32_t n, int wave, float* out,
                               const int32_t* win, void* stream) {
  if (n % 4 != 0) return (int)hipErrorInvalidValue;
  return idwt_forward_any(x, yh, S, n, wave, out, 0, stream, win, 1);
}

static int planes_half_to_tm(const void* planes_cm_half, uint32_t C, uint32_t R, void* planes_tm_half,
                             const int32_t* roi_host, void* stream, const int32_t* spans = nullptr);

// The three windowed calls above restricted to the pieces anything reads (`spans`: device tables as described at
// narrow_rows, on the level's own n x n grid; for the layout change on the R x R plane grid; NULL = no restriction).
// Forward: results outside the pieces are not produced (the output keeps what it held).  Adjoint: the band gradients
// outside the pieces are not produced, the LL gradient is zero there (it is exactly zero when the pieces hold every
// coefficient a data gradient can reach).  Levels that run the tile kernels (n < walk_min_n) ignore the spans.
int tnl_idwt_level_forward_spans(const float* x, const float* yh, uint32_t S, uint32_t n, int wave, void* out,
                                 int half_out, const int32_t* win, int strided, const int32_t* spans, void* stream) {
  if (n % 4 != 0 || win == nullptr) return (int)hipErrorInvalidValue;
  return idwt_forward_any(x, yh, S, n, wave, out, half_out, stream, win, strided, spans);
}

int tnl_planes_half_to_texel_major_spans(const void* planes_roi_half, uint32_t C, uint32_t R, void* planes_tm_half,
                                         const int32_t* roi, const int32_t* spans, void* stream) {
  return planes_half_to_tm(planes_roi_half, C, R, planes_tm_half, roi, stream, spans);
}

int tnl_planes_half_to_texel_major_roi(const void* planes_roi_half, uint32_t C, uint32_t R, void* planes_tm_half,
                                       const int32_t* roi, void* stream) {
  return planes_half_to_tm(planes_roi_half, C, R, planes_tm_half, roi, stream);
}

int tnl_planes_half_to_texel_major(const void* planes_cm_half, uint32_t C, uint32_t R, void* planes_tm_half,
                                   void* stream) {
  return planes_half_to_tm(planes_cm_half, C, R, planes_tm_half, nullptr, stream);
}

static int planes_half_to_tm(const void* planes_cm_half, uint32_t C, uint32_t R, void* planes_tm_half,
                             const int32_t* roi_host, void* stream, const int32_t* spans) {
  if (C == 0 || R == 0) return 0;
  if (C % 8 != 0 || R % 8 != 0) return (int)hipErrorInvalidValue;
  Roi roi;
  if (!make_roi(roi_host, 3 * C, R, roi) || (roi.rw && (roi.spp != (int)C || roi.s0 != 0)))
    return (int)hipErrorInvalidValue;
  // (R % 8 == 0 and window rows in multiples of 64: TY = 4 rows never straddle an 8-row group of the spans)
  constexpr int TY = TNL_LAYOUT_ROWS;
  static_assert(TY == 1 || TY == 2 || TY == 4 || TY == 8, "rows per workgroup must divide the spans' 8-row groups");
  const uint32_t rows = roi.rw ? (uint32_t)roi.rh : R;
  const dim3 grid = roi.rw ? dim3(roi.rw / TX, cdiv(rows, TY), 3) : dim3(cdiv(R, TX), cdiv(rows, TY), 3);
  const size_t lds = (size_t)TY * ((size_t)C * (TX + 8) + (C / 8) * 8) * sizeof(_Float16);
  hipLaunchKernelGGL(k_to_texel_major_h<TY>, grid, dim3(NT), lds, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(planes_cm_half), (int)C, (int)R,
                     reinterpret_cast<_Float16*>(planes_tm_half), roi, roi.rw ? spans : nullptr);
  return (int)hipGetLastError();
}

static int idwt_backward_any(const float* dout, uint32_t S, uint32_t n, int wave, float* dx, float* dyh,
                             const int32_t* roi_host, void* stream, int strided = 0, int32_t* out_rect = nullptr,
                             const int32_t* spans = nullptr);

int tnl_idwt_level_backward_spans(const float* dout, uint32_t S, uint32_t n, int wave, float* dx, float* dyh,
                                  const int32_t* win, int strided, int32_t* out_rect, const int32_t* spans,
                                  void* stream) {
  if (win == nullptr) return (int)hipErrorInvalidValue;
  if (!(n % 8 == 0 && (int)n >= g_walk_min_n)) spans = nullptr;       // tile kernels: everything is produced
  return idwt_backward_any(dout, S, n, wave, dx, dyh, win, stream, strided, out_rect, spans);
}

int tnl_idwt_level_backward(const float* dout, uint32_t S, uint32_t n, int wave, float* dx, float* dyh,
                            void* stream) {
  return idwt_backward_any(dout, S, n, wave, dx, dyh, nullptr, stream);
}

int tnl_idwt_level_backward_roi(const float* dout_roi, uint32_t S, uint32_t n, int wave, float* dx, float* dyh,
                                const int32_t* roi, void* stream) {
  return idwt_backward_any(dout_roi, S, n, wave, dx, dyh, roi, stream);
}

int tnl_idwt_level_backward_win(const float* dout, uint32_t S, uint32_t n, int wave, float* dx, float* dyh,
                                const int32_t* win, int strided, int32_t* out_rect, void* stream) {
  return idwt_backward_any(dout, S, n, wave, dx, dyh, win, stream, strided, out_rect);
}

static int idwt_backward_any(const float* dout, uint32_t S, uint32_t n, int wave, float* dx, float* dyh,
                             const int32_t* roi_host, void* stream, int strided, int32_t* out_rect,
                             const int32_t* spans) {
  if (S == 0 || n == 0) return 0;
  if (S > 65535) return (int)hipErrorInvalidValue;
  Roi roi;
  if (!make_roi(roi_host, S, 2 * n, roi, strided ? 4 : 64)) return (int)hipErrorInvalidValue;
  roi.strided = (roi.rw && strided) ? 1 : 0;
  if (out_rect != nullptr && roi_host == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  switch (wave) {
    case 0: return launch_bwd<0>(dout, S, n, dx, dyh, st, roi, out_rect, spans);
    case 1: return launch_bwd<1>(dout, S, n, dx, dyh, st, roi, out_rect, spans);
    case 2: return launch_bwd<2>(dout, S, n, dx, dyh, st, roi, out_rect, spans);
    case 3: return launch_bwd<3>(dout, S, n, dx, dyh, st, roi, out_rect, spans);
    case 4: return launch_bwd<4>(dout, S, n, dx, dyh, st, roi, out_rect, spans);
    default: return (int)hipErrorInvalidValue;
  }
}

int tnl_idwt_level_backward_live_adam(const float* dout, uint32_t S, uint32_t n, int wave, float* dx, const int32_t* win,
                                      int strided, const int32_t* live_rect, const int32_t* spans, float* p, float* m,
                                      float* v, const int32_t* band_table, uint32_t nb, const float* step_rec, float beta1,
                                      float beta2, float eps, float inv_scale, const float* inv_scale_dev, float l1_coef,
                                      const float* found_inf, float* abs_sum, void* stream) {
  if (S == 0 || n == 0) return 0;
  if (S > 65535 || win == nullptr || live_rect == nullptr || dx == nullptr || step_rec == nullptr) return (int)hipErrorInvalidValue;
  if (!(n % 8 == 0 && (int)n >= g_walk_min_n)) return (int)hipErrorInvalidValue;      // a column-walk level
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(dx)) & 15)
    return (int)hipErrorInvalidValue;                                                 // 16-byte quads
  if ((band_table == nullptr) != (spans == nullptr) || (band_table != nullptr && (nb == 0 || nb > 128))) return (int)hipErrorInvalidValue;
  for (int q = 0; q < 3; q++)        // 16-byte quads of p, m, v; 8-row bands
    if (live_rect[q] % 4 || live_rect[3 + q] % 8 || live_rect[q] < 0 || live_rect[3 + q] < 0 ||
        live_rect[q] + live_rect[6] > (int)n || live_rect[3 + q] + live_rect[7] > (int)n)
      return (int)hipErrorInvalidValue;
  if (live_rect[6] <= 0 || live_rect[7] <= 0 || live_rect[6] % 4 || live_rect[7] % 8) return (int)hipErrorInvalidValue;
  if (band_table != nullptr && (int)nb != live_rect[7] / 8) return (int)hipErrorInvalidValue;
  Roi roi;
  if (!make_roi(win, S, 2 * n, roi, strided ? 4 : 64)) return (int)hipErrorInvalidValue;
  roi.strided = (roi.rw && strided) ? 1 : 0;
  if (!roi.rw) return (int)hipErrorInvalidValue;
  // step_size / bias2_sqrt come from the record on the device
  WalkAdam wa{p, m, v, make_adam_args(0.f, 1.f, beta1, beta2, eps, inv_scale, l1_coef),
              reinterpret_cast<const AdamStepRec*>(step_rec), inv_scale_dev, found_inf, abs_sum, band_table, (int)nb};
  hipStream_t st = (hipStream_t)stream;
  switch (wave) {
    case 0: return launch_bwd_fused<0>(dout, S, n, dx, st, roi, live_rect, spans, wa);
    case 1: return launch_bwd_fused<1>(dout, S, n, dx, st, roi, live_rect, spans, wa);
    case 2: return launch_bwd_fused<2>(dout, S, n, dx, st, roi, live_rect, spans, wa);
    case 3: return launch_bwd_fused<3>(dout, S, n, dx, st, roi, live_rect, spans, wa);
    case 4: return launch_bwd_fused<4>(dout, S, n, dx, st, roi, live_rect, spans, wa);
    default: return (int)hipErrorInvalidValue;
  }
}

int tnl_idwt_level_backward_adam(const float* dout, uint32_t S, uint32_t n, int wave, float* dx, float* p, float* m,
                                 float* v, float* ll_p, float* ll_m, float* ll_v, float step_size, float bias2_sqrt,
                                 float beta1, float beta2, float eps, float inv_scale, const float* inv_scale_dev,
                                 float l1_coef, const float* found_inf, float* abs_sum, void* stream) {
  if (S == 0 || n == 0) return 0;
  if (S > 65535 || n % 2 != 0 || (dx == nullptr && ll_p == nullptr)) return (int)hipErrorInvalidValue;
  FuseAdam fa{p, m, v, ll_p, ll_m, ll_v, make_adam_args(step_size, bias2_sqrt, beta1, beta2, eps, inv_scale, l1_coef),
              inv_scale_dev, found_inf, abs_sum};
  const int tpw = pick_tpw(cdiv(n, TI), cdiv(n, TI), S);
  const dim3 grid(cdiv(cdiv(n, TI), tpw), cdiv(n, TI), S);
  hipStream_t st = (hipStream_t)stream;
#define TNL_BWD_ADAM(WW) \
  hipLaunchKernelGGL((k_idwt_bwd_pipe<WW, true>), grid, dim3(NT), 0, st, dout, (int)n, dx, (float*)nullptr, fa, Roi{}, Roi{}, tpw)
  switch (wave) {
    case 0: TNL_BWD_ADAM(0); break;
    case 1: TNL_BWD_ADAM(1); break;
    case 2: TNL_BWD_ADAM(2); break;
    case 3: TNL_BWD_ADAM(3); break;
    case 4: TNL_BWD_ADAM(4); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef TNL_BWD_ADAM
  return (int)hipGetLastError();
}

static int planes_to_tm(const float* planes_cm, uint32_t C, uint32_t R, int half_out, void* planes_tm,
                        const int32_t* roi_host, void* stream) {
  if (C == 0 || R == 0) return 0;
  Roi roi;
  if (!make_roi(roi_host, 3 * C, R, roi) || (roi.rw && (roi.spp != (int)C || roi.s0 != 0 || roi.rw % TX != 0)))
    return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(planes_cm) | reinterpret_cast<uintptr_t>(planes_tm)) & 15) return (int)hipErrorInvalidValue;
  const dim3 grid = roi.rw ? dim3(roi.rw / TX, roi.rh, 3) : dim3(cdiv(R, TX), R, 3);
  const size_t lds = (size_t)C * (TX + 1) * sizeof(float);
  if (half_out)
    hipLaunchKernelGGL(k_to_texel_major<true>, grid, dim3(NT), lds, (hipStream_t)stream, planes_cm, (int)C, (int)R,
                       planes_tm, roi);
  else
    hipLaunchKernelGGL(k_to_texel_major<false>, grid, dim3(NT), lds, (hipStream_t)stream, planes_cm, (int)C, (int)R,
                       planes_tm, roi);
  return (int)hipGetLastError();
}

int tnl_planes_to_texel_major(const float* planes_cm, uint32_t C, uint32_t R, int half_out, void* planes_tm,
                              void* stream) {
  return planes_to_tm(planes_cm, C, R, half_out, planes_tm, nullptr, stream);
}

int tnl_planes_to_texel_major_win(const float* planes_cm, uint32_t C, uint32_t R, int half_out, void* planes_tm,
                                  const int32_t* roi, void* stream) {
  if (roi == nullptr) return (int)hipErrorInvalidValue;
  return planes_to_tm(planes_cm, C, R, half_out, planes_tm, roi, stream);
}

int tnl_planes_to_channel_major(const float* grad_tm, uint32_t C, uint32_t R, float* grad_cm, void* stream) {
  if (C == 0 || R == 0) return 0;
  const dim3 grid(cdiv(R, TX), R, 3);
  const size_t lds = (size_t)C * (TX + 1) * sizeof(float);
  hipLaunchKernelGGL(k_to_channel_major, grid, dim3(NT), lds, (hipStream_t)stream, grad_tm, (int)C, (int)R, grad_cm);
  return (int)hipGetLastError();
}

}  // extern "C"
