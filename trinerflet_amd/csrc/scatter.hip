// scatter.hip -- plane-gradient accumulation without global float atomics (gfx950).
//
// Replaces torch's grid_sampler_2d_backward (atomicAdd of 12*C floats per sample into the (3,C,R,R)
// gradient, which autograd zero-fills first) for the training step.  The memory-side fp32 atomic unit of
// MI355X adds ~1.3 TB/s chip-wide (MI355X_MICROARCH.md "Global float atomics"); at 1536 B of adds per
// sample that alone was 6 ms per step.  Here instead:
//   1. the field backward writes the feature gradient dF as fp16 [M, 3C] (the precision the reference's
//      autocast Linear backward hands to grid_sample's backward as well),
//   2. samples are counting-sorted by the 32x8-texel tile their bilinear footprint touches, per plane
//      (a footprint that straddles tiles is listed in each of them),
//   3. one workgroup per tile reduces its samples on the matrix cores (separable bilinear weights x dF, fp32
//      accumulators in registers) and writes the finished tile with plain 16-byte stores -- every tile is
//      written exactly once, so the 4*P-byte zero fill of the gradient disappears too.
// HBM traffic per sample: 12 B xyz (the fill pass) + ~3.4 list entries x 12 B x2 (written by the fill pass, read by the
// reduction: sample id + its clipped texel coordinates on the list's plane -- round 3; before, the reduction gathered
// xyz[id], a 12-byte read that costs a sector, i.e. as many requests as the dF row: 0.40 -> 0.35 ms at base) + 6*C B of
// dF, versus 48*C B of atomics; per step additionally the 4*P-byte tile stores that replace the memset (only the ROI's
// with a ROI).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#ifndef TNL_MAIN_PRIO
#define TNL_MAIN_PRIO 0   // A/B builds: static wave priority (s_setprio) of the step's kernels that run beside the side chain
#endif
#define TNL_SET_MAIN_PRIO() do { if (TNL_MAIN_PRIO) __builtin_amdgcn_s_setprio(TNL_MAIN_PRIO); } while (0)
// Static wave priority (s_setprio) of the tile sort passes: on the side stream they share their SIMDs with the step's
// HBM-bound kernels and are short dependent chains -- served first they finish sooner and cost those kernels nothing
// measurable: small 1.948 -> 1.915 ms per step (the forward no longer runs beside the fill pass: 0.40 -> 0.34), base 3.748 ->
// 3.724, large equal (profiles/r06p_ab_wave_priority.txt; the reverse -- priority for the step's kernels -- speeds them by
// 0.1 ms and makes the side chain the critical path: small + 0.11 ms).  0 = none (A/B builds, tools/knob_ci.sh).
#ifndef TNL_SIDE_PRIO
#define TNL_SIDE_PRIO 3
#endif

#include <algorithm>
#include <mutex>
#include <stdlib.h>

#include "../../include/trinerflet_hip.h"
#include "roi_common.h"
#include "bin_common.h"
#include "triplane_common.h"

namespace {

constexpr int NT = 256;
thread_local int g_fill_cap = 0;   // tnl_plane_grad_fill_cap: most workgroups of the sort's fill pass (0 = one thread per sample all at once)

__device__ __forceinline__ uint32_t eff_m(uint32_t M, const int32_t* m_actual) {
  return m_actual ? min(M, (uint32_t)max(*m_actual, 0)) : M;
}

// pass 1 / pass 3: FILL=false counts entries per bin, FILL=true writes sample ids at cursor positions (bin_common.h)
template <bool FILL>
__global__ void __launch_bounds__(NT)
k_bin(const float* __restrict__ xyz, float bound, uint32_t M, const int32_t* __restrict__ m_actual, int R, int TNX,
      int TNY, int* __restrict__ counts_or_cursor, uint32_t* __restrict__ entries, float2* __restrict__ epos) {
  if (TNL_SIDE_PRIO) __builtin_amdgcn_s_setprio(TNL_SIDE_PRIO);
  const uint32_t Me = eff_m(M, m_actual);
  // (a launch of fewer workgroups than M / NT walks the samples with the grid's stride; uniform trip count per wave)
  for (uint32_t i0 = blockIdx.x * NT; i0 < M; i0 += gridDim.x * NT) {
    const uint32_t i = i0 + threadIdx.x;
    const bool live = i < Me;
    const uint32_t il = live ? i : 0;
    bin_sample<FILL>(xyz[(size_t)il * 3], xyz[(size_t)il * 3 + 1], xyz[(size_t)il * 3 + 2], live, i, bound, R, TNX, TNY,
                     counts_or_cursor, entries, threadIdx.x & 63, epos);
  }
}

// exclusive scan of the bin counts; also leaves a copy as the fill cursors.  Two tiny launches: (1) each
// workgroup scans 1024 consecutive counts (coalesced) and publishes its total; (2) every workgroup adds the totals
// of the workgroups before it (<= 48 values at R = 2048).  A single-workgroup version took 107 us of pure latency.
// (256 threads x 4 counts per workgroup since round 3: a 1024-thread workgroup needs a whole CU's worth of free wave
//  slots, and on the side stream, beside the Adam pass's 4096 workgroups, the 5-us kernel waited ~190 us for them --
//  profiles/r03b_step_timeline.txt -- holding back the fill pass behind it.  The step time did not move: 4.26-4.31 vs
//  4.22-4.31 ms over four alternating runs.)
__global__ void __launch_bounds__(256)
k_scan_local(const int* __restrict__ counts, int nb, int* __restrict__ offsets, int* __restrict__ block_tot) {
  __shared__ int wsum[4];
  const int i0 = blockIdx.x * 1024 + 4 * threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = i0 + k < nb ? counts[i0 + k] : 0;
  const int mine = v[0] + v[1] + v[2] + v[3];
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(incl, off);
    if (lane >= off) incl += u;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int b = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int w = wsum[k];
    if (k < wv) b += w;
    tot += w;
  }
  int run = b + incl - mine;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (i0 + k < nb) offsets[i0 + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(256)
k_scan_fix(int nb, int nblk, const int* __restrict__ block_tot, int* __restrict__ offsets, int* __restrict__ cursor) {
  __shared__ int s_base;
  if (threadIdx.x < 64) {
    int part = 0;
    for (int k = threadIdx.x; k < (int)blockIdx.x; k += 64) part += block_tot[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (threadIdx.x == 0) s_base = part;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = blockIdx.x * 1024 + 4 * threadIdx.x + k;
    if (i < nb) {
      const int o = offsets[i] + s_base;
      offsets[i] = o;
      cursor[i] = o;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) offsets[nb] = s_base + block_tot[nblk - 1];
}

// pass 4: one workgroup per (plane, tile of 32 x 8 texels): accumulate the tile's samples, store the tile.
//
// The accumulation  acc[(y,x), c] += sum_q rw_q[y] * cw_q[x] * g_q[c]   (bilinear weights are separable:
// rw = {1-wy, wy} on rows y0,y1 ; cw = {1-wx, wx} on columns x0,x1) is a matrix product over the records q, so
// it runs on the matrix cores instead of LDS atomics (ds_add_f32 measured ~150 cycles per wave-instruction here:
// 9 ms per step) or sort-and-reduce passes (6 barriers per 256 records, LDS-latency-bound: 1.7 ms):
//   per tile row y (one 32-texel block):  D_y[x, c] += A_y[x, q] * G[q, c],   A_y[x, q] = fp16(cw_q[x] * rw_q[y])
// with v_mfma_f32_32x32x16_f16, fp32 accumulators in registers for the whole tile lifetime (wave w owns rows
// 2w, 2w+1).  Records are consumed 256 at a time: phase A (one thread per record; the next chunk's id, xyz and
// fp16 dF slice are prefetched into registers during phase B) computes the tap once and stages dF
// (the image gI, read back through ds_read_b64_tr_b16), the row weights (rwT[y][q]) and the column taps; phase B is
// 16 k-steps x 2 rows of MFMA per wave.  Two barriers per chunk, no atomics, deterministic summation order.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef float f16acc __attribute__((ext_vector_type(16)));

typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;

// The dF stage is an image [32-channel block][record][32 channels] with 64-byte rows whose eight 8-byte chunks are
// XOR-swizzled with the row (the layout of field_bwd.hip's stages): a record's thread stores its channels with
// ds_write_b64, and the MFMA B operand -- 8 records of one channel per lane -- is read with the transposing
// ds_read_b64_tr_b16, conflict-free both ways.  ([channel][record] rows took one ds_write_b16 per channel and record.)
__device__ __forceinline__ int img_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ h4v tr4(const char* p) {
  const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
      (__attribute__((address_space(3))) fp16x4_t*)(const_cast<char*>(p)));
  return __builtin_bit_cast(h4v, v);
}

template <int C>
__global__ void __launch_bounds__(NT)
k_tile_accumulate(const _Float16* __restrict__ dfeat, uint32_t Mcap, const float* __restrict__ xyz, float bound, int R, int TNX,
                  int TNY, const int* __restrict__ offsets, const uint32_t* __restrict__ entries,
                  const float2* __restrict__ epos, float grad_scale,
                  float* __restrict__ grad_out, int layout, int* __restrict__ nonfinite_flag, Roi roi) {
  TNL_SET_MAIN_PRIO();
  // layout: bit 0 = channel-major (3,C,R,R) output; bit 1 = the caller zero-filled the output (one contiguous fill):
  // untouched tiles are then skipped instead of being zeroed here in 128-byte row pieces; bit 2 (with a ROI) = the output
  // is the whole (3,C,R,R) array, of which only the window is written
  const int channel_major = layout & 1;
  constexpr int NTEX = TSX * TSY;
  constexpr int TILE_F = NTEX * C;
    constexpr int NB = (C + 31) / 32;          // 32-channel column blocks
  constexpr int QS = NT + 8;                 // record stride (halfs) of the transposed stages: 16-B aligned rows
  constexpr int XS = TSX + 4;                // epilogue staging stride (floats)
  constexpr int GBLK = NT * 64;              // bytes of one 32-channel block of the dF image
  constexpr size_t STAGE_A = (size_t)NB * GBLK + (size_t)(TSY + TSX) * QS * 2;
  constexpr size_t STAGE_E = (size_t)4 * 2 * 32 * NB * XS * 4;
  constexpr size_t LDS_BYTES = STAGE_A > STAGE_E ? STAGE_A : STAGE_E;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  char* gI = smem;                                                  // [NB][NT][32]  dF image (see img_off)
  _Float16* rwT = reinterpret_cast<_Float16*>(smem + (size_t)NB * GBLK);   // [TSY][QS]  row weights [tile row][record]
  // column weights [tile column][record]: a record has at most two non-zero entries (columns x0, x0+1 if inside the
  // tile), written once by the record's thread in phase A; every wave then reads its A-fragment factor with one
  // 16-byte LDS load instead of rebuilding it from the taps (that rebuild was ~40 VALU instructions per k-step,
  // repeated by all four waves, and made phase B VALU-bound)
  _Float16* cwT = rwT + (size_t)TSY * QS;                           // [TSX][QS]

  // With a ROI only its tiles are launched (samples binned elsewhere are dropped: the caller guarantees there are
  // none) and the output is the compact channel-major window [3C][rh][rw].
  int p, ty, tx;
  if (roi.rw) {
    const int rnx = roi.rw / TSX, rny = roi.rh / TSY;
    p = blockIdx.x / (rnx * rny);
    const int rem = blockIdx.x - p * rnx * rny;
    ty = rem / rnx + roi.oy[p] / TSY;
    tx = rem - (rem / rnx) * rnx + roi.ox[p] / TSX;
  } else {
    p = blockIdx.x / (TNX * TNY);
    const int rem = blockIdx.x - p * TNX * TNY;
    ty = rem / TNX; tx = rem - ty * TNX;
  }
  const int bin = p * TNX * TNY + ty * TNX + tx;
  const int beg = offsets[bin * BIN_SUBS], end = offsets[(bin + 1) * BIN_SUBS];   // all sub-bins of the tile
  const int x_lo = tx * TSX, y_lo = ty * TSY;
  // channel-major output addressing: row stride, slice stride and origin of the (possibly compact) window
  // (layout bit 2: the window's tiles only, but written at their place in the whole (3,C,R,R) array)
  const bool compact = roi.rw != 0 && !(layout & 4);
  const int ow = compact ? roi.rw : R, oh = compact ? roi.rh : R;
  const int x_out = x_lo - (compact ? roi.ox[p] : 0), y_out = y_lo - (compact ? roi.oy[p] : 0);
  if (beg == end) {  // untouched tile: this store replaces the zero fill of the gradient
    if (layout & 2) return;
    if (!channel_major) {
      float* dst = grad_out + (((size_t)p * R + y_lo) * R + x_lo) * C;
      constexpr int ROW_F4 = TSX * C / 4;
      for (int q = threadIdx.x; q < TILE_F / 4; q += NT) {
        const int ry = q / ROW_F4, rq = q - ry * ROW_F4;
        reinterpret_cast<float4*>(dst + (size_t)ry * R * C)[rq] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      constexpr int X4 = TSX / 4;
      for (int q = threadIdx.x; q < C * TSY * X4; q += NT) {
        const int ch = q / (TSY * X4), r2 = q - ch * (TSY * X4);
        const int ry = r2 / X4, lx = (r2 - ry * X4) * 4;
        *reinterpret_cast<float4*>(grad_out + (((size_t)p * C + ch) * oh + y_out + ry) * ow + x_out + lx) =
            make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;

  f16acc acc[2][NB];
#pragma unroll
  for (int b = 0; b < 2; b++)
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
      for (int g = 0; g < 16; g++) acc[b][nb][g] = 0.f;
  // channels beyond C (C = 16 or 48: the padded half of the last block) stay zero for the whole kernel
  if (C % 32 != 0) {
    for (int q = threadIdx.x; q < GBLK / 16; q += NT)
      reinterpret_cast<float4*>(gI + (size_t)(NB - 1) * GBLK)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // transposed-read offsets: lane 4q + p of a 16-lane group addresses record row q of a 4-row block, chunk p of the
  // group's 16 channels
  const int tq = (lane & 15) >> 2, tc = 4 * ((lane >> 4) & 1) + (lane & 3);
  const int t0 = img_off(8 * h + tq, tc), t1 = img_off(8 * h + 4 + tq, tc);
  for (int q = threadIdx.x; q < TSX * QS / 8; q += NT) {
    h8v z;
#pragma unroll
    for (int j = 0; j < 8; j++) z[j] = (_Float16)0.f;
    reinterpret_cast<h8v*>(cwT)[q] = z;
  }
  int prev_c0 = -100;   // first of the (up to) two cwT rows this thread's previous record wrote

  // prefetch registers for one record: its texel coordinates on this plane (from the list) and its dF row
  float px = 0.f, py = 0.f;
  float2 npos = make_float2(0.f, 0.f);
  h8v pg[C / 8];
  bool pv = false;
  // two levels ahead: the record id of chunk t+2 is requested while the data of chunk t+1 (addressed by the id
  // fetched one trip earlier) is in flight -- id -> xyz / dF is a dependent pair of global latencies
  uint32_t nid = 0;
  bool nv = false;
  auto fetch_id = [&](int base) {
    nv = base + (int)threadIdx.x < end;
    if (nv) { nid = entries[base + threadIdx.x]; npos = epos[base + threadIdx.x]; }
  };
  auto prefetch = [&]() {   // data of the chunk whose ids are in (nid, nv)
    pv = nv;
    if (pv) {
      const uint32_t i = nid;
      px = npos.x; py = npos.y;
      const h8v* src = reinterpret_cast<const h8v*>(dfeat + ((size_t)p * Mcap + i) * C);   // plane-major [3][M][C]
#pragma unroll
      for (int k = 0; k < C / 8; k++) pg[k] = src[k];   // (non-temporal loads here: 0.80 -> 0.85 ms, not used)
    }
  };
  fetch_id(beg);
  prefetch();
  fetch_id(beg + NT);
  for (int base = beg; base < end; base += NT) {
    __syncthreads();  // previous chunk's fragments fully read
    // ---- A: one thread per record
    {
      const int q = threadIdx.x;
      // clear what this thread's previous record left in cwT
      if ((unsigned)prev_c0 < (unsigned)TSX) cwT[(size_t)prev_c0 * QS + q] = (_Float16)0.f;
      if ((unsigned)(prev_c0 + 1) < (unsigned)TSX) cwT[(size_t)(prev_c0 + 1) * QS + q] = (_Float16)0.f;
      prev_c0 = -100;
      int ly0 = -1, ly1 = -1;
      float wy0 = 0.f, wy1 = 0.f;
      if (pv) {
        TexelTap t;
        tap_from_texel(px, py, R, t);
        const float wx = t.w01 + t.w11, wy = t.w10 + t.w11;  // weights are (1-wx|wx) x (1-wy|wy)
        const int lx0 = t.x0 - x_lo, lx1 = t.x1 - x_lo;
        ly0 = t.y0 - y_lo; ly1 = (t.y1 != t.y0) ? t.y1 - y_lo : -1;
        if ((unsigned)lx0 < (unsigned)TSX) cwT[(size_t)lx0 * QS + q] = (_Float16)(1.f - wx);
        if ((unsigned)lx1 < (unsigned)TSX && t.x1 != t.x0) cwT[(size_t)lx1 * QS + q] = (_Float16)wx;
        prev_c0 = lx0;   // lx1 is lx0 + 1 whenever it is written
        wy0 = (1.f - wy) * grad_scale; wy1 = wy * grad_scale;
#pragma unroll
        for (int k = 0; k < C / 8; k++) {
          char* blk = gI + (size_t)((8 * k) / 32) * GBLK;
          const int c0 = ((8 * k) % 32) / 4;
          *reinterpret_cast<h4v*>(blk + img_off(q, c0)) = __builtin_shufflevector(pg[k], pg[k], 0, 1, 2, 3);
          *reinterpret_cast<h4v*>(blk + img_off(q, c0 + 1)) = __builtin_shufflevector(pg[k], pg[k], 4, 5, 6, 7);
        }
      } else {
        const h4v z = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < C / 4; k++)   // 0 * stale NaN would poison the MFMA
          *reinterpret_cast<h4v*>(gI + (size_t)((4 * k) / 32) * GBLK + img_off(q, ((4 * k) % 32) / 4)) = z;
      }
#pragma unroll
      for (int y = 0; y < TSY; y++) rwT[(size_t)y * QS + q] = (_Float16)((y == ly0 ? wy0 : 0.f) + (y == ly1 ? wy1 : 0.f));
    }
    prefetch();               // chunk base + NT (pv false past the end)
    fetch_id(base + 2 * NT);
    __syncthreads();
    // ---- B: matrix-core accumulation, wave wv owns tile rows 2wv and 2wv+1
    const int nks = (min(NT, end - base) + 15) / 16;
    for (int ks = 0; ks < nks; ks++) {
      const int q0 = 16 * ks + 8 * h;
      const h8v cw = *reinterpret_cast<const h8v*>(cwT + (size_t)r * QS + q0);   // column r, records q0..q0+7
      h8v bf[NB];
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
        bf[nb] = __builtin_shufflevector(tr4(gI + (size_t)nb * GBLK + t0 + 1024 * ks),
                                         tr4(gI + (size_t)nb * GBLK + t1 + 1024 * ks), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const h8v rw = *reinterpret_cast<const h8v*>(rwT + (size_t)(2 * wv + b) * QS + q0);
        // (skipping the row when no record of the k-step has weight in it -- a wave-uniform ballot -- was measured
        //  slower: 0.54 -> 0.70 ms; the MFMAs are cheaper than the test)
        const h8v af = cw * rw;  // packed fp16 products (v_pk_mul_f16)
#pragma unroll
        for (int nb = 0; nb < NB; nb++)
          acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nb], acc[b][nb], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: D[x = acc_row(g,h)][channel = 32nb + r] of rows 2wv, 2wv+1
  if (nonfinite_flag != nullptr) {
    float chk = 0.f;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int g = 0; g < 16; g++) chk += acc[b][nb][g] * 0.f;  // 0 unless some value is inf/nan
    if (chk != 0.f || chk != chk) *nonfinite_flag = 1;
  }
  if (!channel_major) {
#pragma unroll
    for (int b = 0; b < 2; b++) {
      float* dst = grad_out + (((size_t)p * R + y_lo + 2 * wv + b) * R + x_lo) * C;
#pragma unroll
      for (int nb = 0; nb < NB; nb++) {
        const int c = 32 * nb + r;
        if (c < C) {
#pragma unroll
          for (int g = 0; g < 16; g++) dst[(size_t)((g & 3) + 8 * (g >> 2) + 4 * h) * C + c] = acc[b][nb][g];
        }
      }
    }
  } else {
    __syncthreads();  // the staging area aliases gI / rwT / cwT
    float* stg = reinterpret_cast<float*>(smem) + (size_t)wv * 2 * 32 * NB * XS;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int g = 0; g < 16; g++)
          stg[(size_t)((b * NB + nb) * 32 + r) * XS + (g & 3) + 8 * (g >> 2) + 4 * h] = acc[b][nb][g];
    // a wave reads back only what it wrote: no barrier, but LDS writes must land first
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    constexpr int X4 = TSX / 4;
    for (int q = lane; q < 2 * 32 * NB * X4; q += 64) {
      const int row = q / X4, lx = (q - row * X4) * 4;   // row = (b*NB + nb)*32 + channel-in-block
      const int b = row / (32 * NB), c = row - b * (32 * NB);
      if (c < C) {
        const float* a = stg + (size_t)row * XS + lx;
        *reinterpret_cast<float4*>(grad_out + (((size_t)p * C + c) * oh + y_out + 2 * wv + b) * ow + x_out + lx) =
            make_float4(a[0], a[1], a[2], a[3]);
      }
    }
  }
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace

extern "C" {

// bytes of scratch: counts, offsets(+1), cursor: one int per (plane, tile) each; entries: 12 * M uint32 + 12 * M float2
uint64_t tnl_plane_grad_binned_workspace(uint32_t M, uint32_t R) {
  if (R % TSX != 0) return 0;
  const uint64_t nb = 3ull * (R / TSX) * (R / TSY) * BIN_SUBS;
  return (3 * nb + 8 + (nb + 1023) / 1024 + 8 + 2) * 4 + 12ull * M * 4 + 12ull * M * 8;   // + (fx, fy) per list entry
}

int tnl_plane_grad_binned(const void* dfeat_half, const float* xyz, float bound, uint32_t M,
                          const int32_t* m_actual, uint32_t C, uint32_t R, float grad_scale, float* grad_out,
                          int channel_major, int32_t* nonfinite_flag, void* workspace, void* stream) {
  return tnl_plane_grad_binned_roi(dfeat_half, xyz, bound, M, m_actual, C, R, grad_scale, grad_out, channel_major,
                                   nonfinite_flag, nullptr, workspace, stream);
}

struct SortWs {
  int *counts, *offsets, *cursor, *block_tot;
  uint32_t* entries;
  float2* epos;       // per list entry the sample's clipped texel coordinates on the list's plane
  int nb, nblk;
};

static SortWs sort_ws(void* workspace, uint32_t R, uint32_t M = 0) {
  SortWs w;
  const int TNX = R / TSX, TNY = R / TSY;
  w.nb = 3 * TNX * TNY * BIN_SUBS;   // counters: BIN_SUBS sub-bins per (plane, tile), see bin_common.h
  w.nblk = (w.nb + 1023) / 1024;
  w.counts = reinterpret_cast<int*>(workspace);
  w.offsets = w.counts + w.nb + 1;
  w.cursor = w.offsets + w.nb + 1;
  w.block_tot = w.cursor + w.nb + 1;  // nblk ints, inside the slack before the entry list (see workspace())
  w.entries = reinterpret_cast<uint32_t*>(w.cursor + w.nb + 2 + w.nblk + 8);
  uint32_t* after = w.entries + 12ull * M;                      // 12 entries per sample at most (4 tiles x 3 planes)
  after += (reinterpret_cast<uintptr_t>(after) & 7) ? 1 : 0;    // 8-byte aligned
  w.epos = reinterpret_cast<float2*>(after);
  return w;
}

// Where the tile lists lie inside the workspace, in int32 units: out = {number of bins (sub-bins included), index of
// offsets[0] (nb + 1 entries, the last one = total entries), index of the first list entry, sub-bins per tile}.  For
// callers that post-process the lists -- TrainStep(deterministic=True) orders every tile's list by sample id, so that
// the reduction's summation order (and with it every bit of the plane gradient) no longer depends on the arrival order
// of the fill pass's atomics.
int tnl_plane_grad_sort_layout(uint32_t M, uint32_t R, int64_t* out) {
  if (R % TSX != 0 || out == nullptr) return (int)hipErrorInvalidValue;
  const SortWs w = sort_ws(nullptr, R, M);
  out[0] = w.nb;
  out[1] = w.offsets - w.counts;
  out[2] = reinterpret_cast<int*>(w.entries) - w.counts;
  out[3] = BIN_SUBS;
  out[4] = reinterpret_cast<int*>(w.epos) - w.counts;     // (fx, fy) float pairs, one per list entry
  return 0;
}

// The positions of the list entries lie behind the 12 * M entry slots, so sort, reduce and tnl_plane_grad_sort_layout must
// be given the SAME capacity M (and an 8-byte aligned workspace: the layout call computes the padding from a null base).
// A host-side note of what each workspace was last sorted with lets the reduce refuse a mismatch instead of reading
// positions from the wrong place (no device read-back; a small ring, newest first).
struct SortNote { const void* ws; uint32_t M, R; };
static SortNote g_sort_notes[32];
static unsigned g_sort_note_next = 0;
static std::mutex g_sort_note_mutex;

static void note_sort(const void* ws, uint32_t M, uint32_t R) {
  std::lock_guard<std::mutex> lock(g_sort_note_mutex);
  for (auto& n : g_sort_notes)
    if (n.ws == ws) { n.M = M; n.R = R; return; }
  g_sort_notes[g_sort_note_next++ % 32] = SortNote{ws, M, R};
}
// false only if the workspace is known and was sorted with another capacity / plane size
static bool sort_matches(const void* ws, uint32_t M, uint32_t R) {
  std::lock_guard<std::mutex> lock(g_sort_note_mutex);
  for (const auto& n : g_sort_notes)
    if (n.ws == ws) return n.M == M && n.R == R;
  return true;
}

// Part 1 (needs only the sample positions): counting sort of the samples by (plane, tile).  TrainStep runs it on
// the march's side stream, so it is off the critical path of the step.
static int plane_grad_sort_impl(const float* xyz, float bound, uint32_t M, const int32_t* m_actual, uint32_t R,
                                void* workspace, bool counted, void* stream) {
  if (R % TSX != 0 || (reinterpret_cast<uintptr_t>(workspace) & 7) != 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int TNX = R / TSX, TNY = R / TSY;
  const SortWs w = sort_ws(workspace, R, M);
  note_sort(workspace, M, R);
  if (!counted) {
    hipError_t e = hipMemsetAsync(w.counts, 0, (size_t)(w.nb + 1) * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (M > 0) {
      hipLaunchKernelGGL(k_bin<false>, dim3(cdiv(M, NT)), dim3(NT), 0, st, xyz, bound, M, m_actual, (int)R, TNX, TNY,
                         w.counts, w.entries, w.epos);
    }
  }
  hipLaunchKernelGGL(k_scan_local, dim3(w.nblk), dim3(256), 0, st, w.counts, w.nb, w.offsets, w.block_tot);
  hipLaunchKernelGGL(k_scan_fix, dim3(w.nblk), dim3(256), 0, st, w.nb, w.nblk, w.block_tot, w.offsets, w.cursor);
  if (M > 0) {
    // (tnl_plane_grad_fill_cap: a sort enqueued beside other kernels keeps to about a wave per SIMD)
    const uint32_t fill_blocks = g_fill_cap > 0 ? std::min<uint32_t>(cdiv(M, NT), (uint32_t)g_fill_cap) : cdiv(M, NT);
    hipLaunchKernelGGL(k_bin<true>, dim3(fill_blocks), dim3(NT), 0, st, xyz, bound, M, m_actual, (int)R, TNX, TNY,
                       w.cursor, w.entries, w.epos);
  }
  return (int)hipGetLastError();
}

int tnl_plane_grad_sort(const float* xyz, float bound, uint32_t M, const int32_t* m_actual, uint32_t R,
                        void* workspace, void* stream) {
  return plane_grad_sort_impl(xyz, bound, M, m_actual, R, workspace, false, stream);
}

// The same with the per-bin counts already in the workspace (tnl_march_rays_train_binned counted them while it wrote
// the samples): scan + fill only.
int tnl_plane_grad_fill_cap(int blocks) {
  const int prev = g_fill_cap;
  if (blocks >= 0) g_fill_cap = blocks;
  return prev;
}

int tnl_plane_grad_sort_counted(const float* xyz, float bound, uint32_t M, const int32_t* m_actual, uint32_t R,
                                void* workspace, void* stream) {
  return plane_grad_sort_impl(xyz, bound, M, m_actual, R, workspace, true, stream);
}

// Part 2: one workgroup per (plane, tile) reduces the tile's sorted samples on the matrix cores.
int tnl_plane_grad_reduce(const void* dfeat_half, const float* xyz, float bound, uint32_t M, uint32_t C, uint32_t R,
                          float grad_scale, float* grad_out, int channel_major, int32_t* nonfinite_flag,
                          const int32_t* roi_host, const void* workspace, void* stream) {
  if (R % TSX != 0 || (C != 16 && C != 32 && C != 48)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(workspace) & 7) != 0 || !sort_matches(workspace, M, R)) return (int)hipErrorInvalidValue;
  Roi roi;
  if (!make_roi(roi_host, 3 * C, R, roi) || (roi.rw && (!(channel_major & 1) || roi.spp != (int)C || roi.s0 != 0)))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int TNX = R / TSX, TNY = R / TSY;
  const SortWs w = sort_ws(const_cast<void*>(workspace), R, M);
  const int* offsets = w.offsets;
  const uint32_t* entries = w.entries;
  const float2* epos = w.epos;
  const _Float16* df = reinterpret_cast<const _Float16*>(dfeat_half);
  const int ntiles = roi.rw ? 3 * (roi.rw / TSX) * (roi.rh / TSY) : w.nb / BIN_SUBS;
  if (C == 16)
    hipLaunchKernelGGL(k_tile_accumulate<16>, dim3(ntiles), dim3(NT), 0, st, df, M, xyz, bound, (int)R, TNX, TNY,
                       offsets, entries, epos, grad_scale, grad_out, channel_major, nonfinite_flag, roi);
  else if (C == 32)
    hipLaunchKernelGGL(k_tile_accumulate<32>, dim3(ntiles), dim3(NT), 0, st, df, M, xyz, bound, (int)R, TNX, TNY,
                       offsets, entries, epos, grad_scale, grad_out, channel_major, nonfinite_flag, roi);
  else
    hipLaunchKernelGGL(k_tile_accumulate<48>, dim3(ntiles), dim3(NT), 0, st, df, M, xyz, bound, (int)R, TNX, TNY,
                       offsets, entries, epos, grad_scale, grad_out, channel_major, nonfinite_flag, roi);
  return (int)hipGetLastError();
}

int tnl_plane_grad_binned_roi(const void* dfeat_half, const float* xyz, float bound, uint32_t M,
                              const int32_t* m_actual, uint32_t C, uint32_t R, float grad_scale, float* grad_out,
                              int channel_major, int32_t* nonfinite_flag, const int32_t* roi_host, void* workspace,
                              void* stream) {
  if (R % TSX != 0 || (C != 16 && C != 32 && C != 48)) return (int)hipErrorInvalidValue;
  const int e = tnl_plane_grad_sort(xyz, bound, M, m_actual, R, workspace, stream);
  if (e != 0) return e;
  return tnl_plane_grad_reduce(dfeat_half, xyz, bound, M, C, R, grad_scale, grad_out, channel_major, nonfinite_flag,
                               roi_host, workspace, stream);
}

}  // extern "C"
