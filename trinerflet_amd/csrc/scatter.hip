// scatter.hip -- plane-gradient accumulation without global float atomics (gfx950).
//
// Replaces torch's grid_sampler_2d_backward (atomicAdd of 12*C floats per sample into the (3,C,R,R)
// gradient, which autograd zero-fills first) for the training step.  The memory-side fp32 atomic unit of
// MI355X adds ~1.3 TB/s chip-wide (MI355X_MICROARCH.md "Global float atomics"); at 1536 B of adds per
// sample that alone was 6 ms per step.  Here instead:
//   1. the field backward writes the feature gradient dF as fp16 [M, 3C] (the precision the reference's
//      autocast Linear backward hands to grid_sample's backward as well),
//   2. samples are counting-sorted by the 16x16-texel tile their bilinear footprint touches, per plane
//      (a footprint that straddles tiles is listed in each of them),
//   3. one workgroup per tile accumulates its samples in an fp32 LDS tile (ds_add_f32, lanes = channels)
//      and writes the finished tile with plain 16-byte stores -- every tile is written exactly once, so
//      the 4*P-byte zero fill of the gradient disappears too.
// HBM traffic per sample: 12 B xyz x3 passes + ~3.4 list entries x 4 B x2 + 6*C B of dF, versus 48*C B
// of atomics; per step additionally the 4*P-byte tile stores that replace the memset.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trinerflet_hip.h"
#include "triplane_common.h"

namespace {

constexpr int TSX = 32;  // tile width in texels (128-B rows in the channel-major output)
constexpr int TSY = 8;   // tile height
constexpr int NT = 256;

struct Foot {  // tiles touched by a bilinear footprint on one plane
  int tx0, ty0, tx1, ty1;
};

__device__ __forceinline__ Foot footprint(const TexelTap& t) {
  Foot f;
  f.tx0 = t.x0 / TSX; f.ty0 = t.y0 / TSY;
  f.tx1 = t.x1 / TSX; f.ty1 = t.y1 / TSY;
  return f;
}

__device__ __forceinline__ uint32_t eff_m(uint32_t M, const int32_t* m_actual) {
  return m_actual ? min(M, (uint32_t)max(*m_actual, 0)) : M;
}

// pass 1 / pass 3: FILL=false counts entries per bin, FILL=true writes sample ids at cursor positions.
// Consecutive samples of a ray usually fall into the same tile, so the lanes of a wave form runs with equal
// bins: only the head lane of a run issues the (integer, L2) atomic for the whole run and the members derive
// their slot from it -- ~5x fewer atomics for the primary tile; the rare straddle tiles use one atomic each.
template <bool FILL>
__global__ void __launch_bounds__(NT)
k_bin(const float* __restrict__ xyz, float bound, uint32_t M, const int32_t* __restrict__ m_actual, int R, int TNX,
      int TNY, int* __restrict__ counts_or_cursor, uint32_t* __restrict__ entries) {
  const uint32_t Me = eff_m(M, m_actual);
  const uint32_t i = blockIdx.x * NT + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = i < Me;
  const uint32_t il = live ? i : 0;
  const float x = xyz[(size_t)il * 3], y = xyz[(size_t)il * 3 + 1], z = xyz[(size_t)il * 3 + 2];
#pragma unroll
  for (int p = 0; p < 3; p++) {
    TexelTap t;
    triplane_tap(x, y, z, bound, R, p, t);
    const Foot f = footprint(t);
    const int base = p * TNX * TNY;
    // primary tile, run-aggregated
    const int bin0 = live ? base + f.ty0 * TNX + f.tx0 : -1 - lane;
    const int prev = __shfl_up(bin0, 1);
    const bool head = (lane == 0) || (bin0 != prev);
    const unsigned long long hmask = __ballot(head);
    const unsigned long long below = hmask & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int hp = 63 - __clzll((long long)below);                       // head of my run
    const unsigned long long above = (hp == 63) ? 0ull : (hmask >> (hp + 1)) << (hp + 1);
    const int nh = above ? (__ffsll((long long)above) - 1) : 64;         // head of the next run
    int slot = 0;
    if (head && live) {
      if (FILL) slot = atomicAdd(counts_or_cursor + bin0, nh - hp);
      else atomicAdd(counts_or_cursor + bin0, nh - hp);
    }
    if (FILL) {
      slot = __shfl(slot, hp) + (lane - hp);
      if (live) entries[slot] = i;
    }
    // straddle tiles (footprint crosses a tile edge): one atomic each
    if (live) {
#pragma unroll
      for (int k = 1; k < 4; k++) {
        const int tx = (k & 1) ? f.tx1 : f.tx0, ty = (k & 2) ? f.ty1 : f.ty0;
        const bool dup = ((k & 1) && f.tx1 == f.tx0) || ((k & 2) && f.ty1 == f.ty0);
        if (dup) continue;
        const int bin = base + ty * TNX + tx;
        if (FILL) entries[atomicAdd(counts_or_cursor + bin, 1)] = i;
        else atomicAdd(counts_or_cursor + bin, 1);
      }
    }
  }
}

// exclusive scan of the bin counts (single workgroup); also leaves a copy as the fill cursors
__global__ void __launch_bounds__(1024)
k_scan_bins(const int* __restrict__ counts, int nb, int* __restrict__ offsets, int* __restrict__ cursor) {
  __shared__ int part[1024];
  const int per = (nb + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(lo + per, nb);
  int s = 0;
  for (int k = lo; k < hi; k++) s += counts[k];
  part[threadIdx.x] = s;
  __syncthreads();
  // Hillis-Steele over 1024 partial sums
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int k = lo; k < hi; k++) {
    offsets[k] = run;
    cursor[k] = run;
    run += counts[k];
  }
  if (threadIdx.x == 1023) offsets[nb] = part[1023];
}

// pass 4: one workgroup per (plane, tile): accumulate in LDS, store the tile.
//
// Float LDS atomics are the slow path on this chip (ds_add_f32 measured ~150 cycles per wave-instruction: 9 ms
// per step), and a read-modify-write chain per record is latency-bound, so the tile is reduced by SORTING:
// entries are consumed in chunks of 256 records (one thread per record; the next chunk's id, xyz and fp16 dF
// slice are prefetched into registers while the current chunk is reduced).  Per chunk:
//   A. each thread computes its record's tap once, stages dF in LDS and ranks its (<= 4) in-tile corner
//      contributions per texel with an integer LDS atomic (fast path),
//   B. a 256-entry exclusive scan turns the per-texel counts into offsets,
//   C. contributions (record index, weight) are written in texel order,
//   D. each group of C lanes (lane = channel) walks the texels it owns, sums their contributions in a register
//      -- independent LDS reads, no read-modify-write chain -- and adds the sum to its exclusively owned
//      accumulator words.
template <int C>
__global__ void __launch_bounds__(NT)
k_tile_accumulate(const _Float16* __restrict__ dfeat, const float* __restrict__ xyz, float bound, int R, int TNX,
                  int TNY, const int* __restrict__ offsets, const uint32_t* __restrict__ entries, float grad_scale,
                  float* __restrict__ grad_out, int channel_major) {
  constexpr int NTEX = TSX * TSY;
  constexpr int TILE_F = NTEX * C;
  constexpr int CS = C + 1;  // LDS texel stride: odd, so the transposed read of the epilogue is conflict-free
  constexpr int F = 3 * C;
  constexpr int GL = C <= 16 ? 4 : (C <= 32 ? 8 : 16);    // lanes per texel group, 4 channels per lane
  constexpr int NG = NT / GL;                             // texel groups per workgroup
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  __shared__ __attribute__((aligned(16))) float acc[NTEX * CS];
  __shared__ __attribute__((aligned(16))) _Float16 gbuf[NT][C];
  __shared__ int hist[NTEX];
  __shared__ int offs[NTEX + 1];
  __shared__ int wsum[4];
  __shared__ __attribute__((aligned(8))) float2 list_qw[4 * NT];  // (record index as int bits, weight)
  const int bin = blockIdx.x;
  const int p = bin / (TNX * TNY), rem = bin - p * TNX * TNY;
  const int ty = rem / TNX, tx = rem - ty * TNX;
  const int beg = offsets[bin], end = offsets[bin + 1];
  const int x_lo = tx * TSX, y_lo = ty * TSY;
  // epilogue store (also used for untouched tiles, whose store replaces the zero fill of the gradient):
  //   texel-major  [3][R][R][C]: a tile row is TSX*C contiguous floats
  //   channel-major (3,C,R,R)  : per channel, TSY rows of TSX contiguous floats (128 B)
  auto store_tile = [&](bool zero) {
    if (!channel_major) {
      float* dst = grad_out + (((size_t)p * R + y_lo) * R + x_lo) * C;
      constexpr int ROW_F4 = TSX * C / 4;
      for (int q = threadIdx.x; q < TILE_F / 4; q += NT) {
        const int ry = q / ROW_F4, rq = q - ry * ROW_F4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!zero) {
          const int f = rq * 4, lx = f / C, c0 = f - lx * C;
          const float* a = acc + (ry * TSX + lx) * CS + c0;
          v = make_float4(a[0], a[1], a[2], a[3]);
        }
        reinterpret_cast<float4*>(dst + (size_t)ry * R * C)[rq] = v;
      }
    } else {
      constexpr int X4 = TSX / 4;
      for (int q = threadIdx.x; q < C * TSY * X4; q += NT) {
        const int ch = q / (TSY * X4), r2 = q - ch * (TSY * X4);
        const int ry = r2 / X4, lx = (r2 - ry * X4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!zero) {
          const float* a = acc + (ry * TSX + lx) * CS + ch;
          v = make_float4(a[0], a[CS], a[2 * CS], a[3 * CS]);
        }
        *reinterpret_cast<float4*>(grad_out + (((size_t)p * C + ch) * R + y_lo + ry) * R + x_lo + lx) = v;
      }
    }
  };
  if (beg == end) {
    store_tile(true);
    return;
  }
  // acc holds only the CURRENT chunk's per-texel sums (plain stores in phase D); the running totals live in
  // registers: thread t owns the accumulator words {j*NT + t}
  constexpr int NACC = TILE_F / NT;
  float racc[NACC];
  // accumulator word j of thread t: flat index f = j*NT + t -> (texel f / C, channel f % C) -> LDS texel*CS + c
  int aidx[NACC];
#pragma unroll
  for (int j = 0; j < NACC; j++) {
    const int f = j * NT + threadIdx.x;
    aidx[j] = (f / C) * CS + (f % C);
    racc[j] = 0.f;
    acc[aidx[j]] = 0.f;
  }
  hist[threadIdx.x] = 0;  // NT == NTEX
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int grp = threadIdx.x / GL, c = 4 * (threadIdx.x - grp * GL);  // first of this lane's 4 channels

  // prefetch registers for one record
  float px = 0.f, py = 0.f, pz = 0.f;
  h8 pg[C / 8];
  bool pv = false;
  auto prefetch = [&](int base) {
    pv = base + (int)threadIdx.x < end;
    if (pv) {
      const uint32_t i = entries[base + threadIdx.x];
      px = xyz[(size_t)i * 3]; py = xyz[(size_t)i * 3 + 1]; pz = xyz[(size_t)i * 3 + 2];
      const h8* src = reinterpret_cast<const h8*>(dfeat + (size_t)i * F + p * C);
#pragma unroll
      for (int k = 0; k < C / 8; k++) pg[k] = src[k];
    }
  };
  prefetch(beg);
  for (int base = beg; base < end; base += NT) {
    __syncthreads();  // previous chunk fully reduced; acc / hist initialised on the first trip
    // ---- A: tap, stage dF, rank the in-tile corners
    int key[4], rank[4];
    float wt[4];
    const bool valid = pv;
    if (valid) {
      TexelTap t;
      triplane_tap(px, py, pz, bound, R, p, t);
#pragma unroll
      for (int k = 0; k < C / 8; k++) reinterpret_cast<h8*>(&gbuf[threadIdx.x][0])[k] = pg[k];
      const int lx0 = t.x0 - x_lo, lx1 = t.x1 - x_lo, ly0 = t.y0 - y_lo, ly1 = t.y1 - y_lo;
      const bool ix0 = (unsigned)lx0 < (unsigned)TSX, ix1 = ((unsigned)lx1 < (unsigned)TSX) && (t.x1 != t.x0);
      const bool iy0 = (unsigned)ly0 < (unsigned)TSY, iy1 = ((unsigned)ly1 < (unsigned)TSY) && (t.y1 != t.y0);
      key[0] = (iy0 && ix0) ? ly0 * TSX + lx0 : -1; wt[0] = t.w00 * grad_scale;
      key[1] = (iy0 && ix1) ? ly0 * TSX + lx1 : -1; wt[1] = t.w01 * grad_scale;
      key[2] = (iy1 && ix0) ? ly1 * TSX + lx0 : -1; wt[2] = t.w10 * grad_scale;
      key[3] = (iy1 && ix1) ? ly1 * TSX + lx1 : -1; wt[3] = t.w11 * grad_scale;
#pragma unroll
      for (int k = 0; k < 4; k++) rank[k] = key[k] >= 0 ? atomicAdd(&hist[key[k]], 1) : 0;
    }
    if (base + NT < end) prefetch(base + NT); else pv = false;
    __syncthreads();
    // ---- B: exclusive scan of the 256 per-texel counts
    {
      const int v = hist[threadIdx.x];
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off);
        if (lane >= off) incl += u;
      }
      if (lane == 63) wsum[wv] = incl;
      __syncthreads();
      int b = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) if (k < wv) b += wsum[k];
      offs[threadIdx.x] = b + incl - v;
      if (threadIdx.x == NT - 1) offs[NTEX] = b + incl;
    }
    __syncthreads();
    // ---- C: contributions in texel order
    if (valid) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (key[k] >= 0) {
          const int pos = offs[key[k]] + rank[k];
          list_qw[pos] = make_float2(__int_as_float((int)threadIdx.x | (key[k] << 8)), wt[k]);
        }
      }
    }
    __syncthreads();
    // ---- D: per-texel register reduction (lane = channel), then one add into the owned accumulator words
    if (c < C) {
      // group g owns texels [g*TPG, (g+1)*TPG): its contributions are ONE contiguous run of the sorted list.
      // Stream it 8 entries at a time (independent list + dF reads; a lane covers 4 channels = one 8-byte LDS
      // read), summing in registers and closing a texel with plain stores whenever the key changes.
      constexpr int TPG = NTEX / NG;
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      const int k0 = offs[grp * TPG], k1 = offs[(grp + 1) * TPG];
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int cur = -1;
      for (int k = k0; k < k1; k += 8) {
        float2 qw[8];
        half4 g[8];
#pragma unroll
        for (int u = 0; u < 8; u++) qw[u] = list_qw[min(k + u, 4 * NT - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) g[u] = *reinterpret_cast<const half4*>(&gbuf[__float_as_int(qw[u].x) & (NT - 1)][c]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
          if (k + u < k1) {
            const int key = __float_as_int(qw[u].x) >> 8;
            if (key != cur) {
              if (cur >= 0) {  // each texel is closed once per chunk: plain stores
                float* a = acc + cur * CS + c;
                a[0] = s0; a[1] = s1; a[2] = s2; a[3] = s3;
              }
              s0 = s1 = s2 = s3 = 0.f;
              cur = key;
            }
            const float w = qw[u].y;
            s0 = fmaf((float)g[u][0], w, s0); s1 = fmaf((float)g[u][1], w, s1);
            s2 = fmaf((float)g[u][2], w, s2); s3 = fmaf((float)g[u][3], w, s3);
          }
        }
      }
      if (cur >= 0) {
        float* a = acc + cur * CS + c;
        a[0] = s0; a[1] = s1; a[2] = s2; a[3] = s3;
      }
    }
    __syncthreads();
    // ---- E: fold the chunk sums into the register totals and clear them (independent LDS reads)
#pragma unroll
    for (int j = 0; j < NACC; j++) {
      racc[j] += acc[aidx[j]];
      acc[aidx[j]] = 0.f;
    }
    hist[threadIdx.x] = 0;
  }
  // totals back to LDS, then the (possibly transposing) coalesced store
#pragma unroll
  for (int j = 0; j < NACC; j++) acc[aidx[j]] = racc[j];
  __syncthreads();
  store_tile(false);
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

}  // namespace

extern "C" {

// bytes of scratch: counts, offsets(+1), cursor: one int per (plane, tile) each; entries: 12 * M uint32
uint64_t tnl_plane_grad_binned_workspace(uint32_t M, uint32_t R) {
  if (R % TSX != 0) return 0;
  const uint64_t nb = 3ull * (R / TSX) * (R / TSY);
  return (3 * nb + 8) * 4 + 12ull * M * 4;
}

int tnl_plane_grad_binned(const void* dfeat_half, const float* xyz, float bound, uint32_t M,
                          const int32_t* m_actual, uint32_t C, uint32_t R, float grad_scale, float* grad_out,
                          int channel_major, void* workspace, void* stream) {
  if (R % TSX != 0 || (C != 16 && C != 32 && C != 48)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int TNX = R / TSX, TNY = R / TSY;
  const int nb = 3 * TNX * TNY;
  int* counts = reinterpret_cast<int*>(workspace);
  int* offsets = counts + nb + 1;
  int* cursor = offsets + nb + 1;
  uint32_t* entries = reinterpret_cast<uint32_t*>(cursor + nb + 2);
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)(nb + 1) * sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  if (M > 0) {
    hipLaunchKernelGGL(k_bin<false>, dim3(cdiv(M, NT)), dim3(NT), 0, st, xyz, bound, M, m_actual, (int)R, TNX, TNY,
                       counts, entries);
  }
  hipLaunchKernelGGL(k_scan_bins, dim3(1), dim3(1024), 0, st, counts, nb, offsets, cursor);
  if (M > 0) {
    hipLaunchKernelGGL(k_bin<true>, dim3(cdiv(M, NT)), dim3(NT), 0, st, xyz, bound, M, m_actual, (int)R, TNX, TNY,
                       cursor, entries);
  }
  const _Float16* df = reinterpret_cast<const _Float16*>(dfeat_half);
  if (C == 16)
    hipLaunchKernelGGL(k_tile_accumulate<16>, dim3(nb), dim3(NT), 0, st, df, xyz, bound, (int)R, TNX, TNY, offsets,
                       entries, grad_scale, grad_out, channel_major);
  else if (C == 32)
    hipLaunchKernelGGL(k_tile_accumulate<32>, dim3(nb), dim3(NT), 0, st, df, xyz, bound, (int)R, TNX, TNY, offsets,
                       entries, grad_scale, grad_out, channel_major);
  else
    hipLaunchKernelGGL(k_tile_accumulate<48>, dim3(nb), dim3(NT), 0, st, df, xyz, bound, (int)R, TNX, TNY, offsets,
                       entries, grad_scale, grad_out, channel_major);
  return (int)hipGetLastError();
}

}  // extern "C"
