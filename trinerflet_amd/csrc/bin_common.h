// bin_common.h -- binning of samples into (plane, tile) bins, shared by scatter.hip (k_bin) and raymarch.hip
// (k_march_train_emit counts while it writes the samples).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "triplane_common.h"

constexpr int TSX = 32;  // tile width in texels (128-B rows in the channel-major output)
constexpr int TSY = 8;   // tile height
// Every (plane, tile) bin has BIN_SUBS sub-bins, chosen by the sample's wave ((i >> 6) mod BIN_SUBS): the tiles under
// the centre of the scene are hit by thousands of rays, and atomics on ONE address serialise -- the sort's two atomic
// passes were bound by their hottest counters, not by the number of atomics.  A tile's list is the concatenation of
// its sub-bins (consecutive in the scanned offsets), so the reduction kernel only reads offsets[bin * BIN_SUBS] and
// offsets[(bin + 1) * BIN_SUBS].  Measured (ms per step | side work alone): small 1 sub-bin 2.66 | 1.23, 4: 2.36 | 0.82,
// 8: 2.44 | 0.91, 32: 2.34 | 0.86; base (4x the tiles, cooler counters) unchanged at 5.24-5.31 | 1.01 for 1-16, worse at 32.
constexpr int BIN_SUBS = 4;   // (a constant, not a build knob: tests and tools read the lists through tnl_plane_grad_sort_layout)

struct Foot {  // tiles touched by a bilinear footprint on one plane
  int tx0, ty0, tx1, ty1;
};

__device__ __forceinline__ Foot footprint(const TexelTap& t) {
  Foot f;
  f.tx0 = t.x0 / TSX; f.ty0 = t.y0 / TSY;
  f.tx1 = t.x1 / TSX; f.ty1 = t.y1 / TSY;
  return f;
}

// One sample per lane, all 64 lanes of the wave in the call (wave-level shuffles): FILL = false counts the sample's
// entries per bin, FILL = true writes its id at the cursor positions.  Consecutive samples of a ray usually fall into
// the same tile, so the lanes form runs with equal bins: only the head lane of a run issues the (integer, L2) atomic
// for the whole run and the members derive their slot from it -- ~5x fewer atomics for the primary tile; the rare
// straddle tiles use one atomic per distinct bin of the wave.
//
// FILL issues ALL of a sample's slot-returning atomics (3 planes x (primary + up to 3 straddle groups)) before it
// consumes the first returned slot: who leads which group and every lane's rank inside it follow from ballots alone.
// The first version consumed each slot right behind its atomic -- about a dozen dependent L2 round trips per wave, 85 %
// of the fill kernel's wave cycles parked (profiles/r02b_pmc_sq1.txt).
template <bool FILL>
__device__ __forceinline__ void bin_sample(float x, float y, float z, bool live, uint32_t i, float bound, int R, int TNX,
                                           int TNY, int* __restrict__ counts_or_cursor, uint32_t* __restrict__ entries,
                                           int lane, float2* __restrict__ epos = nullptr) {
  int slot_r[3][4], lead_r[3][4], rank_r[3][4];
  bool act_r[3][4];
  float fxs[3], fys[3];
#pragma unroll
  for (int p = 0; p < 3; p++) {
    TexelTap t;
    triplane_texel(x, y, z, bound, R, p, fxs[p], fys[p]);
    tap_from_texel(fxs[p], fys[p], R, t);
    const Foot f = footprint(t);
    const int base = p * TNX * TNY;
    // primary tile, run-aggregated
    const int sub = (int)((i >> 6) & (uint32_t)(BIN_SUBS - 1));
    const int bin0 = live ? (base + f.ty0 * TNX + f.tx0) * BIN_SUBS + sub : -1 - lane;
    const int prev = __shfl_up(bin0, 1);
    const bool head = (lane == 0) || (bin0 != prev);
    const unsigned long long hmask = __ballot(head);
    const unsigned long long below = hmask & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int hp = 63 - __clzll((long long)below);                       // head of my run
    const unsigned long long above = (hp == 63) ? 0ull : (hmask >> (hp + 1)) << (hp + 1);
    const int nh = above ? (__ffsll((long long)above) - 1) : 64;         // head of the next run
    slot_r[p][0] = 0;
    if (head && live) {
      if (FILL) slot_r[p][0] = atomicAdd(counts_or_cursor + bin0, nh - hp);
      else atomicAdd(counts_or_cursor + bin0, nh - hp);
    }
    lead_r[p][0] = hp; rank_r[p][0] = lane - hp; act_r[p][0] = live;
    // straddle tiles (footprint crosses a tile edge; ~15 % of the samples have one): the lanes that go to the same bin
    // are found by ballot and share one atomic (1-4 distinct bins per wave and direction)
#pragma unroll
    for (int k = 1; k < 4; k++) {
      const int tx = (k & 1) ? f.tx1 : f.tx0, ty = (k & 2) ? f.ty1 : f.ty0;
      const bool dup = ((k & 1) && f.tx1 == f.tx0) || ((k & 2) && f.ty1 == f.ty0);
      const bool act = live && !dup;
      const int bin = (base + ty * TNX + tx) * BIN_SUBS + sub;
      slot_r[p][k] = 0; lead_r[p][k] = lane; rank_r[p][k] = 0; act_r[p][k] = act;
      // groups of equal bins by ballot alone; the atomics follow the loop: a lane leads at most one group, so ONE
      // (exec-masked) atomic instruction per (plane, direction) serves every group of the wave, and no slot-returning
      // atomic sits inside a loop (where the compiler has to wait for each before the next reuses its register)
      bool is_lead = false;
      int gsize = 0;
      unsigned long long todo = __ballot(act);
      while (todo) {   // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const int lbin = __shfl(bin, leader);
        const bool mine = act && bin == lbin;
        const unsigned long long grp = __ballot(mine);
        if (lane == leader) { is_lead = true; gsize = __popcll(grp); }
        if (mine) { lead_r[p][k] = leader; rank_r[p][k] = __popcll(grp & ((1ull << lane) - 1ull)); }
        todo &= ~grp;
      }
      if (is_lead) {
        if (FILL) slot_r[p][k] = atomicAdd(counts_or_cursor + bin, gsize);
        else atomicAdd(counts_or_cursor + bin, gsize);
      }
    }
  }
  if (FILL) {
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int slot = __shfl(slot_r[p][k], lead_r[p][k]);
        if (act_r[p][k]) {
          entries[slot + rank_r[p][k]] = i;
          // the sample's clipped texel coordinates on this plane ride along: the reduction reads them in list order
          // instead of gathering xyz[i] (a 12-byte read that costs a whole sector, as many requests as the dF row)
          if (epos != nullptr) epos[slot + rank_r[p][k]] = make_float2(fxs[p], fys[p]);
        }
      }
    }
  }
}
