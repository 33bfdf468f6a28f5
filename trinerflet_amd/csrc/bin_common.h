// bin_common.h -- binning of samples into (plane, tile) bins, shared by scatter.hip (k_bin) and raymarch.hip
// (k_march_train_emit counts while it writes the samples).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "triplane_common.h"

constexpr int TSX = 32;  // tile width in texels (128-B rows in the channel-major output)
constexpr int TSY = 8;   // tile height

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
// straddle tiles use one atomic each.
template <bool FILL>
__device__ __forceinline__ void bin_sample(float x, float y, float z, bool live, uint32_t i, float bound, int R, int TNX,
                                           int TNY, int* __restrict__ counts_or_cursor, uint32_t* __restrict__ entries,
                                           int lane) {
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
    // straddle tiles (footprint crosses a tile edge; ~15 % of the samples have one): the lanes that go to the same bin
    // are found by ballot and share one atomic (1-4 distinct bins per wave and direction, where every straddling lane
    // used to issue its own)
#pragma unroll
    for (int k = 1; k < 4; k++) {
      const int tx = (k & 1) ? f.tx1 : f.tx0, ty = (k & 2) ? f.ty1 : f.ty0;
      const bool dup = ((k & 1) && f.tx1 == f.tx0) || ((k & 2) && f.ty1 == f.ty0);
      const bool act = live && !dup;
      const int bin = base + ty * TNX + tx;
      unsigned long long todo = __ballot(act);
      while (todo) {   // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const int lbin = __shfl(bin, leader);
        const bool mine = act && bin == lbin;
        const unsigned long long grp = __ballot(mine);
        int slot = 0;
        if (lane == leader) {
          if (FILL) slot = atomicAdd(counts_or_cursor + lbin, __popcll(grp));
          else atomicAdd(counts_or_cursor + lbin, __popcll(grp));
        }
        if (FILL) {
          slot = __shfl(slot, leader);
          if (mine) entries[slot + __popcll(grp & ((1ull << lane) - 1ull))] = i;
        }
        todo &= ~grp;
      }
    }
  }
}
