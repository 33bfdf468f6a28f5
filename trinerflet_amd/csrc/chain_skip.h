// chain_skip.h -- `do { t += dt; } while (t < tt);` in fp32 without walking the chain (round 6).
//
// The empty-cell skip of the march (raymarching.cu:386-398: `do { t += dt; } while (t < tt);` with dt constant when
// dt_gamma = 0, every README configuration) is a chain of DEPENDENT float adds: 7-14 per empty cell at max_steps = 1024,
// 28-55 at the `--test` render's 4096 -- more lane time than the probes of the cells themselves.  The result only depends
// on (t, dt, tt), and inside a binade the chain is an exact arithmetic progression, so it is computed in ~30 instructions
// whatever its length -- the same float to the bit (tests/test_chain_skip_cpu.py compiles this header with gcc and holds it
// against the literal loop on 40 M random cases: ties, binade crossings, sub-ulp steps).  Host and device code.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdbool.h>
#ifdef __HIPCC__
#define CHAIN_HD __device__ __forceinline__
#define CHAIN_F2U(x) __float_as_uint(x)
#define CHAIN_U2F(x) __uint_as_float(x)
#else
#define CHAIN_HD static inline
static inline uint32_t CHAIN_F2U(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float CHAIN_U2F(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
#endif

// Inside one binade [2^e, 2^(e+1)) every float is a multiple of u = 2^(e-23), and fl(t + dt) = t + qu with
// qu = rn(dt / u) u for EVERY t there, provided dt / u is not a tie (then the rounding would follow t's parity) and the
// sum stays below 2^(e+1) (above, the grain doubles).  qu is measured with one real add, the tie excluded by an exact
// test, and the number of steps to the first chain point >= tt is integer arithmetic on the mantissas; steps that would
// leave the binade, ties and chains with a grain-sized step (qu < 256 u) are taken one real add at a time.
// Below ~16 steps the loop itself is the shorter program (A/B at max_steps = 1024, where an empty cell is 7-14 steps: the
// per-lane count pass with the jump on every cell was 5 % SLOWER at small, tools/ab_chain_jump.sh): callers use
// chain_skip_or_walk, which walks short chains and jumps long ones -- the same float either way.
CHAIN_HD float chain_skip(float t, float dt, float tt);
CHAIN_HD float chain_skip_or_walk_n(float t, float dt, float tt, float walk_steps) {
  if (tt - t > walk_steps * dt) return chain_skip(t, dt, tt);
  do { t += dt; } while (t < tt);
  return t;
}
CHAIN_HD float chain_skip_or_walk(float t, float dt, float tt) {
  if (tt - t > 16.0f * dt) return chain_skip(t, dt, tt);
  do { t += dt; } while (t < tt);
  return t;
}

CHAIN_HD float chain_skip(float t, float dt, float tt) {
  bool first = true;
  for (;;) {
    if (!first && !(t < tt)) return t;
    first = false;
    const float t1 = t + dt;                               // the real next chain point
    const uint32_t eb = CHAIN_F2U(t) & 0x7f800000u;
    const float top = CHAIN_U2F(eb + 0x00800000u);          // 2^(e+1)
    if (!(t1 < top) || !(t1 < tt) || eb < (24u << 23)) { t = t1; continue; }
    const float u = CHAIN_U2F(eb - (23u << 23));
    const float qu = t1 - t;                                 // exact: both in one binade
    const int B = (int)(qu / u);                             // exact (division by a power of two)
    if (B < 256 || fabsf(dt - qu) == 0.5f * u) { t = t1; continue; }
    const int A = (int)((CHAIN_F2U(t) & 0x007fffffu) | 0x00800000u);          // t = A u
    const int Tm = tt < top ? (int)((CHAIN_F2U(tt) & 0x007fffffu) | 0x00800000u) : 0x01000000;   // tt >= t: same binade or beyond
    const int D = Tm - A;                                    // > B here (t1 < tt and t1 < top)
    int k = (int)ceilf((float)D * (1.0f / (float)B));
    k += (k * B < D) ? 1 : 0;                                // the reciprocal is good to an ulp: fix the ceiling either way
    k -= ((k - 1) * B >= D) ? 1 : 0;
    if (A + k * B > 0x00ffffff) k -= 1;                      // the last step would leave the binade: it is a real add
    if (k < 1) { t = t1; continue; }
    t = (float)(A + k * B) * u;                              // exact
  }
}
