#include <stdio.h>
#include <stdlib.h>
/* tests/test_chain_skip_cpu.py compiles this with gcc -ffp-contract=off against trinerflet_amd/csrc/chain_skip.h (the header
 * the HIP kernels include) and runs it: chain_skip(t, dt, tt) against the literal `do { t += dt; } while (t < tt);` of the
 * reference's march (raymarching.cu:393-398) on argv[1] random cases -- the two step sizes of the README runs (2 sqrt(3) /
 * 1024 and / 4096) and others, starts near every binade top between 0.25 and 8, spans from less than one step to hundreds. */
#include "chain_skip.h"
static float ref(float t, float dt, float tt) { do { t += dt; } while (t < tt); return t; }
static uint64_t s = 88172645463325252ull;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; }
int main(int argc, char** argv) {
  const long cases = argc > 1 ? atol(argv[1]) : 4000000;
  const float dts[] = {3.4641016151377544f / 1024, 3.4641016151377544f / 4096, 3.4641016151377544f / 512, 0.001f, 0.0123f, 1e-5f, 0.25f};
  long bad = 0, n = 0;
  for (long rep = 0; rep < cases; rep++) {
    float dt = rep % 8 == 7 ? (float)(1e-4 + rnd() * 0.02) : dts[rep % 7];
    float t = (float)(0.05 + rnd() * 9.0);
    if (rep % 5 == 0) { /* near a binade top */ float tops[] = {0.25f, 0.5f, 1.f, 2.f, 4.f, 8.f}; t = tops[rep % 6] - (float)(rnd() * 0.05); }
    double span = rep % 3 == 0 ? rnd() * 0.005 : (rep % 3 == 1 ? rnd() * 0.08 : rnd() * 3.0);
    if (dt < 5e-5f) span *= 0.01;
    float tt = t + (float)span;
    float a = ref(t, dt, tt), b = rep % 2 ? chain_skip(t, dt, tt) : chain_skip_or_walk(t, dt, tt);
    n++;
    if (a != b) { if (bad < 10) printf("MISMATCH t=%.9g dt=%.9g tt=%.9g ref=%.9g got=%.9g\n", t, dt, tt, a, b); bad++; }
  }
  printf("%ld cases, %ld mismatches\n", n, bad);
  return bad != 0;
}
