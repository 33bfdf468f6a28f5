/*
 * trinerflet_oracle.c -- CPU restatement of the TriNeRFLet volume-rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (trinerflet_amd/) may import, link or
 * call this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - inverse DWT (orc_idwt_*): PINNED against PyWavelets pywt.idwt2(mode='zero') and against the
 *     reference's own TriPlaneVolume.build_planes (tests/golden/idwt_*.npz, make_golden.py).
 *   - triplane sample (orc_triplane_*): PINNED against the reference's
 *     TriPlaneVolume.sample_from_planes run in-container (tests/golden/sample_*.npz).
 *   - ray generation (orc_get_rays): PINNED against the reference's get_rays (nerf/utils.py:65-149) run
 *     in-container (tests/golden/trainer_reference.npz, make_golden_trainer.py).
 *   - raymarching / shencoder kernels: the reference is CUDA-only and cannot execute in this
 *     image; these restatements are "parity unpinned" by execution, and are anchored on
 *     line-by-line restatement plus domain properties (see tests/test_oracle_props.py).
 *
 * Floating point: the marching code is float32 with the fused multiply-adds nvcc emits for
 * `a + b * c` written as explicit fmaf(); compile with -ffp-contract=off so that nothing else
 * is contracted.  The HIP kernels use the same explicit form so that per-ray sample counts are
 * bit-identical.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* Wavelet synthesis taps: pywt.Wavelet(name).rec_lo / rec_hi.         */
/* The reference selects them at reconstruction/triplaneencoder/triplane_encoder.py:174-185.    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const char *name; int L; double lo[18]; double hi[18]; } orc_wavelet;

static const orc_wavelet WAVELETS[] = {
  {"haar", 2,
   {0.7071067811865476, 0.7071067811865476},
   {0.7071067811865476, -0.7071067811865476}},
  {"bior2.2", 6,
   {0, 0.3535533905932738, 0.7071067811865476, 0.3535533905932738, 0, 0},
   {0, 0.1767766952966369, 0.3535533905932738, -1.0606601717798212, 0.3535533905932738,
    0.1767766952966369}},
  {"bior4.4", 10,
   {0, -0.06453888262869706, -0.04068941760916406, 0.41809227322161724, 0.7884856164055829,
    0.41809227322161724, -0.04068941760916406, -0.06453888262869706, 0, 0},
   {0, -0.03782845550726404, -0.023849465019556843, 0.11062440441843718, 0.37740285561283066,
    -0.8526986790088938, 0.37740285561283066, 0.11062440441843718, -0.023849465019556843,
    -0.03782845550726404}},
  {"bior2.6", 14,
   {0, 0, 0, 0, 0, 0.3535533905932738, 0.7071067811865476, 0.3535533905932738, 0, 0, 0, 0, 0, 0},
   {0, 0.006905339660024878, 0.013810679320049757, -0.04695630968816917, -0.1077232986963881,
    0.16987135563661201, 0.4474660099696121, -0.966747552403483, 0.4474660099696121,
    0.16987135563661201, -0.1077232986963881, -0.04695630968816917, 0.013810679320049757,
    0.006905339660024878}},
  {"bior6.8", 18,
   {0, 0, 0, 0.014426282505624435, 0.014467504896790148, -0.07872200106262882,
    -0.04036797903033992, 0.41784910915027457, 0.7589077294536541, 0.41784910915027457,
    -0.04036797903033992, -0.07872200106262882, 0.014467504896790148, 0.014426282505624435,
    0, 0, 0, 0},
   {0, -0.0019088317364812906, -0.0019142861290887667, 0.016990639867602342,
    0.01193456527972926, -0.04973290349094079, -0.07726317316720414, 0.09405920349573646,
    0.4207962846098268, -0.8259229974584023, 0.4207962846098268, 0.09405920349573646,
    -0.07726317316720414, -0.04973290349094079, 0.01193456527972926, 0.016990639867602342,
    -0.0019142861290887667, -0.0019088317364812906}},
};
#define N_WAVELETS ((int)(sizeof(WAVELETS) / sizeof(WAVELETS[0])))

/* returns wavelet id or -1; fills L, lo[18], hi[18] when non-NULL */
ORC_API int orc_wavelet_lookup(const char *name, int *L, double *lo, double *hi) {
  for (int i = 0; i < N_WAVELETS; i++) {
    if (strcmp(name, WAVELETS[i].name) == 0) {
      if (L) *L = WAVELETS[i].L;
      for (int k = 0; k < 18; k++) {
        if (lo) lo[k] = k < WAVELETS[i].L ? WAVELETS[i].lo[k] : 0.0;
        if (hi) hi[k] = k < WAVELETS[i].L ? WAVELETS[i].hi[k] : 0.0;
      }
      return i;
    }
  }
  return -1;
}

/* ------------------------------------------------------------------------------------------ */
/* One level of the 2-D inverse DWT as TriPlaneVolume.build_planes performs it                  */
/* (triplane_encoder.py:379,392-394): yl = 2*x ; zero-pad yl and yh by p=(L-2)/4 ;               */
/* x_next = pytorch_wavelets.DWTInverse(wave, mode='zero')((yl,[yh])).                          */
/* Closed form: 1-D synthesis along an axis, K=(L-2)/2:                        */
/*   out[o] = sum_j lo[j]*g0[o-2j+K] + hi[j]*g1[o-2j+K]   (taps outside [0,L) dropped)          */
/* 2-D: lo = syn_H(ll, lh) ; hi = syn_H(hl, hh) ; out = syn_W(lo, hi), yh[:,:,0/1/2]=lh/hl/hh.  */
/* taps_f32 != 0 rounds the taps to float first (pytorch_wavelets stores float32 buffers).      */
/* ------------------------------------------------------------------------------------------ */
static void get_taps(int wave, int taps_f32, double *g0, double *g1) {
  const orc_wavelet *w = &WAVELETS[wave];
  for (int k = 0; k < w->L; k++) {
    g0[k] = taps_f32 ? (double)(float)w->lo[k] : w->lo[k];
    g1[k] = taps_f32 ? (double)(float)w->hi[k] : w->hi[k];
  }
}

/* x:[S][n][n], yh:[S][3][n][n] -> out:[S][2n][2n]; ll_scale is the "2*" of :379 */
ORC_API void orc_idwt_level_f64(const double *x, const double *yh, int S, int n, int wave,
                                int taps_f32, double ll_scale, double *out) {
  const int L = WAVELETS[wave].L, K = (L - 2) / 2, m = 2 * n;
  double g0[18], g1[18];
  get_taps(wave, taps_f32, g0, g1);
  double *lo = (double *)malloc(sizeof(double) * (size_t)m * n);
  double *hi = (double *)malloc(sizeof(double) * (size_t)m * n);
  for (int s = 0; s < S; s++) {
    const double *ll = x + (size_t)s * n * n;
    const double *lh = yh + ((size_t)s * 3 + 0) * n * n;
    const double *hl = yh + ((size_t)s * 3 + 1) * n * n;
    const double *hh = yh + ((size_t)s * 3 + 2) * n * n;
    /* synthesis along H (rows index o, columns untouched) */
    for (int o = 0; o < m; o++) {
      for (int c = 0; c < n; c++) {
        double a = 0, b = 0;
        for (int j = 0; j < n; j++) {
          int k = o - 2 * j + K;
          if (k < 0 || k >= L) continue;
          a += ll_scale * ll[(size_t)j * n + c] * g0[k] + lh[(size_t)j * n + c] * g1[k];
          b += hl[(size_t)j * n + c] * g0[k] + hh[(size_t)j * n + c] * g1[k];
        }
        lo[(size_t)o * n + c] = a;
        hi[(size_t)o * n + c] = b;
      }
    }
    /* synthesis along W */
    double *dst = out + (size_t)s * m * m;
    for (int r = 0; r < m; r++) {
      for (int o = 0; o < m; o++) {
        double a = 0;
        int jlo = (o + K - L + 1) / 2 - 1; if (jlo < 0) jlo = 0;
        int jhi = (o + K) / 2 + 1; if (jhi > n - 1) jhi = n - 1;
        for (int j = jlo; j <= jhi; j++) {
          int k = o - 2 * j + K;
          if (k < 0 || k >= L) continue;
          a += lo[(size_t)r * n + j] * g0[k] + hi[(size_t)r * n + j] * g1[k];
        }
        dst[(size_t)r * m + o] = a;
      }
    }
  }
  free(lo); free(hi);
}

/* float I/O variant (double accumulation, float32 taps: what the reference holds) */
ORC_API void orc_idwt_level_f32(const float *x, const float *yh, int S, int n, int wave,
                                float *out) {
  size_t nx = (size_t)S * n * n;
  double *xd = (double *)malloc(sizeof(double) * nx);
  double *yd = (double *)malloc(sizeof(double) * nx * 3);
  double *od = (double *)malloc(sizeof(double) * nx * 4);
  for (size_t i = 0; i < nx; i++) xd[i] = x[i];
  for (size_t i = 0; i < nx * 3; i++) yd[i] = yh[i];
  orc_idwt_level_f64(xd, yd, S, n, wave, 1, 2.0, od);
  for (size_t i = 0; i < nx * 4; i++) out[i] = (float)od[i];
  free(xd); free(yd); free(od);
}

/* Adjoint (VJP) of one level: dout:[S][2n][2n] -> dx:[S][n][n] (includes the 2x), dyh:[S][3][n][n].
 * This is what autograd of pytorch_wavelets' SFB2D (analysis with the rec_* taps) followed by
 * the crop of F.pad and the 2* produces (SURVEY.md A.1 "Backward"). */
ORC_API void orc_idwt_level_adj_f64(const double *dout, int S, int n, int wave, int taps_f32,
                                    double ll_scale, double *dx, double *dyh) {
  const int L = WAVELETS[wave].L, K = (L - 2) / 2, m = 2 * n;
  double g0[18], g1[18];
  get_taps(wave, taps_f32, g0, g1);
  double *dlo = (double *)malloc(sizeof(double) * (size_t)m * n);
  double *dhi = (double *)malloc(sizeof(double) * (size_t)m * n);
  for (int s = 0; s < S; s++) {
    const double *src = dout + (size_t)s * m * m;
    for (int r = 0; r < m; r++) {
      for (int j = 0; j < n; j++) {
        double a = 0, b = 0;
        for (int k = 0; k < L; k++) {
          int o = 2 * j - K + k;
          if (o < 0 || o >= m) continue;
          a += src[(size_t)r * m + o] * g0[k];
          b += src[(size_t)r * m + o] * g1[k];
        }
        dlo[(size_t)r * n + j] = a;
        dhi[(size_t)r * n + j] = b;
      }
    }
    double *dll = dx + (size_t)s * n * n;
    double *dlh = dyh + ((size_t)s * 3 + 0) * n * n;
    double *dhl = dyh + ((size_t)s * 3 + 1) * n * n;
    double *dhh = dyh + ((size_t)s * 3 + 2) * n * n;
    for (int j = 0; j < n; j++) {
      for (int c = 0; c < n; c++) {
        double a = 0, b = 0, cc = 0, d = 0;
        for (int k = 0; k < L; k++) {
          int o = 2 * j - K + k;
          if (o < 0 || o >= m) continue;
          a += dlo[(size_t)o * n + c] * g0[k];
          b += dlo[(size_t)o * n + c] * g1[k];
          cc += dhi[(size_t)o * n + c] * g0[k];
          d += dhi[(size_t)o * n + c] * g1[k];
        }
        dll[(size_t)j * n + c] = ll_scale * a;
        dlh[(size_t)j * n + c] = b;
        dhl[(size_t)j * n + c] = cc;
        dhh[(size_t)j * n + c] = d;
      }
    }
  }
  free(dlo); free(dhi);
}

ORC_API void orc_idwt_level_adj_f32(const float *dout, int S, int n, int wave, float *dx,
                                    float *dyh) {
  size_t nx = (size_t)S * n * n;
  double *dd = (double *)malloc(sizeof(double) * nx * 4);
  double *xd = (double *)malloc(sizeof(double) * nx);
  double *yd = (double *)malloc(sizeof(double) * nx * 3);
  for (size_t i = 0; i < nx * 4; i++) dd[i] = dout[i];
  orc_idwt_level_adj_f64(dd, S, n, wave, 1, 2.0, xd, yd);
  for (size_t i = 0; i < nx; i++) dx[i] = (float)xd[i];
  for (size_t i = 0; i < nx * 3; i++) dyh[i] = (float)yd[i];
  free(dd); free(xd); free(yd);
}

/* ------------------------------------------------------------------------------------------ */
/* Triplane lookup: TriPlaneVolume.sample_from_planes_aux (triplane_encoder.py:314-332) with    */
/* plane_axes from create_subplanes_trivial_base (:250-289): plane0<-(x,z), plane1<-(x,y),      */
/* plane2<-(y,z); F.grid_sample(bilinear, border, align_corners=True): grid x -> W, y -> H.     */
/* planes:[3][C][R][R] (reference layout), xyz:[N][3] -> out:[N][3C], index plane*C + c.        */
/* Coordinate arithmetic in float as torch does it (u = xyz/bound ; ((u+1)/2)*(R-1) ; clip).    */
/* ------------------------------------------------------------------------------------------ */
static inline void tri_coords(const float *p, float bound, int R, int plane, int *x0, int *y0,
                              int *x1, int *y1, float *wx, float *wy) {
  float ux = p[0] / bound, uy = p[1] / bound, uz = p[2] / bound;
  float gx = plane == 2 ? uy : ux;
  float gy = plane == 0 ? uz : (plane == 1 ? uy : uz);
  float fx = ((gx + 1.f) / 2.f) * (float)(R - 1);
  float fy = ((gy + 1.f) / 2.f) * (float)(R - 1);
  fx = fminf((float)(R - 1), fmaxf(fx, 0.f));
  fy = fminf((float)(R - 1), fmaxf(fy, 0.f));
  float flx = floorf(fx), fly = floorf(fy);
  *x0 = (int)flx; *y0 = (int)fly;
  *x1 = *x0 + 1 > R - 1 ? R - 1 : *x0 + 1;
  *y1 = *y0 + 1 > R - 1 ? R - 1 : *y0 + 1;
  *wx = fx - flx; *wy = fy - fly;
}

ORC_API void orc_triplane_sample(const float *planes, const float *xyz, float bound, int N, int C,
                                 int R, float *out) {
  for (int i = 0; i < N; i++) {
    for (int p = 0; p < 3; p++) {
      int x0, y0, x1, y1; float wx, wy;
      tri_coords(xyz + 3 * (size_t)i, bound, R, p, &x0, &y0, &x1, &y1, &wx, &wy);
      /* torch weights: nw=(x1-x)(y1-y) ne=(x-x0)(y1-y) sw=(x1-x)(y-y0) se=(x-x0)(y-y0) */
      float nw = (1.f - wx) * (1.f - wy), ne = wx * (1.f - wy), sw = (1.f - wx) * wy, se = wx * wy;
      for (int c = 0; c < C; c++) {
        const float *pl = planes + ((size_t)p * C + c) * R * R;
        double v = (double)pl[(size_t)y0 * R + x0] * nw + (double)pl[(size_t)y0 * R + x1] * ne +
                   (double)pl[(size_t)y1 * R + x0] * sw + (double)pl[(size_t)y1 * R + x1] * se;
        out[(size_t)i * 3 * C + p * C + c] = (float)v;
      }
    }
  }
}

/* VJP wrt planes (grid_sampler_2d_backward's scatter).  dplanes:[3][C][R][R] double, accumulated. */
ORC_API void orc_triplane_sample_bwd(const float *dout, const float *xyz, float bound, int N,
                                     int C, int R, double *dplanes) {
  for (int i = 0; i < N; i++) {
    for (int p = 0; p < 3; p++) {
      int x0, y0, x1, y1; float wx, wy;
      tri_coords(xyz + 3 * (size_t)i, bound, R, p, &x0, &y0, &x1, &y1, &wx, &wy);
      float nw = (1.f - wx) * (1.f - wy), ne = wx * (1.f - wy), sw = (1.f - wx) * wy, se = wx * wy;
      /* torch skips out-of-range corners (x0+1 > R-1); their weight is 0 so clamping is equal */
      for (int c = 0; c < C; c++) {
        double g = dout[(size_t)i * 3 * C + p * C + c];
        double *pl = dplanes + ((size_t)p * C + c) * R * R;
        pl[(size_t)y0 * R + x0] += g * nw;
        pl[(size_t)y0 * R + x1] += g * ne;
        pl[(size_t)y1 * R + x0] += g * sw;
        pl[(size_t)y1 * R + x1] += g * se;
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Spherical harmonics, degree 4 (16 values): aux_libs/shencoder/src/shencoder.cu:43-68.        */
/* ------------------------------------------------------------------------------------------ */
ORC_API void orc_sh4(const float *dirs, int N, float *out) {
  for (int i = 0; i < N; i++) {
    float x = dirs[3 * (size_t)i], y = dirs[3 * (size_t)i + 1], z = dirs[3 * (size_t)i + 2];
    float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float *o = out + 16 * (size_t)i;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* raymarching helpers: aux_libs/raymarching/src/raymarching.cu:25-81                           */
/* ------------------------------------------------------------------------------------------ */
static inline float signf_(float x) { return copysignf(1.0f, x); }
static inline float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

static inline int mip_from_pos(float x, float y, float z, float max_cascade) {
  float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  int e; frexpf(mx, &e);
  return (int)fminf(max_cascade - 1, fmaxf(0, (float)e));
}
static inline int mip_from_dt(float dt, float H, float max_cascade) {
  float mx = (float)((double)dt * (double)H * 0.5); /* power-of-two scaling: exact */
  int e; frexpf(mx, &e);
  return (int)fminf(max_cascade - 1, fmaxf(0, (float)e));
}
static inline uint32_t expand_bits(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
static inline uint32_t morton3D_(uint32_t x, uint32_t y, uint32_t z) {
  return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline uint32_t morton3D_invert_(uint32_t x) {
  x = x & 0x49249249;
  x = (x | (x >> 2)) & 0xc30c30c3;
  x = (x | (x >> 4)) & 0x0f00f00f;
  x = (x | (x >> 8)) & 0xff0000ff;
  x = (x | (x >> 16)) & 0x0000ffff;
  return x;
}

/* raymarching.cu:92-145 */
ORC_API void orc_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                                    uint32_t N, float min_near, float *nears, float *fars) {
  for (uint32_t n = 0; n < N; n++) {
    const float *o = rays_o + 3 * (size_t)n, *d = rays_d + 3 * (size_t)n;
    float ox = o[0], oy = o[1], oz = o[2];
    float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, t;
    if (near > far) { t = near; near = far; far = t; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { t = near_y; near_y = far_y; far_y = t; }
    if (near > far_y || near_y > far) { nears[n] = fars[n] = 3.402823466e+38f; continue; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { t = near_z; near_z = far_z; far_z = t; }
    if (near > far_z || near_z > far) { nears[n] = fars[n] = 3.402823466e+38f; continue; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    nears[n] = near; fars[n] = far;
  }
}

/* raymarching.cu:214-226 */
ORC_API void orc_morton3D(const int *coords, uint32_t N, int *indices) {
  for (uint32_t n = 0; n < N; n++)
    indices[n] = (int)morton3D_((uint32_t)coords[3 * (size_t)n], (uint32_t)coords[3 * (size_t)n + 1],
                                (uint32_t)coords[3 * (size_t)n + 2]);
}
/* raymarching.cu:237-254 */
ORC_API void orc_morton3D_invert(const int *indices, uint32_t N, int *coords) {
  for (uint32_t n = 0; n < N; n++) {
    int ind = indices[n];
    coords[3 * (size_t)n + 0] = (int)morton3D_invert_((uint32_t)(ind >> 0));
    coords[3 * (size_t)n + 1] = (int)morton3D_invert_((uint32_t)(ind >> 1));
    coords[3 * (size_t)n + 2] = (int)morton3D_invert_((uint32_t)(ind >> 2));
  }
}
/* raymarching.cu:268-289; N = number of output bytes */
ORC_API void orc_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
  for (uint32_t n = 0; n < N; n++) {
    uint8_t bits = 0;
    for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > density_thresh) ? (uint8_t)(1u << i) : 0;
    bitfield[n] = bits;
  }
}

/* One marching state machine shared by train and inference paths (raymarching.cu:358-398,
 * 430-479, 749-805).  emit==NULL counts only. Returns number of occupied steps taken; *t_io is
 * advanced. */
typedef struct {
  float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, bound, dt_gamma, dt_min, dt_max;
  uint32_t C, H;
  const uint8_t *grid;
} march_ctx;

static void march_ctx_init(march_ctx *m, const float *o, const float *d, float bound,
                           float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                           const uint8_t *grid) {
  m->ox = o[0]; m->oy = o[1]; m->oz = o[2];
  m->dx = d[0]; m->dy = d[1]; m->dz = d[2];
  m->rdx = 1 / m->dx; m->rdy = 1 / m->dy; m->rdz = 1 / m->dz;
  m->rH = 1 / (float)H;
  m->H3 = (float)(H * H * H);
  m->bound = bound; m->dt_gamma = dt_gamma;
  m->dt_min = 2 * 1.7320508075688772f / max_steps;
  m->dt_max = 2 * 1.7320508075688772f * (float)(1 << (C - 1)) / H;
  m->C = C; m->H = H; m->grid = grid;
}

static uint32_t march_run(const march_ctx *m, float *t_io, float far, uint32_t limit,
                          float *xyzs, float *dirs, float *deltas) {
  float t = *t_io, last_t = t;
  uint32_t step = 0;
  const float H = (float)m->H;
  while (t < far && step < limit) {
    const float x = clampf_(fmaf(t, m->dx, m->ox), -m->bound, m->bound);
    const float y = clampf_(fmaf(t, m->dy, m->oy), -m->bound, m->bound);
    const float z = clampf_(fmaf(t, m->dz, m->oz), -m->bound, m->bound);
    const float dt = clampf_(t * m->dt_gamma, m->dt_min, m->dt_max);
    int la = mip_from_pos(x, y, z, (float)m->C), lb = mip_from_dt(dt, H, (float)m->C);
    const int level = la > lb ? la : lb;
    const float mip_bound = fminf(scalbnf(1.0f, level), m->bound);
    const float mip_rbound = 1 / mip_bound;
    /* 0.5 * (x*r + 1) * H is evaluated in double in the reference; 0.5 and H are powers of two,
     * so the float evaluation below is bit-identical */
    const int nx = (int)clampf_(0.5f * fmaf(x, mip_rbound, 1.0f) * H, 0.0f, (float)(m->H - 1));
    const int ny = (int)clampf_(0.5f * fmaf(y, mip_rbound, 1.0f) * H, 0.0f, (float)(m->H - 1));
    const int nz = (int)clampf_(0.5f * fmaf(z, mip_rbound, 1.0f) * H, 0.0f, (float)(m->H - 1));
    const uint32_t index = (uint32_t)((float)level * m->H3) + morton3D_((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const int occ = m->grid[index / 8] & (1 << (index % 8));
    if (occ) {
      if (xyzs) {
        xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
        dirs[0] = m->dx; dirs[1] = m->dy; dirs[2] = m->dz;
      }
      t += dt;
      if (xyzs) {
        deltas[0] = dt; deltas[1] = t - last_t;
        xyzs += 3; dirs += 3; deltas += 2;
      }
      last_t = t;
      step++;
    } else {
      const float tx = (fmaf((((float)nx + 0.5f + 0.5f * signf_(m->dx)) * m->rH * 2 - 1), mip_bound, -x)) * m->rdx;
      const float ty = (fmaf((((float)ny + 0.5f + 0.5f * signf_(m->dy)) * m->rH * 2 - 1), mip_bound, -y)) * m->rdy;
      const float tz = (fmaf((((float)nz + 0.5f + 0.5f * signf_(m->dz)) * m->rH * 2 - 1), mip_bound, -z)) * m->rdz;
      const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
      do { t += clampf_(t * m->dt_gamma, m->dt_min, m->dt_max); } while (t < tt);
    }
  }
  *t_io = t;
  return step;
}

/* raymarching.cu:312-480.  The reference packs rays in atomic arrival order (nondeterministic,
 * SURVEY F10); the oracle uses ray-id order, which is one valid arrival order: rays[n] =
 * (n, exclusive-prefix of num_steps, num_steps).  counter[0] += total steps, counter[1] += N.
 * xyzs/dirs/deltas must be zero-filled by the caller (raymarching.py:205-207). */
ORC_API void orc_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                                  float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                                  uint32_t C, uint32_t H, uint32_t M, const float *nears,
                                  const float *fars, float *xyzs, float *dirs, float *deltas,
                                  int *rays, int *counter, const float *noises) {
  uint32_t point_index = (uint32_t)counter[0];
  uint32_t ray_index = (uint32_t)counter[1];
  for (uint32_t n = 0; n < N; n++) {
    march_ctx m;
    march_ctx_init(&m, rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid);
    const float near = nears[n], far = fars[n], noise = noises[n];
    float t0 = near;
    t0 = fmaf(clampf_(t0 * dt_gamma, m.dt_min, m.dt_max), noise, t0); /* nvcc contracts `t0 += a*b` */
    float t = t0;
    uint32_t num_steps = march_run(&m, &t, far, max_steps, NULL, NULL, NULL);
    uint32_t off = point_index;
    point_index += num_steps;
    rays[3 * (size_t)ray_index + 0] = (int)n;
    rays[3 * (size_t)ray_index + 1] = (int)off;
    rays[3 * (size_t)ray_index + 2] = (int)num_steps;
    ray_index++;
    if (num_steps == 0) continue;
    if (off + num_steps > M) continue;
    t = t0;
    march_run(&m, &t, far, num_steps, xyzs + 3 * (size_t)off, dirs + 3 * (size_t)off, deltas + 2 * (size_t)off);
  }
  counter[0] = (int)point_index;
  counter[1] = (int)ray_index;
}

/* raymarching.cu:501-577 (expf stands in for __expf; compare with a tolerance) */
ORC_API void orc_composite_rays_train_forward(const float *sigmas, const float *rgbs,
                                              const float *deltas, const int *rays, uint32_t M,
                                              uint32_t N, float T_thresh, float *weights_sum,
                                              float *depth, float *image) {
  for (uint32_t n = 0; n < N; n++) {
    uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
             num_steps = (uint32_t)rays[3 * (size_t)n + 2];
    if (num_steps == 0 || offset + num_steps > M) {
      weights_sum[index] = 0; depth[index] = 0;
      image[3 * (size_t)index] = image[3 * (size_t)index + 1] = image[3 * (size_t)index + 2] = 0;
      continue;
    }
    const float *sg = sigmas + offset, *c = rgbs + 3 * (size_t)offset, *dl = deltas + 2 * (size_t)offset;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    for (uint32_t step = 0; step < num_steps; step++) {
      const float alpha = 1.0f - expf(-sg[0] * dl[0]);
      const float weight = alpha * T;
      r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
      t += dl[1]; d += weight * t; ws += weight;
      T *= 1.0f - alpha;
      if (T < T_thresh) break;
      sg++; c += 3; dl += 2;
    }
    weights_sum[index] = ws; depth[index] = d;
    image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
  }
}

/* raymarching.cu:602-682; grad_sigmas / grad_rgbs zero-filled by the caller (raymarching.py:283-284) */
ORC_API void orc_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                               const float *sigmas, const float *rgbs,
                                               const float *deltas, const int *rays,
                                               const float *weights_sum, const float *image,
                                               uint32_t M, uint32_t N, float T_thresh,
                                               float *grad_sigmas, float *grad_rgbs) {
  for (uint32_t n = 0; n < N; n++) {
    uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
             num_steps = (uint32_t)rays[3 * (size_t)n + 2];
    if (num_steps == 0 || offset + num_steps > M) continue;
    const float gws = grad_weights_sum[index];
    const float *gi = grad_image + 3 * (size_t)index;
    const float ws_final = weights_sum[index];
    const float r_final = image[3 * (size_t)index], g_final = image[3 * (size_t)index + 1], b_final = image[3 * (size_t)index + 2];
    const float *sg = sigmas + offset, *c = rgbs + 3 * (size_t)offset, *dl = deltas + 2 * (size_t)offset;
    float *gs = grad_sigmas + offset, *gc = grad_rgbs + 3 * (size_t)offset;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
    for (uint32_t step = 0; step < num_steps; step++) {
      const float alpha = 1.0f - expf(-sg[0] * dl[0]);
      const float weight = alpha * T;
      r += weight * c[0]; g += weight * c[1]; b += weight * c[2]; ws += weight;
      T *= 1.0f - alpha;
      gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
      gs[0] = dl[0] * (gi[0] * (T * c[0] - (r_final - r)) + gi[1] * (T * c[1] - (g_final - g)) +
                       gi[2] * (T * c[2] - (b_final - b)) + gws * (1 - ws_final));
      if (T < T_thresh) break;
      sg++; c += 3; dl += 2; gs++; gc += 3;
    }
  }
}

/* raymarching.cu:701-805; outputs zero-filled by the caller (raymarching.py:333-335) */
ORC_API void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int *rays_alive,
                            const float *rays_t, const float *rays_o, const float *rays_d,
                            float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                            const uint8_t *grid, const float *nears, const float *fars,
                            float *xyzs, float *dirs, float *deltas, const float *noises) {
  (void)nears;
  for (uint32_t n = 0; n < n_alive; n++) {
    const int index = rays_alive[n];
    march_ctx m;
    march_ctx_init(&m, rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, bound, dt_gamma, max_steps, C, H, grid);
    float t = rays_t[index];
    const float far = fars[index];
    t = fmaf(clampf_(t * dt_gamma, m.dt_min, m.dt_max), noises[n], t);
    march_run(&m, &t, far, n_step, xyzs + (size_t)n * n_step * 3, dirs + (size_t)n * n_step * 3,
              deltas + (size_t)n * n_step * 2);
  }
}

/* raymarching.cu:819-905 */
ORC_API void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int *rays_alive,
                                float *rays_t, const float *sigmas, const float *rgbs,
                                const float *deltas, float *weights_sum, float *depth, float *image) {
  for (uint32_t n = 0; n < n_alive; n++) {
    const int index = rays_alive[n];
    const float *sg = sigmas + (size_t)n * n_step, *c = rgbs + (size_t)n * n_step * 3,
                *dl = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index];
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
    uint32_t step = 0;
    while (step < n_step) {
      if (dl[0] == 0) break;
      const float alpha = 1.0f - expf(-sg[0] * dl[0]);
      const float T = 1 - weight_sum;
      const float weight = alpha * T;
      weight_sum += weight;
      t += dl[1];
      d += weight * t;
      r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
      if (T < T_thresh) break;
      sg++; c += 3; dl += 2; step++;
    }
    if (step < n_step) rays_alive[n] = -1; else rays_t[index] = t;
    weights_sum[index] = weight_sum; depth[index] = d;
    image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
  }
}

/* ---------------------------------------------------------------------------------------------
 * Ray generation for flat pixel ids (8(f) rank 2).  reconstruction/nerf/utils.py:65-149 (get_rays):
 *   i = x + 0.5, j = y + 0.5 ; dir = ((i-cx)/fx, (j-cy)/fy, 1) / |.| ; rays_d = dir @ R^T ; rays_o = t.
 * pix[n] = b*H*W + y*W + x (the flattened [B, H*W] index of shuffle_data, utils.py:228-236).
 * ------------------------------------------------------------------------------------------- */
ORC_API void orc_get_rays(const float *poses, const float *intrinsics, uint32_t H, uint32_t W,
                          const int64_t *pix, uint64_t N, float *rays_o, float *rays_d) {
  const float fx = intrinsics[0], fy = intrinsics[1], cx = intrinsics[2], cy = intrinsics[3];
  const uint64_t HW = (uint64_t)H * W;
  for (uint64_t n = 0; n < N; n++) {
    const uint64_t b = (uint64_t)pix[n] / HW, p = (uint64_t)pix[n] % HW;
    const float i = (float)(p % W) + 0.5f, j = (float)(p / W) + 0.5f;
    const float xs = (i - cx) / fx, ys = (j - cy) / fy, zs = 1.0f;
    const float nrm = sqrtf(xs * xs + ys * ys + zs * zs);
    const float d[3] = {xs / nrm, ys / nrm, zs / nrm};
    const float *P = poses + 16 * b;
    for (int k = 0; k < 3; k++) {
      rays_d[3 * n + k] = d[0] * P[4 * k + 0] + d[1] * P[4 * k + 1] + d[2] * P[4 * k + 2];
      rays_o[3 * n + k] = P[4 * k + 3];
    }
  }
}

/* Keyed bijection of [0, total) used instead of a materialised torch.randperm (utils.py:230): balanced Feistel
 * network over the next even power of two, cycle-walked into range.  Own construction (no reference
 * counterpart beyond "a uniform random permutation per epoch"); restated here so the device version can be
 * checked value for value. */
static uint32_t feistel_f_(uint32_t x, uint64_t key, uint32_t round) {
  uint32_t h = x * 0x9E3779B1u + (uint32_t)(key >> (16 * (round & 3))) + round * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
ORC_API uint64_t orc_permute_index(uint64_t g, uint64_t total, uint64_t key) {
  uint32_t half = 1;
  while ((1ull << (2 * half)) < total) half++;
  const uint32_t mask = (uint32_t)((1ull << half) - 1);
  uint64_t v = g;
  do {
    uint32_t l = (uint32_t)(v >> half) & mask, r = (uint32_t)v & mask;
    for (uint32_t k = 0; k < 4; k++) {
      const uint32_t t = l ^ (feistel_f_(r, key, k) & mask);
      l = r; r = t;
    }
    v = ((uint64_t)l << half) | r;
  } while (v >= total);
  return v;
}
