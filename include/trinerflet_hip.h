/*
 * trinerflet_hip.h -- C ABI of libtrinerflet_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for TriNeRFLet's volume-rendering hot path.  Every entry point
 * takes plain device pointers, sizes and a HIP stream (hipStream_t passed as void*; NULL = the
 * null stream) and returns a hipError_t value as int (0 = hipSuccess; launch errors only --
 * like the reference, no argument validation is done on the raymarching entry points,
 * SURVEY.md 8(b) "Error conventions").  The caller allocates every buffer.  No torch types.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the binding a reference maintainer would add.
 */
#ifndef TRINERFLET_HIP_H
#define TRINERFLET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNL_API __attribute__((visibility("default")))

/* ABI version; bumped when a signature changes. */
TNL_API int tnl_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * raymarching: replaces the 10 pybind functions of aux_libs/raymarching/src/raymarching.h:7-17
 * (kernels in aux_libs/raymarching/src/raymarching.cu).  All float buffers are fp32 (the
 * reference wrappers force fp32 with custom_fwd(cast_inputs=torch.float32)).
 * ------------------------------------------------------------------------------------------- */

/* raymarching.h:7  near_far_from_aabb ; kernel raymarching.cu:92-145.
 * rays_o,rays_d:[N,3] aabb:[6] -> nears,fars:[N]; a miss writes FLT_MAX to both. */
TNL_API int tnl_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                                   uint32_t N, float min_near, float *nears, float *fars,
                                   void *stream);

/* raymarching.h:8  sph_from_ray ; kernel raymarching.cu:163-198.  coords:[N,2] */
TNL_API int tnl_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                             float *coords, void *stream);

/* raymarching.h:9  morton3D ; kernel raymarching.cu:214-226.  coords:[N,3] i32 -> indices:[N] */
TNL_API int tnl_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream);

/* raymarching.h:10 morton3D_invert ; kernel raymarching.cu:237-254 */
TNL_API int tnl_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream);

/* raymarching.h:11 packbits ; kernel raymarching.cu:268-289.  grid:[8N] fp32 -> bitfield:[N] u8 */
TNL_API int tnl_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield,
                         void *stream);
/* The same with the threshold min(density_thresh, *mean_density_dev) taken on the device (renderer.py:531-533 reads the
 * mean density back to the host between the grid update and packbits; here the read-back waits until everything of the
 * refresh is enqueued). */
TNL_API int tnl_packbits_dev(const float *grid, uint32_t N, float density_thresh, const float *mean_density_dev,
                             uint8_t *bitfield, void *stream);

/* Bounding box, in cell coordinates, of the occupied cells of each cascade of a Morton-ordered bitfield
 * (the layout tnl_packbits writes): bounds[c] = {min x, y, z, max x, y, z} int32, preset by the caller to
 * {H, H, H, -1, -1, -1}.  TrainStep derives the occupancy window of the planes from it after every density-grid
 * refresh (renderer.py:448-542 changes the bitfield there); no reference counterpart. */
TNL_API int tnl_occupancy_bounds(const uint8_t *bitfield, uint32_t bytes_per_cascade, uint32_t cascades,
                                 int32_t *bounds, void *stream);
/* Per plane p and 8-texel row group g of the R x R feature planes: the column extent ext[(p * R/8 + g) * 2 + {0,1}] =
 * [lo, end) of the bilinear footprints of all positions inside occupied cells (int32, preset by the caller to
 * {INT_MAX, -1}; untouched = no sample can read that row group).  TrainStep derives from it, level by level, which
 * coefficients of a level's live rectangle can receive a gradient or reach a sampled texel at all (the "bands" of
 * tnl_adam_l1_step_live_bands); no reference counterpart. */
TNL_API int tnl_occupancy_row_extents(const uint8_t *bitfield, uint32_t bytes_per_cascade, uint32_t cascades, uint32_t H,
                                      float bound, uint32_t R, int32_t *ext, void *stream);

/* No sample can lie outside the box of the occupied cells, and `far` only enters the march's loop conditions: marching
 * with min(far, the ray's exit from that box) gives the same samples to the bit, without the probe chain through the
 * empty cells behind the object (no reference counterpart; the reference marches every ray to its far).
 *   tnl_occupied_box  box[6] (device) = world-space {min xyz, max xyz} of the occupied cells of all cascades grown by one
 *                     cell; faces within a cell of the volume boundary are +-inf (positions are clamped there); an
 *                     empty bitfield gives an empty box.  bounds_scratch: 6 * cascades int32 (device).
 *   tnl_clip_fars     fars_out[n] = min(fars[n], exit of ray n from the box), -FLT_MAX for a ray that misses it.
 * The box belongs to the bitfield it was computed from: recompute it whenever the bitfield changes. */
TNL_API int tnl_occupied_box(const uint8_t *bitfield, uint32_t bytes_per_cascade, uint32_t cascades, uint32_t H,
                             float bound, int32_t *bounds_scratch, float *box, void *stream);
TNL_API int tnl_clip_fars(const float *rays_o, const float *rays_d, const float *fars, const float *box, uint32_t N,
                          float *fars_out, void *stream);

/* Number of int32 scratch words tnl_march_rays_train needs for N rays: the minimum (the samples are then written
 * by a second march of every ray), and the size with which the count pass can record each sample's t
 * (+ N * max_steps floats; 0 if that exceeds 32 bits) so that the samples are written from the record instead --
 * bit-identical output, the ray is marched once. */
TNL_API uint32_t tnl_march_rays_train_workspace(uint32_t N);
TNL_API uint32_t tnl_march_rays_train_workspace_rec(uint32_t N, uint32_t max_steps);

/* raymarching.h:13 march_rays_train ; kernel raymarching.cu:312-480.
 * Same arguments as the reference plus a scratch buffer of workspace_words int32 (see the two size functions
 * above).  Packing is DETERMINISTIC: rays[n] =
 * (n, exclusive prefix of num_steps, num_steps), i.e. the ray-id arrival order of the
 * reference's atomics (which are nondeterministic there).  counter[0] += total steps,
 * counter[1] += N.  xyzs/dirs/deltas rows that no ray owns are left untouched (the caller
 * zero-fills them, raymarching.py:205-207). */
TNL_API int tnl_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid,
                                 float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                                 uint32_t C, uint32_t H, uint32_t M, const float *nears,
                                 const float *fars, float *xyzs, float *dirs, float *deltas,
                                 int32_t *rays, int32_t *counter, const float *noises,
                                 int32_t *workspace, uint32_t workspace_words, void *stream);
/* tnl_march_rays_train that also performs the first pass of the plane-gradient tile sort (the per-bin counts of
 * tnl_plane_grad_sort for plane resolution R, into sort_workspace = a tnl_plane_grad_binned_workspace(M, R) buffer)
 * while it writes the samples -- the writing wave's lanes are consecutive samples of one ray, the sort's best case.
 * Needs the workspace_rec scratch size.  Follow with tnl_plane_grad_sort_counted on the same stream. */
TNL_API int tnl_march_rays_train_binned(const float *rays_o, const float *rays_d, const uint8_t *grid,
                                 float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                                 uint32_t C, uint32_t H, uint32_t M, const float *nears,
                                 const float *fars, float *xyzs, float *dirs, float *deltas,
                                 int32_t *rays, int32_t *counter, const float *noises,
                                 int32_t *workspace, uint32_t workspace_words, uint32_t R, void *sort_workspace, void *stream);

/* Form of the count pass of the two calls above (no reference predecessor; raymarching.cu:312-398 walks one ray per
 * thread).  0 (default): one WAVEFRONT per ray over 64 consecutive chain points wherever it applies (dt_gamma = 0, at
 * most two cascades, 8-byte aligned bitfield) -- 3x shorter alone (789 -> 263 us at 60 000 base rays), the form for a
 * march the caller waits for; 1: one ray per lane everywhere -- a seventh of the instructions at one wave per SIMD, the
 * form for a march that runs BESIDE other kernels on a second stream (TrainStep's prefetch of the next batch).  Same
 * outputs bit for bit.  Per host thread (thread-local); returns the previous value (any other argument only queries). */
TNL_API int tnl_march_count_form(int form);

/* Launch width of the two wide passes of a march + tile sort that is enqueued BESIDE other kernels (TrainStep's prefetch of
 * the next batch on a second stream): the emit pass of tnl_march_rays_train* (one wavefront per ray, grid-stride over the
 * rays when capped) and the fill pass of tnl_plane_grad_sort* (grid-stride over the samples).  At full width the two flood
 * every CU's wave slots for ~0.5 ms and a main-stream launch of larger workgroups that starts meanwhile waits for slots (base
 * step: 550 us for a 190-us kernel); capped at 2 / 1 workgroups per CU they take about as long by themselves and leave the
 * slots: -0.14 ms per step.  blocks = 0 (default): uncapped, the form for work the caller waits for; < 0 only queries.
 * Per host thread; each returns the previous value.  Same outputs (the fill's order inside a tile list is unordered either way). */
TNL_API int tnl_march_emit_cap(int blocks);
TNL_API int tnl_plane_grad_fill_cap(int blocks);

/* raymarching.h:14 composite_rays_train_forward ; kernel raymarching.cu:501-577.
 * One 64-lane wavefront per ray; transmittance by a wavefront product scan. */
TNL_API int tnl_composite_rays_train_forward(const float *sigmas, const float *rgbs,
                                             const float *deltas, const int32_t *rays, uint32_t M,
                                             uint32_t N, float T_thresh, float *weights_sum,
                                             float *depth, float *image, void *stream);

/* raymarching.h:15 composite_rays_train_backward ; kernel raymarching.cu:602-682.
 * Writes every sample row a ray owns (zero where the reference would have stopped early). */
TNL_API int tnl_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                              const float *sigmas, const float *rgbs,
                                              const float *deltas, const int32_t *rays,
                                              const float *weights_sum, const float *image,
                                              uint32_t M, uint32_t N, float T_thresh,
                                              float *grad_sigmas, float *grad_rgbs, void *stream);

/* raymarching.h:16 march_rays ; kernel raymarching.cu:701-805 */
TNL_API int tnl_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive,
                           const float *rays_t, const float *rays_o, const float *rays_d,
                           float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                           const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                           float *dirs, float *deltas, const float *noises, void *stream);

/* raymarching.h:17 composite_rays ; kernel raymarching.cu:819-905 (in place) */
TNL_API int tnl_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                               int32_t *rays_alive, float *rays_t, const float *sigmas,
                               const float *rgbs, const float *deltas, float *weights_sum,
                               float *depth, float *image, void *stream);

/* Device-side, order-preserving replacement of `rays_alive = rays_alive[rays_alive >= 0]`
 * (reconstruction/nerf/renderer.py:364).  workspace: ceil(n_alive/256) int32.  n_out: device
 * int32 receiving the survivor count. */
TNL_API int tnl_compact_rays(const int32_t *rays_alive, uint32_t n_alive, int32_t *rays_alive_out,
                             int32_t *n_out, int32_t *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * shencoder: replaces sh_encode_forward of aux_libs/shencoder/src/shencoder.h
 * (kernel_sh, shencoder.cu:28-355).  degree C in 1..4 (the path uses 4 -> 16 outputs);
 * returns hipErrorInvalidValue for C > 4 or D != 3.  dy_dx may be NULL (it is on the path).
 * ------------------------------------------------------------------------------------------- */
TNL_API int tnl_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D,
                                  uint32_t C, float *dy_dx, void *stream);
/* shencoder.cu:359-382 (kernel_sh_backward) */
TNL_API int tnl_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D,
                                   uint32_t C, const float *dy_dx, float *grad_inputs, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Wavelet triplane: replaces the torch / pytorch_wavelets operator chain of
 * reconstruction/triplaneencoder/triplane_encoder.py (build_planes :364-405, sample_from_planes_aux
 * :314-332).  wave: 0 haar, 1 bior2.2, 2 bior4.4, 3 bior2.6, 4 bior6.8 (:174-180).
 * ------------------------------------------------------------------------------------------- */

/* One inverse-DWT level: x:[S,n,n], yh:[S,3,n,n] -> out:[S,2n,2n]; computes
 * idwt((pad(2*x), [pad(yh)])) of triplane_encoder.py:379,392-394 (pytorch_wavelets.DWTInverse,
 * mode='zero').  S = 3*channels slices (depthwise). */
TNL_API int tnl_idwt_level_forward(const float *x, const float *yh, uint32_t S, uint32_t n,
                                   int wave, float *out, void *stream);

/* Same level with an fp16 result (n % 4 == 0): used for the finest level, whose only consumer is the fp16
 * texel-major sampler copy, so the fp32 planes are never written. */
TNL_API int tnl_idwt_level_forward_half(const float *x, const float *yh, uint32_t S, uint32_t n,
                                        int wave, void *out_half, void *stream);

/* Adjoint of the above (autograd of SFB2D + pad + 2*): dout:[S,2n,2n] -> dx:[S,n,n], dyh:[S,3,n,n] */
TNL_API int tnl_idwt_level_backward(const float *dout, uint32_t S, uint32_t n, int wave, float *dx,
                                    float *dyh, void *stream);

/* The adjoint level fused with the optimiser (tnl_adam_l1_step's arithmetic and arguments): the level's
 * detail-band gradients are applied to p/m/v:[S,3,n,n] where they are produced and never written; dx:[S,n,n]
 * receives the low-pass gradient for the next level, or, at the coarsest level, pass dx = NULL and the LL
 * parameter ll_p/ll_m/ll_v:[S,n,n] (updated without the L1 term).  abs_sum += sum |p_old| of the detail bands.
 * n % 2 == 0.  Saves the 8 bytes per coefficient of writing and re-reading the gradient. */
TNL_API int tnl_idwt_level_backward_adam(const float *dout, uint32_t S, uint32_t n, int wave, float *dx,
                                         float *p, float *m, float *v, float *ll_p, float *ll_m,
                                         float *ll_v, float step_size, float bias2_sqrt, float beta1,
                                         float beta2, float eps, float inv_scale,
                                         const float *inv_scale_dev, float l1_coef,
                                         const float *found_inf, float *abs_sum, void *stream);

/* Layout change between the reference's (3,C,R,R) planes ("channel-major") and the sampler's
 * texel-major [3,R,R,C] storage.  half_out != 0 stores fp16 (e = 2), else fp32 (e = 4).  Both pointers 16-byte aligned.
 * _win: only the window `roi` (10 host ints {ox[3], oy[3], rw, rh, C, 0}, rw % 64 == 0; as the *_roi entry points below)
 * of each plane is converted -- both arrays keep their whole-plane shape, texels outside the window keep their contents.
 * For the training forward of the module path: no sample of a batch marched through the current occupancy grid lies
 * outside the grid's window (nerf/network.py), so 70 % of the 2.4 GB layout pass at the base geometry is never read. */
TNL_API int tnl_planes_to_texel_major(const float *planes_cm, uint32_t C, uint32_t R, int half_out,
                                      void *planes_tm, void *stream);
TNL_API int tnl_planes_to_texel_major_win(const float *planes_cm, uint32_t C, uint32_t R, int half_out,
                                          void *planes_tm, const int32_t *roi, void *stream);
/* fp16 (3,C,R,R) -> fp16 [3,R,R,C] (C % 8 == 0, R % 8 == 0) */
TNL_API int tnl_planes_half_to_texel_major(const void *planes_cm_half, uint32_t C, uint32_t R,
                                           void *planes_tm_half, void *stream);
/* [3,R,R,C] fp32 gradient -> (3,C,R,R) fp32 */
TNL_API int tnl_planes_to_channel_major(const float *grad_tm, uint32_t C, uint32_t R, float *grad_cm,
                                        void *stream);

/* TriPlaneVolume.forward (triplane_encoder.py:523-530, :314-332): xyz:[N,3] -> feats:[N,3C] fp32,
 * feature index plane*C + c; planes_tm texel-major (fp16 if half_in). */
TNL_API int tnl_triplane_sample_forward(const void *planes_tm, int half_in, const float *xyz,
                                        float bound, uint32_t N, uint32_t C, uint32_t R, float *feats,
                                        void *stream);
/* VJP w.r.t. the planes (torch grid_sampler_2d_backward): atomically accumulates into
 * grad_tm:[3,R,R,C] fp32, which the caller zero-fills. */
TNL_API int tnl_triplane_sample_backward(const float *grad_feats, const float *xyz, float bound,
                                         uint32_t N, uint32_t C, uint32_t R, float *grad_tm,
                                         void *stream);

/* F.grid_sample(bilinear, padding_mode='border', align_corners=True) on texel-major planes [3][R][R][C] for the
 * optional TriPlaneVolume lookups whose coordinates are not the plain axis projection: learn_rotation_axis
 * (triplane_encoder.py:335-362), lbound_auto_scale (:323-326), the nested zoom planes (:453-483).
 * grid: [N][3][CG][2] normalised (gx -> W, gy -> H); CG = 1 (one pair per plane) or C (one per channel).
 * feats: [N][3C].  Backward (grid_sampler_2d_backward semantics): grad_tm [3][R][R][C] fp32 receives atomic adds
 * (caller zero-fills; NULL = not wanted); grad_grid [N][3][CG][2] (NULL = not wanted; zero-filled by the caller when
 * CG = 1, where the channels' contributions are added atomically); a clipped coordinate gets zero gradient. */
TNL_API int tnl_grid_sample_tm_forward(const void *planes_tm, int half_in, const float *grid, uint32_t N, uint32_t C,
                                       uint32_t CG, uint32_t R, float *feats, void *stream);
TNL_API int tnl_grid_sample_tm_backward(const void *planes_tm, int half_in, const float *grid,
                                        const float *grad_feats, uint32_t N, uint32_t C, uint32_t CG, uint32_t R,
                                        float *grad_tm, float *grad_grid, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused field: NeRFNetwork.forward (reconstruction/nerf/network.py:118-147) = triplane lookup
 * + sigma MLP + trunc_exp (activation.py:5-17) + SH-4 + colour MLP + sigmoid, fp16 MFMA with
 * fp32 accumulation (the reference runs its nn.Linear layers under autocast(fp16)).
 * Weights are fp32 masters in the nn.Linear layout [out,in]:
 *   W0:[Hd,3C] W1:[16,Hd] W2:[Hc,31] W3:[Hc,Hc] W4:[3,Hc],  Hd,Hc in {64,128}, C in {16,32,48}.
 * tnl_field_pack converts them to the kernel's fp16 fragment order (call once per optimiser step).
 * ------------------------------------------------------------------------------------------- */
TNL_API uint32_t tnl_field_packed_bytes(uint32_t C, uint32_t Hd, uint32_t Hc);
TNL_API int tnl_field_pack(const float *W0, const float *W1, const float *W2, const float *W3,
                           const float *W4, uint32_t C, uint32_t Hd, uint32_t Hc, void *packed,
                           void *stream);
/* sigma:[M] rgb:[M,3] fp32.  feats_save (tnl_field_feats_save_bytes(M, C, Hd) bytes, may be NULL) keeps
 * the interpolated features for the backward pass (fp16, opaque: ceil(M/32)*32 rows of 3C blocked by the 32-sample
 * tile a wavefront owns, [tile][16-channel step][sample][16], so that each store instruction of the forward covers
 * whole 128-byte lines instead of 32-byte pieces of 32 different rows), followed for hidden 128 by the 16
 * sigma-net outputs per sample (fp16 [M,16]) that the colour half of the split backward starts from.  dirs == NULL or rgb == NULL: density only (NeRFNetwork.density,
 * network.py:149-166); with dirs == NULL and rgb != NULL, rgb receives the 15 geo features ([M,15]).
 * m_actual (device int32, may be NULL): rows >= min(M, *m_actual) are skipped -- march_rays_train's
 * counter[0], so the zero rows that pad the sample buffer to its budget M cost nothing. */
TNL_API uint64_t tnl_field_feats_save_bytes(uint32_t M, uint32_t C, uint32_t Hd);
TNL_API int tnl_field_forward(const void *planes_tm, int half_in, const float *xyz, const float *dirs,
                              float bound, uint32_t M, uint32_t C, uint32_t R, uint32_t Hd,
                              uint32_t Hc, const void *packed, float *sigma, float *rgb,
                              void *feats_save, const int32_t *m_actual, void *stream);
/* grad_sigma:[M], grad_rgb:[M,3] -> grad_tm:[3,R,R,C] fp32 (atomic, caller zero-fills) and
 * gradW:[Hd*3C + 16*Hd + Hc*31 + Hc*Hc + 3*Hc] fp32 in nn.Linear layout, concatenated W0..W4
 * (accumulated, caller zero-fills); workspace bytes from tnl_field_backward_workspace.  sigma / rgb may be
 * NULL (the chain is recomputed from feats_save), except that hidden 128 with dfeat_half != NULL needs
 * sigma (its two-launch backward reads exp(logit) instead of recomputing the sigma net twice).  M must
 * be the M of the forward call that filled feats_save. */
TNL_API uint64_t tnl_field_backward_workspace(uint32_t M, uint32_t C, uint32_t Hd, uint32_t Hc);
TNL_API int tnl_field_backward(const float *grad_sigma, const float *grad_rgb, const float *sigma,
                               const float *rgb, const void *feats_save, const float *xyz,
                               const float *dirs, float bound, uint32_t M, uint32_t C, uint32_t R,
                               uint32_t Hd, uint32_t Hc, const void *packed, float *grad_tm,
                               float *gradW, void *workspace, const int32_t *m_actual, void *dfeat_half,
                               void *stream);

/* Plane-gradient accumulation without global float atomics (csrc/scatter.hip).  When tnl_field_backward is
 * given dfeat_half (fp16, plane-major [3][M][C]) it writes the feature gradient there instead of scattering it; this call
 * then counting-sorts the samples by 32x8-texel tile per plane and lets one workgroup per tile reduce its
 * samples on the matrix cores and store the tile.  EVERY tile of the gradient is written (no zero fill needed): texel-major
 * [3,R,R,C] if channel_major == 0, the reference's (3,C,R,R) otherwise (the adjoint IDWT reads that directly,
 * so the layout-change pass disappears).  channel_major | 2: the caller has zero-filled grad_out (one contiguous fill) and
 * untouched tiles are skipped instead of being zeroed tile by tile -- the faster form for WHOLE planes, of which a scene
 * touches a third (the drop-in autograd path; TrainStep's windowed call keeps the in-kernel zeroes).
 * channel_major | 4 (the *_roi / _reduce entries, with a roi): only the window's tiles are launched and every texel of the
 * window is written, but at its place in the WHOLE (3,C,R,R) array (the rest of grad_out is not touched) -- the gradient
 * the windowed autograd rebuild's backward reads (triplane_encoder._IDWTChainWin).
 * grad_scale multiplies dfeat.  R % 32 == 0, C in {16,32,48}.
 * nonfinite_flag (device int32, may be NULL) is set to 1 if any stored value is inf/nan (GradScaler probe).
 * Replaces torch grid_sampler_2d_backward + the autograd zero fill. */
TNL_API uint64_t tnl_plane_grad_binned_workspace(uint32_t M, uint32_t R);
TNL_API int tnl_plane_grad_binned(const void *dfeat_half, const float *xyz, float bound, uint32_t M,
                                  const int32_t *m_actual, uint32_t C, uint32_t R, float grad_scale,
                                  float *grad_out, int channel_major, int32_t *nonfinite_flag,
                                  void *workspace, void *stream);

/* The two halves of tnl_plane_grad_binned[_roi], for callers that overlap them: _sort (counting sort of the samples
 * by plane tile) needs only the positions and can run as soon as the march has produced them, on another stream;
 * _reduce consumes the sorted workspace together with dfeat.  Same workspace size and contents contract. */
/* Layout of the tile lists inside a tnl_plane_grad_binned_workspace, in int32 units: out[0] = bins (sub-bins
 * included), out[1] = index of offsets[0] (out[0] + 1 entries; a tile's list spans offsets[tile * out[3]] ..
 * offsets[(tile + 1) * out[3]]), out[2] = index of the first list entry (sample ids), out[3] = sub-bins per tile,
 * out[4] = index of the entries' positions (two floats per list entry, parallel to the ids: the sample's clipped
 * texel coordinates on the list's plane, which the reduction reads instead of gathering xyz[id]); M as given to
 * tnl_plane_grad_binned_workspace.  A caller may reorder the entries (ids AND positions alike) INSIDE a tile's span
 * between tnl_plane_grad_sort* and tnl_plane_grad_reduce (trinerflet_amd.nerf.field.order_tile_lists sorts them by
 * sample id: a reproducible summation order). */
TNL_API int tnl_plane_grad_sort_layout(uint32_t M, uint32_t R, int64_t *out);
TNL_API int tnl_plane_grad_sort(const float *xyz, float bound, uint32_t M, const int32_t *m_actual, uint32_t R,
                                void *workspace, void *stream);
/* _sort with the first pass (per-bin counts) already done by tnl_march_rays_train_binned: scan + fill only. */
TNL_API int tnl_plane_grad_sort_counted(const float *xyz, float bound, uint32_t M, const int32_t *m_actual,
                                        uint32_t R, void *workspace, void *stream);
TNL_API int tnl_plane_grad_reduce(const void *dfeat_half, const float *xyz, float bound, uint32_t M, uint32_t C,
                                  uint32_t R,
                                  float grad_scale, float *grad_out, int channel_major, int32_t *nonfinite_flag,
                                  const int32_t *roi, const void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Occupancy ROI variants (no reference predecessor: the reference rebuilds and differentiates whole planes).
 * roi = HOST array of 10 int32 {ox0,ox1,ox2, oy0,oy1,oy2, rw, rh, spp, s0}: per-plane origin and common size of
 * the window of the R x R plane grid that every sample footprint falls into (multiples of 64), spp = slices
 * (channels) per plane and s0 = global index of the first slice passed to a depthwise call (0 unless a rank owns
 * a sub-range of the 3*spp slices); NULL = the whole plane (then each call equals its non-ROI counterpart).
 * ROI-side arrays are COMPACT: [S][rh][rw].
 *   tnl_idwt_level_forward_half_roi    finest level only over the window -> fp16 [S][rh][rw]   (S = 3C slices)
 *   tnl_planes_half_to_texel_major_roi compact fp16 window -> the window of the full fp16 [3,R,R,C] array
 *                                      (texels outside keep their previous contents)
 *   tnl_plane_grad_binned_roi          tiles of the window only -> compact fp32 (3C,rh,rw); channel_major must
 *                                      be 1; samples whose footprint leaves the window are DROPPED (caller's
 *                                      contract: the window covers the occupancy grid's footprint)
 *   tnl_idwt_level_backward_roi        adjoint of the finest level reading the compact gradient (zero outside);
 *                                      bit-identical to the full adjoint of the zero-extended gradient
 * ------------------------------------------------------------------------------------------- */
TNL_API int tnl_idwt_level_forward_half_roi(const float *x, const float *yh, uint32_t S, uint32_t n, int wave,
                                            void *out_half, const int32_t *roi, void *stream);
TNL_API int tnl_planes_half_to_texel_major_roi(const void *planes_roi_half, uint32_t C, uint32_t R,
                                               void *planes_tm_half, const int32_t *roi, void *stream);
TNL_API int tnl_plane_grad_binned_roi(const void *dfeat_half, const float *xyz, float bound, uint32_t M,
                                      const int32_t *m_actual, uint32_t C, uint32_t R, float grad_scale,
                                      float *grad_out, int channel_major, int32_t *nonfinite_flag,
                                      const int32_t *roi, void *workspace, void *stream);
TNL_API int tnl_idwt_level_backward_roi(const float *dout_roi, uint32_t S, uint32_t n, int wave, float *dx,
                                        float *dyh, const int32_t *roi, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused optimiser for the wavelet coefficients: torch.optim.Adam(betas, eps, no weight decay)
 * (reconstruction/main_nerf.py:119) with the wavelet L1 regulariser's gradient
 * (reconstruction/nerf/utils.py:639-655) and GradScaler's unscale folded in:
 *   g = grad * inv_scale + l1_coef * sign(p) ;  m,v,p <- Adam(g) ;  abs_sum += sum |p_old|.
 * step_size = lr / (1 - beta1^t), bias2_sqrt = sqrt(1 - beta2^t).  inv_scale_dev (device float, may be
 * NULL) multiplies inv_scale, so a dynamic loss scale never has to be read back by the host.  If
 * found_inf[0] != 0 nothing is updated (GradScaler.step skip).  grad may be zeroed afterwards (zero_grad != 0).
 * ------------------------------------------------------------------------------------------- */
TNL_API int tnl_adam_l1_step(float *p, float *grad, float *m, float *v, uint64_t n, float step_size,
                             float bias2_sqrt, float beta1, float beta2, float eps, float inv_scale,
                             const float *inv_scale_dev, float l1_coef, const float *found_inf,
                             float *abs_sum, int zero_grad, void *stream);
/* Same update with the bias corrections taken from a DEVICE counter: opt_step_dev[0] = number of optimiser steps
 * taken so far (torch.optim.Adam's `step` state, which GradScaler.step leaves unchanged on a skipped iteration,
 * torch/amp/grad_scaler.py); t = opt_step_dev[0] + 1, step_size = lr / (1 - beta1^t), bias2_sqrt =
 * sqrt(1 - beta2^t) evaluated in double on the device.  The caller advances the counter by (1 - found_inf). */
TNL_API int tnl_adam_l1_step_dev(float *p, float *grad, float *m, float *v, uint64_t n, float lr,
                                 const float *opt_step_dev, float beta1, float beta2, float eps, float inv_scale,
                                 const float *inv_scale_dev, float l1_coef, const float *found_inf,
                                 float *abs_sum, int zero_grad, void *stream);
/* zero_grad: bit 0 = write zeros over the gradient after reading it; bit 1 = store p / m / v even for wavefronts whose
 * p = m = v = g are all zero (the update's fixed point, normally skipped: the same bits) -- for timing probes. */
/* tnl_adam_l1_step_dev with a second L1 coefficient that exists only on the device (optim.FusedAdamL1, fold_l1):
 * l1_scaled_dev[0] = d(scaled loss) / d(sum |p|), collected from the backward of the reference's regulariser
 * (nerf/utils.py:639-655: v.abs().mean() * weight) instead of materialising sign(p) * s as a gradient and adding it to
 * the data gradient (two whole-array passes); g = (grad + l1_scaled_dev[0] * sign(p)) * inv_scale_dev[0] + l1_coef *
 * sign(p), evaluated as grad * inv + (l1 + s * inv) * sign(p).  A non-finite l1_scaled_dev[0] skips the update like
 * found_inf.  May be NULL. */
TNL_API int tnl_adam_l1_step_sink(float *p, float *grad, float *m, float *v, uint64_t n, float lr,
                                  const float *opt_step_dev, float beta1, float beta2, float eps,
                                  const float *inv_scale_dev, float l1_coef, const float *l1_scaled_dev,
                                  const float *found_inf, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Background mix + MSE + gradient in one pass (csrc/loss.hip): pred = image + (1 - weights_sum) * bg
 * (reconstruction/nerf/renderer.py:317), loss = mean over rays and channels of (pred - gt)^2
 * (nerf/utils.py:595,633), grad_pred = d(scale * loss)/d pred, grad_weights_sum = -sum_c grad_pred * bg.
 * bg = bg_rays[n] ([N,3], --train_rand_bg) if non-NULL else bg_color; inv_norm = 1 / (3 * rays in the global batch);
 * scale_dev (device float, may be NULL) = the GradScaler's loss scale; mse_accum (device float) += this launch's share
 * of the loss (the caller zeroes it).
 * ------------------------------------------------------------------------------------------- */
TNL_API int tnl_mse_loss(const float *image, const float *weights_sum, const float *gt_rgb, float bg_color,
                         const float *bg_rays, uint32_t N, float inv_norm, const float *scale_dev, float *pred,
                         float *grad_pred, float *grad_weights_sum, float *mse_accum, void *stream);

/* mean |x| of a coefficient tensor and its gradient (the wavelet L1 regulariser's term, nerf/utils.py:639-655
 * `val.abs().mean()`), for the drop-in autograd path: forward = one read pass (fixed partition of x over <= 2048
 * workgroups, partial sums in double: reproducible), backward grad_x = sign(x) * grad_out[0] / n (sign(0) = 0) = one read
 * and one write pass -- instead of torch's abs, mean, sign, mul kernels.  x, grad_x 16-byte aligned; workspace of
 * tnl_abs_mean_workspace() bytes; out / grad_out device floats. */
TNL_API uint64_t tnl_abs_mean_workspace(void);
TNL_API int tnl_abs_mean_forward(const float *x, uint64_t n, void *workspace, float *out, void *stream);
TNL_API int tnl_abs_mean_backward(const float *x, uint64_t n, const float *grad_out, float *grad_x, void *stream);
/* found_inf[0] = 1 if any of x[0..n) is inf or nan (left as it is otherwise): the inf check of
 * torch.amp.GradScaler.step (grad_scaler.py: _check_inf_per_device -> _amp_foreach_non_finite_check_and_unscale_ with a
 * scale of 1, which also writes every element back) as a read-only pass, 4 B per element.  x 16-byte aligned. */
TNL_API int tnl_nonfinite_check(const float *x, uint64_t n, float *found_inf, void *stream);

/* Measurement aid (bench.py): streaming copy of `bytes` (a multiple of 16) from src to dst, 16 bytes per lane,
 * non-temporal -- what this box's memory system gives a plain copy, printed beside the 8 TB/s HBM3E spec.  No
 * reference counterpart. */
TNL_API int tnl_copy_probe(const void *src, void *dst, uint64_t bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Scalar bookkeeping of one optimisation step (csrc/stepstate.hip), replacing torch.cuda.amp.GradScaler's
 * unscale_/step/update bookkeeping and the wavelet-L1 value (reconstruction/nerf/utils.py:1158-1166, :641-655)
 * -- two dozen tiny dependent launches -- by three.  All pointers are device memory.
 *   prologue: small_grad[0..n) = 0 (the MLP gradient), *inv_scale = 1 / *scale, *abs_sum = 0, *nonfinite = 0,
 *             *mse = 0.
 *   probe:    *probe = sum |g0| + sum |g1| (g1 may be NULL with n1 = 0), +inf if *nonfinite != 0 (NULL: ignored);
 *             *found_inf = 1.0 when *probe is not finite else 0.0 (GradScaler.unscale_'s found_inf).  With several
 *             ranks the caller all-reduces *probe and recomputes found_inf.
 *   epilogue: *opt_steps += 1 - *found_inf; if update_scale, GradScaler.update() with the given growth / backoff
 *             factors and interval (_amp_update_scale_ semantics); *reg = *abs_sum * l1_coef (abs_sum may be NULL).
 * ------------------------------------------------------------------------------------------- */
TNL_API int tnl_step_prologue(const float *scale, float *inv_scale, float *abs_sum, int32_t *nonfinite, float *mse,
                              float *small_grad, uint32_t n, void *stream);
TNL_API int tnl_scaler_probe(const float *g0, uint32_t n0, const float *g1, uint32_t n1, const int32_t *nonfinite,
                             float *probe, float *found_inf, void *stream);
TNL_API int tnl_step_epilogue(const float *found_inf, float *opt_steps, float *scale, int32_t *growth_tracker,
                              float growth, float backoff, int32_t growth_interval, int32_t update_scale,
                              const float *abs_sum, float l1_coef, float *reg, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Ray batches from a device-resident pixel pool.  One launch replaces get_rays over whole
 * images (reconstruction/nerf/utils.py:65-149), shuffle_data (CPU randperm + gather of every tensor, :228-236),
 * select_batch (slice + H2D, :238-243) and the background blend of train_step / eval_step (:559-577, :690-695).
 *   poses [B,4,4] fp32 cam2world (device); intrinsics_host = {fx, fy, cx, cy} (HOST); images [B,H,W,channels]
 *   fp32 in [0,1] or uint8 (images_u8 != 0: value / 255), channels 3 or 4, may be NULL when gt_rgb is NULL.
 *   Pixel of ray n:  pix[n] if pix != NULL (device int64, b*H*W + y*W + x);  else g = first + n and
 *   perm_total == 0 ? g : permute(g)  with perm_total = B*H*W and the epoch's perm_key (tnl_permute_index: a keyed
 *   bijection of [0, total), so consecutive batches of one epoch partition the pool like shuffle_data's randperm).
 *   gt_rgb (may be NULL) = rgb*a + bg*(1-a) for 4 channels (bg = bg_rand[n] if bg_rand != NULL else bg_color),
 *   rgb for 3.  pix_out (may be NULL) receives the pixel ids.
 * ------------------------------------------------------------------------------------------- */
TNL_API uint64_t tnl_permute_index(uint64_t g, uint64_t total, uint64_t key);
TNL_API int tnl_ray_batch(const float *poses, const float *intrinsics_host, uint32_t B, uint32_t H, uint32_t W,
                          const void *images, int channels, int images_u8, const int64_t *pix, uint64_t first,
                          uint64_t perm_total, uint64_t perm_key, uint32_t N, float bg_color,
                          const float *bg_rand, float *rays_o, float *rays_d, float *gt_rgb, int64_t *pix_out,
                          void *stream);

/* Marching work one ray may spend per trip of the one-kernel render before it pauses and rides along without a sample
 * (a probe counts 8 units, an add of the empty-cell skip 1; default 96; 0 = unbounded: a ray marches to its next sample in
 * one go, the form of rounds 4-5; -1 only queries).  A ray entering the volume at max_steps = 4096 walks ~30 empty cells of
 * 28-55 dependent adds each: unbounded, the wave's other 31 rays wait for it every time a slot is refilled.  The samples
 * are the same to the bit either way.  Process-wide tuning knob; returns the previous value. */
TNL_API int tnl_render_work(int units);

/* ---------------------------------------------------------------------------------------------
 * The inference render as one persistent kernel (csrc/render.hip; replaces the alive-ray loop of
 * reconstruction/nerf/renderer.py:338-372 and its march_rays / forward / composite_rays launches): every ray is marched
 * (the loop kernels' state machine), evaluated (the fused field on 32-sample tiles of 32 different rays) and composited
 * (raymarching.cu:853-904) inside the kernel; rays are taken from a queue.  rays_o / rays_d [N,3], nears / fars [N]
 * (fars may be clipped to the occupied box, tnl_clip_fars), noises [N] or NULL (the first iteration's perturbation),
 * queue: one int32 of device scratch.  Outputs as tnl_composite_rays leaves them: weights_sum [N], depth [N] (sum of
 * weight * t), image [N,3], every element written.  planes_tm / packed / C / R / Hd as tnl_field_forward.
 * A ray ends when it leaves [near, far) without a sample, after the sample at which its transmittance falls below
 * T_thresh, or after max_steps samples.
 * ------------------------------------------------------------------------------------------- */
TNL_API int tnl_render_rays(const void *planes_tm, int half_in, uint32_t C, uint32_t R, uint32_t Hd, uint32_t Hc,
                            const void *packed, const float *rays_o, const float *rays_d, const float *nears,
                            const float *fars, uint32_t N, const uint8_t *grid, float bound, float dt_gamma,
                            uint32_t max_steps, uint32_t cascades, uint32_t H, float T_thresh, float density_scale,
                            const float *noises, int32_t *queue, float *weights_sum, float *depth, float *image,
                            void *stream);

/* ---------------------------------------------------------------------------------------------
 * Inference loop with its state on the device (reference loop: reconstruction/nerf/renderer.py:338-372, which reads
 * the survivor count back every iteration).  state = device int32[4] {n_alive, n_step, step, rows}; the caller
 * initialises {N, 0, 0, 0}.  One iteration = plan -> march -> (field forward over `rows` = state[3], passed as its
 * m_actual) -> composite -> compact.  Every launch is sized for N rays and reads the live sizes from `state`:
 *   tnl_infer_plan          n_alive = 0 once step >= max_steps; n_step = max(min(N / n_alive, 8), min_step) (min_step = 1 is
 *                           the reference's rule; up to 8 regroups the same sample sequences into fewer, wider
 *                           iterations: identical colours for every ray that ends before the max_steps cap; buffers
 *                           then need min_step * N + 128 rows); rows = n_alive*n_step.  min_step > 8 returns
 *                           hipErrorInvalidValue (n_step is at most 8 everywhere in this loop)
 *   tnl_march_rays_dev      tnl_march_rays for the first n_alive entries of rays_alive; zero-fills the n_alive*n_step
 *                           sample rows it owns first (raymarching.py:337-339); noises may be NULL (no perturbation).
 *                           t_scratch (rows_cap floats, rows_cap = the row capacity of xyzs / dirs / deltas; may be
 *                           NULL): the march then only records each sample's t and a second kernel writes the rows with
 *                           one thread per row -- the same bits, contiguous stores instead of 16 scattered 4-byte
 *                           stores per sample
 *   tnl_composite_rays_dev  tnl_composite_rays
 *   tnl_compact_rays_dev    ordered compaction of the survivors into rays_alive_out, then step += n_step and
 *                           n_alive = survivors; workspace: (N + 255) / 256 + 2 int32
 * Buffers: xyzs/dirs [min_step*N + 128, 3], deltas [min_step*N + 128, 2] (n_alive * n_step <= max(min_step, 1) * N).  The host may poll state[0]
 * whenever it likes; iterations enqueued after the rays ran out do nothing.
 * ------------------------------------------------------------------------------------------- */
TNL_API int tnl_infer_plan(int32_t *state, uint32_t N, uint32_t max_steps, uint32_t min_step, void *stream);
TNL_API int tnl_march_rays_dev(const int32_t *state, uint32_t N, const int32_t *rays_alive, const float *rays_t,
                               const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                               uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid, const float *fars,
                               float *xyzs, float *dirs, float *deltas, const float *noises, float *t_scratch,
                               uint32_t rows_cap, void *stream);
TNL_API int tnl_composite_rays_dev(const int32_t *state, uint32_t N, float T_thresh, int32_t *rays_alive,
                                   float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                                   float *weights_sum, float *depth, float *image, void *stream);
TNL_API int tnl_compact_rays_dev(int32_t *state, uint32_t N, const int32_t *rays_alive, int32_t *rays_alive_out,
                                 int32_t *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Gradient-support chain (no reference predecessor).  With an occupancy window the plane gradient is zero outside
 * it, so every level's coefficient gradient is zero outside the rectangle of coarse tiles the window reaches:
 *   tnl_idwt_level_backward_win  adjoint of one level whose fine-side input is valid only inside `win` (10 ints as
 *       for the *_roi calls; strided = 0: compact [S][rh][rw] array, 1: full-size array, window multiples of 4).
 *       If out_rect (HOST, 8 ints: ox[3], oy[3], w, h in coarse coordinates) is given it receives the rectangle of
 *       reachable tiles (common size over the planes) and NOTHING is stored outside it -- dx / dyh keep their old
 *       contents there; the next (coarser) level takes out_rect as its strided window.
 *   tnl_adam_l1_step_rect        tnl_adam_l1_step_dev over one level [S][bands][n][n] (n a power of two) that
 *       reads the gradient only inside the rectangle and uses 0 outside (24 instead of 28 bytes per coefficient).
 * Together: bit-identical to the whole-plane adjoint + Adam, without storing or re-reading the zeros.
 * ------------------------------------------------------------------------------------------- */
/* Forward counterpart for the levels below the finest: only the window `win` (10 ints, multiples of 64, fine
 * coordinates of this level) of the full-size fp32 output (S, 2n, 2n) is computed; the rest keeps its contents.
 * The caller chooses each level's window as the next level's window halved and grown by the filter halo. */
TNL_API int tnl_idwt_level_forward_win(const float *x, const float *yh, uint32_t S, uint32_t n, int wave, float *out,
                                       const int32_t *win, void *stream);
TNL_API int tnl_idwt_level_backward_win(const float *dout, uint32_t S, uint32_t n, int wave, float *dx, float *dyh,
                                        const int32_t *win, int strided, int32_t *out_rect, void *stream);

/* Levels with n >= walk_min_n (and n % 8 == 0) run the column-walk IDWT / adjoint kernels (a thread owns a column and
 * slides a register window down the plane: no staged halo tile), smaller ones the LDS-tiled kernels.  Default 512
 * (0 restores it).  A tuning / test knob (tests force the walk kernels on small planes); process-global, set it
 * before launching work.  The kernel choice depends on n only, so windowed and whole-plane calls of one level always
 * take the same kernel (their results are bit-identical). */
TNL_API int tnl_idwt_set_walk_min_n(uint32_t walk_min_n);
/* the value in force (after the defaulting rule of the setter): levels with n >= it and n % 8 == 0 are column-walk levels */
TNL_API int tnl_idwt_get_walk_min_n(void);
/* The windowed level calls restricted to the pieces anything reads.  `spans`: device int32 [3][n / 8][2] on the level's
 * own n x n (coarse) grid -- for plane pl and coarse rows 8g .. 8g+7 the coarse columns [lo, end) -- or, for the
 * layout change, [3][R / 8][2] on the R x R plane grid (what tnl_occupancy_row_extents writes); NULL = no restriction.
 *   forward   results outside the pieces are not produced (the output keeps what it held); half_out / win / strided
 *             as tnl_idwt_level_forward_half_roi (1, roi, 0) and tnl_idwt_level_forward_win (0, win, 1)
 *   backward  arguments of tnl_idwt_level_backward_win; the band gradients outside the pieces are not produced, the
 *             LL gradient dx is written as zero there (which it is, exactly, when the pieces hold every coefficient
 *             a data gradient can reach)
 * Levels that run the tile kernels (n < walk_min_n) ignore the spans.  TrainStep builds the tables level by level from
 * the occupied cells' projection; no reference counterpart. */
TNL_API int tnl_idwt_level_forward_spans(const float *x, const float *yh, uint32_t S, uint32_t n, int wave, void *out,
                                         int half_out, const int32_t *win, int strided, const int32_t *spans,
                                         void *stream);
TNL_API int tnl_idwt_level_backward_spans(const float *dout, uint32_t S, uint32_t n, int wave, float *dx, float *dyh,
                                          const int32_t *win, int strided, int32_t *out_rect, const int32_t *spans,
                                          void *stream);
/* A column-walk level of the windowed adjoint with the optimiser in its epilogue (TrainStep's live / deferred split,
 * steady state): the band gradients of the level's LIVE pieces are applied to p / m / v:[S,3,n,n] where they are produced
 * and never written -- tnl_adam_l1_step_live_bands' arithmetic with the step's scalars read from `step_rec` (one slot of
 * the ring tnl_adam_record_step writes; record the step BEFORE this call) -- 24 instead of 32 bytes per live coefficient
 * over the adjoint + optimiser pair.  live_rect: the level's live rectangle (8 host ints {x[3], y[3], w, h}; x, w % 4 == 0,
 * y, h % 8 == 0), a superset of the gradient's support (the rectangle tnl_idwt_level_backward_spans returns for this
 * window): outside the support the gradient is exactly zero, as the unfused pass takes it.  band_table / spans: the live
 * band pieces in the two forms of tnl_adam_l1_step_live_bands and tnl_idwt_level_backward_spans (device), or both NULL:
 * the whole rectangle.  dx:[S,n,n] receives the low-pass gradient over the live rectangle.  abs_sum += sum |p| of the
 * updated pieces.  Replaces reconstruction/nerf/utils.py:1166-1173 (scaler.step(optimizer)) for this level's
 * coefficients together with autograd's backward of triplane_encoder.py:392-394.  n must be a column-walk level. */
TNL_API int tnl_idwt_level_backward_live_adam(const float *dout, uint32_t S, uint32_t n, int wave, float *dx,
                                              const int32_t *win, int strided, const int32_t *live_rect,
                                              const int32_t *spans, float *p, float *m, float *v,
                                              const int32_t *band_table, uint32_t nb, const float *step_rec,
                                              float beta1, float beta2, float eps, float inv_scale,
                                              const float *inv_scale_dev, float l1_coef, const float *found_inf,
                                              float *abs_sum, void *stream);
TNL_API int tnl_planes_half_to_texel_major_spans(const void *planes_roi_half, uint32_t C, uint32_t R,
                                                 void *planes_tm_half, const int32_t *roi, const int32_t *spans,
                                                 void *stream);
/* Launch-shape knobs of the walk kernels, for A/B measurements (tools/bench_idwt.py); results do not depend on them.
 * key 1: coarse rows per phase of the forward kernel (4 or 8); key 2: XCD-aware block order (0 / 1);
 * key 3: coarse rows per workgroup (multiple of 8; 0 = automatic); key 4: form of the forward kernel -- 0 = a thread owns one
 * coarse column, 2 / 4 = the one-wave pair form (two columns per thread, 8-byte loads) with 2 / 4 coarse rows per phase,
 * -1 = the build's default. */
TNL_API int tnl_idwt_set_tuning(int key, int value);
TNL_API int tnl_adam_l1_step_rect(float *p, float *grad, float *m, float *v, uint32_t S, uint32_t bands, uint32_t n,
                                  uint32_t spp, uint32_t s0, const int32_t *rect, float lr,
                                  const float *opt_step_dev, float beta1, float beta2, float eps, float inv_scale,
                                  const float *inv_scale_dev, float l1_coef, const float *found_inf,
                                  float *abs_sum, void *stream);
/* For TrainStep's captured steps (train.py, graph=True): the same passes with nothing passed by value that changes from
 * step to step.  _record_step_dev: tnl_adam_record_step with the learning rate read from device memory; _step_rec /
 * _step_rect_rec: tnl_adam_l1_step_dev / _rect with the step's scalars read from the ring slot `step_rec` that record
 * wrote (4 floats: lr / (1 - beta1^t), sqrt(1 - beta2^t), skip, pad -- the expressions the kernels otherwise evaluate
 * themselves: the same bits). */
TNL_API int tnl_adam_record_step_dev(float *ring, int32_t slot, const float *lr_dev, const float *opt_step_dev, float beta1,
                                     float beta2, const float *found_inf, void *stream);
TNL_API int tnl_adam_l1_step_rec(float *p, float *grad, float *m, float *v, uint64_t n, const float *step_rec, float beta1,
                                 float beta2, float eps, const float *inv_scale_dev, float l1_coef,
                                 const float *found_inf, float *abs_sum, void *stream);
TNL_API int tnl_adam_l1_step_rect_rec(float *p, float *grad, float *m, float *v, uint32_t S, uint32_t bands, uint32_t n,
                                      uint32_t spp, uint32_t s0, const int32_t *rect, const float *step_rec, float beta1,
                                      float beta2, float eps, const float *inv_scale_dev, float l1_coef,
                                      const float *found_inf, float *abs_sum, void *stream);
/* Live / deferred split of a level's optimiser pass between two density-grid refreshes (TrainStep; no reference
 * counterpart -- the reference runs torch.optim.Adam over every coefficient every step, main_nerf.py:119).
 * A coefficient outside `live` (8 host ints like `rect`: the footprint the windowed plane rebuild reads united with
 * the gradient's rectangle) is neither read nor reached by a data gradient until the window changes, and its update
 * p, m, v <- adam(p, l1 sign(p), m, v) depends on nothing but its own three numbers and the step's scalars:
 *   tnl_adam_l1_step_live   ONE launch over n_levels (<= 8) levels of the flat arrays p / grad / m / v: level k
 *                           starts at element offsets[k], is [S][bands[k]][sizes[k]][sizes[k]], is updated inside
 *                           live[8k .. 8k+8) only (the whole level: {0,0,0, 0,0,0, n, n}), reads its gradient inside
 *                           grad_rect[8k ..) (0 elsewhere) and uses l1_coefs[k].  step_rec (may be NULL): the ring
 *                           slot tnl_adam_record_step wrote for this step -- the bias-corrected scalars are then read
 *                           instead of being evaluated by every workgroup.  Nothing outside `live` is touched.
 *   tnl_adam_record_step    ring[slot] (4 floats per slot, 16 slots) = this step's {lr / (1 - beta1^t),
 *                           sqrt(1 - beta2^t), found_inf != 0} from the same device counter, BEFORE the step's
 *                           epilogue advances it.
 *   tnl_adam_l1_catchup     replays ring[0 .. count) (count <= 16), oldest first, for every coefficient outside
 *                           `live`: one 24-byte pass instead of `count`; abs_sums[r] += sum |p| as step r saw it.
 * The same operations in the same order as the per-step pass: p, m, v are bit-identical after the catch-up. */
TNL_API int tnl_adam_l1_step_live(float *p, float *grad, float *m, float *v, uint32_t S, uint32_t spp, uint32_t s0,
                                  uint32_t n_levels, const uint64_t *offsets, const uint32_t *sizes,
                                  const uint32_t *bands, const int32_t *live, const int32_t *grad_rect,
                                  const float *l1_coefs, float lr, const float *opt_step_dev, const float *step_rec,
                                  float beta1, float beta2, float eps, float inv_scale, const float *inv_scale_dev,
                                  const float *found_inf, float *abs_sum, void *stream);
TNL_API int tnl_adam_record_step(float *ring, int32_t slot, float lr, const float *opt_step_dev, float beta1,
                                 float beta2, const float *found_inf, void *stream);
/* tnl_adam_record_step for optim.FusedAdamL1's live / deferred split (round 5): the record's fourth float carries the
 * step's folded L1 coefficient in true units, l1_scaled_dev[0] * inv_scale_dev[0] (tnl_adam_l1_step_sink's product; a
 * non-finite one marks the step skipped); tnl_adam_l1_step_live[_bands] with that record as step_rec and
 * tnl_adam_l1_catchup[_bands] add it to their l1_coef -- records written by tnl_adam_record_step carry 0 there. */
TNL_API int tnl_adam_record_step_l1(float *ring, int32_t slot, float lr, const float *opt_step_dev, float beta1,
                                    float beta2, const float *found_inf, const float *l1_scaled_dev,
                                    const float *inv_scale_dev, void *stream);
TNL_API int tnl_adam_l1_catchup(float *p, float *m, float *v, uint32_t S, uint32_t bands, uint32_t n, uint32_t spp,
                                uint32_t s0, const int32_t *live, const float *ring, int32_t count, float beta1,
                                float beta2, float eps, float l1_coef, float *abs_sums, void *stream);
/* The same two passes over a live SET finer than the rectangle: the rectangle's rows in groups of 8 ("bands",
 * nb = live h / 8 <= 128), each with its own column piece.  A band table is device int32[5 nb + 1]:
 *   [0 .. nb]              prefix sums of the bands' float4 counts 8 * w_b / 4   ([nb] = float4s per slice = band_quads)
 *   [nb + 1 + b]           w_b / 4 (0 = nothing live in the band)
 *   [2 nb + 1 + pl nb + b] first column of band b on plane pl (multiple of 4, piece inside the rectangle)
 * band_tables[k] / band_table NULL = the whole rectangle (the functions above).  TrainStep builds the tables from
 * tnl_occupancy_row_extents, halving and growing by the filter reach level by level. */
TNL_API int tnl_adam_l1_step_live_bands(float *p, float *grad, float *m, float *v, uint32_t S, uint32_t spp, uint32_t s0,
                                        uint32_t n_levels, const uint64_t *offsets, const uint32_t *sizes,
                                        const uint32_t *bands, const int32_t *live, const int32_t *grad_rect,
                                        const int32_t *const *band_tables, const uint32_t *band_quads,
                                        const float *l1_coefs, float lr, const float *opt_step_dev, const float *step_rec,
                                        float beta1, float beta2, float eps, float inv_scale, const float *inv_scale_dev,
                                        const float *found_inf, float *abs_sum, void *stream);
TNL_API int tnl_adam_l1_catchup_bands(float *p, float *m, float *v, uint32_t S, uint32_t bands, uint32_t n, uint32_t spp,
                                      uint32_t s0, const int32_t *live, const int32_t *band_table, const float *ring,
                                      int32_t count, float beta1, float beta2, float eps, float l1_coef, float *abs_sums,
                                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRINERFLET_HIP_H */
