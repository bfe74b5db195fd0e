/*
 * emernerf_hip.h -- C ABI of libemernerf_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * This is the drop-in boundary of the EmerNeRF volumetric-rendering hot path.  Each entry point
 * replaces one native call the reference makes into tiny-cuda-nn / nerfacc / cuBLAS-via-torch;
 * the reference interface it replaces is cited next to it (paths relative to /root/reference).
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless named host_*;
 *   - the caller owns every buffer (the library never allocates device memory, never syncs);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); kernels are enqueued on it;
 *   - return 0 on success, a negative EMER_E_* code on failure; emer_last_error() gives the text;
 *   - re-entrant across streams/devices (no mutable globals except the thread-local error string).
 */
#ifndef EMERNERF_HIP_H
#define EMERNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMER_MAX_LEVELS 32

#define EMER_OK 0
#define EMER_E_INVALID (-1)  /* bad argument / unsupported configuration */
#define EMER_E_LAUNCH (-2)   /* hipLaunchKernel failed                   */

/* dtype tags for table / gradient storage */
#define EMER_F32 0
#define EMER_F16 1

/* activation tags (epilogues of emer_linear_*) */
#define EMER_ACT_NONE 0
#define EMER_ACT_RELU 1
#define EMER_ACT_SIGMOID 2
#define EMER_ACT_TRUNC_EXP 3 /* y = exp(x - 1); bwd g*min(y, e^15)  (nerf_utils.py:59-75, radiance_field.py:28) */

/* s->t transforms (third_party/nerfacc_prop_net.py:299-315) */
#define EMER_STOT_UNIFORM 0
#define EMER_STOT_UNIFORM_LINDISP 1
#define EMER_STOT_LINDISP 2
#define EMER_STOT_SQRT 3
#define EMER_STOT_LOG 4
#define EMER_STOT_UNIFORM_LINDISP_0 5

const char *emer_last_error(void);
/* Measurement hook: the next emer_hashgrid_fwd / emer_hashgrid_bwd_params_sliced call of this thread records the two
 * caller-owned HIP events (hipEvent_t) immediately around its main kernel.  One-shot. */
int emer_profile_next(void *start_event, void *stop_event);
int emer_version(void);

/* ------------------------------------------------------------------------------------------------
 * Multiresolution hash grid (replaces tcnn.Encoding{otype: HashGrid}:
 *   radiance_fields/encodings.py:133-160, third_party/tcnn_modules.py:122 (fwd), :161-163 (bwd),
 *   :420-423 (_C.create_encoding)).  Level table rules: SURVEY.md Appendix A.1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct emer_grid_desc {
    uint32_t n_dims;            /* D: 2..4                                  */
    uint32_t n_levels;          /* L <= EMER_MAX_LEVELS                     */
    uint32_t n_features;        /* F: 1, 2, 4 or 8                          */
    uint32_t log2_hashmap_size; /* T                                        */
    uint32_t base_resolution;
    float per_level_scale;
    float scale[EMER_MAX_LEVELS];
    uint32_t res[EMER_MAX_LEVELS];
    uint32_t size[EMER_MAX_LEVELS];   /* entries in the level                */
    uint32_t offset[EMER_MAX_LEVELS]; /* first entry of the level            */
    uint32_t hashed[EMER_MAX_LEVELS]; /* 1 -> coherent-prime hash, 0 -> dense */
    uint32_t n_entries;               /* n_params = n_entries * n_features   */
} emer_grid_desc;

/* Host-only: fill the level table (replaces _C.create_encoding + native.n_params()). */
int emer_grid_desc_init(emer_grid_desc *host_desc, uint32_t n_dims, uint32_t n_levels,
                        uint32_t n_features, uint32_t log2_hashmap_size, uint32_t base_resolution,
                        float per_level_scale);

/* Encode.  x [N,D] f32 in [0,1]; params flat (level, entry, feature), dtype param_dtype.
 * out element (n, l, f) is written at out[n*out_stride_n + l*out_stride_l + f] (f32):
 *   row-major [N, L*F] (the reference's layout): stride_n = L*F, stride_l = F;
 *   level-major [L][N][F] (coalesced, what the fused heads read): stride_n = F, stride_l = N*F.
 * slice_masks (may be NULL): [L][rows][ceil(N/64)] u64 (rows = emer_hashgrid_mask_rows) by-product consumed by emer_hashgrid_bwd_params_sliced:
 *   one bitmap per (level l, LDS slice s); bit (n % 64) of word n/64 is set iff a corner of sample n
 *   lives in slice s of level l.  The buffer must hold EMER_SLICE_MASK_SCRATCH more words behind the bitmaps
 *   (work cursors of the backward).
 * Replaces native.fwd (tcnn_modules.py:122). */
#define EMER_SLICE_MASK_SCRATCH 2064 /* 8 work cursors + (builds with owner pacing) 64 x 64 trip counters, as 32-bit words in 64-bit units */
int emer_hashgrid_fwd(const emer_grid_desc *host_desc, const float *x, const void *params,
                      int param_dtype, float *out, int64_t out_stride_n, int64_t out_stride_l,
                      uint64_t *slice_masks, int64_t n, void *stream);

/* dParams[l, idx, f] += w_corner * dOut[n, l, f]  (atomic scatter; grad is NOT zeroed here).
 * grad dtype: EMER_F32 (f32 atomics) or EMER_F16 (packed half2 atomics, F even).
 * Replaces the params path of native.bwd (tcnn_modules.py:161-163). */
int emer_hashgrid_bwd_params(const emer_grid_desc *host_desc, const float *x, const float *dout,
                             int64_t dout_stride_n, int64_t dout_stride_l, void *grad,
                             int grad_dtype, int64_t n, void *stream);

/* Same result as emer_hashgrid_bwd_params with an f32 gradient table, but OVERWRITES grad (no
 * memset needed) and uses no global atomics: each workgroup owns one LDS-resident table slice (accumulated in double) and
 * streams its 1-bit-per-sample slice bitmap ("owner computes"; see csrc/hashgrid.hip).  The training path.
 * slice_masks [L][rows][ceil(N/64)] (+ EMER_SLICE_MASK_SCRATCH words the call overwrites), rows = emer_hashgrid_mask_rows(desc):
 * from emer_hashgrid_fwd / emer_hashgrid_slice_masks for the same x.
 * dout is level-major: dout_stride_n == n_features (dout_stride_l free); n < 2^28. */
int emer_hashgrid_bwd_params_sliced(const emer_grid_desc *host_desc, const float *x,
                                    const float *dout, int64_t dout_stride_n,
                                    int64_t dout_stride_l, uint64_t *slice_masks,
                                    float *grad, int64_t n, void *stream);
/* [r5] The work-item plan of the owner-computes backward (host arithmetic, no GPU): slices and sample ranges per level, total items as
 * the return value (negative: error).  n_ranges > 1 on a hashed level = the tail items of a grid without enough dense filler (the xyzt
 * tables: their finest levels are cut in 2 or 4 so that the last round of the 256 owners is full).  Arrays of n_levels entries or NULL. */
int emer_hashgrid_sliced_plan(const emer_grid_desc *g, uint32_t *n_slices, uint32_t *n_ranges);
/* [r5] emer_hashgrid_bwd_params_sliced that ADDS to grad instead of overwriting it: the second and later evaluations of one encoder in
 * a step (the flow table is evaluated at the sample positions and at the warped positions, radiance_field.py:553-620; chunked training).
 * Replaces a table-sized temporary and autograd's add. */
int emer_hashgrid_bwd_params_sliced_add(const emer_grid_desc *g, const float *x, const float *dout, int64_t stride_n,
                                        int64_t stride_l, uint64_t *slice_masks, float *grad, int64_t n, void *stream);
/* The same for the levels [level_begin, level_end) only (a contiguous range of the table: entries offset[level_begin] ..): the other
 * levels' entries are neither read nor written.  Calls that partition the levels give the one-call result; a data-parallel trainer
 * starts the collective of the first call's range while the second call computes. */
int emer_hashgrid_bwd_params_sliced_levels(const emer_grid_desc *host_desc, const float *x, const float *dout,
                                           int64_t dout_stride_n, int64_t dout_stride_l, uint64_t *slice_masks,
                                           float *grad, int64_t n, int32_t level_begin, int32_t level_end, void *stream);
/* The level k at which to cut for two such calls, [k, n_levels) first: the finest levels whose work items fill one round of the
 * resident owner workgroups (a cut elsewhere lengthens the pair of launches by a round); 0 = do not cut.  Host arithmetic only. */
int emer_hashgrid_sliced_split_level(const emer_grid_desc *host_desc);
int emer_hashgrid_slice_masks(const emer_grid_desc *host_desc, const float *x,
                              uint64_t *slice_masks, int64_t n, void *stream);
/* 1 when the owner-computes backward covers the grid: every level cuts into LDS slices (128 KiB of double accumulators
 * each) that share at most 256 bitmaps (one per slice up to 256 slices per level, the case of every shipped grid; beyond
 * that 2^k neighbouring slices share one), i.e. up to 16384 slices per level.  Else use emer_hashgrid_bwd_params. */
int emer_hashgrid_sliced_supported(const emer_grid_desc *host_desc);
/* Bitmap rows per level: 64, or 256 when a level has more than 64 LDS slices (T = 2^20 with 4 features: every slice
 * keeps its own bitmap).  The bitmaps hold n_levels * rows * ceil(n / 64) words + EMER_SLICE_MASK_SCRATCH.  0: unsupported grid. */
int emer_hashgrid_mask_rows(const emer_grid_desc *host_desc);

/* dX[n, d] = sum_l scale_l sum_f dOut * d(interp)/dx.  Replaces the input path of native.bwd
 * (needed by the flow configs, radiance_fields/radiance_field.py:572-608). */
int emer_hashgrid_bwd_input(const emer_grid_desc *host_desc, const float *x, const void *params,
                            int param_dtype, const float *dout, int64_t dout_stride_n,
                            int64_t dout_stride_l, float *dx, int64_t n, void *stream);
/* The same gradient without a second gather pass [r4]: emer_hashgrid_fwd_jac is emer_hashgrid_fwd on fp32 tables that also stores,
 * for the rows jac_row0 .. n - 1 (the rows whose position needs a gradient), jac [n_levels][n - jac_row0][n_features][n_dims] =
 * d out / d x; emer_hashgrid_bwd_input_jac contracts it with dOut (dout: pointer of the first of the n_rows rows, same strides).
 * The encoding is bitwise emer_hashgrid_fwd's.  Same reference path as above (native.bwd's input gradient). */
int emer_hashgrid_fwd_jac(const emer_grid_desc *host_desc, const float *x, const float *params, float *out,
                          int64_t out_stride_n, int64_t out_stride_l, uint64_t *slice_masks, float *jac,
                          int64_t jac_row0, int64_t n, void *stream);
int emer_hashgrid_bwd_input_jac(const emer_grid_desc *host_desc, const float *jac, const float *dout,
                                int64_t dout_stride_n, int64_t dout_stride_l, float *dx, int64_t n_rows, void *stream);

/* Layout glue: level-major [L][N][F] <-> row-major [N, L*F] (the layout tcnn_modules.py:263 returns).
 * to_row_major != 0: src is level-major, dst row-major; 0: the reverse.  src != dst. */
int emer_layout_transpose(const float *src, float *dst, int32_t n_levels, int64_t n,
                          int32_t n_features, int to_row_major, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Scene contraction + ray sample points
 *   (radiance_fields/nerf_utils.py:13-28, radiance_field.py:278-300 and :828-835,
 *    render_utils.py:316-318,341).
 * ---------------------------------------------------------------------------------------------- */
/* out = contract(pos) with rows outside (0,1)^3 zeroed.  aabb: 6 floats (device). */
int emer_contract_fwd(const float *pos, const float *aabb, int unbounded, float *out, int64_t n,
                      void *stream);
/* dpos = J^T dout (zero for rejected rows). */
int emer_contract_bwd(const float *pos, const float *aabb, int unbounded, const float *dout,
                      float *dpos, int64_t n, void *stream);
/* Flow warp of the temporal aggregation (radiance_fields/radiance_field.py:567-580): the xyzt query points of the flow branch in one
 * launch.  x3 [3n][4] = [normed | t], [contract(positions + flow[:, 0:3] * noise) | clamp(t + time_diff * noise, 0, 1)],
 * [contract(positions + flow[:, 3:6] * noise) | clamp(t - time_diff * noise, 0, 1)]; x2 [2n][4] = rows n .. 3n of x3 again (the flow
 * table's query as its own tensor).  positions / normed [n][3], timestamps / noise [n], flow [n][6], aabb [6]. */
int emer_flow_warp_fwd(const float *positions, const float *normed, const float *timestamps, const float *flow,
                       const float *noise, float time_diff, const float *aabb, int unbounded, float *x3, float *x2,
                       int64_t n, void *stream);
/* dflow [n][6] from the input gradients of the two consumers, dx3 [3n][4] (rows < n ignored) and dx2 [2n][4] (either may be NULL);
 * positions, timestamps and noise carry no gradient. */
int emer_flow_warp_bwd(const float *positions, const float *flow, const float *noise, const float *aabb, int unbounded,
                       const float *dx3, const float *dx2, float *dflow, int64_t n, void *stream);
/* positions[r,s,:] = origins[r] + dirs[r] * (t_starts[r,s] + t_ends[r,s]) / 2, then contract.
 * Writes normed [R*S, out_dim] (out_dim 3, or 4 with times[r] appended as the 4th column);
 * positions_out (optional, may be NULL) receives the un-contracted world positions [R*S,3]. */
int emer_ray_points(const float *origins, const float *dirs, const float *t_starts,
                    const float *t_ends, const float *times, const float *aabb, int unbounded,
                    float *normed, int out_dim, float *positions_out, int64_t n_rays,
                    int32_t n_samples, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Proposal sampler (replaces nerfacc.pdf.importance_sampling + _transform_stot:
 *   third_party/nerfacc_prop_net.py:153,156,172-173,299-339).  Frozen spec: SURVEY.md A.2.
 * ---------------------------------------------------------------------------------------------- */
/* Floats of LDS per ray available to emer_importance_sample_points (device LDS per workgroup / 4 rays; 10240 on gfx950): the caller
 * takes emer_importance_sample + emer_ray_points when 2 * n_edges_in + n_intervals_out + 1 exceeds it.  0: no device. */
int64_t emer_importance_sample_points_capacity(void);
/* [r5] emer_importance_sample (interval form: t_starts / t_ends [R][n]) and, in the same launch, the sample points of the new intervals:
 * what emer_ray_points computes from its result (render_utils.py:316-318,341) -- normed [R][n][3] (scene contraction by aabb /
 * unbounded), positions [R][n][3] or NULL.  Bitwise the two separate calls.  2 m + n + 1 <= 10240. */
int emer_importance_sample_points(const float *vals, const float *cdfs, int64_t n_rays, int32_t n_edges_in, int32_t n_intervals_out,
                                  const float *jitter, float *s_out, float *t_starts, float *t_ends, float t_min, float t_max,
                                  int stot_type, const float *origins, const float *dirs, const float *aabb, int unbounded,
                                  float *normed, float *positions, void *stream);
/* vals/cdfs [R,m] -> s_out [R,n+1] (sorted edges in s) and, if t_out != NULL, t = stot(s_out):
 *   t_ends == NULL: t_out [R,n+1] = the edges;
 *   t_ends != NULL: t_out [R,n] = interval starts, t_ends [R,n] = interval ends (what PropNetEstimator.sampling
 *                   returns, nerfacc_prop_net.py:176-179, without slicing copies).
 * jitter: NULL (centre of bin) or [R] U(0,1) per ray (stratified). Bit-exact vs the oracle. */
int emer_importance_sample(const float *vals, const float *cdfs, int64_t n_rays, int32_t m,
                           int32_t n_intervals, const float *jitter, float *s_out, float *t_out,
                           float *t_ends, float t_min, float t_max, int stot_type, void *stream);
int emer_stot(const float *s, int64_t n, float t_min, float t_max, int stot_type, float *t,
              void *stream);

/* ------------------------------------------------------------------------------------------------
 * Proposal-network supervision (replaces PropNetEstimator.compute_loss and its helpers:
 *   third_party/nerfacc_prop_net.py:22-34 blur_stepfun, :37-60 sorted_interp_quad, :181-238 compute_loss,
 *   :342-362 _pdf_loss).  One launch per proposal level computes the loss AND its gradient.
 * ---------------------------------------------------------------------------------------------- */
/* s_final [R,n+1] sample edges of the final level (s space), trans [R,n] its transmittance (cdf = 1 - [trans, 0],
 * no gradient); s_prop / cdf_prop [R,m+1] edges and cdf of one proposal level.
 *   anti_aliased != 0: zip-NeRF loss with blur half-width pulse_width:
 *       sum_rays sum_j max(w_s - w_p, 0)^2 / (w_p + 1e-5),  w_p = diff(cdf_prop),
 *       w_s = diff(quadratic interpolation of the blurred final histogram at s_prop);
 *   anti_aliased == 0: _pdf_loss, sum over the n final intervals of max(w - w_outer, 0)^2 / (w + 1e-7).
 * loss_rays [R] (may be NULL) = per-ray sums * scale; loss_out (may be NULL; needs loss_rays) = their sum in a fixed
 * order (accumulate != 0: added to the value already there); d_cdf_prop [R,m+1] (may be NULL) = d(scale * sum)/d cdf_prop.
 * The caller passes scale = loss_scaler / (R * m) (resp. R * n) for the reference's .mean() * loss_scaler. */
int emer_prop_loss(const float *s_final, const float *trans, int32_t n_final, const float *s_prop,
                   const float *cdf_prop, int32_t n_prop, float pulse_width, int anti_aliased, int64_t n_rays,
                   float scale, float *loss_rays, float *loss_out, int accumulate, float *d_cdf_prop,
                   void *stream);
/* out[0] = (accumulate ? out[0] : 0) + sum(x[0..n)): one workgroup, fixed order, double accumulation. */
int emer_reduce_sum(const float *x, int64_t n, int accumulate, float *out, void *stream);
/* y = x * dev_scalar[0] * host_scale (dev_scalar may be NULL): gradients scaled by an upstream 0-dim gradient
 * without a host round trip. */
int emer_scale(const float *x, const float *dev_scalar, float host_scale, float *y, int64_t n, void *stream);
/* dst_f16[i] = (half) src[i], round to nearest even: the fp16 copy of an fp32 master table that the encoders read in
 * half-precision mode (tcnn casts the master parameters per call, third_party/tcnn_modules.py:223-233,257-260).  dst 16-byte
 * aligned; src may sit at any float (a table inside a flat parameter buffer). */
int emer_cast_f32_f16(const float *src, void *dst_f16, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Dense volume rendering (replaces nerfacc.render_transmittance_from_density /
 *   render_weight_from_density / accumulate_along_rays on (R,S) tensors:
 *   radiance_fields/render_utils.py:35-43,73-77,103-115,159-282; nerfacc_prop_net.py:165-168).
 * ---------------------------------------------------------------------------------------------- */
/* Per ray: sigma_dt = sigma*(t_end-t_start); T = exp(-excl_cumsum); alpha = 1-exp(-sigma_dt);
 * w = T*alpha.  Any of weights/trans/alphas/cdfs/ray_stats may be NULL.
 *   cdfs [R,S+1]  = 1 - [T, 0]                        (nerfacc_prop_net.py:166-168)
 *   ray_stats [R,4] = (sum w, sum w*mid, median_depth, reserved), mid = (t_start+t_end)/2;
 *   median_depth follows render_utils.py:107-115 (first s with cumsum(w) >= 0.5, clamped);
 *   t_mid / t_dist [R,S] (may be NULL) = (t_start+t_end)/2, t_end-t_start: extras["t_vals"], ["t_dist"] (:84-85). */
int emer_render_weights_fwd(const float *t_starts, const float *t_ends, const float *sigma,
                            int64_t n_rays, int32_t n_samples, float *weights, float *trans,
                            float *alphas, float *cdfs, float *ray_stats, float *t_mid, float *t_dist,
                            void *stream);
/* Given dL/dweights, dL/dtrans, dL/dalphas (any may be NULL) and dL/d(sum w), dL/d(sum w*mid) per ray
 * (columns 0, 1 of d_ray_stats [R,4], may be NULL) produce dL/dsigma.  alphas is a differentiable output because the
 * reference forms weights = trans * alphas itself (render_utils.py:73-77). */
int emer_render_weights_bwd(const float *t_starts, const float *t_ends, const float *sigma,
                            const float *d_weights, const float *d_trans, const float *d_alphas,
                            const float *d_ray_stats, int64_t n_rays, int32_t n_samples, float *d_sigma,
                            void *stream);

/* [r4] The static model's `rendering` as ONE launch each way (render_utils.py:73-122,158-159,217-220): the scan of
 * emer_render_weights_fwd, the 3-channel emer_accumulate_fwd and emer_ray_epilogue_fwd (opacity = clamp(sum w, 1e-6, 1),
 * depth = sum(w mid) / opacity, median depth, rgb_out = sum(w rgb) + rgb_sky (1 - opacity)); results bitwise those of the three
 * calls.  rgb / rgb_out NULL: geometry only.  trans, t_mid, t_dist, median_depth, rgb_sky may be NULL.  ray_stats [R,4] is saved
 * for the backward.  The backward takes the gradients of rgb_out [R,3], opacity [R], depth [R] and -- from other consumers of the
 * extras -- weights / trans [R,S] (each may be NULL) and writes d_sigma [R,S], d_rgb [R,S,3] and d_rgb_sky [R,3] (may be NULL). */
int emer_composite_rgb_fwd(const float *t_starts, const float *t_ends, const float *sigma, const float *rgb,
                           const float *rgb_sky, int64_t n_rays, int32_t n_samples, float *weights, float *trans,
                           float *t_mid, float *t_dist, float *ray_stats, float *opacity, float *depth,
                           float *median_depth, float *rgb_out, void *stream);
int emer_composite_rgb_bwd(const float *t_starts, const float *t_ends, const float *sigma, const float *rgb,
                           const float *rgb_sky, const float *weights, const float *ray_stats, const float *d_rgb_out,
                           const float *d_opacity, const float *d_depth, const float *d_weights, const float *d_trans,
                           int64_t n_rays, int32_t n_samples, float *d_sigma, float *d_rgb, float *d_rgb_sky, void *stream);
/* Static / dynamic / shadow colour blend + accumulation of `rendering` (radiance_fields/render_utils.py:125-175):
 *   a = static_density / (density + 1e-6), b = dynamic_density / (density + 1e-6),
 *   acc_rgb[r] = sum_s w (a rgb_s (1 - shadow) + b rgb_d),   acc_shadow_sq[r] = sum_s w shadow^2
 * weights / densities / shadow_ratio [R,S] (shadow_ratio may be NULL = 0), rgb [R,S,3]; acc_rgb [R,3], acc_shadow_sq [R]
 * (may be NULL). */
int emer_blend_accumulate_fwd(const float *weights, const float *density, const float *static_density,
                              const float *dynamic_density, const float *static_rgb, const float *dynamic_rgb,
                              const float *shadow_ratio, int64_t n_rays, int32_t n_samples, float *acc_rgb,
                              float *acc_shadow_sq, void *stream);
/* Gradients of the above w.r.t. every input (each output pointer may be NULL); d_acc_* may be NULL (zero). */
int emer_blend_accumulate_bwd(const float *weights, const float *density, const float *static_density,
                              const float *dynamic_density, const float *static_rgb, const float *dynamic_rgb,
                              const float *shadow_ratio, const float *d_acc_rgb, const float *d_acc_shadow_sq,
                              int64_t n_rays, int32_t n_samples, float *d_weights, float *d_density,
                              float *d_static_density, float *d_dynamic_density, float *d_static_rgb,
                              float *d_dynamic_rgb, float *d_shadow_ratio, void *stream);

/* [r4] The same blend for the C-channel features of the decomposed feature head (render_utils.py:247-252, `dino_feat`):
 * acc [R,C] = sum_s w (sigma_s / (sigma + 1e-6) feat_s + sigma_d / (sigma + 1e-6) feat_d); feat_* [R,S,C].  Backward: any output
 * pointer may be NULL. */
int emer_blend_accumulate_wide_fwd(const float *weights, const float *density, const float *static_density,
                                   const float *dynamic_density, const float *static_feat, const float *dynamic_feat,
                                   int64_t n_rays, int32_t n_samples, int32_t n_channels, float *acc, void *stream);
int emer_blend_accumulate_wide_bwd(const float *weights, const float *density, const float *static_density,
                                   const float *dynamic_density, const float *static_feat, const float *dynamic_feat,
                                   const float *d_acc, int64_t n_rays, int32_t n_samples, int32_t n_channels,
                                   float *d_weights, float *d_density, float *d_static_density, float *d_dynamic_density,
                                   float *d_static_feat, float *d_dynamic_feat, void *stream);
/* Per-ray epilogue of `rendering` (radiance_fields/render_utils.py:102-105,217-226):
 *   opacity = clamp(sum w, 1e-6, 1); depth = (sum w*mid) / opacity; median_depth = ray_stats[:,2];
 *   rgb = acc_rgb + rgb_sky * (1 - opacity)   (rgb_sky NULL: rgb = acc_rgb; rgb NULL: geometry only, lidar rays).
 * ray_stats [R,4] from emer_render_weights_fwd; acc_rgb / rgb_sky / rgb [R,3]; opacity / depth / median_depth [R]. */
int emer_ray_epilogue_fwd(const float *ray_stats, const float *acc_rgb, const float *rgb_sky, int64_t n_rays,
                          float *opacity, float *depth, float *median_depth, float *rgb, void *stream);
/* d_ray_stats [R,4] = gradient of ray_stats for emer_render_weights_bwd (columns 2, 3 zero); d_rgb_sky [R,3] (may be NULL).
 * The gradient of acc_rgb is d_rgb itself.  Any of d_opacity / d_depth / d_rgb may be NULL (zero). */
int emer_ray_epilogue_bwd(const float *ray_stats, const float *rgb_sky, const float *d_opacity, const float *d_depth,
                          const float *d_rgb, int64_t n_rays, float *d_ray_stats, float *d_rgb_sky, void *stream);
/* Pixel losses of a training step (loss/base.py:83-146 rgb L2, :149-185 opacity-based sky loss):
 *   loss = w_rgb * mean((rgb - pixels)^2) + w_sky * mean(binary_cross_entropy(opacity, 1 - sky_mask))
 * with torch's clamping of the logs at -100.  rgb/pixels [R,3] (NULL: no rgb term), opacity/sky_mask [R] (NULL: no
 * sky term).  loss_rays [R] scratch, loss_out [1] (fixed summation order). */
int emer_pixel_loss_fwd(const float *rgb, const float *pixels, const float *opacity, const float *sky_mask,
                        int64_t n_rays, float w_rgb, float w_sky, float *loss_rays, float *loss_out, void *stream);
/* d_rgb [R,3], d_opacity [R] (either may be NULL), multiplied by the device scalar upstream[0] (NULL: 1). */
int emer_pixel_loss_bwd(const float *rgb, const float *pixels, const float *opacity, const float *sky_mask,
                        int64_t n_rays, float w_rgb, float w_sky, const float *upstream, float *d_rgb,
                        float *d_opacity, void *stream);
/* Mean-type regularisers of the dynamic / flow / feature models in one pass:
 *   loss = (base ? base[0] : 0)
 *        + c_dyn    * mean(dyn_density[0..n_dyn))                       dynamic-density sparsity, loss/base.py:394-398 (called at
 *                                                                       train_emernerf.py:683-688; lidar step :797-802)
 *        + c_shadow * mean(shadow[0..n_shadow))                         shadow sparsity, same class (train_emernerf.py:689-694)
 *        + c_feat   * mean((feat - feat_gt)^2 over n_feat)              feature L2, loss/base.py:83-146 (train_emernerf.py:676-682)
 *        + c_cycle  * mean((fwd_flow + fwd_pred_bwd_flow)^2 + (bwd_flow + bwd_pred_fwd_flow)^2 over n_flow)
 *                                                                       flow cycle consistency, train_emernerf.py:700-716
 *                                                                       (the reference's 0.5 * 0.01 enters through c_cycle).
 * A term is absent when its first pointer (dyn_density / shadow / feat / fwd_pred_bwd_flow) is NULL.  workspace: at least
 * EMER_REG_MAX_BLOCKS floats; loss_out [1]; fixed summation order (bit-reproducible). */
#define EMER_REG_MAX_BLOCKS 1024
int emer_reg_losses_fwd(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow,
                        float c_shadow, const float *feat, const float *feat_gt, int64_t n_feat, float c_feat,
                        const float *fwd_flow, const float *fwd_pred_bwd_flow, const float *bwd_flow,
                        const float *bwd_pred_fwd_flow, int64_t n_flow, float c_cycle, const float *base, float *workspace,
                        float *loss_out, void *stream);
/* Gradients of the four terms, each multiplied by upstream[0] (NULL: 1) * grad_scale; every output may be NULL.  fwd_flow /
 * bwd_flow receive no gradient (detached in the reference). */
int emer_reg_losses_bwd(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow,
                        float c_shadow, const float *feat, const float *feat_gt, int64_t n_feat, float c_feat,
                        const float *fwd_flow, const float *fwd_pred_bwd_flow, const float *bwd_flow,
                        const float *bwd_pred_fwd_flow, int64_t n_flow, float c_cycle, const float *upstream,
                        float grad_scale, float *d_dyn_density, float *d_shadow, float *d_feat,
                        float *d_fwd_pred_bwd_flow, float *d_bwd_pred_fwd_flow, void *stream);
/* [r5] The same pair with the cycle term read straight from the flow MLP's outputs (no slice copies, ONE gradient tensor): flow6
 * [n_rows][6] = (forward | backward flow) at the sample positions (constants), flow2_6 [2 n_rows][6] = the flow at the
 * forward-warped points (rows 0..n_rows: its columns 3..5 are forward_pred_backward_flow) and at the backward-warped points (rows
 * n_rows..: columns 0..2 are backward_pred_forward_flow), radiance_field.py:585-592, train_emernerf.py:700-716.  Same value and
 * summation order as emer_reg_losses_fwd on the four slices; d_flow2_6 [2 n_rows][6] is written entirely (zeros in the unread
 * column blocks).  flow2_6 == NULL: no cycle term. */
int emer_reg_losses_fwd6(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow,
                         float c_shadow, const float *feat, const float *feat_gt, int64_t n_feat, float c_feat,
                         const float *flow6, const float *flow2_6, int64_t n_rows, float c_cycle, const float *base,
                         float *workspace, float *loss_out, void *stream);
int emer_reg_losses_bwd6(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow,
                         float c_shadow, const float *feat, const float *feat_gt, int64_t n_feat, float c_feat,
                         const float *flow6, const float *flow2_6, int64_t n_rows, float c_cycle, const float *upstream,
                         float grad_scale, float *d_dyn_density, float *d_shadow, float *d_feat, float *d_flow2_6,
                         void *stream);
/* Lidar-ray supervision (train_emernerf.py:770-808): depth loss (loss/base.py:188-271, "l2", normalised by max_depth,
 * mean over rays with 0.01 < range < max_depth) + line-of-sight loss (loss/base.py:430-464: empty-space and near-surface
 * terms with margin epsilon, times the fraction of rays with range > 0).
 *   loss = w_depth * depth_loss + w_sight * sight_loss   (w_sight = loss_coef * coef_decay)
 * depth / lidar_ranges [R], weights / t_vals [R,S].  workspace: R + 2 floats.  loss_out [1] (may be NULL);
 * d_depth [R] / d_weights [R,S] (may be NULL) are multiplied by the device scalar upstream[0] (NULL: 1). */
int emer_lidar_loss(const float *depth, const float *lidar_ranges, const float *weights, const float *t_vals,
                    int64_t n_rays, int32_t n_samples, float epsilon, float max_depth, float w_depth, float w_sight,
                    const float *upstream, float *workspace, float *loss_out, float *d_depth, float *d_weights,
                    void *stream);
/* out[r,c] = sum_s w[r,s] * values[r,s,c]   (values == NULL: C = 1, out[r] = sum_s w). */
int emer_accumulate_fwd(const float *weights, const float *values, int64_t n_rays,
                        int32_t n_samples, int32_t n_channels, float *out, void *stream);
/* d_weights[r,s] (+)= sum_c d_out[r,c]*values[r,s,c]; d_values[r,s,c] = w[r,s]*d_out[r,c]. */
int emer_accumulate_bwd(const float *weights, const float *values, const float *d_out,
                        int64_t n_rays, int32_t n_samples, int32_t n_channels, float *d_weights,
                        float *d_values, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Small MLP heads (replaces the torch.nn.Linear / ReLU / Sigmoid / trunc_exp chains of
 *   radiance_fields/radiance_field.py:74-198 and radiance_fields/mlp.py:7-46, i.e. cuBLAS GEMM +
 *   elementwise launches).  fp32-exact MFMA (v_mfma_f32_32x32x2_f32): bitwise an fmaf chain.
 * ---------------------------------------------------------------------------------------------- */
/* Y[M,N] = act(X[M,K] @ W[N,K]^T + bias[N]).  ldx/ldy: row strides (elements).  W row-major
 * [N,K] (torch Linear layout).  bias may be NULL.  aux_density (may be NULL): receives
 * exp(pre[:,0] - 1), the density read off geometry feature 0 (radiance_field.py:28,422). */
int emer_linear_fwd(const float *x, int64_t ldx, const float *w, const float *bias, float *y,
                    int64_t ldy, int64_t m, int32_t n, int32_t k, int act, float *aux_density,
                    void *stream);
/* Backward of the above.  dy [M,N] (ld lddy), y [M,N] = saved forward output (for act').
 *   dpre = dy * act'(y)  [+ on column 0: d_aux_density * min(aux_density, e^15)]  is formed on the fly while
 *          operand tiles are staged -- it is never written to HBM (dy may be NULL when only d_aux_density flows);
 *   dx [M,K] = dpre @ W                                       (skipped when dx == NULL)
 *   dw [N,K] += dpre^T @ X ; dbias [N] += column sums of dpre (skipped when dw == NULL)
 * workspace: emer_linear_bwd_workspace(m, n, k) floats (per-row-block partial sums of dw/dbias; a second
 * kernel reduces them -- no global atomics). */
int64_t emer_linear_bwd_workspace(int64_t m, int32_t n, int32_t k);
int emer_linear_bwd(const float *dy, int64_t lddy, const float *y, int64_t ldy, const float *x,
                    int64_t ldx, const float *w, float *workspace, float *dx, int64_t lddx,
                    float *dw, float *dbias, int64_t m, int32_t n, int32_t k, int act,
                    const float *d_aux_density, const float *aux_density, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused MLP chains.  One launch evaluates up to EMER_CHAIN_MAX_LAYERS dense layers on 16-row tiles with
 * every activation kept in LDS between layers (replaces the per-layer nn.Linear / ReLU / cat / Sigmoid
 * launches of radiance_field.py:74-198,315,622-658 and mlp.py:38-46, and their autograd).  The same entry
 * point runs the dgrad chains of the backward: weights are addressed through (w_sn, w_sk) strides, so
 * W^T needs no copy, and `mask` applies relu' from a saved activation.
 * ---------------------------------------------------------------------------------------------- */
#define EMER_CHAIN_MAX_LAYERS 6
#define EMER_CHAIN_MAX_SEGS 3

/* A column segment [col, col+width) of the per-row working buffer, filled from global memory before the
 * layers run (several segments = a virtual torch.cat).
 *   mode 0: value(row, c) = ptr[(row / row_div) * ld + c]         (row_div = samples per ray for per-ray data)
 *   mode 1: level-major grid encoding [L][n_total][f]: ptr[((c / f) * n_total + row) * f + c % f]
 * fix_a/fix_b (may be NULL): buffer[row][col] += fix_a[row] * min(fix_b[row], e^15)   (density gradient merge) */
typedef struct emer_chain_seg {
    const float *ptr;
    int64_t ld;
    int64_t n_total;
    const float *fix_a;
    const float *fix_b;
    int32_t col, width, row_div, mode, f;
    int32_t dst_col; /* emer_wgrad_segmented only: column of dw where this segment's first column lands */
} emer_chain_seg;

/* out[:, out_col:out_col+N] (op)= act(in[:, in_col:in_col+K] @ W^T + bias), W element (n,k) = w[n*w_sn + k*w_sk].
 * accumulate != 0: out += result.  mask (may be NULL): out *= (mask[row*mask_ld + col] > 0) after act.
 * store (may be NULL): copy out[:, store_col:store_col+store_n] to global; store_mode 0 row-major (store_ld),
 *   1 level-major [L][store_ntotal][store_f].  store_exp0 (may be NULL): exp(out[row][0] - 1) per row. */
typedef struct emer_chain_layer {
    const float *w;
    int64_t w_sn, w_sk;
    const float *bias;
    const float *mask;
    int64_t mask_ld;
    float *store;
    int64_t store_ld, store_ntotal;
    float *store_exp0;
    int32_t in_col, K, out_col, N, act, accumulate;
    int32_t store_col, store_n, store_mode, store_f;
} emer_chain_layer;

typedef struct emer_chain_desc {
    emer_chain_seg segs[EMER_CHAIN_MAX_SEGS];
    emer_chain_layer layers[EMER_CHAIN_MAX_LAYERS];
    int32_t n_segs, n_layers;
    int32_t buf_cols; /* columns of the per-row working buffer (<= 512) */
    int32_t _pad;
} emer_chain_desc;

int emer_mlp_chain(const emer_chain_desc *host_desc, int64_t n_rows, void *stream);

/* dW[N,K] += dpre[M,N]^T @ X[M,K], dbias[N] += colsum(dpre), with X given as up to EMER_CHAIN_MAX_SEGS
 * column segments (modes 0 and 1; a virtual concat).  col0 (may be NULL): [m] values that REPLACE column 0 of dpre
 * (the geometry-feature-0 gradient with the density gradient merged in, as emer_neck_bwd writes it).
 * dw has row stride ld_dw and segment s lands at columns [dst_col_s, dst_col_s + width_s) -- so the gradient of a
 * virtually concatenated operand can be ACCUMULATED straight into column blocks of a wider weight-gradient matrix
 * (e.g. a parameter's .grad).  workspace: emer_linear_bwd_workspace(m, n, k) floats. */
int emer_wgrad_segmented(const float *dpre, int64_t ld_dpre, const float *col0,
                         const emer_chain_seg *host_segs, int32_t n_segs, float *workspace, float *dw,
                         int64_t ld_dw, float *dbias, int64_t m, int32_t n, int32_t k, void *stream);

/* ---- register-resident fused heads (hidden width 64; csrc/mlp_fused.hip) ------------------------------
 * The per-sample hot heads of RadianceField / DensityField as single kernels in which a wave keeps its 16 rows
 * in registers across all layers ("transposed chaining"; fp32 results from exact three-term bf16 splits on the bf16
 * matrix pipe, csrc/mlp_fused.hip).  Callers fall back to emer_mlp_chain for
 * shapes these do not cover (emer_neck_supported == 0).
 *
 * Neck: base_mlp = Sequential(Linear(L*F, 64), ReLU, Linear(64, n_out)) fed by the LEVEL-MAJOR grid encoding
 * enc_lm [L][n][F] (radiance_field.py:74-80,89-96,302-318; DensityField :808-812,836-840).
 *   n_out 64 / 128: out0 [n][64] = features 0..63, out1 [n][64] = features 64..127 (the reference's
 *     geo / semantic split, :400), dens [n] = exp(feature0 - 1) (trunc_exp, :28,315) when non-NULL;
 *   n_out 1: dens [n] = exp(out - 1) only (proposal network).
 *   h1 [n][64] receives the post-ReLU hidden activations (saved for the backward; may be NULL). */
int emer_neck_supported(int32_t n_levels, int32_t n_feat, int32_t hidden, int32_t n_out);
int emer_neck_fwd(const float *enc_lm, int32_t n_levels, int32_t n_feat, int64_t n, const float *w0,
                  const float *b0, const float *w1, const float *b1, int32_t n_out, float *h1,
                  float *out0, float *out1, float *dens, void *stream);
/* Data gradients of emer_neck_fwd.  d0 / d1 [n][64]: gradients of features 0..63 / 64..127 (NULL = zero);
 * ddens [n] (NULL = none) enters feature 0 as ddens * min(dens, e^15) (nerf_utils.py:69-72).
 * Writes dpre0 [n][64] (gradient at the hidden pre-activation, the wgrad operand), denc_lm [L][n][F] and,
 * for n_out == 1, dpre1 [n] (gradient at the single output's pre-activation).  dcol0 [n] (may be NULL) receives
 * d0[:, 0] + ddens * min(dens, e^15): column 0 of the output-layer wgrad operand (emer_wgrad_segmented's col0). */
int emer_neck_bwd(const float *d0, const float *d1, const float *ddens, const float *dens, const float *h1,
                  int32_t n_levels, int32_t n_feat, int64_t n, const float *w0, const float *w1,
                  int32_t n_out, float *dpre1, float *dcol0, float *dpre0, float *denc_lm, void *stream);

/* Backward of emer_neck_fwd (n_out == 64, or [r5] 128: geometry | semantic features of the feature configs, d1 = the gradient of
 * outputs 64..127) INCLUDING the weight gradients (autograd of radiance_field.py:74-80,89-96): one
 * kernel reads d0 / d1 / ddens and the forward's input enc_lm once, RECOMPUTES the hidden layer from enc_lm (the forward need not
 * store it: h1 = NULL there), writes denc_lm [L][n][F] and keeps dW1 = d^T h1, db1, dW0 = dpre0^T enc, db0 in the waves'
 * accumulators (neither h1 nor the pre-activation gradient dpre0 ever goes to memory); per-workgroup partials are summed
 * into dw1 [n_out][ld_dw1 >= 64], db1 [n_out], dw0 [64][ld_dw0 >= L*F], db0 [64] with += semantics (what AccumulateGrad does; the
 * targets may be a parameter's .grad).  workspace: emer_neck_bwd_fused_workspace(...) floats; n * L * F < 2^30.
 * emer_neck_bwd_fused_supported == 0 (the density MLP): emer_density_bwd_fused, or emer_neck_bwd + emer_wgrad_segmented. */
int emer_neck_bwd_fused_supported(int32_t n_levels, int32_t n_feat, int32_t hidden, int32_t n_out);
int64_t emer_neck_bwd_fused_workspace(int32_t n_levels, int32_t n_feat, int64_t n, int32_t n_out);
int emer_neck_bwd_fused(const float *d0, const float *d1, const float *ddens, const float *dens, const float *enc_lm, int32_t n_levels,
                        int32_t n_feat, int64_t n, const float *w0, const float *b0, const float *w1, int32_t n_out,
                        float *denc_lm, float *workspace, float *dw0, int64_t ld_dw0, float *db0, float *dw1, int64_t ld_dw1,
                        float *db1, void *stream);

/* Plain heads: nn.Sequential(Linear(k0, 64), ReLU, [Linear(64, 64), ReLU,] Linear(64, n_out)[, Sigmoid]) -- the flow MLP
 * (radiance_field.py:101-111, fed by the level-major xyzt encoding), the shadow head (:148-153) and the feature heads
 * (:192-198, dino_head / dino_sky_head) -- register-resident like the neck.  x: row-major [n][ldx] (n_feat == 0; k0 and
 * ldx multiples of 4) or level-major [n_levels][n][n_feat] (n_feat == 4, k0 = n_levels * n_feat).  Weights in torch
 * Linear layout; for two layers pass the output layer as w1 / b1 and w2 = b2 = NULL.  h1 / h2 [n][64]: post-ReLU
 * activations saved for the backward (NULL: inference).  emer_rmlp_supported == 0: use emer_mlp_chain. */
int emer_rmlp_supported(int32_t n_layers, int32_t k0, int32_t n_feat, int32_t hidden, int32_t n_out);
int emer_rmlp_fwd(const float *x, int64_t ldx, int32_t n_levels, int32_t n_feat, int32_t k0, int64_t n, int32_t n_layers,
                  const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                  int32_t n_out, int32_t final_act, float *h1, float *h2, float *out, int64_t ldo, void *stream);
/* Data gradients of emer_rmlp_fwd.  dlast [n][ldd]: gradient at the LAST pre-activation (the caller multiplies by
 * sigmoid').  Writes dpre0 [n][64] (and dpre1 [n][64] for three layers), the operands of emer_wgrad_segmented, and -- when
 * dx is non-NULL -- the input gradient in the input's layout (row-major [n][lddx] or level-major). */
int emer_rmlp_bwd(const float *dlast, int64_t ldd, const float *h1, const float *h2, int32_t n_levels, int32_t n_feat,
                  int32_t k0, int64_t n, int32_t n_layers, const float *w0, const float *w1, const float *w2, int32_t n_out,
                  float *dpre1, float *dpre0, float *dx, int64_t lddx, void *stream);
/* [r5] The whole backward of emer_rmlp_fwd in ONE kernel, weight gradients included (replaces autograd of the flow MLP,
 * radiance_field.py:101-111, and of the shadow head, :148-153: torch's sigmoid backward, emer_rmlp_bwd, one weight-gradient launch
 * and one reduction per layer).  Stacks with n_out <= 16, or three layers on a row-major input with n_out <= 64 in multiples of 4 --
 * the feature heads, :192-198 (emer_rmlp_bwd_fused_supported; wide outputs need 16-byte aligned rows of dout / out).  dout [n][ldd]: gradient of the OUTPUT;
 * sigmoid' is applied from the saved `out` [n][ldo] (final_act == EMER_ACT_NONE: out may be NULL).  The hidden layers are
 * recomputed from x (pass h1 = h2 = NULL to emer_rmlp_fwd).  Writes dx in x's layout when non-NULL; ACCUMULATES (+=) dw0
 * [64][ld_dw0 >= k0], db0 [64], dw1 / db1 (three layers: [64][ld_dw1 >= 64], [64]; two layers: the output layer, [n_out][ld_dw1], [n_out])
 * and, for three layers, dw2 [n_out][ld_dw2 >= 64], db2 [n_out]; bias gradient pointers may be NULL.  workspace:
 * emer_rmlp_bwd_fused_workspace(...) floats (0: this call is not covered -- use emer_rmlp_bwd). */
int emer_rmlp_bwd_fused_supported(int32_t n_layers, int32_t k0, int32_t n_feat, int32_t hidden, int32_t n_out);
int64_t emer_rmlp_bwd_fused_workspace(int32_t n_layers, int32_t k0, int32_t n_feat, int64_t n, int32_t n_out);
int emer_rmlp_bwd_fused(const float *dout, int64_t ldd, const float *out, int64_t ldo, const float *x, int64_t ldx, int32_t n_levels,
                        int32_t n_feat, int32_t k0, int64_t n, int32_t n_layers, const float *w0, const float *b0, const float *w1,
                        const float *b1, const float *w2, int32_t n_out, int32_t final_act, float *dx, int64_t lddx, float *workspace,
                        float *dw0, int64_t ld_dw0, float *db0, float *dw1, int64_t ld_dw1, float *db1, float *dw2, int64_t ld_dw2,
                        float *db2, void *stream);

/* rgb head: mlp.MLP(in = kh + 64, hidden 64, 3 layers, skip connection at layer 1) + sigmoid
 * (radiance_field.py:130-143,622-658, mlp.py:20-46) on input [hray[ray] | geo[sample]], where hray
 * (dir-PE | appearance embedding, kh columns) is constant along a ray.  The per-ray part arrives as
 * pre-activations rb0 = hray W0[:, :kh]^T + b0 and rb1 = hray W1[:, 64:64+kh]^T + b1 ([rays][64], row stride ld_rb); the kernel
 * does the per-sample part: a1 = relu(geo W0[:, kh:]^T + rb0), a2 = relu(a1 W1[:, :64]^T + geo W1[:, 64+kh:]^T
 * + rb1), out = sigmoid(a2 W2^T + b2).  w0 [64][kh+64], w1 [64][64+kh+64], w2 [3][64] are the torch Linear
 * weights; rows of ray r are r*S .. r*S+S-1 and S % 16 == 0.  a1 / a2 [n][64] are saved for the backward (a2 NULL, or both NULL: left to emer_rgb_head_bwd_recompute;
 * inference, nothing is stored). */
int emer_rgb_head_fwd(const float *geo, int64_t ld_geo, const float *rb0, const float *rb1, int64_t ld_rb, int64_t n_rays,
                      int32_t samples_per_ray, int32_t kh, const float *w0, const float *w1,
                      const float *w2, const float *b2, float *a1, float *a2, float *out, void *stream);
/* Backward of the density MLP (emer_neck_fwd with n_out = 1: DensityField.base_mlp + trunc_exp, radiance_field.py:808-812,
 * 836-840) INCLUDING its weight gradients, for narrow inputs (L * F <= 16: the proposal networks, L8 / F1): writes denc_lm
 * [L][n][F] and ACCUMULATES (+=) dw0 [64][ld_dw0 >= L F], db0 [64], dw1 [1][64], db1 [1].  The hidden layer is recomputed from
 * enc_lm, so the forward need not store it (h1 = NULL).  workspace: emer_density_bwd_fused_workspace floats (0: unsupported
 * shape, use emer_neck_bwd + emer_wgrad_segmented). */
int64_t emer_density_bwd_fused_workspace(int32_t n_levels, int32_t n_feat, int64_t n);
int emer_density_bwd_fused(const float *ddens, const float *dens, const float *enc_lm, int32_t n_levels,
                           int32_t n_feat, int64_t n, const float *w0, const float *b0, const float *w1,
                           float *denc_lm, float *workspace, float *dw0, int64_t ld_dw0, float *db0,
                           float *dw1, float *db1, void *stream);

/* emer_neck_fwd (n_out = 64, hidden layer not stored) followed by emer_rgb_head_fwd as ONE launch (RadianceField.forward of the
 * static model: radiance_field.py:302-318,400 then :622-658): a wave keeps its 16 rows in registers from the grid encoding to
 * the colour; the geometry features are written once (the backward needs them) and never read back.  enc_lm [L][n][F] with
 * n = n_rays * samples_per_ray, nw0 [64][L F], nw1 [64][64] (the neck), the other arguments as in emer_rgb_head_fwd.
 * Outputs geo [n][64], dens [n] = exp(geo[:, 0] - 1), a1 / a2 [n][64] (a2 or both NULL: inference, or a recomputing backward), out [n][3].  The results are
 * bit-identical to the two separate calls. */
int emer_field_fwd_supported(int32_t n_levels, int32_t n_feat);
int emer_field_fwd(const float *enc_lm, int32_t n_levels, int32_t n_feat, int64_t n_rays, int32_t samples_per_ray,
                   const float *nw0, const float *nb0, const float *nw1, const float *nb1, const float *rb0,
                   const float *rb1, int64_t ld_rb, int32_t kh, const float *w0, const float *w1,
                   const float *w2, const float *b2, float *geo, float *dens, float *a1, float *a2,
                   float *out, void *stream);
/* Data gradients: dpre2 [n][3] = dout * out * (1 - out), dpre1 / dpre0 [n][64] (pre-activation gradients of
 * layers 1 / 0, the wgrad operands), dgeo [n][64], and s1 / s0 [rays][64] = sums of dpre1 / dpre0 over the
 * samples of each ray (all that hray, W0[:, :kh], W1[:, 64:64+kh], b0 and b1 need).
 * dw2 (may be NULL) [3][ld_dw2 >= 64] += dpre2^T a2 and db2 (may be NULL) [3] += column sums of dpre2: the output layer's
 * weight gradient rides along (a2 and dpre2 are in registers here; a separate pass would re-read 280 MB for 195 numbers);
 * needs emer_rgb_head_bwd_workspace(n_rays) floats of workspace. */
int64_t emer_rgb_head_bwd_workspace(int64_t n_rays);
int emer_rgb_head_bwd(const float *dout, const float *out, const float *a1, const float *a2, int64_t n_rays,
                      int32_t samples_per_ray, int32_t kh, const float *w0, const float *w1,
                      const float *w2, float *dpre2, float *dpre1, float *dpre0, float *dgeo, float *s1,
                      float *s0, float *workspace, float *dw2, int64_t ld_dw2, float *db2, void *stream);

/* The same backward INCLUDING the weight gradients of the per-sample column blocks of layers 0 / 1 (autograd of
 * mlp.py:38-46 for radiance_field.py:130-143,622-658): neither dpre1 nor dpre0 reaches memory and no separate weight-gradient
 * pass reads them back.  Writes dgeo [n][64] and s1 / s0 [rays][64]; ACCUMULATES (+=)
 *   dw1[:, 0:64] += dpre1^T a1,  dw1[:, 64 + kh : 128 + kh] += dpre1^T geo   (dw1 [64][ld_dw1 >= 128 + kh]: layers.1.weight),
 *   dw0[:, kh : kh + 64] += dpre0^T geo                                        (dw0 [64][ld_dw0 >= 64 + kh]: layers.0.weight),
 *   dw2 [3][ld_dw2 >= 64] += dpre2^T a2,  db2 [3] += column sums of dpre2.
 * The per-ray column blocks (hray) and b0 / b1 follow from s1 / s0 (emer_ray_wgrad, emer_ray_pre_bwd).  geo [n][ld_geo] is the
 * forward's input.  samples_per_ray % 16 == 0; workspace: emer_rgb_head_bwd_fused_workspace(n_rays, samples_per_ray) floats. */
int emer_rgb_head_bwd_fused_supported(int32_t samples_per_ray);
int64_t emer_rgb_head_bwd_fused_workspace(int64_t n_rays, int32_t samples_per_ray);
int emer_rgb_head_bwd_fused(const float *dout, const float *out, const float *a1, const float *a2, const float *geo,
                            int64_t ld_geo, int64_t n_rays, int32_t samples_per_ray, int32_t kh, const float *w0,
                            const float *w1, const float *w2, float *dgeo, float *s1, float *s0, float *workspace,
                            float *dw0, int64_t ld_dw0, float *dw1, int64_t ld_dw1, float *dw2, int64_t ld_dw2,
                            float *db2, void *stream);
/* [r6] The same backward WITHOUT saved activations: a1 / a2 are recomputed in the kernel from geo and the per-ray pre-activations
 * rb0 / rb1 [n_rays][64] (row stride ld_rb; emer_ray_pre_fwd's output, bias included), bitwise the forward's values -- the forward
 * (emer_rgb_head_fwd / emer_field_fwd) is then called with a1 = a2 = NULL and writes 512 B / sample less, this kernel reads 512 B /
 * sample less.  a1 != NULL: the cheaper half -- the forward stored a1 [n][64] (and not a2), only a2 is recomputed.  Outputs and
 * accumulation targets as emer_rgb_head_bwd_fused, bitwise.  Workspace: emer_rgb_head_bwd_fused_workspace.
 * Autograd of radiance_fields/mlp.py:38-46 as instantiated at radiance_field.py:130-143 (rgb_head), same as the entries above. */
int emer_rgb_head_bwd_recompute(const float *dout, const float *out, const float *a1, const float *geo, int64_t ld_geo, const float *rb0,
                                const float *rb1, int64_t ld_rb, int64_t n_rays, int32_t samples_per_ray, int32_t kh,
                                const float *w0, const float *w1, const float *w2, float *dgeo, float *s1, float *s0,
                                float *workspace, float *dw0, int64_t ld_dw0, float *dw1, int64_t ld_dw1, float *dw2,
                                int64_t ld_dw2, float *db2, void *stream);

/* density_activation of the reference: y[i] = exp(x[i*x_stride] - 1); backward
 * dx[i*dx_stride] = dy[i] * min(y[i], e^15)   (radiance_field.py:28,461; nerf_utils.py:59-75). */
int emer_trunc_exp_fwd(const float *x, int64_t x_stride, float *y, int64_t n, void *stream);
int emer_trunc_exp_bwd(const float *dy, const float *y, float *dx, int64_t dx_stride, int64_t n,
                       void *stream);

/* Temporal aggregation of the flow branch (radiance_field.py:553-620) on a batch laid out [current | forward-warped |
 * backward-warped]: x3 holds three thirds of n floats each (n % 4 == 0, 16-byte aligned),
 *   fwd: out[i] = (x3[i] + 0.5 x3[n + i] + 0.5 x3[2 n + i]) / 2          bwd: dx3 = [g / 2 | g / 4 | g / 4]. */
int emer_aggregate3_fwd(const float *x3, int64_t n, float *out, void *stream);
int emer_aggregate3_bwd(const float *g, int64_t n, float *dx3, void *stream);
/* [r4] The same with the density activation of the aggregated features' column 0 in the launch (radiance_field.py:461,595-613):
 * x3 [3 n_rows, n_cols] (n_cols % 4 == 0), out [n_rows, n_cols], density [n_rows] = exp(out[:, 0] - 1); the backward adds
 * d_density * exp(min(out[:, 0] - 1, 15)) to column 0 of g (either gradient may be NULL) before the split into thirds. */
int emer_aggregate3_density_fwd(const float *x3, int64_t n_rows, int32_t n_cols, float *out, float *density, void *stream);
int emer_aggregate3_density_bwd(const float *g, const float *d_density, const float *density, int64_t n_rows,
                                int32_t n_cols, float *dx3, void *stream);

/* Direction encoding used by the rgb / sky heads: d -> (d+1)/2 -> [x, sin(2^i x), sin(2^i x + pi/2)]
 * i = 0..max_deg (radiance_fields/encodings.py:60-104, radiance_field.py:629-632).
 * dirs [n,3] -> out [n, 3*(1+2*(max_deg+1))].  remap != 0 applies the (d+1)/2 step first. */
int emer_dir_encode(const float *dirs, float *out, int64_t n, int32_t max_deg, int remap,
                    void *stream);

/* ---- per-ray side of the colour heads (csrc/rayinputs.hip) ---------------------------------------------
 * Input rows of the rgb head (radiance_field.py:622-643: PE of the remapped direction | appearance embedding of the ray's
 * image / camera) and of the sky head (:660-674: PE of the raw direction | the same embedding) in one launch.
 * dirs [n_rays, >=3] (row stride ld_dirs); idx [n_rays] int64 with element stride idx_stride (NULL: row 0);
 * emb [n_emb, emb_dim] (emb_dim 0: no embedding); out_* [n_rays, 3*(1+2*(max_deg+1)) + emb_dim] (either may be NULL).
 * An index outside [0, n_emb) writes NaN into the row (torch raises a device assert). */
int emer_ray_inputs_fwd(const float *dirs, int64_t ld_dirs, const int64_t *idx, int64_t idx_stride,
                        const float *emb, int32_t n_emb, int32_t emb_dim, int32_t max_deg,
                        int64_t n_rays, float *out_rgb, int64_t ld_rgb, float *out_sky,
                        int64_t ld_sky, void *stream);
/* Gradient of the embedding table (nn.Embedding backward, radiance_field.py:134-141): dw[e] += sum over the rays with
 * idx == e of g_a[ray] + g_b[ray]; g_* point at the FIRST embedding column of the two consumers' input gradients (either
 * may be NULL).  Deterministic (fixed summation order), no atomics, no workspace. */
int emer_embed_grad(const float *g_a, int64_t ld_a, const float *g_b, int64_t ld_b, const int64_t *idx,
                    int64_t idx_stride, int64_t n_rays, int32_t n_emb, int32_t emb_dim, float *dw,
                    void *stream);
/* The per-ray operand's share of layers 0 and 1 of the rgb head (mlp.py:38-46, skip connection into layer 1): h [n_rays, kh]
 * multiplies column block [0, kh) of W0 and [n_hidden, n_hidden + kh) of W1 for every sample of the ray, so it is applied
 * once per ray.  wa / wb point at those blocks INSIDE the weight matrices (row strides ld_wa / ld_wb).
 *   fwd: rb[r] = [wa h[r] + ba | wb h[r] + bb]   [n_rays, 2 n_hidden]   (the offsets emer_rgb_head_fwd adds per ray)
 *   bwd: dh[r] = s0[r] wa + s1[r] wb             s0 / s1 [n_rays, n_hidden]: per-ray sums of dpre0 / dpre1 */
int emer_ray_pre_fwd(const float *h, int64_t ld_h, int64_t n_rays, int32_t kh, int32_t n_hidden,
                     const float *wa, int64_t ld_wa, const float *ba, const float *wb, int64_t ld_wb,
                     const float *bb, float *rb, int64_t ld_rb, void *stream);
int emer_ray_pre_bwd(const float *s0, const float *s1, int64_t ld_s, int64_t n_rays, int32_t kh,
                     int32_t n_hidden, const float *wa, int64_t ld_wa, const float *wb, int64_t ld_wb,
                     float *dh, int64_t ld_dh, void *stream);

/* The per-ray skip MLP itself (the sky head, radiance_field.py:156-187,660-686: mlp.MLP(num_layers=3, skip_connections=[1]),
 * hidden width 64, n_out <= 16) after emer_ray_pre_fwd has applied the input x to layers 0 and 1 (wa = W0, wb = W1[:, 64:]):
 *   fwd: a1 = relu(rb[:, :64]); a2 = relu(W1[:, :64] a1 + rb[:, 64:]); out = act(W2 a2 + b2)   act: EMER_ACT_NONE / _SIGMOID
 *   bwd: dpre2 = dout * act'(out); dpre1 = (a2 > 0) (dpre2 W2); dpre0 = (a1 > 0) (dpre1 W1[:, :64])
 * a1 / a2 / dpre1 / dpre0 [n_rows, 64], out / dout / dpre2 [n_rows, n_out], all contiguous.  The input gradient is
 * emer_ray_pre_bwd(dpre0, dpre1, ...); the weight gradients are emer_wgrad_segmented on dpre0 / dpre1 / dpre2. */
int emer_ray_head_fwd(const float *rb, int64_t ld_rb, int64_t n_rows, const float *w1, int64_t ld_w1,
                      const float *w2, const float *b2, int32_t n_out, int act, float *a1, float *a2,
                      float *out, void *stream);
int emer_ray_head_bwd(const float *dout, const float *out, const float *a1, const float *a2,
                      int64_t n_rows, const float *w1, int64_t ld_w1, const float *w2, int32_t n_out,
                      int act, float *dpre2, float *dpre1, float *dpre0, void *stream);

/* Weight gradients of per-ray layers (a few thousand rows), up to EMER_RAY_WGRAD_MAX_JOBS layers per launch -- the three
 * layers of the sky head, or the per-ray column blocks of the rgb head's layers 0 and 1 (autograd of mlp.py:38-46):
 *   dw[i][dst_col_s + j] += sum_rows dy[row][i] x_s[row][j]   (s = 0 .. n_segs-1: the column blocks of a virtual concat)
 *   dbias[i]             += sum_rows dy[row][i]               (dbias may be NULL)
 * dy [m, n], n <= 64; operand blocks of at most 64 columns, 127 in total; all jobs share m.  Row chunks combine with relaxed float
 * atomics (summation order not fixed, like emer_wgrad_segmented's reduction). */
#define EMER_RAY_WGRAD_MAX_JOBS 8
typedef struct emer_ray_wgrad_job {
    const float *dy;
    int64_t ld_dy;
    const float *x[2];
    int64_t ld_x[2];
    float *dw;
    int64_t ld_dw;
    float *dbias;
    int32_t n, n_segs;
    int32_t width[2], dst_col[2];
} emer_ray_wgrad_job;
int emer_ray_wgrad(const emer_ray_wgrad_job *jobs, int32_t n_jobs, int64_t m, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Training-ray generation (replaces datasets/base/pixel_source.py:39-76 get_rays and
 * :564-731 sample_uniform_rays / sample_important_rays / the gathers of get_train_rays) on device-resident
 * dataset tensors.  Random numbers: splitmix64(seed_word[0] ^ salt, counter) -- seed_word lives in device memory so a
 * captured hipGraph draws new rays by bumping it.
 * ---------------------------------------------------------------------------------------------- */
/* n uniform pixels: image = candidates[randint(n_candidates)] (candidates NULL: 0..n_candidates-1), x = randint(width),
 * y = randint(height)  (pixel_source.py:622-668). */
int emer_sample_uniform(const uint64_t *seed_word, uint64_t salt, int64_t n, const int64_t *candidates,
                        int32_t n_candidates, int32_t height, int32_t width, int64_t *img_idx, int64_t *y,
                        int64_t *x, void *stream);
/* Lidar training rays (datasets/base/lidar_source.py:223-308: sample_uniform_rays' torch.randint over the cached scans + the four
 * gathers of get_train_rays) in one launch: idx = randint(n_points) (or idx_in[i] when given: gather only), out_* = cached_*[idx].
 * origins / directions [n_points][3], ranges / timestamps [n_points]; idx_out (may be NULL) receives the drawn indices. */
int emer_lidar_sample_rays(const uint64_t *seed_word, uint64_t salt, int64_t n, int64_t n_points, const int64_t *idx_in,
                           const float *origins, const float *directions, const float *ranges, const float *timestamps,
                           int64_t *idx_out, float *out_origins, float *out_directions, float *out_ranges,
                           float *out_timestamps, void *stream);
/* k DISTINCT indices into weights[0..n_weights) with probability proportional to the weights, without replacement
 * (torch.multinomial(w, k, replacement=False), pixel_source.py:588-592): Efraimidis-Spirakis keys -log(u)/w, the k
 * smallest found by a 3-pass radix select.  workspace: 4 + 2048 uint32 words.  Order of flat_out is unspecified. */
int emer_sample_importance(const float *weights, int64_t n_weights, const uint64_t *seed_word, uint64_t salt,
                           int64_t k, uint32_t *workspace, int64_t *flat_out, void *stream);
/* flat index into the [n_candidates][buffer_height][buffer_width] error buffer -> (image, y, x) at full resolution with
 * a random offset inside the downscaled cell, clamped to the image (pixel_source.py:593-620). */
int emer_buffer_to_pixels(const int64_t *flat, int64_t n, int32_t buffer_height, int32_t buffer_width,
                          int32_t downscale, const int64_t *candidates, int32_t height, int32_t width,
                          const uint64_t *seed_word, uint64_t salt, int64_t *img_idx, int64_t *y, int64_t *x,
                          void *stream);
/* (img, y, x) -> ray through pixel centre (x + 0.5, y + 0.5): origins / viewdirs [n,3], direction_norms [n]
 * (get_rays), pixel_coords [n,2] = (y/H, x/W), and the gathers pixels = images[img,y,x,:] [n,3], sky = sky_masks[img,y,x],
 * ray_timestamps = timestamps[img], ray_cam_ids = cam_ids[img] (each dataset tensor may be NULL; then its output is
 * not written).  cam_to_worlds [n_imgs,4,4], intrinsics [n_imgs,3,3] row-major. */
int emer_gen_rays(const int64_t *img_idx, const int64_t *y, const int64_t *x, const float *cam_to_worlds,
                  const float *intrinsics, const float *images, const float *sky_masks, const float *timestamps,
                  const int64_t *cam_ids, int64_t n, int32_t height, int32_t width, float *origins,
                  float *viewdirs, float *direction_norms, float *pixel_coords, float *pixels, float *sky,
                  float *ray_timestamps, int64_t *ray_cam_ids, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer (torch.optim.Adam as configured in builders.py:50-60,114-120: eps 1e-15,
 * betas (0.9, 0.99), weight_decay as L2) over one flat fp32 buffer; grad_scale multiplies the
 * gradient first (1/world_size after the RCCL all-reduce; GradScaler quirk).
 * ---------------------------------------------------------------------------------------------- */
int emer_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale, int32_t step, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EMERNERF_HIP_H */
