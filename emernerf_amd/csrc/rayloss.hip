// Per-ray epilogue of the volume renderer and the pixel losses for gfx950.
//
// Replaces the per-ray torch chains at the end of `rendering` (radiance_fields/render_utils.py:102-105: opacity clamp,
// expected depth = sum(w t) / opacity; :217-226: rgb += rgb_sky * (1 - opacity)) and the two pixel losses every
// EmerNeRF config trains with (loss/base.py:83-146 RealValueLoss "rgb" with L2, coefficient 1; :149-185 SkyLoss
// "opacity_based" = binary cross entropy between opacity and 1 - sky_mask, coefficient 0.001 in
// configs/default_config.yaml) together with their autograd graphs: about 25 elementwise launches per step on
// [R, 1..3] tensors become two forward and two backward launches.  SURVEY.md section 8f row N4 (per-ray part).
#include "common.h"

namespace emer {

// ---------------------------------------------------------------------------------------------- ray epilogue
// stats [R,4] = (sum w, sum w*mid, median depth, -) from emer_render_weights_fwd.
__global__ __launch_bounds__(256) void ray_epilogue_fwd_kernel(const float *__restrict__ stats, const float *__restrict__ acc_rgb,
                                                               const float *__restrict__ rgb_sky, int64_t R,
                                                               float *__restrict__ opacity, float *__restrict__ depth,
                                                               float *__restrict__ median, float *__restrict__ rgb) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float4 st = *reinterpret_cast<const float4 *>(stats + r * 4);
    const float o = fminf(fmaxf(st.x, 1e-6f), 1.0f);  // torch.clamp(1e-6, 1.0)
    opacity[r] = o;
    depth[r] = st.y / o;
    if (median) median[r] = st.z;
    if (rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc_rgb[r * 3 + c];
            if (rgb_sky) v = v + rgb_sky[r * 3 + c] * (1.0f - o);
            rgb[r * 3 + c] = v;
        }
    }
}

// d_stats [R,4] = gradient of ray_stats (columns 2, 3 zero); d_rgb_sky [R,3].  The gradient of acc_rgb is d_rgb itself.
__global__ __launch_bounds__(256) void ray_epilogue_bwd_kernel(const float *__restrict__ stats, const float *__restrict__ rgb_sky,
                                                               const float *__restrict__ d_opacity, const float *__restrict__ d_depth,
                                                               const float *__restrict__ d_rgb, int64_t R, float *__restrict__ d_stats,
                                                               float *__restrict__ d_rgb_sky) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float4 st = *reinterpret_cast<const float4 *>(stats + r * 4);
    const float o = fminf(fmaxf(st.x, 1e-6f), 1.0f);
    float go = d_opacity ? d_opacity[r] : 0.0f;
    const float gd = d_depth ? d_depth[r] : 0.0f;
    go -= gd * st.y / (o * o);
    if (d_rgb && rgb_sky) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = d_rgb[r * 3 + c];
            go -= g * rgb_sky[r * 3 + c];
            if (d_rgb_sky) d_rgb_sky[r * 3 + c] = g * (1.0f - o);
        }
    }
    // clamp passes the gradient where min <= x <= max (torch semantics, bounds included)
    *reinterpret_cast<float4 *>(d_stats + r * 4) = make_float4((st.x >= 1e-6f && st.x <= 1.0f) ? go : 0.0f, gd / o, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------------ pixel losses
// per-ray partial: w_rgb * sum_c (rgb - pix)^2 / (3 R) + w_sky * bce(opacity, 1 - sky) / R
// bce(o, t) = -(t * max(log o, -100) + (1 - t) * max(log(1 - o), -100))   (torch.nn.functional.binary_cross_entropy)
__global__ __launch_bounds__(256) void pixel_loss_fwd_kernel(const float *__restrict__ rgb, const float *__restrict__ pixels,
                                                             const float *__restrict__ opacity, const float *__restrict__ sky_mask,
                                                             int64_t R, float w_rgb, float w_sky, float *__restrict__ loss_rays) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float l = 0.0f;
    if (rgb) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = rgb[r * 3 + c] - pixels[r * 3 + c]; s += d * d; }
        l += w_rgb * s / (3.0f * (float)R);
    }
    if (opacity && sky_mask) {
        const float o = opacity[r], t = 1.0f - sky_mask[r];
        const float bce = -(t * fmaxf(logf(o), -100.0f) + (1.0f - t) * fmaxf(logf(1.0f - o), -100.0f));
        l += w_sky * bce / (float)R;
    }
    loss_rays[r] = l;
}

// gradients times the upstream scalar g[0]: d_rgb = w_rgb * 2 (rgb - pix) / (3R), d_opacity = w_sky * (o - t) / max(o (1 - o), 1e-12) / R
__global__ __launch_bounds__(256) void pixel_loss_bwd_kernel(const float *__restrict__ rgb, const float *__restrict__ pixels,
                                                             const float *__restrict__ opacity, const float *__restrict__ sky_mask,
                                                             int64_t R, float w_rgb, float w_sky, const float *__restrict__ g,
                                                             float *__restrict__ d_rgb, float *__restrict__ d_opacity) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float up = g ? g[0] : 1.0f;
    if (d_rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) d_rgb[r * 3 + c] = up * w_rgb * 2.0f * (rgb[r * 3 + c] - pixels[r * 3 + c]) / (3.0f * (float)R);
    }
    if (d_opacity) {
        const float o = opacity[r], t = 1.0f - sky_mask[r];
        d_opacity[r] = up * w_sky * (o - t) / fmaxf(o * (1.0f - o), 1e-12f) / (float)R;
    }
}


// ------------------------------------------------------------------------------------------------ lidar losses
// Depth + line-of-sight supervision of a lidar-ray batch (loss/base.py:188-271 DepthLoss "l2", :295-345 LineOfSightLoss,
// :430-464 compute_line_of_sight_loss; called at train_emernerf.py:770-808).  With gt = lidar range, t = sample
// midpoints, w = rendering weights:
//   depth:  mean over VALID rays (0.01 < gt < max_depth) of (clamp(pred / max, 0, 1) - clamp(gt / max, 0, 1))^2
//   sight:  [ mean_r sum_s w^2 [t < gt - eps]  +  mean_r sum_s (w - N(t - gt; sigma = eps / 3))^2 [|t - gt| < eps] ]
//           * mean_r [gt > 0]
// (the reference multiplies the SCALAR sum of the two ray-means by the per-ray mask gt > 0 and then averages: the
// product of two means, reproduced as is).  counts = (#rays with gt > 0, #valid rays) from lidar_counts_kernel.
__global__ __launch_bounds__(1024) void lidar_counts_kernel(const float *__restrict__ gt, int64_t R, float max_depth, float *__restrict__ counts) {
    __shared__ float part[2][16];
    float a = 0.0f, b = 0.0f;
    for (int64_t i = threadIdx.x; i < R; i += 1024) {
        const float g = gt[i];
        a += g > 0.0f ? 1.0f : 0.0f;
        b += (g > 0.01f && g < max_depth) ? 1.0f : 0.0f;
    }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = a; part[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.0f, sb = 0.0f;
        for (int i = 0; i < 16; ++i) { sa += part[0][i]; sb += part[1][i]; }
        counts[0] = sa; counts[1] = sb;
    }
}

// one wave per ray; loss_rays [R] = this ray's share of the total; d_depth [R], d_weights [R,S] times upstream g[0]
__global__ __launch_bounds__(256) void lidar_loss_kernel(const float *__restrict__ depth, const float *__restrict__ gt,
                                                         const float *__restrict__ weights, const float *__restrict__ t_vals,
                                                         int64_t R, int32_t S, float eps, float max_depth, float w_depth, float w_sight,
                                                         const float *__restrict__ counts, const float *__restrict__ g,
                                                         float *__restrict__ loss_rays, float *__restrict__ d_depth,
                                                         float *__restrict__ d_weights) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= R) return;
    const float up = g ? g[0] : 1.0f;
    const float n_pos = counts[0], n_valid = counts[1];
    const float gd = gt[r];
    // ---- depth term (per valid ray)
    float l = 0.0f;
    const bool valid = gd > 0.01f && gd < max_depth;
    if (valid && w_depth != 0.0f) {
        const float pn = depth[r] / max_depth, gn = fminf(fmaxf(gd / max_depth, 0.0f), 1.0f);
        const float pc = fminf(fmaxf(pn, 0.0f), 1.0f);
        const float d = pc - gn;
        l += w_depth * d * d / n_valid;
        if (d_depth && lane == 0) d_depth[r] = (pn >= 0.0f && pn <= 1.0f) ? up * w_depth * 2.0f * d / (max_depth * n_valid) : 0.0f;
    } else if (d_depth && lane == 0) {
        d_depth[r] = 0.0f;
    }
    // ---- line of sight: scale = w_sight * mean[gt > 0] / R
    const float scale = w_sight * (n_pos / (float)R) / (float)R;
    const float sigma = eps / 3.0f;
    const float norm = 1.0f / sqrtf(2.0f * 3.14159265358979323846f * sigma * sigma), inv2s2 = 1.0f / (2.0f * sigma * sigma);
    float acc = 0.0f;
    for (int32_t s = lane; s < S; s += kWave) {
        const int64_t i = r * S + s;
        const float w = weights[i], t = t_vals[i];
        float term = 0.0f, dw = 0.0f;
        if (t < gd - eps) { term = w * w; dw = 2.0f * w; }
        else if (t > gd - eps && t < gd + eps) {
            const float x = t - gd;
            const float e = w - norm * expf(-(x * x) * inv2s2);
            term = e * e; dw = 2.0f * e;
        }
        acc += term;
        if (d_weights) d_weights[i] = up * scale * dw;
    }
    acc = wave_sum(acc);
    if (lane == 0) loss_rays[r] = l + scale * acc;
}

// ------------------------------------------------------------------------------------------------ regularisers
// The mean-type regularisers of the dynamic / flow / feature models:
//   dynamic-density sparsity  c * mean(dynamic_density [R,S])                 loss/base.py:394-398 ("sparsity", no mask),
//   shadow sparsity           c * mean(shadow_ratio [R,1])                    same class (train_emernerf.py:689-694),
//   feature L2                c * mean((dino_feat - features)^2 [R,E])        loss/base.py:83-146 (train_emernerf.py:676-682),
//   flow cycle consistency    c * mean((ff + fpb)^2 + (bf + bpf)^2 [R,S,3])   train_emernerf.py:700-716 (ff / bf detached there:
//                                                                              gradients go to the two predictions only).
// The reference evaluates each as a chain of elementwise torch ops, a mean and a scalar multiply-add (~25 launches with their
// autograd twins); here all four are one streaming pass each way.  Fixed summation order: per-thread strided sums, wave / block
// tree, per-block partials summed by one workgroup in double.
struct RegArgs {
    const float *dyn; int64_t n_dyn; float c_dyn;
    const float *shadow; int64_t n_shadow; float c_shadow;
    const float *feat, *feat_gt; int64_t n_feat; float c_feat;
    const float *ff, *fpb, *bf, *bpf; int64_t n_flow; float c_cycle;
    // [r5] packed != 0: ff = the flow MLP's output at the sample positions [n_flow / 3][6] = (forward | backward flow) and fpb = its output at
    // the two warped sets [2 n_flow / 3][6] = (rows 0..N: at the forward-warped points, rows N..2N: at the backward-warped points); the four
    // operands of the cycle term are column blocks of those two tensors (no slice copies, and ONE gradient tensor in the backward)
    int32_t packed;
};
__device__ __forceinline__ void cycle_operands(const RegArgs &a, int64_t i, float &ff, float &fpb, float &bf, float &bpf) {
    if (a.packed) {
        const int64_t row = i / 3, n_rows = a.n_flow / 3;
        const int c = (int)(i - 3 * row);
        ff = a.ff[row * 6 + c]; bf = a.ff[row * 6 + 3 + c];
        fpb = a.fpb[row * 6 + 3 + c]; bpf = a.fpb[(n_rows + row) * 6 + c];
    } else {
        ff = a.ff[i]; fpb = a.fpb[i]; bf = a.bf[i]; bpf = a.bpf[i];
    }
}

constexpr int kRegThreads = 256;

__global__ __launch_bounds__(kRegThreads) void reg_losses_fwd_kernel(const RegArgs a, float *__restrict__ partials) {
    const int64_t tid = (int64_t)blockIdx.x * kRegThreads + threadIdx.x, stride = (int64_t)gridDim.x * kRegThreads;
    float l = 0.0f;
    if (a.dyn) {
        float s = 0.0f;
        for (int64_t i = tid; i < a.n_dyn; i += stride) s += a.dyn[i];
        l += a.c_dyn / (float)a.n_dyn * s;
    }
    if (a.shadow) {
        float s = 0.0f;
        for (int64_t i = tid; i < a.n_shadow; i += stride) s += a.shadow[i];
        l += a.c_shadow / (float)a.n_shadow * s;
    }
    if (a.feat) {
        float s = 0.0f;
        for (int64_t i = tid; i < a.n_feat; i += stride) { const float d = a.feat[i] - a.feat_gt[i]; s += d * d; }
        l += a.c_feat / (float)a.n_feat * s;
    }
    if (a.fpb) {
        float s = 0.0f;
        for (int64_t i = tid; i < a.n_flow; i += stride) {
            float ff, fpb, bf, bpf;
            cycle_operands(a, i, ff, fpb, bf, bpf);
            const float u = ff + fpb, v = bf + bpf;
            s += u * u + v * v;
        }
        l += a.c_cycle / (float)a.n_flow * s;
    }
    __shared__ float part[kRegThreads / 64];
    l = wave_sum(l);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kRegThreads / 64; ++w) t += part[w];
        partials[blockIdx.x] = t;
    }
}

// out[0] = (base ? base[0] : 0) + sum(partials[0..n)), one workgroup, fixed order, double accumulation
__global__ __launch_bounds__(256) void reg_losses_finish_kernel(const float *__restrict__ partials, int32_t n, const float *__restrict__ base,
                                                                float *__restrict__ out) {
    __shared__ double part[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((base ? (double)base[0] : 0.0) + ((part[0] + part[1]) + (part[2] + part[3])));
}

// gradients of the four terms times up = upstream[0] * grad_scale; every output pointer may be null
__global__ __launch_bounds__(kRegThreads) void reg_losses_bwd_kernel(const RegArgs a, const float *__restrict__ upstream, float grad_scale,
                                                                     float *__restrict__ d_dyn, float *__restrict__ d_shadow,
                                                                     float *__restrict__ d_feat, float *__restrict__ d_fpb,
                                                                     float *__restrict__ d_bpf) {
    const int64_t tid = (int64_t)blockIdx.x * kRegThreads + threadIdx.x, stride = (int64_t)gridDim.x * kRegThreads;
    const float up = (upstream ? upstream[0] : 1.0f) * grad_scale;
    if (d_dyn) {
        const float g = up * (a.c_dyn / (float)a.n_dyn);
        for (int64_t i = tid; i < a.n_dyn; i += stride) d_dyn[i] = g;
    }
    if (d_shadow) {
        const float g = up * (a.c_shadow / (float)a.n_shadow);
        for (int64_t i = tid; i < a.n_shadow; i += stride) d_shadow[i] = g;
    }
    if (d_feat) {
        const float g = up * (2.0f * a.c_feat / (float)a.n_feat);
        for (int64_t i = tid; i < a.n_feat; i += stride) d_feat[i] = g * (a.feat[i] - a.feat_gt[i]);
    }
    if (a.packed && d_fpb) {   // d_fpb: the gradient of the whole [2 N][6] tensor (zero in the column blocks the loss does not read)
        const float g = up * (2.0f * a.c_cycle / (float)a.n_flow);
        const int64_t n_rows = a.n_flow / 3;
        for (int64_t j = tid; j < 12 * n_rows; j += stride) {
            const int64_t row2 = j / 6;
            const int c = (int)(j - 6 * row2);
            float v = 0.0f;
            if (row2 < n_rows) { if (c >= 3) v = g * (a.ff[row2 * 6 + c - 3] + a.fpb[j]); }
            else if (c < 3) v = g * (a.ff[(row2 - n_rows) * 6 + 3 + c] + a.fpb[j]);
            d_fpb[j] = v;
        }
    } else if (d_fpb || d_bpf) {
        const float g = up * (2.0f * a.c_cycle / (float)a.n_flow);
        for (int64_t i = tid; i < a.n_flow; i += stride) {
            if (d_fpb) d_fpb[i] = g * (a.ff[i] + a.fpb[i]);
            if (d_bpf) d_bpf[i] = g * (a.bf[i] + a.bpf[i]);
        }
    }
}

}  // namespace emer

using namespace emer;

extern "C" int emer_reduce_sum(const float *x, int64_t n, int accumulate, float *out, void *stream);

extern "C" int emer_ray_epilogue_fwd(const float *ray_stats, const float *acc_rgb, const float *rgb_sky, int64_t n_rays, float *opacity,
                                     float *depth, float *median_depth, float *rgb, void *stream) {
    EMER_REQUIRE(n_rays >= 0, "ray_epilogue_fwd: negative n_rays");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(ray_stats && opacity && depth && (!rgb || acc_rgb), "ray_epilogue_fwd: null pointer");
    hipLaunchKernelGGL(ray_epilogue_fwd_kernel, dim3((uint32_t)ceil_div(n_rays, 256)), dim3(256), 0, as_stream(stream), ray_stats, acc_rgb,
                       rgb_sky, n_rays, opacity, depth, median_depth, rgb);
    return check_launch("ray_epilogue_fwd");
}

extern "C" int emer_ray_epilogue_bwd(const float *ray_stats, const float *rgb_sky, const float *d_opacity, const float *d_depth,
                                     const float *d_rgb, int64_t n_rays, float *d_ray_stats, float *d_rgb_sky, void *stream) {
    EMER_REQUIRE(n_rays >= 0, "ray_epilogue_bwd: negative n_rays");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(ray_stats && d_ray_stats, "ray_epilogue_bwd: null pointer");
    hipLaunchKernelGGL(ray_epilogue_bwd_kernel, dim3((uint32_t)ceil_div(n_rays, 256)), dim3(256), 0, as_stream(stream), ray_stats, rgb_sky,
                       d_opacity, d_depth, d_rgb, n_rays, d_ray_stats, d_rgb_sky);
    return check_launch("ray_epilogue_bwd");
}

extern "C" int emer_pixel_loss_fwd(const float *rgb, const float *pixels, const float *opacity, const float *sky_mask, int64_t n_rays,
                                   float w_rgb, float w_sky, float *loss_rays, float *loss_out, void *stream) {
    EMER_REQUIRE(n_rays >= 1, "pixel_loss_fwd: needs at least one ray");
    EMER_REQUIRE(loss_rays && loss_out && (!rgb || pixels) && (!opacity || sky_mask), "pixel_loss_fwd: null pointer");
    hipLaunchKernelGGL(pixel_loss_fwd_kernel, dim3((uint32_t)ceil_div(n_rays, 256)), dim3(256), 0, as_stream(stream), rgb, pixels, opacity,
                       sky_mask, n_rays, w_rgb, w_sky, loss_rays);
    if (int rc = check_launch("pixel_loss_fwd")) return rc;
    return emer_reduce_sum(loss_rays, n_rays, 0, loss_out, stream);
}

extern "C" int emer_pixel_loss_bwd(const float *rgb, const float *pixels, const float *opacity, const float *sky_mask, int64_t n_rays,
                                   float w_rgb, float w_sky, const float *upstream, float *d_rgb, float *d_opacity, void *stream) {
    EMER_REQUIRE(n_rays >= 1, "pixel_loss_bwd: needs at least one ray");
    EMER_REQUIRE((!d_rgb || (rgb && pixels)) && (!d_opacity || (opacity && sky_mask)), "pixel_loss_bwd: null pointer");
    hipLaunchKernelGGL(pixel_loss_bwd_kernel, dim3((uint32_t)ceil_div(n_rays, 256)), dim3(256), 0, as_stream(stream), rgb, pixels, opacity,
                       sky_mask, n_rays, w_rgb, w_sky, upstream, d_rgb, d_opacity);
    return check_launch("pixel_loss_bwd");
}

// workspace: n_rays + 2 floats (per-ray partials + the two counts)
extern "C" int emer_lidar_loss(const float *depth, const float *lidar_ranges, const float *weights, const float *t_vals, int64_t n_rays,
                               int32_t n_samples, float epsilon, float max_depth, float w_depth, float w_sight, const float *upstream,
                               float *workspace, float *loss_out, float *d_depth, float *d_weights, void *stream) {
    EMER_REQUIRE(n_rays >= 1 && n_samples >= 1 && epsilon > 0.0f && max_depth > 0.0f, "lidar_loss: bad arguments");
    EMER_REQUIRE(depth && lidar_ranges && weights && t_vals && workspace, "lidar_loss: null pointer");
    hipStream_t st = as_stream(stream);
    float *counts = workspace + n_rays;
    hipLaunchKernelGGL(lidar_counts_kernel, dim3(1), dim3(1024), 0, st, lidar_ranges, n_rays, max_depth, counts);
    hipLaunchKernelGGL(lidar_loss_kernel, dim3((uint32_t)ceil_div(n_rays, 4)), dim3(256), 0, st, depth, lidar_ranges, weights, t_vals, n_rays,
                       n_samples, epsilon, max_depth, w_depth, w_sight, counts, upstream, workspace, d_depth, d_weights);
    if (int rc = check_launch("lidar_loss")) return rc;
    if (loss_out) return emer_reduce_sum(workspace, n_rays, 0, loss_out, stream);
    return EMER_OK;
}

static inline RegArgs make_reg_args(const float *dyn, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow, float c_shadow,
                                    const float *feat, const float *feat_gt, int64_t n_feat, float c_feat, const float *ff, const float *fpb,
                                    const float *bf, const float *bpf, int64_t n_flow, float c_cycle) {
    return RegArgs{dyn, n_dyn, c_dyn, shadow, n_shadow, c_shadow, feat, feat_gt, n_feat, c_feat, ff, fpb, bf, bpf, n_flow, c_cycle, 0};
}
static inline uint32_t reg_blocks(const RegArgs &a) {
    int64_t n = 1;
    if (a.dyn && a.n_dyn > n) n = a.n_dyn;
    if (a.shadow && a.n_shadow > n) n = a.n_shadow;
    if (a.feat && a.n_feat > n) n = a.n_feat;
    if (a.fpb && a.n_flow * (a.packed ? 4 : 1) > n) n = a.n_flow * (a.packed ? 4 : 1);
    const int64_t b = ceil_div(n, (int64_t)kRegThreads * 4);   // >= 4 elements per thread; at most 1024 blocks
    return (uint32_t)(b < 1 ? 1 : b > EMER_REG_MAX_BLOCKS ? EMER_REG_MAX_BLOCKS : b);
}
static bool reg_args_ok(const RegArgs &a) {
    return (!a.dyn || a.n_dyn >= 1) && (!a.shadow || a.n_shadow >= 1) && (!a.feat || (a.feat_gt && a.n_feat >= 1)) &&
           (!a.fpb || (a.ff && (a.packed || (a.bf && a.bpf)) && a.n_flow >= 1));
}

extern "C" int emer_reg_losses_fwd(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow, float c_shadow,
                                   const float *feat, const float *feat_gt, int64_t n_feat, float c_feat, const float *fwd_flow,
                                   const float *fwd_pred_bwd_flow, const float *bwd_flow, const float *bwd_pred_fwd_flow, int64_t n_flow,
                                   float c_cycle, const float *base, float *workspace, float *loss_out, void *stream) {
    const RegArgs a = make_reg_args(dyn_density, n_dyn, c_dyn, shadow, n_shadow, c_shadow, feat, feat_gt, n_feat, c_feat, fwd_flow,
                                    fwd_pred_bwd_flow, bwd_flow, bwd_pred_fwd_flow, n_flow, c_cycle);
    EMER_REQUIRE(reg_args_ok(a), "reg_losses_fwd: a term's companion pointer is null or its count is < 1");
    EMER_REQUIRE(workspace && loss_out, "reg_losses_fwd: null pointer");
    const uint32_t blocks = reg_blocks(a);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(reg_losses_fwd_kernel, dim3(blocks), dim3(kRegThreads), 0, st, a, workspace);
    hipLaunchKernelGGL(reg_losses_finish_kernel, dim3(1), dim3(256), 0, st, workspace, (int32_t)blocks, base, loss_out);
    return check_launch("reg_losses_fwd");
}

extern "C" int emer_reg_losses_bwd(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow, float c_shadow,
                                   const float *feat, const float *feat_gt, int64_t n_feat, float c_feat, const float *fwd_flow,
                                   const float *fwd_pred_bwd_flow, const float *bwd_flow, const float *bwd_pred_fwd_flow, int64_t n_flow,
                                   float c_cycle, const float *upstream, float grad_scale, float *d_dyn_density, float *d_shadow,
                                   float *d_feat, float *d_fwd_pred_bwd_flow, float *d_bwd_pred_fwd_flow, void *stream) {
    const RegArgs a = make_reg_args(dyn_density, n_dyn, c_dyn, shadow, n_shadow, c_shadow, feat, feat_gt, n_feat, c_feat, fwd_flow,
                                    fwd_pred_bwd_flow, bwd_flow, bwd_pred_fwd_flow, n_flow, c_cycle);
    EMER_REQUIRE(reg_args_ok(a), "reg_losses_bwd: a term's companion pointer is null or its count is < 1");
    EMER_REQUIRE((!d_dyn_density || n_dyn >= 1) && (!d_shadow || n_shadow >= 1) && (!d_feat || feat) &&
                 ((!d_fwd_pred_bwd_flow && !d_bwd_pred_fwd_flow) || fwd_pred_bwd_flow), "reg_losses_bwd: gradient requested for an absent term");
    hipLaunchKernelGGL(reg_losses_bwd_kernel, dim3(reg_blocks(a)), dim3(kRegThreads), 0, as_stream(stream), a, upstream, grad_scale,
                       d_dyn_density, d_shadow, d_feat, d_fwd_pred_bwd_flow, d_bwd_pred_fwd_flow);
    return check_launch("reg_losses_bwd");
}

// [r5] The same pair with the flow cycle term read from the flow MLP's own outputs: flow6 [n_rows][6] (at the sample positions: forward |
// backward flow, constants) and flow2_6 [2 n_rows][6] (at the forward-warped, then the backward-warped points: the predicted backward flow is
// columns 3..5 of the first half, the predicted forward flow columns 0..2 of the second).  Same value as emer_reg_losses_fwd on the four
// slices (same summation order); the backward writes the gradient of the WHOLE flow2_6 tensor.  flow2_6 == NULL: no cycle term.
extern "C" int emer_reg_losses_fwd6(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow, float c_shadow,
                                    const float *feat, const float *feat_gt, int64_t n_feat, float c_feat, const float *flow6, const float *flow2_6,
                                    int64_t n_rows, float c_cycle, const float *base, float *workspace, float *loss_out, void *stream) {
    RegArgs a = make_reg_args(dyn_density, n_dyn, c_dyn, shadow, n_shadow, c_shadow, feat, feat_gt, n_feat, c_feat, flow6, flow2_6, nullptr, nullptr,
                              3 * n_rows, c_cycle);
    a.packed = 1;
    EMER_REQUIRE(reg_args_ok(a), "reg_losses_fwd6: a term's companion pointer is null or its count is < 1");
    EMER_REQUIRE(workspace && loss_out, "reg_losses_fwd6: null pointer");
    const uint32_t blocks = reg_blocks(a);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(reg_losses_fwd_kernel, dim3(blocks), dim3(kRegThreads), 0, st, a, workspace);
    hipLaunchKernelGGL(reg_losses_finish_kernel, dim3(1), dim3(256), 0, st, workspace, (int32_t)blocks, base, loss_out);
    return check_launch("reg_losses_fwd6");
}

extern "C" int emer_reg_losses_bwd6(const float *dyn_density, int64_t n_dyn, float c_dyn, const float *shadow, int64_t n_shadow, float c_shadow,
                                    const float *feat, const float *feat_gt, int64_t n_feat, float c_feat, const float *flow6, const float *flow2_6,
                                    int64_t n_rows, float c_cycle, const float *upstream, float grad_scale, float *d_dyn_density, float *d_shadow,
                                    float *d_feat, float *d_flow2_6, void *stream) {
    RegArgs a = make_reg_args(dyn_density, n_dyn, c_dyn, shadow, n_shadow, c_shadow, feat, feat_gt, n_feat, c_feat, flow6, flow2_6, nullptr, nullptr,
                              3 * n_rows, c_cycle);
    a.packed = 1;
    EMER_REQUIRE(reg_args_ok(a), "reg_losses_bwd6: a term's companion pointer is null or its count is < 1");
    EMER_REQUIRE((!d_dyn_density || n_dyn >= 1) && (!d_shadow || n_shadow >= 1) && (!d_feat || feat) && (!d_flow2_6 || flow2_6),
                 "reg_losses_bwd6: gradient requested for an absent term");
    hipLaunchKernelGGL(reg_losses_bwd_kernel, dim3(reg_blocks(a)), dim3(kRegThreads), 0, as_stream(stream), a, upstream, grad_scale,
                       d_dyn_density, d_shadow, d_feat, d_flow2_6, (float *)nullptr);
    return check_launch("reg_losses_bwd6");
}
