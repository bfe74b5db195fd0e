// Alpha-compositing volume renderer for gfx950.
//
// Replaces nerfacc's dense-tensor volrend as EmerNeRF calls it (render_transmittance_from_density,
// render_weight_from_density, accumulate_along_rays: radiance_fields/render_utils.py:35-43,73-77,
// 103-115,159-282; third_party/nerfacc_prop_net.py:165-168; loss/base.py:454-460).  Semantics:
// SURVEY.md Appendix A.2; CPU restatement: oracle/emer_oracle.c (orc_render_weights, orc_accumulate).
//
// Mapping: one 64-lane wavefront owns one ray; lane i holds sample 64*c + i of chunk c.  The
// exclusive cumsum of sigma*dt (and the reverse-mode suffix sums) are wave scans built from DPP
// shuffles -- no LDS traffic, no atomics, fully coalesced (R,S) loads/stores.  S is unbounded
// (chunks of 64 with a carried prefix); at the metric shape S=128 a ray is exactly two chunks.
#include "common.h"

namespace emer {

constexpr int kRaysPerBlockC = 4;

__global__ __launch_bounds__(256) void render_weights_fwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                                 const float *__restrict__ sigma, int64_t R, int32_t S,
                                                                 float *__restrict__ weights, float *__restrict__ trans,
                                                                 float *__restrict__ alphas, float *__restrict__ cdfs,
                                                                 float *__restrict__ ray_stats, float *__restrict__ t_mid,
                                                                 float *__restrict__ t_dist) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    if (r >= R) return;
    float carry = 0.0f;          // sum of sigma*dt over previous chunks
    float wsum = 0.0f, wmid = 0.0f, wcarry = 0.0f;
    float median = 0.0f;
    bool found = false;
    float last_mid = 0.0f;
    for (int32_t base = 0; base < S; base += kWave) {
        const int32_t s = base + lane;
        const bool ok = s < S;
        const int64_t i = r * S + s;
        const float a = ok ? ts[i] : 0.0f, b = ok ? te[i] : 0.0f, sg = ok ? sigma[i] : 0.0f;
        const float sdt = sg * (b - a);
        const float incl = wave_inclusive_sum(sdt, lane);
        const float excl = carry + (incl - sdt);
        const float T = expf(-excl);
        const float al = 1.0f - expf(-sdt);
        const float w = ok ? T * al : 0.0f;
        if (ok) {
            if (weights) weights[i] = w;
            if (trans) trans[i] = T;
            if (alphas) alphas[i] = al;
            if (cdfs) cdfs[r * (int64_t)(S + 1) + s] = 1.0f - T;
            if (t_mid) t_mid[i] = (a + b) / 2.0f;  // extras["t_vals"], extras["t_dist"] of rendering (render_utils.py:84-85)
            if (t_dist) t_dist[i] = b - a;
        }
        carry += __shfl(incl, kWave - 1, kWave);
        if (ray_stats) {
            const float mid = (a + b) / 2.0f;
            // median depth: first sample whose inclusive cumsum(w) reaches 0.5 (render_utils.py:107-115)
            const float cw = wcarry + wave_inclusive_sum(w, lane);
            const unsigned long long hit = __ballot(ok && cw >= 0.5f);
            if (!found && hit) {
                const int first = __ffsll((long long)hit) - 1;
                median = __shfl(mid, first, kWave);
                found = true;
            }
            const int last_lane = (S - 1 - base) < (kWave - 1) ? (S - 1 - base) : (kWave - 1);
            last_mid = __shfl(mid, last_lane, kWave);
            wcarry = __shfl(cw, kWave - 1, kWave);
            wsum += w;
            wmid += w * mid;
        }
    }
    if (cdfs && lane == 0) cdfs[r * (int64_t)(S + 1) + S] = 1.0f;  // 1 - [T, 0]
    if (ray_stats) {
        wsum = wave_sum(wsum);
        wmid = wave_sum(wmid);
        if (lane == 0) {
            float4 st = make_float4(wsum, wmid, found ? median : last_mid, 0.0f);
            *reinterpret_cast<float4 *>(ray_stats + r * 4) = st;
        }
    }
}

// Reverse mode.  With a_i = sigma_i*dt_i, T_i = exp(-sum_{j<i} a_j), w_i = T_i (1 - exp(-a_i)):
//   dL/da_i = gw_i * T_{i+1} - sum_{k>i} (gw_k w_k + gT_k T_k) + galpha_i * exp(-a_i),   T_{i+1} = T_i exp(-a_i)
// with gw_i = d_weights_i + g(sum w) + g(sum w*mid) * mid_i (alpha_i = 1 - exp(-a_i) is a differentiable output too:
// the reference's render_utils.py:73-77 forms weights = trans * alphas itself).
__global__ __launch_bounds__(256) void render_weights_bwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                                 const float *__restrict__ sigma,
                                                                 const float *__restrict__ d_weights,
                                                                 const float *__restrict__ d_trans,
                                                                 const float *__restrict__ d_alphas,
                                                                 const float *__restrict__ d_ray_stats, int64_t R, int32_t S,
                                                                 float *__restrict__ d_sigma) {
    __shared__ float chunk_base[kRaysPerBlockC][64];  // per-wave exclusive prefix of each 64-chunk (S <= 4096)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    const bool active = r < R;
    const int32_t n_chunks = (S + kWave - 1) / kWave;
    if (active) {
        float carry = 0.0f;
        for (int32_t c = 0; c < n_chunks; ++c) {
            const int32_t s = c * kWave + lane;
            const int64_t i = r * S + s;
            const float sdt = s < S ? sigma[i] * (te[i] - ts[i]) : 0.0f;
            if (lane == 0) chunk_base[wave][c] = carry;
            carry += wave_sum(sdt);
        }
    }
    __syncthreads();
    if (!active) return;
    const float g0 = d_ray_stats ? d_ray_stats[r * 4 + 0] : 0.0f;  // [R,4] like ray_stats (columns 2, 3 carry no gradient)
    const float g1 = d_ray_stats ? d_ray_stats[r * 4 + 1] : 0.0f;
    float suffix = 0.0f;  // sum over later chunks of (gw w + gT T)
    for (int32_t c = n_chunks - 1; c >= 0; --c) {
        const int32_t s = c * kWave + lane;
        const bool ok = s < S;
        const int64_t i = r * S + s;
        const float a = ok ? ts[i] : 0.0f, b = ok ? te[i] : 0.0f, sg = ok ? sigma[i] : 0.0f;
        const float dt = b - a;
        const float sdt = sg * dt;
        const float incl = wave_inclusive_sum(sdt, lane);
        const float excl = chunk_base[wave][c] + (incl - sdt);
        const float T = expf(-excl);
        const float e = expf(-sdt);
        const float w = T * (1.0f - e);
        float gw = (ok && d_weights) ? d_weights[i] : 0.0f;
        gw += g0 + g1 * ((a + b) / 2.0f);
        const float gT = (ok && d_trans) ? d_trans[i] : 0.0f;
        const float gA = (ok && d_alphas) ? d_alphas[i] : 0.0f;
        const float term = ok ? gw * w + gT * T : 0.0f;
        const float sfx_incl = wave_inclusive_suffix_sum(term, lane);
        const float later = suffix + (sfx_incl - term);  // strictly-later samples
        if (ok) d_sigma[i] = dt * (gw * T * e - later + gA * e);
        suffix += __shfl(sfx_incl, 0, kWave);
    }
}

// out[r,c] = sum_s w[r,s] * values[r,s,c].  Few channels (<= 8): lanes walk samples, one wave
// reduction per channel.  Many channels: lanes walk channels (coalesced rows), w broadcast.
template <int C>
__global__ __launch_bounds__(256) void accumulate_fwd_small_kernel(const float *__restrict__ w, const float *__restrict__ v,
                                                                   int64_t R, int32_t S, float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    if (r >= R) return;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0f;
    for (int32_t s = lane; s < S; s += kWave) {
        const float ws = w[r * S + s];
        if (v) {
            const float *vp = v + (r * S + s) * (int64_t)C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += ws * vp[c];
        } else {
            acc[0] += ws;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) out[r * C + c] = acc[c];
    }
}

__global__ __launch_bounds__(256) void accumulate_fwd_wide_kernel(const float *__restrict__ w, const float *__restrict__ v,
                                                                  int64_t R, int32_t S, int32_t C, float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    if (r >= R) return;
    for (int32_t c0 = 0; c0 < C; c0 += kWave) {
        const int32_t c = c0 + lane;
        float acc = 0.0f;
        if (c < C)
            for (int32_t s = 0; s < S; ++s) acc += w[r * S + s] * v[(r * S + s) * (int64_t)C + c];
        if (c < C) out[r * (int64_t)C + c] = acc;
    }
}

template <int C>
__global__ __launch_bounds__(256) void accumulate_bwd_small_kernel(const float *__restrict__ w, const float *__restrict__ v,
                                                                   const float *__restrict__ d_out, int64_t R, int32_t S,
                                                                   float *__restrict__ d_w, float *__restrict__ d_v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // flat (r, s)
    if (i >= R * S) return;
    const int64_t r = i / S;
    float g[C];
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = d_out[r * C + c];
    if (d_w) {
        float a = 0.0f;
        if (v) {
#pragma unroll
            for (int c = 0; c < C; ++c) a += g[c] * v[i * C + c];
        } else {
            a = g[0];
        }
        d_w[i] = a;
    }
    if (d_v) {
        const float ws = w[i];
#pragma unroll
        for (int c = 0; c < C; ++c) d_v[i * C + c] = ws * g[c];
    }
}

__global__ __launch_bounds__(256) void accumulate_bwd_wide_kernel(const float *__restrict__ w, const float *__restrict__ v,
                                                                  const float *__restrict__ d_out, int64_t R, int32_t S,
                                                                  int32_t C, float *__restrict__ d_w, float *__restrict__ d_v) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    if (r >= R) return;
    for (int32_t s = 0; s < S; ++s) {
        const float ws = w[r * S + s];
        float dot = 0.0f;
        for (int32_t c = lane; c < C; c += kWave) {
            const float g = d_out[r * (int64_t)C + c];
            const int64_t j = (r * S + s) * (int64_t)C + c;
            if (d_w) dot += g * v[j];
            if (d_v) d_v[j] = ws * g;
        }
        if (d_w) {
            dot = wave_sum(dot);
            if (lane == 0) d_w[r * S + s] = dot;
        }
    }
}


// ---------------------------------------------------------------------------- static / dynamic / shadow blend
// rendering's decomposed colour path (radiance_fields/render_utils.py:125-173) in one pass per direction:
//   a = sigma_s / (sigma + 1e-6), b = sigma_d / (sigma + 1e-6)                       (:131-136)
//   rgb = a * rgb_s * (1 - shadow) + b * rgb_d                                        (:169-173)
//   acc_rgb[r] = sum_s w * rgb,   acc_shadow[r] = sum_s w * shadow^2                  (:165-168,175)
// The reference materialises a, b and the blended [R,S,3] colour (plus their autograd graph: ~30 elementwise
// launches on 12 MB tensors); here one wave owns a ray, lanes walk its samples.
__global__ __launch_bounds__(256) void blend_accumulate_fwd_kernel(const float *__restrict__ w, const float *__restrict__ sig,
                                                                   const float *__restrict__ sig_s, const float *__restrict__ sig_d,
                                                                   const float *__restrict__ rgb_s, const float *__restrict__ rgb_d,
                                                                   const float *__restrict__ shadow, int64_t R, int32_t S,
                                                                   float *__restrict__ acc_rgb, float *__restrict__ acc_shadow) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    if (r >= R) return;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, cs = 0.0f;
    for (int32_t s = lane; s < S; s += kWave) {
        const int64_t i = r * S + s;
        const float wi = w[i], inv = 1.0f / (sig[i] + 1e-6f);
        const float a = sig_s[i] * inv, b = sig_d[i] * inv;
        const float sh = shadow ? shadow[i] : 0.0f;
        const float ka = a * (1.0f - sh);
        c0 += wi * (ka * rgb_s[i * 3 + 0] + b * rgb_d[i * 3 + 0]);
        c1 += wi * (ka * rgb_s[i * 3 + 1] + b * rgb_d[i * 3 + 1]);
        c2 += wi * (ka * rgb_s[i * 3 + 2] + b * rgb_d[i * 3 + 2]);
        cs += wi * sh * sh;
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2); cs = wave_sum(cs);
    if (lane == 0) {
        acc_rgb[r * 3 + 0] = c0; acc_rgb[r * 3 + 1] = c1; acc_rgb[r * 3 + 2] = c2;
        if (acc_shadow) acc_shadow[r] = cs;
    }
}

__global__ __launch_bounds__(256) void blend_accumulate_bwd_kernel(const float *__restrict__ w, const float *__restrict__ sig,
                                                                   const float *__restrict__ sig_s, const float *__restrict__ sig_d,
                                                                   const float *__restrict__ rgb_s, const float *__restrict__ rgb_d,
                                                                   const float *__restrict__ shadow, const float *__restrict__ g_rgb,
                                                                   const float *__restrict__ g_shadow, int64_t R, int32_t S,
                                                                   float *__restrict__ d_w, float *__restrict__ d_sig,
                                                                   float *__restrict__ d_sig_s, float *__restrict__ d_sig_d,
                                                                   float *__restrict__ d_rgb_s, float *__restrict__ d_rgb_d,
                                                                   float *__restrict__ d_shadow) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    if (r >= R) return;
    const float g0 = g_rgb ? g_rgb[r * 3 + 0] : 0.0f, g1 = g_rgb ? g_rgb[r * 3 + 1] : 0.0f, g2 = g_rgb ? g_rgb[r * 3 + 2] : 0.0f;
    const float gs = g_shadow ? g_shadow[r] : 0.0f;
    for (int32_t s = lane; s < S; s += kWave) {
        const int64_t i = r * S + s;
        const float wi = w[i], inv = 1.0f / (sig[i] + 1e-6f);
        const float ss = sig_s[i], sd = sig_d[i];
        const float a = ss * inv, b = sd * inv;
        const float sh = shadow ? shadow[i] : 0.0f;
        const float s0 = rgb_s[i * 3 + 0], s1 = rgb_s[i * 3 + 1], s2 = rgb_s[i * 3 + 2];
        const float e0 = rgb_d[i * 3 + 0], e1 = rgb_d[i * 3 + 1], e2 = rgb_d[i * 3 + 2];
        const float ka = a * (1.0f - sh);
        const float gS = g0 * s0 + g1 * s1 + g2 * s2;  // <g, rgb_s>
        const float gD = g0 * e0 + g1 * e1 + g2 * e2;  // <g, rgb_d>
        if (d_w) d_w[i] = ka * gS + b * gD + gs * sh * sh;
        const float wka = wi * ka, wb = wi * b;
        if (d_rgb_s) { d_rgb_s[i * 3 + 0] = g0 * wka; d_rgb_s[i * 3 + 1] = g1 * wka; d_rgb_s[i * 3 + 2] = g2 * wka; }
        if (d_rgb_d) { d_rgb_d[i * 3 + 0] = g0 * wb; d_rgb_d[i * 3 + 1] = g1 * wb; d_rgb_d[i * 3 + 2] = g2 * wb; }
        if (d_shadow) d_shadow[i] = wi * (2.0f * gs * sh - a * gS);
        const float da = wi * (1.0f - sh) * gS, db = wi * gD;
        if (d_sig_s) d_sig_s[i] = da * inv;
        if (d_sig_d) d_sig_d[i] = db * inv;
        if (d_sig) d_sig[i] = -(da * ss + db * sd) * inv * inv;
    }
}

// [r4] The same blend for the WIDE feature channels of the decomposed feature head (render_utils.py:247-252):
//   acc[r][c] = sum_s w * (a * feat_s[c] + b * feat_d[c]),  a, b as above.
// The reference materialises both ratios with a trailing axis, two [R,S,C] products, their sum and the accumulation -- with autograd,
// ~15 elementwise launches on 67 MB tensors at the 2048-ray shard.  One wave owns a ray, lanes walk the channels (coalesced rows), the
// per-sample scalars are wave-uniform loads; the backward needs two wave reductions per sample (<g, feat_s>, <g, feat_d>).
__global__ __launch_bounds__(256) void blend_accumulate_wide_fwd_kernel(const float *__restrict__ w, const float *__restrict__ sig,
                                                                        const float *__restrict__ sig_s, const float *__restrict__ sig_d,
                                                                        const float *__restrict__ f_s, const float *__restrict__ f_d, int64_t R,
                                                                        int32_t S, int32_t C, float *__restrict__ acc) {
    // one workgroup per ray: wave k takes the samples k, k + 4, ... (two at a time: two independent chains), the four partial sums meet in LDS
    __shared__ float part[4][kWave];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x;
    for (int32_t c0 = 0; c0 < C; c0 += kWave) {
        const int32_t c = c0 + lane;
        const bool on = c < C;
        float a0 = 0.0f, a1 = 0.0f;
        int32_t s = wave;
        for (; s + 4 < S; s += 8) {
            const int64_t i = r * S + s, i2 = i + 4;
            const float inv0 = 1.0f / (sig[i] + 1e-6f), inv1 = 1.0f / (sig[i2] + 1e-6f);
            const float wa0 = sig_s[i] * inv0, wb0 = sig_d[i] * inv0, wa1 = sig_s[i2] * inv1, wb1 = sig_d[i2] * inv1;
            const float fs0 = on ? f_s[i * C + c] : 0.0f, fd0 = on ? f_d[i * C + c] : 0.0f;
            const float fs1 = on ? f_s[i2 * C + c] : 0.0f, fd1 = on ? f_d[i2 * C + c] : 0.0f;
            a0 += w[i] * (wa0 * fs0 + wb0 * fd0);
            a1 += w[i2] * (wa1 * fs1 + wb1 * fd1);
        }
        if (s < S) {
            const int64_t i = r * S + s;
            const float inv = 1.0f / (sig[i] + 1e-6f);
            const float fs0 = on ? f_s[i * C + c] : 0.0f, fd0 = on ? f_d[i * C + c] : 0.0f;
            a0 += w[i] * (sig_s[i] * inv * fs0 + sig_d[i] * inv * fd0);
        }
        part[wave][lane] = a0 + a1;
        __syncthreads();
        if (wave == 0 && on) acc[r * (int64_t)C + c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void blend_accumulate_wide_bwd_kernel(const float *__restrict__ w, const float *__restrict__ sig,
                                                                        const float *__restrict__ sig_s, const float *__restrict__ sig_d,
                                                                        const float *__restrict__ f_s, const float *__restrict__ f_d,
                                                                        const float *__restrict__ g_acc, int64_t R, int32_t S, int32_t C,
                                                                        float *__restrict__ d_w, float *__restrict__ d_sig,
                                                                        float *__restrict__ d_sig_s, float *__restrict__ d_sig_d,
                                                                        float *__restrict__ d_f_s, float *__restrict__ d_f_d) {
    // samples are independent in the backward: a wave takes FOUR consecutive samples at a time (flat over (ray, sample); their eight
    // feature rows are in flight together, the eight dot products reduce interleaved, lanes 0..3 write the per-sample scalars)
    const int lane = threadIdx.x & 63;
    const int64_t n = R * (int64_t)S, n_quads = (n + 3) >> 2, n_waves = (int64_t)gridDim.x * 4;
    for (int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_quads; q += n_waves) {
        float gS[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gD[4] = {0.0f, 0.0f, 0.0f, 0.0f}, wa[4], wb[4];
        int64_t rr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = q * 4 + u < n ? q * 4 + u : n - 1;
            rr[u] = i / S;
            const float inv = 1.0f / (sig[i] + 1e-6f);
            wa[u] = w[i] * sig_s[i] * inv;
            wb[u] = w[i] * sig_d[i] * inv;
        }
        for (int32_t c = lane; c < C; c += kWave) {
            float fs[4], fd[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = q * 4 + u < n ? q * 4 + u : n - 1;
                g[u] = g_acc[rr[u] * (int64_t)C + c];
                fs[u] = f_s[i * C + c];
                fd[u] = f_d[i * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = q * 4 + u;
                gS[u] += g[u] * fs[u];
                gD[u] += g[u] * fd[u];
                if (i < n) {
                    if (d_f_s) d_f_s[i * C + c] = g[u] * wa[u];
                    if (d_f_d) d_f_d[i * C + c] = g[u] * wb[u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { gS[u] = wave_sum(gS[u]); gD[u] = wave_sum(gD[u]); }
        const int u = lane & 3;
        const int64_t i = q * 4 + u;
        if (lane < 4 && i < n) {
            const float mS = u == 0 ? gS[0] : u == 1 ? gS[1] : u == 2 ? gS[2] : gS[3];
            const float mD = u == 0 ? gD[0] : u == 1 ? gD[1] : u == 2 ? gD[2] : gD[3];
            const float wi = w[i], ss = sig_s[i], sd = sig_d[i], inv = 1.0f / (sig[i] + 1e-6f);
            const float a_ = ss * inv, b_ = sd * inv;
            if (d_w) d_w[i] = a_ * mS + b_ * mD;
            const float da = wi * mS, db = wi * mD;
            if (d_sig_s) d_sig_s[i] = da * inv;
            if (d_sig_d) d_sig_d[i] = db * inv;
            if (d_sig) d_sig[i] = -(da * ss + db * sd) * inv * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ [r4]
// The static model's `rendering` (render_utils.py:73-122,158-159,217-220) as ONE launch each way: weights / transmittance scan,
// accumulation of the 3-channel colour, and the per-ray epilogue (opacity clamp, expected depth, median depth, sky composite) --
// three launches forward and three backward before, on tensors of a few MB.  Same arithmetic, same order as the three kernels
// (render_weights_fwd / accumulate_fwd_small<3> / ray_epilogue_fwd): results are bitwise theirs.  rgb == nullptr: geometry only
// (the density-only render of the lidar step).
__global__ __launch_bounds__(256) void composite_rgb_fwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                                const float *__restrict__ sigma, const float *__restrict__ rgb,
                                                                const float *__restrict__ rgb_sky, int64_t R, int32_t S,
                                                                float *__restrict__ weights, float *__restrict__ trans,
                                                                float *__restrict__ t_mid, float *__restrict__ t_dist,
                                                                float *__restrict__ ray_stats, float *__restrict__ opacity,
                                                                float *__restrict__ depth, float *__restrict__ median_out,
                                                                float *__restrict__ rgb_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    if (r >= R) return;
    float carry = 0.0f, wsum = 0.0f, wmid = 0.0f, wcarry = 0.0f, median = 0.0f, last_mid = 0.0f;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    bool found = false;
    for (int32_t base = 0; base < S; base += kWave) {
        const int32_t s = base + lane;
        const bool ok = s < S;
        const int64_t i = r * S + s;
        const float a = ok ? ts[i] : 0.0f, b = ok ? te[i] : 0.0f, sg = ok ? sigma[i] : 0.0f;
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        if (ok && rgb) { c0 = rgb[i * 3 + 0]; c1 = rgb[i * 3 + 1]; c2 = rgb[i * 3 + 2]; }
        const float sdt = sg * (b - a);
        const float incl = wave_inclusive_sum(sdt, lane);
        const float excl = carry + (incl - sdt);
        const float T = expf(-excl);
        const float al = 1.0f - expf(-sdt);
        const float w = ok ? T * al : 0.0f;
        const float mid = (a + b) / 2.0f;
        if (ok) {
            weights[i] = w;
            if (trans) trans[i] = T;
            if (t_mid) t_mid[i] = mid;
            if (t_dist) t_dist[i] = b - a;
            acc[0] += w * c0; acc[1] += w * c1; acc[2] += w * c2;
        }
        carry += __shfl(incl, kWave - 1, kWave);
        const float cw = wcarry + wave_inclusive_sum(w, lane);
        const unsigned long long hit = __ballot(ok && cw >= 0.5f);
        if (!found && hit) {
            const int first = __ffsll((long long)hit) - 1;
            median = __shfl(mid, first, kWave);
            found = true;
        }
        const int last_lane = (S - 1 - base) < (kWave - 1) ? (S - 1 - base) : (kWave - 1);
        last_mid = __shfl(mid, last_lane, kWave);
        wcarry = __shfl(cw, kWave - 1, kWave);
        wsum += w;
        wmid += w * mid;
    }
    wsum = wave_sum(wsum);
    wmid = wave_sum(wmid);
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
        const float med = found ? median : last_mid;
        *reinterpret_cast<float4 *>(ray_stats + r * 4) = make_float4(wsum, wmid, med, 0.0f);
        const float o = fminf(fmaxf(wsum, 1e-6f), 1.0f);  // torch.clamp(1e-6, 1.0)
        opacity[r] = o;
        depth[r] = wmid / o;
        if (median_out) median_out[r] = med;
        if (rgb_out) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = acc[c];
                if (rgb_sky) v = v + rgb_sky[r * 3 + c] * (1.0f - o);
                rgb_out[r * 3 + c] = v;
            }
        }
    }
}

// Reverse of the above: ray_epilogue_bwd (per ray, by every lane), accumulate_bwd_small<3> and render_weights_bwd in one pass.
// d_weights / d_trans: gradients that reach the extras from other consumers (may be null).
__global__ __launch_bounds__(256) void composite_rgb_bwd_kernel(const float *__restrict__ ts, const float *__restrict__ te,
                                                                const float *__restrict__ sigma, const float *__restrict__ rgb,
                                                                const float *__restrict__ rgb_sky, const float *__restrict__ weights,
                                                                const float *__restrict__ ray_stats, const float *__restrict__ d_rgb_out,
                                                                const float *__restrict__ d_opacity, const float *__restrict__ d_depth,
                                                                const float *__restrict__ d_weights, const float *__restrict__ d_trans,
                                                                int64_t R, int32_t S, float *__restrict__ d_sigma,
                                                                float *__restrict__ d_rgb, float *__restrict__ d_rgb_sky) {
    __shared__ float chunk_base[kRaysPerBlockC][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlockC + wave;
    const bool active = r < R;
    const int32_t n_chunks = (S + kWave - 1) / kWave;
    if (active) {
        float carry = 0.0f;
        for (int32_t c = 0; c < n_chunks; ++c) {
            const int32_t s = c * kWave + lane;
            const int64_t i = r * S + s;
            const float sdt = s < S ? sigma[i] * (te[i] - ts[i]) : 0.0f;
            if (lane == 0) chunk_base[wave][c] = carry;
            carry += wave_sum(sdt);
        }
    }
    __syncthreads();
    if (!active) return;
    // per-ray part (ray_epilogue_bwd): gradient of (sum w, sum w mid) and of the accumulated colour
    const float4 st = *reinterpret_cast<const float4 *>(ray_stats + r * 4);
    const float o = fminf(fmaxf(st.x, 1e-6f), 1.0f);
    float go = d_opacity ? d_opacity[r] : 0.0f;
    const float gd = d_depth ? d_depth[r] : 0.0f;
    go -= gd * st.y / (o * o);
    float g[3] = {0.0f, 0.0f, 0.0f};
    if (d_rgb_out) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g[c] = d_rgb_out[r * 3 + c];
            if (rgb_sky) {
                go -= g[c] * rgb_sky[r * 3 + c];
                if (d_rgb_sky && lane == 0) d_rgb_sky[r * 3 + c] = g[c] * (1.0f - o);
            }
        }
    }
    const float g0 = (st.x >= 1e-6f && st.x <= 1.0f) ? go : 0.0f;  // clamp passes the gradient where min <= x <= max (torch semantics)
    const float g1 = gd / o;
    float suffix = 0.0f;
    for (int32_t c = n_chunks - 1; c >= 0; --c) {
        const int32_t s = c * kWave + lane;
        const bool ok = s < S;
        const int64_t i = r * S + s;
        const float a = ok ? ts[i] : 0.0f, b = ok ? te[i] : 0.0f, sg = ok ? sigma[i] : 0.0f;
        const float dt = b - a;
        const float sdt = sg * dt;
        const float incl = wave_inclusive_sum(sdt, lane);
        const float excl = chunk_base[wave][c] + (incl - sdt);
        const float T = expf(-excl);
        const float e = expf(-sdt);
        const float w = T * (1.0f - e);
        float gw = (ok && d_weights) ? d_weights[i] : 0.0f;
        if (ok && rgb && d_rgb_out) {   // accumulate_bwd: d_w += sum_c g_c rgb_c,  d_rgb = w g
            float acc = 0.0f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) acc += g[ch] * rgb[i * 3 + ch];
            gw += acc;
            if (d_rgb) {
                const float ws = weights[i];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) d_rgb[i * 3 + ch] = ws * g[ch];
            }
        }
        gw += g0 + g1 * ((a + b) / 2.0f);
        const float gT = (ok && d_trans) ? d_trans[i] : 0.0f;
        const float term = ok ? gw * w + gT * T : 0.0f;
        const float sfx_incl = wave_inclusive_suffix_sum(term, lane);
        const float later = suffix + (sfx_incl - term);
        if (ok) d_sigma[i] = dt * (gw * T * e - later);
        suffix += __shfl(sfx_incl, 0, kWave);
    }
}

}  // namespace emer

using namespace emer;

extern "C" int emer_composite_rgb_fwd(const float *ts, const float *te, const float *sigma, const float *rgb, const float *rgb_sky, int64_t R,
                                      int32_t S, float *weights, float *trans, float *t_mid, float *t_dist, float *ray_stats, float *opacity,
                                      float *depth, float *median_depth, float *rgb_out, void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1, "composite_rgb_fwd: bad sizes R=%lld S=%d", (long long)R, S);
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(ts && te && sigma && weights && ray_stats && opacity && depth && (!rgb_out || rgb), "composite_rgb_fwd: null pointer");
    hipLaunchKernelGGL(composite_rgb_fwd_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlockC)), dim3(256), 0, as_stream(stream), ts, te, sigma,
                       rgb_out ? rgb : nullptr, rgb_sky, R, S, weights, trans, t_mid, t_dist, ray_stats, opacity, depth, median_depth, rgb_out);
    return check_launch("composite_rgb_fwd");
}

extern "C" int emer_composite_rgb_bwd(const float *ts, const float *te, const float *sigma, const float *rgb, const float *rgb_sky,
                                      const float *weights, const float *ray_stats, const float *d_rgb_out, const float *d_opacity,
                                      const float *d_depth, const float *d_weights, const float *d_trans, int64_t R, int32_t S,
                                      float *d_sigma, float *d_rgb, float *d_rgb_sky, void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1 && S <= 4096, "composite_rgb_bwd: bad sizes R=%lld S=%d (S <= 4096)", (long long)R, S);
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(ts && te && sigma && ray_stats && d_sigma && (!d_rgb || (rgb && weights && d_rgb_out)), "composite_rgb_bwd: null pointer");
    hipLaunchKernelGGL(composite_rgb_bwd_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlockC)), dim3(256), 0, as_stream(stream), ts, te, sigma, rgb,
                       rgb_sky, weights, ray_stats, d_rgb_out, d_opacity, d_depth, d_weights, d_trans, R, S, d_sigma, d_rgb, d_rgb_sky);
    return check_launch("composite_rgb_bwd");
}

extern "C" int emer_blend_accumulate_wide_fwd(const float *weights, const float *density, const float *static_density, const float *dynamic_density,
                                             const float *static_feat, const float *dynamic_feat, int64_t R, int32_t S, int32_t C, float *acc,
                                             void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1 && C >= 1, "blend_accumulate_wide_fwd: bad sizes");
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(weights && density && static_density && dynamic_density && static_feat && dynamic_feat && acc, "blend_accumulate_wide_fwd: null pointer");
    hipLaunchKernelGGL(blend_accumulate_wide_fwd_kernel, dim3((uint32_t)R), dim3(256), 0, as_stream(stream), weights, density,
                       static_density, dynamic_density, static_feat, dynamic_feat, R, S, C, acc);
    return check_launch("blend_accumulate_wide_fwd");
}

extern "C" int emer_blend_accumulate_wide_bwd(const float *weights, const float *density, const float *static_density, const float *dynamic_density,
                                             const float *static_feat, const float *dynamic_feat, const float *d_acc, int64_t R, int32_t S, int32_t C,
                                             float *d_weights, float *d_density, float *d_static_density, float *d_dynamic_density,
                                             float *d_static_feat, float *d_dynamic_feat, void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1 && C >= 1, "blend_accumulate_wide_bwd: bad sizes");
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(weights && density && static_density && dynamic_density && static_feat && dynamic_feat && d_acc, "blend_accumulate_wide_bwd: null pointer");
    const int64_t wg = ceil_div(R * (int64_t)S, 4 * 8);   // ~two quads of samples per wave
    hipLaunchKernelGGL(blend_accumulate_wide_bwd_kernel, dim3((uint32_t)(wg > 16384 ? 16384 : wg)), dim3(256), 0, as_stream(stream), weights, density,
                       static_density, dynamic_density, static_feat, dynamic_feat, d_acc, R, S, C, d_weights, d_density, d_static_density,
                       d_dynamic_density, d_static_feat, d_dynamic_feat);
    return check_launch("blend_accumulate_wide_bwd");
}

extern "C" int emer_render_weights_fwd(const float *ts, const float *te, const float *sigma, int64_t R, int32_t S,
                                       float *weights, float *trans, float *alphas, float *cdfs, float *ray_stats,
                                       float *t_mid, float *t_dist, void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1, "render_weights_fwd: bad sizes R=%lld S=%d", (long long)R, S);
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(ts && te && sigma, "render_weights_fwd: null input");
    hipLaunchKernelGGL(render_weights_fwd_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlockC)), dim3(256), 0, as_stream(stream),
                       ts, te, sigma, R, S, weights, trans, alphas, cdfs, ray_stats, t_mid, t_dist);
    return check_launch("render_weights_fwd");
}

extern "C" int emer_render_weights_bwd(const float *ts, const float *te, const float *sigma, const float *d_weights,
                                       const float *d_trans, const float *d_alphas, const float *d_ray_stats, int64_t R,
                                       int32_t S, float *d_sigma, void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1 && S <= 4096, "render_weights_bwd: bad sizes R=%lld S=%d (S <= 4096)", (long long)R, S);
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(ts && te && sigma && d_sigma, "render_weights_bwd: null pointer");
    hipLaunchKernelGGL(render_weights_bwd_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlockC)), dim3(256), 0, as_stream(stream),
                       ts, te, sigma, d_weights, d_trans, d_alphas, d_ray_stats, R, S, d_sigma);
    return check_launch("render_weights_bwd");
}

extern "C" int emer_accumulate_fwd(const float *w, const float *v, int64_t R, int32_t S, int32_t C, float *out,
                                   void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1 && C >= 1, "accumulate_fwd: bad sizes");
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(w && out, "accumulate_fwd: null pointer");
    EMER_REQUIRE(v || C == 1, "accumulate_fwd: values == NULL requires n_channels == 1");
    const dim3 grid((uint32_t)ceil_div(R, kRaysPerBlockC)), block(256);
    hipStream_t st = as_stream(stream);
    switch (C) {
#define EMER_ACC(c) case c: hipLaunchKernelGGL(accumulate_fwd_small_kernel<c>, grid, block, 0, st, w, v, R, S, out); break;
        EMER_ACC(1) EMER_ACC(2) EMER_ACC(3) EMER_ACC(4) EMER_ACC(5) EMER_ACC(6) EMER_ACC(7) EMER_ACC(8)
#undef EMER_ACC
        default: hipLaunchKernelGGL(accumulate_fwd_wide_kernel, grid, block, 0, st, w, v, R, S, C, out);
    }
    return check_launch("accumulate_fwd");
}

extern "C" int emer_accumulate_bwd(const float *w, const float *v, const float *d_out, int64_t R, int32_t S, int32_t C,
                                   float *d_w, float *d_v, void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1 && C >= 1, "accumulate_bwd: bad sizes");
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(w && d_out, "accumulate_bwd: null pointer");
    EMER_REQUIRE(v || (C == 1 && !d_v), "accumulate_bwd: values == NULL requires n_channels == 1 and d_values == NULL");
    hipStream_t st = as_stream(stream);
    const dim3 block(256);
    if (C <= 8) {
        const dim3 grid((uint32_t)ceil_div(R * S, 256));
        switch (C) {
#define EMER_ACC(c) case c: hipLaunchKernelGGL(accumulate_bwd_small_kernel<c>, grid, block, 0, st, w, v, d_out, R, S, d_w, d_v); break;
            EMER_ACC(1) EMER_ACC(2) EMER_ACC(3) EMER_ACC(4) EMER_ACC(5) EMER_ACC(6) EMER_ACC(7) EMER_ACC(8)
#undef EMER_ACC
        }
    } else {
        hipLaunchKernelGGL(accumulate_bwd_wide_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlockC)), block, 0, st, w, v, d_out, R,
                           S, C, d_w, d_v);
    }
    return check_launch("accumulate_bwd");
}

extern "C" int emer_blend_accumulate_fwd(const float *weights, const float *density, const float *static_density,
                                         const float *dynamic_density, const float *static_rgb, const float *dynamic_rgb,
                                         const float *shadow_ratio, int64_t R, int32_t S, float *acc_rgb, float *acc_shadow_sq,
                                         void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1, "blend_accumulate_fwd: bad sizes");
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(weights && density && static_density && dynamic_density && static_rgb && dynamic_rgb && acc_rgb, "blend_accumulate_fwd: null pointer");
    EMER_REQUIRE(!acc_shadow_sq || shadow_ratio, "blend_accumulate_fwd: acc_shadow_sq needs shadow_ratio");
    hipLaunchKernelGGL(blend_accumulate_fwd_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlockC)), dim3(256), 0, as_stream(stream), weights,
                       density, static_density, dynamic_density, static_rgb, dynamic_rgb, shadow_ratio, R, S, acc_rgb, acc_shadow_sq);
    return check_launch("blend_accumulate_fwd");
}

extern "C" int emer_blend_accumulate_bwd(const float *weights, const float *density, const float *static_density,
                                         const float *dynamic_density, const float *static_rgb, const float *dynamic_rgb,
                                         const float *shadow_ratio, const float *d_acc_rgb, const float *d_acc_shadow_sq, int64_t R,
                                         int32_t S, float *d_weights, float *d_density, float *d_static_density,
                                         float *d_dynamic_density, float *d_static_rgb, float *d_dynamic_rgb, float *d_shadow_ratio,
                                         void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1, "blend_accumulate_bwd: bad sizes");
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(weights && density && static_density && dynamic_density && static_rgb && dynamic_rgb, "blend_accumulate_bwd: null pointer");
    EMER_REQUIRE(!d_shadow_ratio || shadow_ratio, "blend_accumulate_bwd: d_shadow_ratio needs shadow_ratio");
    hipLaunchKernelGGL(blend_accumulate_bwd_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlockC)), dim3(256), 0, as_stream(stream), weights,
                       density, static_density, dynamic_density, static_rgb, dynamic_rgb, shadow_ratio, d_acc_rgb, d_acc_shadow_sq, R, S,
                       d_weights, d_density, d_static_density, d_dynamic_density, d_static_rgb, d_dynamic_rgb, d_shadow_ratio);
    return check_launch("blend_accumulate_bwd");
}
