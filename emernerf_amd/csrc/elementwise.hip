// Per-sample / per-parameter elementwise kernels of the hot path (gfx950):
//   scene contraction fwd/bwd, ray sample points, sinusoidal direction encoding, fused Adam.
// All are pure HBM-streaming kernels (grid-stride, 256-thread blocks, capped at 2048 blocks).
//
// FMA contraction is disabled in this file: sample positions and contracted coordinates decide
// which grid cell a sample falls in, so they follow the reference's torch expression order exactly
// (radiance_fields/render_utils.py:318,341; nerf_utils.py:13-28; radiance_field.py:278-300).
#include "common.h"

#pragma clang fp contract(off)

#include "contract.h"

namespace emer {

__global__ __launch_bounds__(256) void contract_fwd_kernel(const float *__restrict__ pos, const float *__restrict__ aabb,
                                                           int unbounded, float *__restrict__ out, int64_t n) {
    const Aabb bb = load_aabb(aabb);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float p[3] = {pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]}, v[3];
        contract_point(bb, unbounded != 0, p, v);
        out[i * 3] = v[0]; out[i * 3 + 1] = v[1]; out[i * 3 + 2] = v[2];
    }
}

// gradient of contract_point w.r.t. the un-contracted position: g = d(loss)/d(contracted) in, d(loss)/d(p) out
__device__ __forceinline__ void contract_point_bwd(const Aabb &bb, bool unbounded, const float (&p)[3], float (&g)[3]) {
    float v[3];
    const bool inside = contract_point(bb, unbounded, p, v);
    if (!inside) { g[0] = 0.0f; g[1] = 0.0f; g[2] = 0.0f; return; }
    if (unbounded) {
        float u[3], mag = 0.0f;
        int k = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            u[d] = (p[d] - bb.lo[d]) / (bb.hi[d] - bb.lo[d]) * 2.0f - 1.0f;
            if (fabsf(u[d]) > mag) { mag = fabsf(u[d]); k = d; }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) g[d] = g[d] / 4.0f;  // y = c/4 + 0.5
        if (!(mag < 1.0f)) {
            // c_d = f(mag) u_d, f = 2/mag - 1/mag^2, mag = |u_k|
            const float inv = 1.0f / mag;
            const float f = 2.0f * inv - inv * inv;
            const float df = -2.0f * inv * inv + 2.0f * inv * inv * inv;
            const float dotug = u[0] * g[0] + u[1] * g[1] + u[2] * g[2];
            const float sgn = u[k] >= 0.0f ? 1.0f : -1.0f;
#pragma unroll
            for (int d = 0; d < 3; ++d) g[d] = f * g[d] + (d == k ? sgn * df * dotug : 0.0f);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) g[d] = g[d] * 2.0f;  // u = 2v - 1
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) g[d] = g[d] / (bb.hi[d] - bb.lo[d]);
}

__global__ __launch_bounds__(256) void contract_bwd_kernel(const float *__restrict__ pos, const float *__restrict__ aabb,
                                                           int unbounded, const float *__restrict__ dout,
                                                           float *__restrict__ dpos, int64_t n) {
    const Aabb bb = load_aabb(aabb);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float p[3] = {pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]};
        float g[3] = {dout[i * 3], dout[i * 3 + 1], dout[i * 3 + 2]};
        contract_point_bwd(bb, unbounded != 0, p, g);
        dpos[i * 3] = g[0]; dpos[i * 3 + 1] = g[1]; dpos[i * 3 + 2] = g[2];
    }
}

// ---- flow warp of the temporal aggregation (radiance_field.py:567-580): the xyzt query points of the batched flow branch -----------
// x3 [3 n][4] = [ normed | t ],  [ contract(pos + fwd_flow * noise) | clamp(t + dt * noise, 0, 1) ],  [ contract(pos + bwd_flow * noise) |
// clamp(t - dt * noise, 0, 1) ] and x2 [2 n][4] = the last two thirds again (the flow table's query is its own autograd output, so that
// the two consumers' input gradients arrive as two tensors instead of being padded, copied and added).  Replaces rand-free part of the
// torch chain: 2 mul + 2 add + 2 contractions + 2 scalar mul + add + sub + 2 clamp + 4 cat = 14 launches and their ~18 autograd twins.
__global__ __launch_bounds__(256) void flow_warp_fwd_kernel(const float *__restrict__ pos, const float *__restrict__ normed, const float *__restrict__ ts,
                                                            const float *__restrict__ flow, const float *__restrict__ noise, float dt,
                                                            const float *__restrict__ aabb, int unbounded, float4 *__restrict__ x3,
                                                            float4 *__restrict__ x2, int64_t n) {
    const Aabb bb = load_aabb(aabb);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float t = ts[i], nz = noise[i];
        x3[i] = make_float4(normed[i * 3], normed[i * 3 + 1], normed[i * 3 + 2], t);
        const float p[3] = {pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float q[3], v[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = p[d] + flow[i * 6 + 3 * h + d] * nz;   // positions + flow * noise (mul, then add: no FMA in this file)
            contract_point(bb, unbounded != 0, q, v);
            const float step = dt * nz;
            const float tw = fminf(fmaxf(h == 0 ? t + step : t - step, 0.0f), 1.0f);   // torch.clamp(t +- time_diff * noise, 0, 1.0)
            const float4 o = make_float4(v[0], v[1], v[2], tw);
            x3[(int64_t)(h + 1) * n + i] = o;
            x2[(int64_t)h * n + i] = o;
        }
    }
}

// dflow [n][6] from the input gradients of the two consumers: dx3 [3 n][4] (rows < n unused; may be null) and dx2 [2 n][4] (may be null);
// positions, timestamps and the noise carry no gradient
__global__ __launch_bounds__(256) void flow_warp_bwd_kernel(const float *__restrict__ pos, const float *__restrict__ flow, const float *__restrict__ noise,
                                                            const float *__restrict__ aabb, int unbounded, const float4 *__restrict__ dx3,
                                                            const float4 *__restrict__ dx2, float *__restrict__ dflow, int64_t n) {
    const Aabb bb = load_aabb(aabb);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float nz = noise[i];
        const float p[3] = {pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2]};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float q[3], g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int d = 0; d < 3; ++d) q[d] = p[d] + flow[i * 6 + 3 * h + d] * nz;
            if (dx3) { const float4 a = dx3[(int64_t)(h + 1) * n + i]; g[0] = a.x; g[1] = a.y; g[2] = a.z; }
            if (dx2) { const float4 a = dx2[(int64_t)h * n + i]; g[0] = g[0] + a.x; g[1] = g[1] + a.y; g[2] = g[2] + a.z; }
            contract_point_bwd(bb, unbounded != 0, q, g);
#pragma unroll
            for (int d = 0; d < 3; ++d) dflow[i * 6 + 3 * h + d] = g[d] * nz;
        }
    }
}

// positions = o + d * (t0 + t1) / 2 (evaluated as ((d * (t0 + t1)) / 2) + o, render_utils.py:341)
__global__ __launch_bounds__(256) void ray_points_kernel(const float *__restrict__ origins, const float *__restrict__ dirs,
                                                         const float *__restrict__ ts, const float *__restrict__ te,
                                                         const float *__restrict__ times, const float *__restrict__ aabb,
                                                         int unbounded, float *__restrict__ normed, int out_dim,
                                                         float *__restrict__ positions_out, int64_t R, int32_t S) {
    const Aabb bb = load_aabb(aabb);
    const int64_t n = R * S;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / S;
        const float tsum = ts[i] + te[i];
        float p[3], v[3];
        ray_point(origins + r * 3, dirs + r * 3, tsum, p);
        contract_point(bb, unbounded != 0, p, v);
        if (out_dim == 4) {
            *reinterpret_cast<float4 *>(normed + i * 4) = make_float4(v[0], v[1], v[2], times[r]);
        } else {
            normed[i * 3] = v[0]; normed[i * 3 + 1] = v[1]; normed[i * 3 + 2] = v[2];
        }
        if (positions_out) { positions_out[i * 3] = p[0]; positions_out[i * 3 + 1] = p[1]; positions_out[i * 3 + 2] = p[2]; }
    }
}

// SinusoidalEncoder(min_deg=0, max_deg) on (d+1)/2 (encodings.py:86-104; radiance_field.py:629):
//   out = [x (3), sin(2^i x_d) (i-major, 3 per degree), sin(2^i x_d + pi/2)]
__global__ __launch_bounds__(256) void dir_encode_kernel(const float *__restrict__ dirs, float *__restrict__ out, int64_t n,
                                                         int32_t max_deg, int remap) {
    const int32_t n_deg = max_deg + 1, width = max_deg == 0 ? 3 : 3 * (1 + 2 * n_deg);  // identity rows are 3 wide
    const float half_pi = 0.5f * 3.14159265358979323846f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = remap ? (dirs[i * 3 + d] + 1.0f) / 2.0f : dirs[i * 3 + d];
        float *o = out + i * width;
        if (max_deg == 0) {  // encoder degenerates to identity when max_deg == min_deg (encodings.py:93-94)
            o[0] = x[0]; o[1] = x[1]; o[2] = x[2];
            continue;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = x[d];
        float scale = 1.0f;
        for (int32_t k = 0; k < n_deg; ++k) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float xb = x[d] * scale;
                o[3 + k * 3 + d] = sinf(xb);
                o[3 + 3 * n_deg + k * 3 + d] = sinf(xb + half_pi);
            }
            scale *= 2.0f;
        }
    }
}

// torch.optim.Adam (weight_decay as L2, no amsgrad) on a flat buffer; builders.py:50-60.
__global__ __launch_bounds__(256) void adam_scalar_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                          float *__restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                          float wd, float gscale, float bc1, float bc2_sqrt) {
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.0f) gi = gi + wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * (mi / denom);
    }
}

__device__ __forceinline__ void adam_one(float &pi, float gi, float &mi, float &vi, float lr_step, float b1, float b2, float eps, float wd,
                                         float gscale, float bc2_sqrt) {
    gi = gi * gscale;
    if (wd != 0.0f) gi = gi + wd * pi;
    mi = mi + (gi - mi) * (1.0f - b1);            // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * b2 + (1.0f - b2) * gi * gi;          // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - lr_step * (mi / denom);
}
// [r4] 16 bytes per lane and array (the four arrays are views at the SAME offset of equally aligned buffers: one scalar head of
// `head` < 4 elements aligns them all), two vectors per thread and trip: three tables of a flow model (970 MB) 252 -> see DESIGN 5.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, int32_t head, float lr, float b1, float b2, float eps,
                                                   float wd, float gscale, float bc1, float bc2_sqrt) {
    const float step_size = lr / bc1;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthreads = (int64_t)gridDim.x * 256;
    const int64_t n4 = (n - head) >> 2;   // whole vectors behind the head
    float4 *p4 = reinterpret_cast<float4 *>(p + head), *m4 = reinterpret_cast<float4 *>(m + head), *v4 = reinterpret_cast<float4 *>(v + head);
    const float4 *g4 = reinterpret_cast<const float4 *>(g + head);
    for (int64_t i = tid; i < n4; i += 2 * nthreads) {
        const int64_t j = i + nthreads;
        const bool two = j < n4;
        float4 pa = p4[i], ga = g4[i], ma = m4[i], va = v4[i];
        float4 pb = pa, gb = ga, mb = ma, vb = va;
        if (two) { pb = p4[j]; gb = g4[j]; mb = m4[j]; vb = v4[j]; }
        adam_one(pa.x, ga.x, ma.x, va.x, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
        adam_one(pa.y, ga.y, ma.y, va.y, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
        adam_one(pa.z, ga.z, ma.z, va.z, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
        adam_one(pa.w, ga.w, ma.w, va.w, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
        m4[i] = ma; v4[i] = va; p4[i] = pa;
        if (two) {
            adam_one(pb.x, gb.x, mb.x, vb.x, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
            adam_one(pb.y, gb.y, mb.y, vb.y, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
            adam_one(pb.z, gb.z, mb.z, vb.z, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
            adam_one(pb.w, gb.w, mb.w, vb.w, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
            m4[j] = mb; v4[j] = vb; p4[j] = pb;
        }
    }
    // the (< 4)-element head and the (< 4)-element tail
    const int64_t tail0 = head + 4 * n4;
    if (tid < head + (n - tail0)) {
        const int64_t i = tid < head ? tid : tail0 + (tid - head);
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, step_size, b1, b2, eps, wd, gscale, bc2_sqrt);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}


// Layout glue between the grid kernels (level-major [L][N][F], coalesced per level) and the
// row-major [N, L*F] tensor the reference API exposes (tcnn_modules.py:263).  64-sample tiles go
// through LDS (pitch LF+1) so both the global reads and the global writes are fully coalesced.
__global__ __launch_bounds__(256) void layout_transpose_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                               int32_t L, int64_t N, int32_t F, int to_row_major) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int32_t LF = L * F, pitch = LF + 1;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int32_t rows = (int32_t)((N - n0) < 64 ? (N - n0) : 64);
    const int32_t per_level = rows * F;  // contiguous floats of one level inside this tile
    if (to_row_major) {
        for (int idx = threadIdx.x; idx < L * per_level; idx += 256) {
            const int l = idx / per_level, rem = idx - l * per_level;
            const int r = rem / F, f = rem - r * F;
            tile[r * pitch + l * F + f] = src[((int64_t)l * N + n0) * F + rem];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < rows * LF; idx += 256) {
            const int r = idx / LF, c = idx - r * LF;
            dst[n0 * LF + idx] = tile[r * pitch + c];
        }
    } else {
        for (int idx = threadIdx.x; idx < rows * LF; idx += 256) {
            const int r = idx / LF, c = idx - r * LF;
            tile[r * pitch + c] = src[n0 * LF + idx];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < L * per_level; idx += 256) {
            const int l = idx / per_level, rem = idx - l * per_level;
            const int r = rem / F, f = rem - r * F;
            dst[((int64_t)l * N + n0) * F + rem] = tile[r * pitch + l * F + f];
        }
    }
}


// density_activation = trunc_exp(x - 1) (radiance_field.py:28, nerf_utils.py:59-75) on a strided column
__global__ __launch_bounds__(256) void trunc_exp_fwd_kernel(const float *__restrict__ x, int64_t stride, float *__restrict__ y,
                                                            int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = expf(x[i * stride] - 1.0f);
}
__global__ __launch_bounds__(256) void trunc_exp_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                            float *__restrict__ dx, int64_t stride, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i * stride] = dy[i] * fminf(y[i], 3269017.3724721107f);  // g * exp(min(x - 1, 15))
}

// Temporal aggregation of the flow branch (radiance_field.py:553-620: features at the current, forward- and backward-warped
// positions, evaluated as ONE 3 N-row batch): out[i] = (x[i] + 0.5 x[N + i] + 0.5 x[2 N + i]) / 2 -- the reference's expression,
// term for term (the halvings are exact, so are the fused multiply-adds) -- and its backward dx = [g / 2 | g / 4 | g / 4].
// float4 streams: one read of the three thirds and one write, instead of five elementwise launches each way and a 3 N-row cat.
__global__ __launch_bounds__(256) void aggregate3_fwd_kernel(const float4 *__restrict__ x, int64_t n4, float4 *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = x[i], b = x[n4 + i], c = x[2 * n4 + i];
        float4 r;
        r.x = ((a.x + 0.5f * b.x) + 0.5f * c.x) / 2.0f;
        r.y = ((a.y + 0.5f * b.y) + 0.5f * c.y) / 2.0f;
        r.z = ((a.z + 0.5f * b.z) + 0.5f * c.z) / 2.0f;
        r.w = ((a.w + 0.5f * b.w) + 0.5f * c.w) / 2.0f;
        out[i] = r;
    }
}
// [r4] ... with the density read off column 0 of the aggregated features in the same launch (radiance_field.py:461: trunc_exp(f[..., 0] - 1)
// = exp(f - 1); its backward g * exp(min(f - 1, 15)) joins column 0 of the incoming gradient instead of travelling as a second, mostly zero
// [N, C] tensor that autograd has to add: the separate path cost a zero fill, a strided store and a 67 MB add per flow step).
// c4 = float4 groups per row; dens / d_dens [rows].
__global__ __launch_bounds__(256) void aggregate3_density_fwd_kernel(const float4 *__restrict__ x, int64_t n4, int32_t c4, float4 *__restrict__ out,
                                                                     float *__restrict__ dens) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 a = x[i], b = x[n4 + i], c = x[2 * n4 + i];
        float4 r;
        r.x = ((a.x + 0.5f * b.x) + 0.5f * c.x) / 2.0f;
        r.y = ((a.y + 0.5f * b.y) + 0.5f * c.y) / 2.0f;
        r.z = ((a.z + 0.5f * b.z) + 0.5f * c.z) / 2.0f;
        r.w = ((a.w + 0.5f * b.w) + 0.5f * c.w) / 2.0f;
        out[i] = r;
        if (i % c4 == 0) dens[i / c4] = expf(r.x - 1.0f);
    }
}
__global__ __launch_bounds__(256) void aggregate3_density_bwd_kernel(const float4 *__restrict__ g, const float *__restrict__ d_dens,
                                                                     const float *__restrict__ dens, int64_t n4, int32_t c4,
                                                                     float4 *__restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = g ? g[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (d_dens && i % c4 == 0) v.x = v.x + d_dens[i / c4] * fminf(dens[i / c4], 3269017.3724721107f);  // g * exp(min(x - 1, 15))
        const float4 h = {v.x / 2.0f, v.y / 2.0f, v.z / 2.0f, v.w / 2.0f};
        const float4 q = {0.5f * h.x, 0.5f * h.y, 0.5f * h.z, 0.5f * h.w};
        dx[i] = h;
        dx[n4 + i] = q;
        dx[2 * n4 + i] = q;
    }
}
__global__ __launch_bounds__(256) void aggregate3_bwd_kernel(const float4 *__restrict__ g, int64_t n4, float4 *__restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = g[i];
        const float4 h = {v.x / 2.0f, v.y / 2.0f, v.z / 2.0f, v.w / 2.0f};
        const float4 q = {0.5f * h.x, 0.5f * h.y, 0.5f * h.z, 0.5f * h.w};
        dx[i] = h;
        dx[n4 + i] = q;
        dx[2 * n4 + i] = q;
    }
}

static inline uint32_t stream_blocks(int64_t n) {
    const int64_t b = ceil_div(n, 256);
    return (uint32_t)(b < 2048 ? b : 2048);
}

}  // namespace emer

using namespace emer;

extern "C" int emer_contract_fwd(const float *pos, const float *aabb, int unbounded, float *out, int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0, "contract_fwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(pos && aabb && out, "contract_fwd: null pointer");
    hipLaunchKernelGGL(contract_fwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), pos, aabb, unbounded, out, n);
    return check_launch("contract_fwd");
}

extern "C" int emer_contract_bwd(const float *pos, const float *aabb, int unbounded, const float *dout, float *dpos,
                                 int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0, "contract_bwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(pos && aabb && dout && dpos, "contract_bwd: null pointer");
    hipLaunchKernelGGL(contract_bwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), pos, aabb, unbounded, dout,
                       dpos, n);
    return check_launch("contract_bwd");
}

extern "C" int emer_flow_warp_fwd(const float *positions, const float *normed, const float *timestamps, const float *flow, const float *noise,
                                  float time_diff, const float *aabb, int unbounded, float *x3, float *x2, int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0, "flow_warp_fwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(positions && normed && timestamps && flow && noise && aabb && x3 && x2, "flow_warp_fwd: null pointer");
    hipLaunchKernelGGL(flow_warp_fwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), positions, normed, timestamps, flow, noise,
                       time_diff, aabb, unbounded, reinterpret_cast<float4 *>(x3), reinterpret_cast<float4 *>(x2), n);
    return check_launch("flow_warp_fwd");
}

extern "C" int emer_flow_warp_bwd(const float *positions, const float *flow, const float *noise, const float *aabb, int unbounded,
                                  const float *dx3, const float *dx2, float *dflow, int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0, "flow_warp_bwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(positions && flow && noise && aabb && dflow, "flow_warp_bwd: null pointer");
    hipLaunchKernelGGL(flow_warp_bwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), positions, flow, noise, aabb, unbounded,
                       reinterpret_cast<const float4 *>(dx3), reinterpret_cast<const float4 *>(dx2), dflow, n);
    return check_launch("flow_warp_bwd");
}

extern "C" int emer_ray_points(const float *origins, const float *dirs, const float *ts, const float *te,
                               const float *times, const float *aabb, int unbounded, float *normed, int out_dim,
                               float *positions_out, int64_t R, int32_t S, void *stream) {
    EMER_REQUIRE(R >= 0 && S >= 1, "ray_points: bad sizes");
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(origins && dirs && ts && te && aabb && normed, "ray_points: null pointer");
    EMER_REQUIRE(out_dim == 3 || (out_dim == 4 && times), "ray_points: out_dim must be 3, or 4 with times != NULL");
    hipLaunchKernelGGL(ray_points_kernel, dim3(stream_blocks(R * S)), dim3(256), 0, as_stream(stream), origins, dirs, ts, te,
                       times, aabb, unbounded, normed, out_dim, positions_out, R, S);
    return check_launch("ray_points");
}

extern "C" int emer_dir_encode(const float *dirs, float *out, int64_t n, int32_t max_deg, int remap, void *stream) {
    EMER_REQUIRE(n >= 0 && max_deg >= 0 && max_deg <= 16, "dir_encode: bad arguments");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(dirs && out, "dir_encode: null pointer");
    hipLaunchKernelGGL(dir_encode_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), dirs, out, n, max_deg, remap);
    return check_launch("dir_encode");
}

extern "C" int emer_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float grad_scale, int32_t step,
                              void *stream) {
    EMER_REQUIRE(n >= 0 && step >= 1, "adam_step: bad arguments (step counts from 1)");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(params && grads && exp_avg && exp_avg_sq, "adam_step: null pointer");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    // the vector body needs all four arrays 16-byte aligned behind a common scalar head; otherwise everything is "head / tail" (scalar)
    const uintptr_t mis = (uintptr_t)params & 15u;
    const bool same = (((uintptr_t)grads & 15u) == mis) && (((uintptr_t)exp_avg & 15u) == mis) && (((uintptr_t)exp_avg_sq & 15u) == mis) && (mis & 3u) == 0;
    if (!same) {   // (never the case for the trainer's flat buffers) one element per thread
        hipLaunchKernelGGL(adam_scalar_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), params, grads, exp_avg, exp_avg_sq, n,
                           lr, beta1, beta2, eps, weight_decay, grad_scale, (float)bc1, (float)sqrt(bc2));
        return check_launch("adam_step");
    }
    int64_t head = mis ? (int64_t)((16u - mis) >> 2) : 0;
    if (head > n) head = n;
    hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks((n + 7) / 8 + 8)), dim3(256), 0, as_stream(stream), params, grads, exp_avg, exp_avg_sq, n,
                       (int32_t)head, lr, beta1, beta2, eps, weight_decay, grad_scale, (float)bc1, (float)sqrt(bc2));
    return check_launch("adam_step");
}

extern "C" int emer_layout_transpose(const float *src, float *dst, int32_t n_levels, int64_t n, int32_t n_features,
                                     int to_row_major, void *stream) {
    EMER_REQUIRE(n >= 0 && n_levels >= 1 && n_features >= 1 && n_levels * n_features <= 512, "layout_transpose: bad sizes");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(src && dst && src != dst, "layout_transpose: null or aliased pointers");
    const size_t lds = (size_t)64 * (n_levels * n_features + 1) * sizeof(float);  // up to 128.25 KiB at L*F = 512
    if (int rc = reserve_lds(reinterpret_cast<const void *>(layout_transpose_kernel), lds, "layout_transpose")) return rc;
    hipLaunchKernelGGL(layout_transpose_kernel, dim3((uint32_t)ceil_div(n, 64)), dim3(256), lds, as_stream(stream), src, dst,
                       n_levels, n, n_features, to_row_major);
    return check_launch("layout_transpose");
}

extern "C" int emer_trunc_exp_fwd(const float *x, int64_t x_stride, float *y, int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0 && x_stride >= 1, "trunc_exp_fwd: bad arguments");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x && y, "trunc_exp_fwd: null pointer");
    hipLaunchKernelGGL(trunc_exp_fwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), x, x_stride, y, n);
    return check_launch("trunc_exp_fwd");
}

extern "C" int emer_trunc_exp_bwd(const float *dy, const float *y, float *dx, int64_t dx_stride, int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0 && dx_stride >= 1, "trunc_exp_bwd: bad arguments");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(dy && y && dx, "trunc_exp_bwd: null pointer");
    hipLaunchKernelGGL(trunc_exp_bwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, as_stream(stream), dy, y, dx, dx_stride, n);
    return check_launch("trunc_exp_bwd");
}

// fp32 master table -> the fp16 copy the encoders read in half-precision mode (tcnn casts its fp32 master parameters to the
// parameter precision on every call: third_party/tcnn_modules.py:223-233,257-260).  Eight entries per lane: 32 B in, 16 B out.
namespace emer {
__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float *__restrict__ src, __half *__restrict__ dst, int64_t n) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0;   // a table inside a flat parameter buffer may sit at any float
    if (i0 + 8 <= n) {
        float4 a, b;
        if (aligned) {
            a = *reinterpret_cast<const float4 *>(src + i0); b = *reinterpret_cast<const float4 *>(src + i0 + 4);
        } else {
            a = make_float4(src[i0], src[i0 + 1], src[i0 + 2], src[i0 + 3]); b = make_float4(src[i0 + 4], src[i0 + 5], src[i0 + 6], src[i0 + 7]);
        }
        union { __half2 h[4]; uint4 u; } o;
        o.h[0] = __floats2half2_rn(a.x, a.y); o.h[1] = __floats2half2_rn(a.z, a.w);
        o.h[2] = __floats2half2_rn(b.x, b.y); o.h[3] = __floats2half2_rn(b.z, b.w);
        *reinterpret_cast<uint4 *>(dst + i0) = o.u;
    } else {
        for (int64_t i = i0; i < n; ++i) dst[i] = __float2half_rn(src[i]);
    }
}
}  // namespace emer

extern "C" int emer_cast_f32_f16(const float *src, void *dst_f16, int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0, "cast_f32_f16: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(src && dst_f16 && ((uintptr_t)dst_f16 % 16) == 0, "cast_f32_f16: null pointer or unaligned destination");
    hipLaunchKernelGGL(emer::cast_f32_f16_kernel, dim3((uint32_t)emer::ceil_div(n, 256 * 8)), dim3(256), 0, emer::as_stream(stream), src,
                       reinterpret_cast<__half *>(dst_f16), n);
    return emer::check_launch("cast_f32_f16");
}


extern "C" int emer_aggregate3_fwd(const float *x3, int64_t n, float *out, void *stream) {
    EMER_REQUIRE(n >= 0 && n % 4 == 0, "aggregate3_fwd: the element count of one third must be a multiple of 4");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x3 && out && (uintptr_t)x3 % 16 == 0 && (uintptr_t)out % 16 == 0, "aggregate3_fwd: null or unaligned pointer");
    hipLaunchKernelGGL(aggregate3_fwd_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(x3), n / 4,
                       reinterpret_cast<float4 *>(out));
    return check_launch("aggregate3_fwd");
}

extern "C" int emer_aggregate3_density_fwd(const float *x3, int64_t n_rows, int32_t n_cols, float *out, float *density, void *stream) {
    EMER_REQUIRE(n_rows >= 0 && n_cols >= 4 && n_cols % 4 == 0, "aggregate3_density_fwd: the row width must be a multiple of 4");
    if (n_rows == 0) return EMER_OK;
    EMER_REQUIRE(x3 && out && density && (uintptr_t)x3 % 16 == 0 && (uintptr_t)out % 16 == 0, "aggregate3_density_fwd: null or unaligned pointer");
    const int64_t n4 = n_rows * (n_cols / 4);
    hipLaunchKernelGGL(aggregate3_density_fwd_kernel, dim3(stream_blocks(n4)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(x3), n4,
                       n_cols / 4, reinterpret_cast<float4 *>(out), density);
    return check_launch("aggregate3_density_fwd");
}

extern "C" int emer_aggregate3_density_bwd(const float *g, const float *d_density, const float *density, int64_t n_rows, int32_t n_cols, float *dx3,
                                           void *stream) {
    EMER_REQUIRE(n_rows >= 0 && n_cols >= 4 && n_cols % 4 == 0, "aggregate3_density_bwd: the row width must be a multiple of 4");
    if (n_rows == 0) return EMER_OK;
    EMER_REQUIRE(dx3 && (!d_density || density) && (!g || (uintptr_t)g % 16 == 0) && (uintptr_t)dx3 % 16 == 0, "aggregate3_density_bwd: null or unaligned pointer");
    const int64_t n4 = n_rows * (n_cols / 4);
    hipLaunchKernelGGL(aggregate3_density_bwd_kernel, dim3(stream_blocks(n4)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(g),
                       d_density, density, n4, n_cols / 4, reinterpret_cast<float4 *>(dx3));
    return check_launch("aggregate3_density_bwd");
}

extern "C" int emer_aggregate3_bwd(const float *g, int64_t n, float *dx3, void *stream) {
    EMER_REQUIRE(n >= 0 && n % 4 == 0, "aggregate3_bwd: the element count of one third must be a multiple of 4");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(g && dx3 && (uintptr_t)g % 16 == 0 && (uintptr_t)dx3 % 16 == 0, "aggregate3_bwd: null or unaligned pointer");
    hipLaunchKernelGGL(aggregate3_bwd_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4 *>(g), n / 4,
                       reinterpret_cast<float4 *>(dx3));
    return check_launch("aggregate3_bwd");
}
