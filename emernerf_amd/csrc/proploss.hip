// Proposal-network supervision losses for gfx950: the zip-NeRF anti-aliased interlevel loss and the plain
// histogram ("pdf") loss, forward value and gradient in ONE launch per proposal level.
//
// Replaces, behind PropNetEstimator.compute_loss (third_party/nerfacc_prop_net.py:181-238): blur_stepfun (:22-34),
// sorted_interp_quad (:37-60), the hinge loss (:222-226) and _pdf_loss (:342-362) -- in the reference a torch.sort of
// 2(S+1) values per ray (a 105 us radix sort at the metric shape), two cumsums, [R, 2S+2, m] boolean masks with four
// fp32 temporaries of that shape (O(R m S) memory: 1.1 GB each at R=8192, S=128, m=129) and their autograd graph.
//
// Mapping: one 64-lane wavefront owns one ray; everything lives in wave-private LDS (about 5 KB per ray at S=128).
//   1. w_n = diff(cdf) / diff(s) of the FINAL samples (cdf = 1 - [trans, 0], no gradient: :186-187,203-205);
//   2. the blurred step function needs sort(cat[s - r, s + r]).  Both halves are already sorted, so the sort is a
//      MERGE: element j of one half lands at j + (number of elements of the other half before it), one binary search
//      each -- no sorting network, no radix sort.  Ties may be ordered either way: a tie contributes a zero-width
//      segment to every sum below;
//   3. three chunked wave scans (slope -> blurred pdf, clamped at 0 -> blurred cdf), DPP/bpermute adds, carries in
//      registers;
//   4. every proposal edge finds its bracketing knots by binary search (the reference's masked max / min over ALL
//      knots selects exactly the last knot <= x and the first knot > x of a sorted sequence), quadratic interpolation,
//      diff, hinge^2 / (w_p + 1e-5);
//   5. the only tensor with a gradient is the proposal cdf (:213: wp = diff(prop_cdfs)), so the backward is local to
//      the ray: d cdf_j = g_{j-1} - g_j with g = d term / d wp.  Written in the same launch.
// Per-ray partial losses are reduced by a second, deterministic single-workgroup kernel (fixed summation order).
#include "common.h"

namespace emer {

constexpr int kPLRays = 4;  // waves (rays) per workgroup

__device__ __forceinline__ int upper_bound_lds(const float *a, int n, float v) {  // number of a[i] <= v (a sorted)
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int lower_bound_lds(const float *a, int n, float v) {  // number of a[i] < v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// mode 0: anti-aliased interlevel loss with pulse half-width `pulse`; mode 1: _pdf_loss (eps 1e-7)
__global__ __launch_bounds__(256) void prop_loss_kernel(const float *__restrict__ s_fin, const float *__restrict__ trans, int32_t n,
                                                        const float *__restrict__ s_prop, const float *__restrict__ c_prop, int32_t m,
                                                        float pulse, int mode, int64_t R, float scale,
                                                        float *__restrict__ loss_rays, float *__restrict__ d_cprop) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kPLRays + wave;
    const bool live = r < R;
    const int ne = n + 1, me = m + 1, K = 2 * ne;
    // wave-private arrays
    float *A = smem + (size_t)wave * (3 * ne + 3 * K + 4 * me);  // s - r   (mode 1: s)
    float *B = A + ne;                                           // s + r   (mode 1: cdf of the final samples)
    float *Y1 = B + ne;                                          // slope events
    float *XR = Y1 + ne;                                         // merged knots
    float *W = XR + K;                                           // slope sums, then blurred pdf at the knots
    float *CD = W + K;                                           // blurred cdf at the knots
    float *Q = CD + K;                                           // proposal edges
    float *PC = Q + me;                                          // proposal cdf
    float *CI = PC + me;                                         // blurred cdf interpolated at the proposal edges
    float *G = CI + me;                                          // d term / d wp per interval (mode 1: d loss / d cdf_key)

    if (live) {
        for (int j = lane; j < ne; j += kWave) {
            const float x = s_fin[r * (int64_t)ne + j];
            const float c = j < n ? 1.0f - trans[r * (int64_t)n + j] : 1.0f;
            if (mode == 0) { A[j] = x - pulse; B[j] = x + pulse; XR[j] = x; W[j] = c; }  // (XR / W borrowed as scratch for x / cdf)
            else { A[j] = x; B[j] = c; }
        }
        for (int j = lane; j < me; j += kWave) { Q[j] = s_prop[r * (int64_t)me + j]; PC[j] = c_prop[r * (int64_t)me + j]; }
    }
    __syncthreads();
    float part = 0.0f;
    if (mode == 0) {
        // ---- 1. normalised weights and their jumps at the edges: y1_j = (w_j - w_{j-1}) / (2 r), w_{-1} = w_n = 0
        if (live) {
            for (int j = lane; j < ne; j += kWave) {
                const float wr = j < n ? (W[j + 1] - W[j]) / (XR[j + 1] - XR[j]) : 0.0f;
                const float wl = j > 0 ? (W[j] - W[j - 1]) / (XR[j] - XR[j - 1]) : 0.0f;
                Y1[j] = (wr - wl) / (2.0f * pulse);
            }
        }
        __syncthreads();
        // ---- 2. merge the two sorted halves (A before B on ties); W receives the slope event of each knot
        if (live) {
            for (int j = lane; j < ne; j += kWave) {
                const float a = A[j], b = B[j], y = Y1[j];
                const int pa = j + lower_bound_lds(B, ne, a);
                const int pb = j + upper_bound_lds(A, ne, b);
                XR[pa] = a; W[pa] = y;
                XR[pb] = b; W[pb] = -y;
            }
        }
        __syncthreads();
        // ---- 3. scans over the K - 1 segments
        if (live) {
            float carry = 0.0f;  // slope = cumsum(events)
            for (int base = 0; base < K - 1; base += kWave) {
                const int t = base + lane;
                const bool ok = t < K - 1;
                const float incl = wave_inclusive_sum(ok ? W[t] : 0.0f, lane) + carry;
                if (ok) CD[t] = incl;  // (CD as scratch: slope of segment t)
                carry = __shfl(incl, kWave - 1, kWave);
            }
        }
        __syncthreads();
        if (live) {
            float carry = 0.0f;  // blurred pdf at knot t + 1 = max(cumsum(dx * slope), 0); 0 at knot 0
            for (int base = 0; base < K - 1; base += kWave) {
                const int t = base + lane;
                const bool ok = t < K - 1;
                const float v = ok ? (XR[t + 1] - XR[t]) * CD[t] : 0.0f;
                const float incl = wave_inclusive_sum(v, lane) + carry;
                if (ok) W[t + 1] = fmaxf(incl, 0.0f);
                carry = __shfl(incl, kWave - 1, kWave);
            }
            if (lane == 0) W[0] = 0.0f;
        }
        __syncthreads();
        if (live) {
            float carry = 0.0f;  // blurred cdf at knot t + 1 = cumsum of trapezoid areas; 0 at knot 0
            for (int base = 0; base < K - 1; base += kWave) {
                const int t = base + lane;
                const bool ok = t < K - 1;
                const float v = ok ? 0.5f * (W[t + 1] + W[t]) * (XR[t + 1] - XR[t]) : 0.0f;
                const float incl = wave_inclusive_sum(v, lane) + carry;
                if (ok) CD[t + 1] = incl;  // (the slopes CD held were consumed by the previous scan, behind a barrier)
                carry = __shfl(incl, kWave - 1, kWave);
            }
            if (lane == 0) CD[0] = 0.0f;
        }
        __syncthreads();
        // ---- 4. quadratic interpolation of the blurred cdf at the proposal edges
        if (live) {
            for (int j = lane; j < me; j += kWave) {
                const float q = Q[j];
                const int k = upper_bound_lds(XR, K, q);
                const int i0 = k - 1 < 0 ? 0 : k - 1, i1 = k > K - 1 ? K - 1 : k;
                const float x0 = XR[i0], x1 = XR[i1], p0 = W[i0], p1 = W[i1];
                const float num = q - x0, den = x1 - x0;
                float off;  // clip(nan_to_num(num / den, 0), 0, 1): 0/0 -> 0, +x/0 -> 1, -x/0 -> 0
                if (den == 0.0f) off = num > 0.0f ? 1.0f : 0.0f;
                else { off = num / den; off = off < 0.0f ? 0.0f : (off > 1.0f ? 1.0f : off); }
                CI[j] = CD[i0] + num * (p0 + p1 * off + p0 * (1.0f - off)) / 2.0f;
            }
        }
        __syncthreads();
        // ---- 5. hinge loss on the interval weights and its gradient w.r.t. the proposal cdf
        if (live) {
            for (int j = lane; j < m; j += kWave) {
                const float ws = CI[j + 1] - CI[j], wp = PC[j + 1] - PC[j];
                const float d = fmaxf(ws - wp, 0.0f), den = wp + 1e-5f;
                part += d * d / den;
                G[j] = -2.0f * d / den - d * d / (den * den);
            }
        }
    } else {
        // ---- _pdf_loss: query = final samples (A = edges, B = cdf), key = proposal (Q = edges, PC = cdf)
        //   ids_right = searchsorted(key, query, right) , ids_left = ids_right - 1, both clamped to [0, m];
        //   w = diff(cdf_query); w_outer = cdf_key[ids_right[1:]] - cdf_key[ids_left[:-1]]; clip(w - w_outer, 0)^2 / (w + eps)
        // the gradient goes to cdf_key: -g at ids_left[j], +g ... accumulated per ray in LDS (G as accumulator).
        if (live) for (int j = lane; j < me; j += kWave) G[j] = 0.0f;
        __syncthreads();
        if (live) {
            for (int j = lane; j < n; j += kWave) {
                int il = upper_bound_lds(Q, me, A[j]) - 1, ir = upper_bound_lds(Q, me, A[j + 1]);
                il = il < 0 ? 0 : (il > m ? m : il);
                ir = ir > m ? m : ir;
                const float w = B[j + 1] - B[j], wo = PC[ir] - PC[il];
                const float d = fmaxf(w - wo, 0.0f), den = w + 1e-7f;
                part += d * d / den;
                const float g = -2.0f * d / den;  // d term / d w_outer
                if (g != 0.0f) { atomicAdd(G + ir, g); atomicAdd(G + il, -g); }
            }
        }
    }
    __syncthreads();
    if (!live) return;
    part = wave_sum(part);
    if (loss_rays && lane == 0) loss_rays[r] = part * scale;
    if (d_cprop) {
        if (mode == 0) {
            for (int j = lane; j < me; j += kWave) {
                const float gm = j > 0 ? G[j - 1] : 0.0f, gj = j < m ? G[j] : 0.0f;
                d_cprop[r * (int64_t)me + j] = (gm - gj) * scale;
            }
        } else {
            for (int j = lane; j < me; j += kWave) d_cprop[r * (int64_t)me + j] = G[j] * scale;
        }
    }
}

// out[0] = (accumulate ? out[0] : 0) + sum(in[0..n)) -- one workgroup, fixed order, double accumulation
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float *__restrict__ in, int64_t n, int accumulate, float *__restrict__ out) {
    __shared__ double part[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += (double)in[i];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += part[i];
        out[0] = (float)(t + (accumulate ? (double)out[0] : 0.0));
    }
}

// y = x * s[0] * host_scale (s may be null)
__global__ __launch_bounds__(256) void scale_kernel(const float *__restrict__ x, const float *__restrict__ s, float host_scale,
                                                    float *__restrict__ y, int64_t n) {
    const float f = (s ? s[0] : 1.0f) * host_scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * f;
}

}  // namespace emer

using namespace emer;

extern "C" int emer_prop_loss(const float *s_final, const float *trans, int32_t n_final, const float *s_prop, const float *cdf_prop,
                              int32_t n_prop, float pulse_width, int anti_aliased, int64_t n_rays, float scale, float *loss_rays,
                              float *loss_out, int accumulate, float *d_cdf_prop, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && n_final >= 1 && n_prop >= 1 && n_final <= 2048 && n_prop <= 512,
                 "prop_loss: bad sizes (n_final <= 2048, n_prop <= 512)");
    EMER_REQUIRE(!anti_aliased || pulse_width > 0.0f, "prop_loss: the anti-aliased loss needs a positive pulse width");
    EMER_REQUIRE(!loss_out || loss_rays, "prop_loss: loss_out needs the per-ray buffer loss_rays");
    if (n_rays == 0) {
        if (loss_out && !accumulate) return hipMemsetAsync(loss_out, 0, sizeof(float), as_stream(stream)) == hipSuccess ? EMER_OK : EMER_E_LAUNCH;
        return EMER_OK;
    }
    EMER_REQUIRE(s_final && trans && s_prop && cdf_prop, "prop_loss: null pointer");
    const int ne = n_final + 1, me = n_prop + 1, K = 2 * ne;
    const size_t lds = (size_t)kPLRays * (3 * ne + 3 * K + 4 * me) * sizeof(float);
    EMER_REQUIRE(lds <= 160 * 1024, "prop_loss: %zu B of LDS needed", lds);
    if (int rc = reserve_lds(reinterpret_cast<const void *>(prop_loss_kernel), lds, "prop_loss")) return rc;
    hipLaunchKernelGGL(prop_loss_kernel, dim3((uint32_t)ceil_div(n_rays, kPLRays)), dim3(256), lds, as_stream(stream), s_final, trans,
                       n_final, s_prop, cdf_prop, n_prop, pulse_width, anti_aliased ? 0 : 1, n_rays, scale, loss_rays, d_cdf_prop);
    if (int rc = check_launch("prop_loss")) return rc;
    if (loss_out) {
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, as_stream(stream), loss_rays, n_rays, accumulate, loss_out);
        return check_launch("prop_loss(reduce)");
    }
    return EMER_OK;
}

extern "C" int emer_reduce_sum(const float *x, int64_t n, int accumulate, float *out, void *stream) {
    EMER_REQUIRE(n >= 0 && out && (x || n == 0), "reduce_sum: bad arguments");
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, n, accumulate, out);
    return check_launch("reduce_sum");
}

extern "C" int emer_scale(const float *x, const float *dev_scalar, float host_scale, float *y, int64_t n, void *stream) {
    EMER_REQUIRE(n >= 0, "scale: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x && y, "scale: null pointer");
    int64_t blocks = ceil_div(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_kernel, dim3((uint32_t)blocks), dim3(256), 0, as_stream(stream), x, dev_scalar, host_scale, y, n);
    return check_launch("scale");
}
