// Proposal-network importance sampler for gfx950.
//
// Replaces nerfacc.pdf.importance_sampling (batched mode) and _transform_stot as called from
// third_party/nerfacc_prop_net.py:153-157,172-173.  Frozen spec: SURVEY.md Appendix A.2; CPU
// restatement: oracle/emer_oracle.c (orc_importance_sample / orc_stot).
//
// BIT-EXACTNESS: sample offsets must match the oracle bit for bit, so this translation unit
// forbids FMA contraction (the oracle is built with -ffp-contract=off) and uses only IEEE
// + - * / on fp32 (hipcc's default fp32 division is correctly rounded).
//
// Mapping: one 64-lane wavefront owns one ray.  The ray's m (<= 4096) CDF edges are staged once
// into a wave-private LDS slice with coalesced loads; each lane then inverts the CDF for outputs
// k = lane, lane+64, ... with a branch-free binary search over LDS.  No cross-lane traffic, no
// atomics; output stores are coalesced (consecutive lanes -> consecutive k).
#include "common.h"

#pragma clang fp contract(off)

#include "contract.h"

namespace emer {

__host__ __device__ __forceinline__ float stot_fwd_map(int type, float t) {
    if (type == EMER_STOT_UNIFORM_LINDISP) return t < 200.0f ? t / 400.0f : 1.0f - 1.0f / (2.0f * t / 200.0f);
    if (type == EMER_STOT_LINDISP) return 1.0f / t;
    if (type == EMER_STOT_SQRT) return sqrtf(t);
    if (type == EMER_STOT_LOG) return logf(t);
    if (type == EMER_STOT_UNIFORM_LINDISP_0) return t < 1.0f ? t / 2.0f : 1.0f - 1.0f / (2.0f * t);
    return t;
}
__device__ __forceinline__ float stot_inv_map(int type, float s) {
    // torch evaluates `200 / (2 - 2*x)` (nerfacc_prop_net.py:308) as (2 - 2*x).reciprocal() * 200
    if (type == EMER_STOT_UNIFORM_LINDISP) return s < 0.5f ? s * 400.0f : (1.0f / (2.0f - 2.0f * s)) * 200.0f;
    if (type == EMER_STOT_LINDISP) return 1.0f / s;
    if (type == EMER_STOT_SQRT) return s * s;
    if (type == EMER_STOT_LOG) return expf(s);  // (libm-dependent last bit: compared by tolerance, not bit-exact)
    if (type == EMER_STOT_UNIFORM_LINDISP_0) return s < 0.5f ? 2.0f * s : 1.0f / (2.0f - 2.0f * s);
    return s;
}
__device__ __forceinline__ float stot_apply(int type, float s, float s_min, float s_max) {
    return stot_inv_map(type, s * s_max + (1.0f - s) * s_min);
}

constexpr int kRaysPerBlock = 4;  // 4 waves per workgroup

__global__ __launch_bounds__(256) void importance_sample_kernel(const float *__restrict__ vals, const float *__restrict__ cdfs,
                                                                int64_t R, int32_t m, int32_t n,
                                                                const float *__restrict__ jitter, float *__restrict__ s_out,
                                                                float *__restrict__ t_out, float *__restrict__ t_ends, float s_min, float s_max, int type) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlock + wave;
    float *c = smem + (size_t)wave * 2 * m, *v = c + m;
    if (r < R) {
        for (int i = lane; i < m; i += kWave) { c[i] = cdfs[r * m + i]; v[i] = vals[r * m + i]; }
    }
    __syncthreads();
    if (r >= R) return;
    const float c0 = c[0], cl = c[m - 1];
    const float step = (cl - c0) / (float)(n + 1);
    const float beta = jitter ? jitter[r] : 0.5f;
    for (int k = lane; k <= n; k += kWave) {
        const float u = c0 + ((float)k + beta) * step;
        int lo = 0, hi = m;  // first j with c[j] > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (c[mid] <= u) lo = mid + 1; else hi = mid;
        }
        // nerfacc pdf.cu clamps the two bracketing edges separately (p0 = clamp(p - 1), p1 = clamp(p) over [0, m - 1]): an
        // u at or beyond the last CDF value brackets (m - 1, m - 1) and returns v[m - 1] through the d < 1e-10 branch
        const int p0 = lo > 0 ? lo - 1 : 0, p1 = lo < m ? lo : m - 1;
        const float cp = c[p0], cq = c[p1], vp = v[p0], vq = v[p1];
        const float d = cq - cp;
        float s;
        if (d < 1e-10f) s = (vp + vq) * 0.5f;
        else s = (u - cp) * ((vq - vp) / d) + vp;
        const int64_t o = r * (int64_t)(n + 1) + k;
        s_out[o] = s;
        if (t_out) {
            const float t = stot_apply(type, s, s_min, s_max);
            if (t_ends) {  // interval form: t_out = starts [R,n], t_ends = ends [R,n] (edge k starts interval k and ends k-1)
                if (k < n) t_out[r * (int64_t)n + k] = t;
                if (k > 0) t_ends[r * (int64_t)n + k - 1] = t;
            } else {
                t_out[o] = t;
            }
        }
    }
}

// [r5] The same sampler followed, in the launch, by the sample POINTS of the new intervals: p = o + d (t0 + t1) / 2 and its scene
// contraction (render_utils.py:316-318,341), i.e. what emer_ray_points computes from this kernel's output one launch later.  The edges'
// t values stay in a wave-private LDS row, lane k takes interval k.  Same device functions, same expression order, contraction off in
// both translation units: the outputs are bitwise those of emer_importance_sample + emer_ray_points.
__global__ __launch_bounds__(256) void importance_sample_points_kernel(const float *__restrict__ vals, const float *__restrict__ cdfs, int64_t R,
                                                                       int32_t m, int32_t n, const float *__restrict__ jitter,
                                                                       float *__restrict__ s_out, float *__restrict__ t_out,
                                                                       float *__restrict__ t_ends, float s_min, float s_max, int type,
                                                                       const float *__restrict__ origins, const float *__restrict__ dirs,
                                                                       const float *__restrict__ aabb, int unbounded,
                                                                       float *__restrict__ normed, float *__restrict__ positions) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kRaysPerBlock + wave;
    float *c = smem + (size_t)wave * (2 * m + n + 1), *v = c + m, *te = v + m;
    if (r < R) {
        for (int i = lane; i < m; i += kWave) { c[i] = cdfs[r * m + i]; v[i] = vals[r * m + i]; }
    }
    __syncthreads();
    if (r >= R) return;
    const float c0 = c[0], cl = c[m - 1];
    const float step = (cl - c0) / (float)(n + 1);
    const float beta = jitter ? jitter[r] : 0.5f;
    for (int k = lane; k <= n; k += kWave) {
        const float u = c0 + ((float)k + beta) * step;
        int lo = 0, hi = m;  // first j with c[j] > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (c[mid] <= u) lo = mid + 1; else hi = mid;
        }
        // nerfacc pdf.cu clamps the two bracketing edges separately (p0 = clamp(p - 1), p1 = clamp(p) over [0, m - 1]): an
        // u at or beyond the last CDF value brackets (m - 1, m - 1) and returns v[m - 1] through the d < 1e-10 branch
        const int p0 = lo > 0 ? lo - 1 : 0, p1 = lo < m ? lo : m - 1;
        const float cp = c[p0], cq = c[p1], vp = v[p0], vq = v[p1];
        const float d = cq - cp;
        float s;
        if (d < 1e-10f) s = (vp + vq) * 0.5f;
        else s = (u - cp) * ((vq - vp) / d) + vp;
        s_out[r * (int64_t)(n + 1) + k] = s;
        const float t = stot_apply(type, s, s_min, s_max);
        te[k] = t;
        if (k < n) t_out[r * (int64_t)n + k] = t;
        if (k > 0) t_ends[r * (int64_t)n + k - 1] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the row of edge times is complete (one wave per ray)
    const Aabb bb = load_aabb(aabb);
    for (int k = lane; k < n; k += kWave) {
        const float tsum = te[k] + te[k + 1];
        float p[3], q[3];
        ray_point(origins + r * 3, dirs + r * 3, tsum, p);
        contract_point(bb, unbounded != 0, p, q);
        const int64_t i = r * (int64_t)n + k;
        normed[i * 3] = q[0]; normed[i * 3 + 1] = q[1]; normed[i * 3 + 2] = q[2];
        if (positions) { positions[i * 3] = p[0]; positions[i * 3 + 1] = p[1]; positions[i * 3 + 2] = p[2]; }
    }
}

__global__ __launch_bounds__(256) void stot_kernel(const float *__restrict__ s, int64_t n, float s_min, float s_max, int type,
                                                   float *__restrict__ t) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        t[i] = stot_apply(type, s[i], s_min, s_max);
}

}  // namespace emer

using namespace emer;

extern "C" int emer_importance_sample(const float *vals, const float *cdfs, int64_t R, int32_t m, int32_t n,
                                      const float *jitter, float *s_out, float *t_out, float *t_ends, float t_min,
                                      float t_max, int stot_type, void *stream) {
    EMER_REQUIRE(R >= 0 && n >= 1, "importance_sample: bad sizes R=%lld n=%d", (long long)R, n);
    EMER_REQUIRE(m >= 2 && m <= 4096, "importance_sample: m=%d edges per ray not in 2..4096", m);
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(vals && cdfs && s_out, "importance_sample: null pointer");
    EMER_REQUIRE(!t_ends || t_out, "importance_sample: t_ends needs t_out (the interval starts)");
    EMER_REQUIRE(stot_type >= 0 && stot_type <= 5, "importance_sample: unknown stot_type %d", stot_type);
    const float s_min = stot_fwd_map(stot_type, t_min), s_max = stot_fwd_map(stot_type, t_max);
    const size_t lds = (size_t)kRaysPerBlock * 2 * m * sizeof(float);
    hipLaunchKernelGGL(importance_sample_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlock)), dim3(256), lds, as_stream(stream),
                       vals, cdfs, R, m, n, jitter, s_out, t_out, t_ends, s_min, s_max, stot_type);
    return check_launch("importance_sample");
}

// Floats of LDS a ray may use in emer_importance_sample_points (2 m + n + 1 must not exceed it): the device's LDS per workgroup over
// the four rays of a workgroup -- 10240 on gfx950 (160 KiB), asked from the runtime rather than assumed.
extern "C" int64_t emer_importance_sample_points_capacity(void) {
    int dev = 0, bytes = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
    return (int64_t)bytes / (kRaysPerBlock * (int64_t)sizeof(float));
}

// emer_importance_sample (interval form) + emer_ray_points of its result in one launch.  normed [R][n][3], positions [R][n][3] or NULL.
// Needs 2 m + n + 1 <= emer_importance_sample_points_capacity() floats of LDS per ray: larger histograms take the two separate calls.
extern "C" int emer_importance_sample_points(const float *vals, const float *cdfs, int64_t R, int32_t m, int32_t n, const float *jitter,
                                             float *s_out, float *t_starts, float *t_ends, float t_min, float t_max, int stot_type,
                                             const float *origins, const float *dirs, const float *aabb, int unbounded, float *normed,
                                             float *positions, void *stream) {
    EMER_REQUIRE(R >= 0 && n >= 1, "importance_sample_points: bad sizes R=%lld n=%d", (long long)R, n);
    EMER_REQUIRE(m >= 2 && m <= 4096 && 2 * (int64_t)m + n + 1 <= emer_importance_sample_points_capacity(),
                 "importance_sample_points: m=%d edges, n=%d intervals per ray do not fit the LDS", m, n);
    if (R == 0) return EMER_OK;
    EMER_REQUIRE(vals && cdfs && s_out && t_starts && t_ends && origins && dirs && aabb && normed, "importance_sample_points: null pointer");
    EMER_REQUIRE(stot_type >= 0 && stot_type <= 5, "importance_sample_points: unknown stot_type %d", stot_type);
    const float s_min = stot_fwd_map(stot_type, t_min), s_max = stot_fwd_map(stot_type, t_max);
    const size_t lds = (size_t)kRaysPerBlock * (2 * (size_t)m + n + 1) * sizeof(float);
    if (int rc = reserve_lds(reinterpret_cast<const void *>(importance_sample_points_kernel), lds, "importance_sample_points")) return rc;
    hipLaunchKernelGGL(importance_sample_points_kernel, dim3((uint32_t)ceil_div(R, kRaysPerBlock)), dim3(256), lds, as_stream(stream), vals, cdfs,
                       R, m, n, jitter, s_out, t_starts, t_ends, s_min, s_max, stot_type, origins, dirs, aabb, unbounded, normed, positions);
    return check_launch("importance_sample_points");
}

extern "C" int emer_stot(const float *s, int64_t n, float t_min, float t_max, int stot_type, float *t, void *stream) {
    EMER_REQUIRE(n >= 0, "stot: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(s && t, "stot: null pointer");
    EMER_REQUIRE(stot_type >= 0 && stot_type <= 5, "stot: unknown stot_type %d", stot_type);
    const float s_min = stot_fwd_map(stot_type, t_min), s_max = stot_fwd_map(stot_type, t_max);
    const uint32_t blocks = (uint32_t)(ceil_div(n, 256) < 2048 ? ceil_div(n, 256) : 2048);
    hipLaunchKernelGGL(stot_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), s, n, s_min, s_max, stot_type, t);
    return check_launch("stot");
}
