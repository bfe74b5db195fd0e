// Scene contraction of a sample point (nerf_utils.py:13-28, radiance_field.py:278-300) and the ray sample point itself
// (render_utils.py:318,341), shared by the translation units that compute sample positions (elementwise.hip: emer_contract_*,
// emer_ray_points, emer_flow_warp_*; sampler.hip: emer_importance_sample_points).  Sample positions decide which grid cell a sample falls
// in, so every includer disables FMA contraction (`#pragma clang fp contract(off)` at file scope) and these functions follow the
// reference's torch expression order exactly.
#pragma once
#include "common.h"

namespace emer {

struct Aabb { float lo[3], hi[3]; };

__device__ __forceinline__ Aabb load_aabb(const float *__restrict__ aabb) {
    Aabb a;
#pragma unroll
    for (int d = 0; d < 3; ++d) { a.lo[d] = aabb[d]; a.hi[d] = aabb[3 + d]; }
    return a;
}

// returns inside flag; v = contracted coords (already zeroed when outside)
__device__ __forceinline__ bool contract_point(const Aabb &bb, bool unbounded, const float (&p)[3], float (&v)[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) v[d] = (p[d] - bb.lo[d]) / (bb.hi[d] - bb.lo[d]);
    if (unbounded) {
        float mag = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) { v[d] = v[d] * 2.0f - 1.0f; mag = fmaxf(mag, fabsf(v[d])); }
        if (!(mag < 1.0f)) {
            const float s = 2.0f - 1.0f / mag;
#pragma unroll
            for (int d = 0; d < 3; ++d) v[d] = s * (v[d] / mag);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] = v[d] / 4.0f + 0.5f;
    }
    bool inside = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) inside = inside && (v[d] > 0.0f) && (v[d] < 1.0f);
    if (!inside) {
#pragma unroll
        for (int d = 0; d < 3; ++d) v[d] = v[d] * 0.0f;
    }
    return inside;
}

// p = o + d * (t0 + t1) / 2, evaluated as ((d * (t0 + t1)) / 2) + o (render_utils.py:341)
__device__ __forceinline__ void ray_point(const float *__restrict__ o, const float *__restrict__ d, float tsum, float (&p)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = o[k] + d[k] * tsum / 2.0f;
}

}  // namespace emer
