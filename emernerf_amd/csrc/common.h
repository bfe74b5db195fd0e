// Shared helpers for the gfx950 kernels of libemernerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/emernerf_hip.h"

namespace emer {

constexpr int kWave = 64;  // CDNA wavefront width (hard-coded on purpose: gfx950 only)

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Every launch goes through this so a failed launch surfaces as EMER_E_LAUNCH, not silence.
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return EMER_E_LAUNCH;
    }
    return EMER_OK;
}

#define EMER_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            ::emer::set_error(__VA_ARGS__); \
            return EMER_E_INVALID;         \
        }                                  \
    } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Raise a kernel's dynamic-LDS limit above 48 KiB.  Cached per (kernel, device): the attribute call is made once, not on
// every launch -- it is not a stream operation, so it must not run while the launch stream is being captured into a
// hipGraph (core.hip).
int reserve_lds(const void *kernel, size_t bytes, const char *what);

// csrc/mlp.hip: dw (+=) and dbias (+=) from per-workgroup partials ([n][k] | [n] at the start of each `stride`-float partial)
int launch_dw_reduce(const float *partials, int32_t n_blocks, int64_t stride, int32_t n, int32_t k, float *dw, int64_t ld_dw,
                     float *dbias, hipStream_t st);

// ... with column blocks of the [n][k] partial scattered to dw[row * ld_dw + dst[s] + (column - col[s])]; no bias part
int launch_dw_reduce_cols(const float *partials, int32_t n_blocks, int64_t stride, int32_t n, int32_t k, float *dw, int64_t ld_dw,
                          int32_t n_segs, const int32_t *col, const int32_t *width, const int32_t *dst, hipStream_t st);

// ... and several gradients of one partial buffer in ONE launch [r4]
#define EMER_DW_MAX_JOBS 3
struct DwReduceJob {
    int64_t off;                 // float offset of this gradient inside a partial
    int32_t n, k;                // dW [n][k], followed by dbias [n] when db != nullptr
    float *dw; int64_t ld_dw; float *db;
    int32_t n_segs;              // 0: plain; else column blocks col[s] .. col[s] + width[s] - 1 land at dw[row * ld_dw + dst[s] ..]
    int32_t col[4], width[4], dst[4];
};
int launch_dw_reduce_multi(const float *partials, int32_t n_blocks, int64_t stride, int n_jobs, const DwReduceJob *jobs, hipStream_t st);

// Measurement hook (emer_profile_next): a pair of caller-owned HIP events that the NEXT instrumented launch of this thread
// records immediately before / after its kernel (hipExtLaunchKernelGGL), so bench.py times the kernel itself and not
// the host's enqueue latency around it.  One-shot; both null when not armed.
struct ProfileEvents { hipEvent_t start, stop; };
ProfileEvents take_profile_events();
// plain launch unless a measurement armed the events (the extended launch is kept out of hipGraph captures)
#define EMER_LAUNCH_PROFILED(ev, kern, grid, block, lds, st, ...)                                                  \
    do {                                                                                                           \
        if ((ev).start) hipExtLaunchKernelGGL(kern, grid, block, lds, st, (ev).start, (ev).stop, 0, __VA_ARGS__);   \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                          \
    } while (0)

// ---- wave-level scans on __shfl (ds_bpermute: the LDS crossbar, no LDS memory).  The grid kernels use DPP-based scans
// instead (csrc/hashgrid.hip); these serve the per-ray kernels, where a scan is a small part of the work. ----------------
__device__ __forceinline__ float wave_inclusive_sum(float v, int lane) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        float o = __shfl_up(v, off, kWave);
        if (lane >= off) v += o;
    }
    return v;
}
// inclusive suffix sum: lane i receives sum_{j >= i} v_j
__device__ __forceinline__ float wave_inclusive_suffix_sum(float v, int lane) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        float o = __shfl_down(v, off, kWave);
        if (lane + off < kWave) v += o;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

}  // namespace emer
