// Standalone probe (no torch): how fast can MI355X scatter-add?  Guides kernel design for
// hashgrid_bwd_params.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: random f32 atomic; 1: random pk f16 atomic; 2: lane-coherent f32 atomic (adjacent lanes adjacent addresses)
// 3: random non-atomic RMW (racy, for raw memory-path speed); 4: random f32 atomic with return
// 5: random f32 x2 (two consecutive floats, like F=2); 6: f64 atomic (two floats packed? no: plain double add)
template <int MODE>
__global__ __launch_bounds__(256) void scatter_kernel(float *table, uint32_t mask, uint32_t per_thread, float *sink) {
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (uint32_t i = 0; i < per_thread; ++i) {
        uint32_t h = hash32(tid * 9781u + i * 6271u + 12345u);
        uint32_t idx;
        if (MODE == 2) idx = ((hash32((tid >> 6) * 31u + i) & mask) & ~63u) + (tid & 63);
        else idx = h & mask;
        const float v = 1.0f + (h >> 28);
        if (MODE == 0 || MODE == 2) __hip_atomic_fetch_add(table + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) unsafeAtomicAdd(reinterpret_cast<__half2 *>(table) + idx, __floats2half2_rn(v, v));
        if (MODE == 3) table[idx] += v;
        if (MODE == 4) acc += __hip_atomic_fetch_add(table + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 5) { idx &= ~1u; __hip_atomic_fetch_add(table + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(table + idx + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (MODE == 6) { idx &= ~1u; __hip_atomic_fetch_add(reinterpret_cast<double *>(table + idx), (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (MODE == 7) __hip_atomic_fetch_add(table + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (MODE == 4 && acc == -1.f) sink[0] = acc;
}

// LDS-privatised scatter: each block owns a private LDS table of `lds_entries` floats.
__global__ __launch_bounds__(256) void lds_scatter_kernel(float *table, uint32_t lds_entries, uint32_t per_thread) {
    extern __shared__ float lds[];
    for (uint32_t i = threadIdx.x; i < lds_entries; i += 256) lds[i] = 0.f;
    __syncthreads();
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t i = 0; i < per_thread; ++i) {
        uint32_t h = hash32(tid * 9781u + i * 6271u + 12345u);
        atomicAdd(&lds[h % lds_entries], 1.0f);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < lds_entries; i += 256) __hip_atomic_fetch_add(table + i, lds[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename F>
float time_us(F &&fn, int iters = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const size_t max_entries = 1u << 26;
    float *table, *sink; CK(hipMalloc(&table, max_entries * 4)); CK(hipMalloc(&sink, 4)); CK(hipMemset(table, 0, max_entries * 4));
    const uint32_t blocks = 8192, per_thread = 8;  // 16.8M adds per launch
    const double n_adds = (double)blocks * 256 * per_thread;
    const char *names[] = {"f32_random", "pkf16_random", "f32_lane_coherent", "rmw_nonatomic", "f32_return", "f32x2_pair", "f64_random", "f32_wgscope"};
    for (int lg = 12; lg <= 24; lg += 2) {
        const uint32_t mask = (1u << lg) - 1;
        printf("table=2^%d floats (%.1f KiB):", lg, (1u << lg) * 4 / 1024.0);
#define RUN(M) { float us = time_us([&] { hipLaunchKernelGGL(scatter_kernel<M>, dim3(blocks), dim3(256), 0, 0, table, mask, per_thread, sink); }); printf("  %s %.0fus (%.1f G/s)", names[M], us, n_adds / us / 1e3); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
        printf("\n");
    }
    for (uint32_t e : {4096u, 8192u, 16384u, 32768u}) {
        float us = time_us([&] { hipLaunchKernelGGL(lds_scatter_kernel, dim3(2048), dim3(256), e * 4, 0, table, e, 32); });
        printf("lds_private entries=%u: %.0fus (%.1f G adds/s)\n", e, us, 2048.0 * 256 * 32 / us / 1e3);
    }
    return 0;
}
