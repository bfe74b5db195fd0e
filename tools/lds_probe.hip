// LDS accumulate-rate probe for MI355X: which LDS primitive can absorb a scatter-add fastest?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: ds_add_f32 random; 1: ds_add_f32 conflict-free (lane-linear, rotating); 2: ds_add_u32 random; 3: non-atomic RMW random;
// 4: ds_add_f32 random with only 16 of 64 lanes active; 5: ds_add_f64 random; 6: ds_add_rtn_f32 random; 7: ds_add_f32, all lanes same address
// 8: two ds_add_f32 to adjacent floats (F=2 pattern)
template <int MODE>
__global__ __launch_bounds__(1024) void lds_kernel(float *out, uint32_t entries, uint32_t iters) {
    extern __shared__ float lds[];
    for (uint32_t i = threadIdx.x; i < entries; i += 1024) lds[i] = 0.f;
    __syncthreads();
    const uint32_t tid = blockIdx.x * 1024 + threadIdx.x;
    uint32_t idx[8];
    float acc = 0.f;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // precompute 8 addresses so the loop body is LDS-bound, not VALU-bound
            uint32_t h = hash32(tid * 9781u + (it * 8 + k) * 6271u + 12345u);
            if (MODE == 1) idx[k] = ((threadIdx.x & 63) + 64 * ((h >> 8) % (entries / 64))) % entries;
            else if (MODE == 7) idx[k] = (it * 8 + k) % entries;
            else if (MODE == 5 || MODE == 8) idx[k] = (h % (entries / 2)) * 2;
            else idx[k] = h % entries;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0 || MODE == 1 || MODE == 7) atomicAdd(&lds[idx[k]], 1.0f);
            if (MODE == 2) atomicAdd(reinterpret_cast<uint32_t *>(lds) + idx[k], 1u);
            if (MODE == 3) lds[idx[k]] += 1.0f;
            if (MODE == 4) { if ((threadIdx.x & 3) == 0) atomicAdd(&lds[idx[k]], 1.0f); }
            if (MODE == 5) atomicAdd(reinterpret_cast<double *>(&lds[idx[k]]), 1.0);
            if (MODE == 6) acc += atomicAdd(&lds[idx[k]], 1.0f);
            if (MODE == 8) { atomicAdd(&lds[idx[k]], 1.0f); atomicAdd(&lds[idx[k] + 1], 2.0f); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[1] + acc;
}
template <typename F> float time_us(F &&fn, int iters = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); fn(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f); }
    std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main() {
    float *out; CK(hipMalloc(&out, 4096 * 4));
    const uint32_t blocks = 256, iters = 64;  // 256 blocks (1/CU) x 1024 thr x 64 x 8 = 134M lane-ops
    const double ops = (double)blocks * 1024 * iters * 8;
    const char *names[] = {"add_f32_random", "add_f32_conflict_free", "add_u32_random", "rmw_nonatomic", "add_f32_16lanes", "add_f64_random", "add_rtn_f32", "add_f32_same_addr", "add_f32x2_pair"};
    for (uint32_t entries : {4096u, 32768u}) {
        printf("entries=%u\n", entries);
#define RUN(M) { CK(hipFuncSetAttribute((const void*)lds_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000)); float us = time_us([&] { hipLaunchKernelGGL(lds_kernel<M>, dim3(blocks), dim3(1024), entries * 4, 0, out, entries, iters); }); double f = (M == 4 ? 0.25 : 1.0); printf("  %-24s %8.0f us  %7.1f G lane-ops/s  (%.2f lane-ops/clk/CU @2.1GHz)\n", names[M], us, ops * f / us / 1e3, ops * f / us / 1e3 / 256 / 2.1); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    }
    return 0;
}
