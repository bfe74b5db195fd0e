// Issue-rate probe for the MFMA shapes the fused heads use (MI355X): cycles per instruction per SIMD with 4 independent accumulators,
// one wave per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x4 c[4]; f32x16 d[2];
    for (int i = 0; i < 4; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) d[i] = f32x16{0};
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
    bf16x8 a8 = __builtin_bit_cast(bf16x8, uint4{threadIdx.x, 1, 2, 3}), b8 = __builtin_bit_cast(bf16x8, uint4{3, 2, 1, threadIdx.x});
    float fa = threadIdx.x, fb = 1.0f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[i], 0, 0, 0);
                if (MODE == 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c[i], 0, 0, 0);
                if (MODE == 2 && i < 2) d[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, d[i], 0, 0, 0);
                if (MODE == 3) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c[i], 0, 0, 0);
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c[i][0];
    for (int i = 0; i < 2; ++i) s += d[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
int main() {
    float *out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 2000;
    const char *names[] = {"16x16x16_bf16 (legacy K=16)", "16x16x32_bf16", "32x32x16_bf16", "16x16x4_f32"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, iters);
            hipDeviceSynchronize();
        }
        float cyc; hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
        const double n = (double)iters * 8 * (mode == 2 ? 2 : 4);
        printf("%-30s %.1f clock64 ticks per instruction (one wave per SIMD)\n", names[mode], cyc / n);
    }
    return 0;
}
