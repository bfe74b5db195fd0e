// Register-resident fused heads for gfx950 (hidden width 64): neck (grid encoding -> 64 -> 64/128 [+ density]),
// proposal density MLP (grid -> 64 -> 1 -> trunc_exp) and the rgb head (3-layer skip MLP + sigmoid), forward and
// data-gradient chains.  Replaces the nn.Sequential / mlp.MLP stacks of radiance_field.py:74-198,808-840 and
// mlp.py:7-46 on the per-sample hot path.
//
// Transposed chaining.  Every layer is computed as Y^T = W X^T on v_mfma_f32_16x16x4_f32 with the WEIGHTS as the
// A operand (lane (n = lane & 15, g = lane >> 4) supplies W[16t'+n][k]) and the ACTIVATIONS as the B operand (lane
// (m = lane & 15, g) supplies X[m][k]).  The 16x16 result tile t' leaves lane (m, g) holding features
// 16t' + 4g + i (i = 0..3) of row m -- which is exactly a legal B operand of the next layer if its reduction
// index is enumerated as k = 16t + 4g + i (any bijection of k is a valid GEMM as long as A uses the same one).
// So a wave owns 16 rows END TO END in registers: no activation ever touches LDS, there is no barrier after the
// weights are staged, and occupancy is bounded by VGPRs (~100) instead of a 16 KB-per-wave LDS row buffer.
// LDS holds only the weights, row-major [n][kpad + 4]: one ds_read_b128 per (t', t) yields the A operands of four
// MFMA steps, and the +4 pitch spreads the 16 rows of a read over all 64 banks.
//
// Global traffic is the algorithmic minimum: each lane loads / stores 16 B pieces (row-major tensors: the four g
// lanes of a row cover 64 contiguous bytes; level-major grid encodings: 16 lanes cover 16 consecutive rows of one
// level), the next tile's input is prefetched into registers while the current tile is in the matrix pipe.
//
// rgb head.  The reference concatenates [dir-PE | appearance embedding | geo] per SAMPLE (radiance_field.py:629-658)
// although the first two are per-RAY constants.  Here the per-ray part enters as a per-ray pre-activation
// (rb = hray W_h^T + b, an 8192-row GEMM instead of a 1M-row one) and the per-sample GEMMs shrink from K = 113 / 177
// to K = 64 / 128.  In the backward the kernel reduces dPre0 / dPre1 over the samples of each ray, which is all the
// per-ray operands need (dhray, dW_h and the biases are tiny per-ray GEMMs on those sums).
#include "common.h"

namespace emer {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kFThreads = 384;  // rgb head: 6 waves; two workgroups per CU = 3 waves per SIMD = 170 VGPRs each
constexpr int kNThreads = 512;  // neck: 8 waves; two workgroups per CU = 4 waves per SIMD = 128 VGPRs each
constexpr int kNeckChunk = 8;  // consecutive 16-row tiles a wave processes per work item

struct WSrc {
    const float *w;
    int64_t sn, sk;  // element (n, k) at w[n * sn + k * sk]
    int32_t n, k;    // real extents (zero padded in LDS)
};

__device__ __forceinline__ void stage_w(float *dst, int pitch, int npad, int kpad, const WSrc s) {
    for (int idx = threadIdx.x; idx < npad * kpad; idx += (int)blockDim.x) {
        const int n = idx / kpad, k = idx - n * kpad;
        dst[n * pitch + k] = (n < s.n && k < s.k) ? s.w[n * s.sn + k * s.sk] : 0.0f;
    }
}
__device__ __forceinline__ void stage_b(float *dst, int npad, const float *b, int n) {
    for (int i = threadIdx.x; i < npad; i += (int)blockDim.x) dst[i] = (b && i < n) ? b[i] : 0.0f;
}

// acc[p] (output tile p) += sum over input tiles t, steps i of W[16p + n][16t + 4g + i] * in[t][i]
// wl already points at this lane's (n = lane & 15, 4 * g) corner of the LDS matrix.
template <int KT, int NT, bool PIPE = true>
__device__ __forceinline__ void tgemm(const float *wl, int pitch, const f32x4 (&in)[KT], f32x4 (&acc)[NT]) {
    // PIPE: software pipelined by hand -- the A fragments of input tile t + 1 are read while tile t is in the matrix
    // pipe (two fragment sets used alternately, no register rotation).  !PIPE: one fragment set (16 VGPRs less), for
    // kernels that hide the LDS latency with a fourth wave per SIMD instead.  Either way the scheduling barrier keeps
    // the compiler from hoisting ALL weight reads of the chain (it would otherwise trade ~100 VGPRs of fragments for
    // latency it does not need to hide -- several waves share the SIMD).
    f32x4 a[PIPE ? 2 : 1][NT];
    if (PIPE) {
#pragma unroll
        for (int p = 0; p < NT; ++p) a[0][p] = *reinterpret_cast<const f32x4 *>(wl + p * 16 * pitch);
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (PIPE) {
            if (t + 1 < KT) {
#pragma unroll
                for (int p = 0; p < NT; ++p) a[(t + 1) & 1][p] = *reinterpret_cast<const f32x4 *>(wl + p * 16 * pitch + (t + 1) * 16);
            }
        } else {
#pragma unroll
            for (int p = 0; p < NT; ++p) a[0][p] = *reinterpret_cast<const f32x4 *>(wl + p * 16 * pitch + t * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int p = 0; p < NT; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[PIPE ? (t & 1) : 0][p][i], in[t][i], acc[p], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT>
__device__ __forceinline__ void init_bias(const float *bl, int g, f32x4 (&acc)[NT]) {
#pragma unroll
    for (int p = 0; p < NT; ++p) acc[p] = *reinterpret_cast<const f32x4 *>(bl + p * 16 + 4 * g);
}
template <int NT>
__device__ __forceinline__ void zero(f32x4 (&acc)[NT]) {
#pragma unroll
    for (int p = 0; p < NT; ++p) acc[p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
template <int NT>
__device__ __forceinline__ void relu(f32x4 (&v)[NT]) {
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[p][i] = v[p][i] > 0.0f ? v[p][i] : 0.0f;
}
// v = mask > 0 ? v : 0   (relu' through the saved post-activation)
template <int NT>
__device__ __forceinline__ void relu_mask(f32x4 (&v)[NT], const f32x4 (&mk)[NT]) {
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[p][i] = mk[p][i] > 0.0f ? v[p][i] : 0.0f;
}

// row-major [rows][>= 16 * NT] tensor: lane (m, g) holds columns 16p + 4g .. + 3 of its row
template <int NT>
__device__ __forceinline__ void ld_rm(const float *rowp, bool ok, int g, f32x4 (&v)[NT]) {
#pragma unroll
    for (int p = 0; p < NT; ++p) v[p] = ok ? *reinterpret_cast<const f32x4 *>(rowp + p * 16 + 4 * g) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
template <int NT>
__device__ __forceinline__ void st_rm(float *rowp, bool ok, int g, const f32x4 (&v)[NT]) {
    if (!ok) return;
#pragma unroll
    for (int p = 0; p < NT; ++p) *reinterpret_cast<f32x4 *>(rowp + p * 16 + 4 * g) = v[p];
}

// level-major grid encoding [L][n_total][F]: feature k = level * F + f
template <int KT, int F>
__device__ __forceinline__ void ld_lm(const float *enc, int64_t n_total, int n_levels, int64_t row, bool ok, int g, f32x4 (&v)[KT]) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        v[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (F == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lv = 16 * t + 4 * g + i;
                if (ok && lv < n_levels) v[t][i] = enc[(int64_t)lv * n_total + row];
            }
        } else if (F == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lv = 8 * t + 2 * g + j;
                if (ok && lv < n_levels) {
                    const float2 x = *reinterpret_cast<const float2 *>(enc + ((int64_t)lv * n_total + row) * 2);
                    v[t][2 * j] = x.x; v[t][2 * j + 1] = x.y;
                }
            }
        } else if (F == 4) {
            const int lv = 4 * t + g;
            if (ok && lv < n_levels) v[t] = *reinterpret_cast<const f32x4 *>(enc + ((int64_t)lv * n_total + row) * 4);
        } else {  // F == 8
            const int lv = 2 * t + (g >> 1);
            if (ok && lv < n_levels) v[t] = *reinterpret_cast<const f32x4 *>(enc + ((int64_t)lv * n_total + row) * 8 + 4 * (g & 1));
        }
    }
}
template <int KT, int F>
__device__ __forceinline__ void st_lm(float *enc, int64_t n_total, int n_levels, int64_t row, bool ok, int g, const f32x4 (&v)[KT]) {
    if (!ok) return;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (F == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lv = 16 * t + 4 * g + i;
                if (lv < n_levels) enc[(int64_t)lv * n_total + row] = v[t][i];
            }
        } else if (F == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lv = 8 * t + 2 * g + j;
                if (lv < n_levels) *reinterpret_cast<float2 *>(enc + ((int64_t)lv * n_total + row) * 2) = make_float2(v[t][2 * j], v[t][2 * j + 1]);
            }
        } else if (F == 4) {
            const int lv = 4 * t + g;
            if (lv < n_levels) *reinterpret_cast<f32x4 *>(enc + ((int64_t)lv * n_total + row) * 4) = v[t];
        } else {
            const int lv = 2 * t + (g >> 1);
            if (lv < n_levels) *reinterpret_cast<f32x4 *>(enc + ((int64_t)lv * n_total + row) * 8 + 4 * (g & 1)) = v[t];
        }
    }
}

// ------------------------------------------------------------------------------------------------ neck forward
struct NeckFwdArgs {
    const float *enc; int64_t n; int32_t n_levels;
    WSrc w0, w1; const float *b0, *b1;
    float *h1;    // [n][64] post-ReLU hidden (may be null when no backward will follow)
    float *out0;  // [n][64] output features 0..63   (null in density mode)
    float *out1;  // [n][64] output features 64..127 (NT1 == 8 only)
    float *dens;  // [n] exp(feature0 - 1)
};

// NT1 = output tiles of the second layer: 4 (64 features), 8 (128 features), 1 (density only: 1 feature -> trunc_exp)
template <int KT0, int F, int NT1>
__global__ __launch_bounds__(kNThreads, 4) void neck_fwd_kernel(const NeckFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P0 = KT0 * 16 + 4, P1 = 64 + 4;
    float *w0l = smem, *w1l = w0l + 64 * P0, *b0l = w1l + NT1 * 16 * P1, *b1l = b0l + 64;
    stage_w(w0l, P0, 64, KT0 * 16, a.w0);
    stage_w(w1l, P1, NT1 * 16, 64, a.w1);
    stage_b(b0l, 64, a.b0, a.w0.n);
    stage_b(b1l, NT1 * 16, a.b1, a.w1.n);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const float *w0p = w0l + m * P0 + 4 * g, *w1p = w1l + m * P1 + 4 * g;
    const int64_t n_tiles = (a.n + 15) >> 4, n_chunks = (n_tiles + kNeckChunk - 1) / kNeckChunk;
    for (int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; c < n_chunks; c += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t t0 = c * kNeckChunk;
        f32x4 xn[KT0];
        ld_lm<KT0, F>(a.enc, a.n, a.n_levels, t0 * 16 + m, t0 * 16 + m < a.n, g, xn);
        for (int j = 0; j < kNeckChunk && t0 + j < n_tiles; ++j) {
            const int64_t row = (t0 + j) * 16 + m;
            const bool ok = row < a.n;
            f32x4 x[KT0];
#pragma unroll
            for (int t = 0; t < KT0; ++t) x[t] = xn[t];
            if (j + 1 < kNeckChunk && t0 + j + 1 < n_tiles) ld_lm<KT0, F>(a.enc, a.n, a.n_levels, row + 16, row + 16 < a.n, g, xn);
            f32x4 h[4];
            init_bias<4>(b0l, g, h);
            tgemm<KT0, 4>(w0p, P0, x, h);
            relu<4>(h);
            if (a.h1) st_rm<4>(a.h1 + row * 64, ok, g, h);
            if constexpr (NT1 == 1) {
                f32x4 o[1];
                init_bias<1>(b1l, g, o);
                tgemm<4, 1>(w1p, P1, h, o);
                if (ok && g == 0) a.dens[row] = expf(o[0][0] - 1.0f);
            } else {
                // 64 output features at a time (16 live accumulators instead of 32)
                f32x4 o[4];
                init_bias<4>(b1l, g, o);
                tgemm<4, 4>(w1p, P1, h, o);
                st_rm<4>(a.out0 + row * 64, ok, g, o);
                if (a.dens && ok && g == 0) a.dens[row] = expf(o[0][0] - 1.0f);
                if constexpr (NT1 == 8) {
                    init_bias<4>(b1l + 64, g, o);
                    tgemm<4, 4>(w1p + 64 * P1, P1, h, o);
                    st_rm<4>(a.out1 + row * 64, ok, g, o);
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------- neck backward
struct NeckBwdArgs {
    const float *d0;     // [n][64] gradient of output features 0..63 (null: zero)
    const float *d1;     // [n][64] gradient of output features 64..127 (KT1 == 8 only)
    const float *ddens;  // [n] gradient of the density (null: none)
    const float *dens;   // [n] saved density (trunc_exp backward: ddens * min(dens, e^15) joins feature 0)
    const float *h1;     // [n][64] saved hidden activations
    int64_t n; int32_t n_levels;
    WSrc w1t, w0t;       // W1^T (64 x K1), W0^T (K0 x 64)
    float *dpre1;        // density mode: [n] pre-activation gradient of the single output (for its wgrad)
    float *dcol0;        // [n] d0[:, 0] + density fix (may be null)
    float *dpre0;        // [n][64]
    float *denc;         // level-major [L][n][F]
};

// KT1 = input tiles of the transposed second layer: 4 / 8 (neck), 0 (density mode: rank-1, no MFMA)
template <int KT0, int F, int KT1>
__global__ __launch_bounds__(kNThreads, 4) void neck_bwd_kernel(const NeckBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K1 = (KT1 == 0 ? 1 : KT1) * 16;
    constexpr int P1 = K1 + 4, P0 = 64 + 4;
    float *w1l = smem, *w0l = w1l + 64 * P1;
    stage_w(w1l, P1, 64, K1, a.w1t);
    stage_w(w0l, P0, KT0 * 16, 64, a.w0t);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const float *w1p = w1l + m * P1 + 4 * g, *w0p = w0l + m * P0 + 4 * g;
    const int64_t n_tiles = (a.n + 15) >> 4, n_chunks = (n_tiles + kNeckChunk - 1) / kNeckChunk;
    for (int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; c < n_chunks; c += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t t0 = c * kNeckChunk;
        for (int j = 0; j < kNeckChunk && t0 + j < n_tiles; ++j) {
            const int64_t row = (t0 + j) * 16 + m;
            const bool ok = row < a.n;
            f32x4 mk[4];
            ld_rm<4>(a.h1 + row * 64, ok, g, mk);
            float fix = 0.0f;
            if (a.ddens && ok) fix = a.ddens[row] * fminf(a.dens[row], 3269017.3724721107f);
            f32x4 da[4];
            if constexpr (KT1 == 0) {
                // rank-1: dA[n] = W1[0][n] * dPre1
                if (a.dpre1 && ok && g == 0) a.dpre1[row] = fix;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) da[p][i] = w1l[(16 * p + 4 * g + i) * P1] * fix;
            } else {
                f32x4 d[KT1];
                {
                    f32x4 lo[4];
                    if (a.d0) ld_rm<4>(a.d0 + row * 64, ok, g, lo); else zero<4>(lo);
                    if (g == 0) {
                        lo[0][0] += fix;
                        if (a.dcol0 && ok) a.dcol0[row] = lo[0][0];
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) d[p] = lo[p];
                }
                if constexpr (KT1 == 8) {
                    f32x4 hi[4];
                    ld_rm<4>(a.d1 + row * 64, ok, g, hi);
#pragma unroll
                    for (int p = 0; p < 4; ++p) d[4 + p] = hi[p];
                }
                zero<4>(da);
                tgemm<KT1, 4>(w1p, P1, d, da);
            }
            relu_mask<4>(da, mk);
            st_rm<4>(a.dpre0 + row * 64, ok, g, da);
            f32x4 de[KT0];
            zero<KT0>(de);
            tgemm<4, KT0>(w0p, P0, da, de);
            st_lm<KT0, F>(a.denc, a.n, a.n_levels, row, ok, g, de);
        }
    }
}

// ------------------------------------------------------------------------------------------------- rgb forward
struct RgbFwdArgs {
    const float *geo; int64_t ld_geo;  // [n][>= 64]
    const float *rb0, *rb1; int64_t ld_rb;  // [rays][64] (row stride ld_rb) per-ray pre-activations, bias included
    int32_t tiles_per_ray; int64_t n_rays;
    WSrc w0g, w1a, w1g, w2; const float *b2;
    float *a1, *a2;                    // [n][64]
    float *out;                        // [n][3]
};

__global__ __launch_bounds__(kNThreads, 4) void rgb_fwd_kernel(const RgbFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = 64 + 4;
    float *w0l = smem, *w1al = w0l + 64 * P, *w1gl = w1al + 64 * P, *w2l = w1gl + 64 * P, *b2l = w2l + 16 * P;
    stage_w(w0l, P, 64, 64, a.w0g);
    stage_w(w1al, P, 64, 64, a.w1a);
    stage_w(w1gl, P, 64, 64, a.w1g);
    stage_w(w2l, P, 16, 64, a.w2);
    stage_b(b2l, 16, a.b2, a.w2.n);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const int off = m * P + 4 * g;
    const int tpr = a.tiles_per_ray;
    for (int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; ray < a.n_rays; ray += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        // the per-ray pre-activations are re-read for every tile (L1/L2 hits) instead of living in 32 VGPRs for the
        // whole ray: the kernel then fits 128 VGPRs = 4 waves per SIMD
        const float *rb0 = a.rb0 + ray * a.ld_rb, *rb1 = a.rb1 + ray * a.ld_rb;
        const int64_t row_base = ray * tpr * 16 + m;
        f32x4 xn[4];
        ld_rm<4>(a.geo + row_base * a.ld_geo, true, g, xn);
        for (int j = 0; j < tpr; ++j) {
            const int64_t row = row_base + (int64_t)j * 16;
            f32x4 x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = xn[t];
            if (j + 1 < tpr) ld_rm<4>(a.geo + (row + 16) * a.ld_geo, true, g, xn);
            f32x4 h[4];
            ld_rm<4>(rb0, true, g, h);
            tgemm<4, 4, false>(w0l + off, P, x, h);
            relu<4>(h);
            st_rm<4>(a.a1 + row * 64, true, g, h);
            f32x4 h2[4];
            ld_rm<4>(rb1, true, g, h2);
            tgemm<4, 4, false>(w1gl + off, P, x, h2);   // geo part first: x dies here
            tgemm<4, 4, false>(w1al + off, P, h, h2);
            relu<4>(h2);
            st_rm<4>(a.a2 + row * 64, true, g, h2);
            f32x4 o[1];
            init_bias<1>(b2l, g, o);
            tgemm<4, 1, false>(w2l + off, P, h2, o);
            if (g == 0) {
                float *op = a.out + row * 3;
#pragma unroll
                for (int i = 0; i < 3; ++i) op[i] = 1.0f / (1.0f + expf(-o[0][i]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ rgb backward
struct RgbBwdArgs {
    const float *dout, *out;           // [n][3] gradient of / saved sigmoid output
    const float *a1, *a2;              // [n][64] saved activations
    int32_t tiles_per_ray; int64_t n_rays;
    WSrc w2t, w1at, w1gt, w0gt;        // transposed views: W2^T (64 x 3), W1a^T, W1g^T, W0g^T (64 x 64)
    float *dpre2;                      // [n][3]
    float *dpre1, *dpre0, *dgeo;       // [n][64]
    float *s1, *s0;                    // [rays][64] sums of dpre1 / dpre0 over the samples of each ray
};

__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes that share g (a DPP row)
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

__global__ __launch_bounds__(kNThreads, 4) void rgb_bwd_kernel(const RgbBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P = 64 + 4, P2 = 16 + 4;
    float *w2l = smem, *w1al = w2l + 64 * P2, *w1gl = w1al + 64 * P, *w0l = w1gl + 64 * P;
    stage_w(w2l, P2, 64, 16, a.w2t);
    stage_w(w1al, P, 64, 64, a.w1at);
    stage_w(w1gl, P, 64, 64, a.w1gt);
    stage_w(w0l, P, 64, 64, a.w0gt);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const int off = m * P + 4 * g;
    const int tpr = a.tiles_per_ray;
    for (int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; ray < a.n_rays; ray += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        f32x4 s1[4], s0[4];
        zero<4>(s1); zero<4>(s0);
        const int64_t row_base = ray * tpr * 16 + m;
        for (int j = 0; j < tpr; ++j) {
            const int64_t row = row_base + (int64_t)j * 16;
            f32x4 m2[4];
            ld_rm<4>(a.a2 + row * 64, true, g, m2);
            f32x4 d2[1];
            d2[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float y = a.out[row * 3 + i];
                    d2[0][i] = a.dout[row * 3 + i] * y * (1.0f - y);  // sigmoid'
                    a.dpre2[row * 3 + i] = d2[0][i];
                }
            }
            f32x4 d1[4];
            zero<4>(d1);
            tgemm<1, 4, false>(w2l + m * P2 + 4 * g, P2, d2, d1);
            relu_mask<4>(d1, m2);
            st_rm<4>(a.dpre1 + row * 64, true, g, d1);
            f32x4 m1[4];  // loaded here (behind tgemm's scheduling barrier): its live range does not overlap a2's mask
            ld_rm<4>(a.a1 + row * 64, true, g, m1);
            f32x4 d0[4];
            zero<4>(d0);
            tgemm<4, 4, false>(w1al + off, P, d1, d0);
            relu_mask<4>(d0, m1);
            st_rm<4>(a.dpre0 + row * 64, true, g, d0);
            f32x4 dg[4];
            zero<4>(dg);
            tgemm<4, 4, false>(w1gl + off, P, d1, dg);
            tgemm<4, 4, false>(w0l + off, P, d0, dg);
            st_rm<4>(a.dgeo + row * 64, true, g, dg);
#pragma unroll
            for (int p = 0; p < 4; ++p) { s1[p] += d1[p]; s0[p] += d0[p]; }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1[p][i] = row16_sum(s1[p][i]); s0[p][i] = row16_sum(s0[p][i]); }
        if (m == 0) {
            st_rm<4>(a.s1 + ray * 64, true, g, s1);
            st_rm<4>(a.s0 + ray * 64, true, g, s0);
        }
    }
}

// --------------------------------------------------------------- plain 2- / 3-layer heads (hidden width 64)
// out = act(W_last relu(... relu(W0 x + b0) ...) + b_last): the flow MLP (xyzt grid 40 -> 64 -> 64 -> 6,
// radiance_field.py:101-111), the shadow head (64 -> 64 -> 1 + sigmoid, :148-153) and the feature heads (64 -> 64 -> 64 ->
// E, :192-198) on the same register-resident transposed chaining as the neck: a wave owns 16 rows end to end, LDS holds
// the weights only.  (Round 1 ran these on the generic LDS-staged chain kernel: 16 KB of row buffer per wave, 6-8 waves
// per CU, ~4x slower.)  Input: row-major [n][ldx] (F == 0) or a level-major grid encoding [L][n][F].
struct RMlpFwdArgs {
    const float *x; int64_t ldx;
    int64_t n; int32_t n_levels, k0, n_out, final_act;
    WSrc w0, w1, w2; const float *b0, *b1, *b2;   // NL == 2: w0, w1;  NL == 3: w0, w1, w2
    float *h1, *h2;                               // [n][64] saved post-ReLU activations (null: not needed)
    float *out; int64_t ldo;                      // [n][ldo >= n_out]
};

// row-major [rows][ldx] input with k0 <= 16 * KT valid columns (k0 and ldx multiples of 4)
template <int KT>
__device__ __forceinline__ void ld_rm_k(const float *rowp, bool ok, int g, int k0, f32x4 (&v)[KT]) {
#pragma unroll
    for (int t = 0; t < KT; ++t)
        v[t] = (ok && 16 * t + 4 * g < k0) ? *reinterpret_cast<const f32x4 *>(rowp + t * 16 + 4 * g) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
template <int KT>
__device__ __forceinline__ void st_rm_k(float *rowp, bool ok, int g, int k0, const f32x4 (&v)[KT]) {
    if (!ok) return;
#pragma unroll
    for (int t = 0; t < KT; ++t)
        if (16 * t + 4 * g < k0) *reinterpret_cast<f32x4 *>(rowp + t * 16 + 4 * g) = v[t];
}
// [rows][ld] tensor with n_valid <= 16 * NT columns, any ld: scalar accesses (narrow outputs: 1, 3, 6 channels)
template <int NT>
__device__ __forceinline__ void ld_narrow(const float *rowp, bool ok, int g, int n_valid, f32x4 (&v)[NT]) {
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[p][i] = (ok && 16 * p + 4 * g + i < n_valid) ? rowp[16 * p + 4 * g + i] : 0.0f;
}
template <int NT>
__device__ __forceinline__ void st_narrow(float *rowp, bool ok, int g, int n_valid, const f32x4 (&v)[NT]) {
    if (!ok) return;
#pragma unroll
    for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (16 * p + 4 * g + i < n_valid) rowp[16 * p + 4 * g + i] = v[p][i];
}

template <int KT0, int F, int NL, int NTO>
__global__ __launch_bounds__(kNThreads, 4) void rmlp_fwd_kernel(const RMlpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int P0 = KT0 * 16 + 4, P = 64 + 4;
    float *w0l = smem, *w1l = w0l + 64 * P0, *w2l = w1l + (NL == 3 ? 64 : NTO * 16) * P;
    float *b0l = w2l + (NL == 3 ? NTO * 16 * P : 0), *b1l = b0l + 64, *b2l = b1l + 64;
    stage_w(w0l, P0, 64, KT0 * 16, a.w0);
    stage_w(w1l, P, NL == 3 ? 64 : NTO * 16, 64, a.w1);
    if (NL == 3) stage_w(w2l, P, NTO * 16, 64, a.w2);
    stage_b(b0l, 64, a.b0, a.w0.n);
    stage_b(b1l, 64, a.b1, a.w1.n);
    if (NL == 3) stage_b(b2l, 64, a.b2, a.w2.n);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const float *w0p = w0l + m * P0 + 4 * g, *w1p = w1l + m * P + 4 * g, *w2p = w2l + m * P + 4 * g;
    const bool wide_out = (a.n_out & 3) == 0 && (a.ldo & 3) == 0;
    const int64_t n_tiles = (a.n + 15) >> 4, n_chunks = (n_tiles + kNeckChunk - 1) / kNeckChunk;
    auto load_x = [&](int64_t row, f32x4 (&v)[KT0]) {
        if constexpr (F == 0) ld_rm_k<KT0>(a.x + row * a.ldx, row < a.n, g, a.k0, v);
        else ld_lm<KT0, F>(a.x, a.n, a.n_levels, row, row < a.n, g, v);
    };
    for (int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; c < n_chunks; c += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t t0 = c * kNeckChunk;
        f32x4 xn[KT0];
        load_x(t0 * 16 + m, xn);
        for (int j = 0; j < kNeckChunk && t0 + j < n_tiles; ++j) {
            const int64_t row = (t0 + j) * 16 + m;
            const bool ok = row < a.n;
            f32x4 x[KT0];
#pragma unroll
            for (int t = 0; t < KT0; ++t) x[t] = xn[t];
            if (j + 1 < kNeckChunk && t0 + j + 1 < n_tiles) load_x(row + 16, xn);
            f32x4 h[4];
            init_bias<4>(b0l, g, h);
            tgemm<KT0, 4, false>(w0p, P0, x, h);
            relu<4>(h);
            if (a.h1) st_rm<4>(a.h1 + row * 64, ok, g, h);
            if constexpr (NL == 3) {
                f32x4 h2[4];
                init_bias<4>(b1l, g, h2);
                tgemm<4, 4, false>(w1p, P, h, h2);
                relu<4>(h2);
                if (a.h2) st_rm<4>(a.h2 + row * 64, ok, g, h2);
#pragma unroll
                for (int p = 0; p < 4; ++p) h[p] = h2[p];
            }
            f32x4 o[NTO];
            init_bias<NTO>(NL == 3 ? b2l : b1l, g, o);
            tgemm<4, NTO, false>(NL == 3 ? w2p : w1p, P, h, o);
            if (a.final_act == EMER_ACT_SIGMOID) {
#pragma unroll
                for (int p = 0; p < NTO; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[p][i] = 1.0f / (1.0f + expf(-o[p][i]));
            }
            if (wide_out) st_rm_k<NTO>(a.out + row * a.ldo, ok, g, a.n_out, o);
            else st_narrow<NTO>(a.out + row * a.ldo, ok, g, a.n_out, o);
        }
    }
}

struct RMlpBwdArgs {
    const float *dlast; int64_t ldd;   // [n][ldd >= n_out] gradient at the LAST pre-activation (sigmoid' applied by the caller)
    const float *h1, *h2;              // saved activations
    int64_t n; int32_t n_levels, k0, n_out;
    WSrc wlt, w1t, w0t;                // W_last^T (64 x n_out), W1^T (64 x 64, NL == 3), W0^T (k0 x 64)
    float *dpre1, *dpre0;              // [n][64] gradients at the hidden pre-activations (dpre1: NL == 3 only)
    float *dx; int64_t lddx;           // gradient of the input: row-major [n][lddx] or level-major; null: not needed
};

template <int KT0, int F, int NL, int NTO>
__global__ __launch_bounds__(kNThreads, 4) void rmlp_bwd_kernel(const RMlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PL = NTO * 16 + 4, P = 64 + 4;
    float *wll = smem, *w1l = wll + 64 * PL, *w0l = w1l + (NL == 3 ? 64 * P : 0);
    stage_w(wll, PL, 64, NTO * 16, a.wlt);
    if (NL == 3) stage_w(w1l, P, 64, 64, a.w1t);
    stage_w(w0l, P, KT0 * 16, 64, a.w0t);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const float *wlp = wll + m * PL + 4 * g, *w1p = w1l + m * P + 4 * g, *w0p = w0l + m * P + 4 * g;
    const bool wide_in = (a.n_out & 3) == 0 && (a.ldd & 3) == 0;
    const int64_t n_tiles = (a.n + 15) >> 4;
    for (int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; t < n_tiles; t += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t row = t * 16 + m;
        const bool ok = row < a.n;
        f32x4 d[NTO];
        if (wide_in) ld_rm_k<NTO>(a.dlast + row * a.ldd, ok, g, a.n_out, d);
        else ld_narrow<NTO>(a.dlast + row * a.ldd, ok, g, a.n_out, d);
        f32x4 da[4];
        zero<4>(da);
        tgemm<NTO, 4, false>(wlp, PL, d, da);
        if constexpr (NL == 3) {
            f32x4 mk2[4];
            ld_rm<4>(a.h2 + row * 64, ok, g, mk2);
            relu_mask<4>(da, mk2);
            st_rm<4>(a.dpre1 + row * 64, ok, g, da);
            f32x4 db[4];
            zero<4>(db);
            tgemm<4, 4, false>(w1p, P, da, db);
#pragma unroll
            for (int p = 0; p < 4; ++p) da[p] = db[p];
        }
        f32x4 mk1[4];
        ld_rm<4>(a.h1 + row * 64, ok, g, mk1);
        relu_mask<4>(da, mk1);
        st_rm<4>(a.dpre0 + row * 64, ok, g, da);
        if (a.dx) {
            f32x4 de[KT0];
            zero<KT0>(de);
            tgemm<4, KT0, false>(w0p, P, da, de);
            if constexpr (F == 0) st_rm_k<KT0>(a.dx + row * a.lddx, ok, g, a.k0, de);
            else st_lm<KT0, F>(a.dx, a.n, a.n_levels, row, ok, g, de);
        }
    }
}

static inline uint32_t fused_grid(int64_t work_items, int threads = kFThreads) {
    const int waves = threads / 64;
    int64_t blocks = (work_items + waves - 1) / waves;
    if (blocks > 512) blocks = 512;  // persistent: 2 workgroups per CU
    return (uint32_t)(blocks < 1 ? 1 : blocks);
}

template <typename K>
static int set_lds(K kern, size_t lds, const char *what) {
    return reserve_lds(reinterpret_cast<const void *>(kern), lds, what);
}

}  // namespace emer

using namespace emer;

// 1 when the register-resident kernels cover a neck / density MLP of this shape
extern "C" int emer_neck_supported(int32_t n_levels, int32_t n_feat, int32_t hidden, int32_t n_out) {
    const int k0 = n_levels * n_feat;
    const bool f_ok = n_feat == 1 || n_feat == 2 || n_feat == 4 || n_feat == 8;
    return (f_ok && k0 >= 1 && k0 <= 64 && hidden == 64 && (n_out == 1 || n_out == 64 || n_out == 128)) ? 1 : 0;
}

#define EMER_NECK_DISPATCH(KERNEL, NTX, ARGS, LDS, WHAT)                                                        \
    do {                                                                                                       \
        const int kt0 = (k0 + 15) / 16;                                                                        \
        int rc_ = EMER_E_INVALID;                                                                                  \
        auto go = [&](auto kern) {                                                                             \
            if (int r = set_lds(kern, LDS(kt0), WHAT)) return r;                                               \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kNThreads), LDS(kt0), st, ARGS);                         \
            return check_launch(WHAT);                                                                         \
        };                                                                                                     \
        if (n_feat == 1) { if (kt0 == 1) rc_ = go(KERNEL<1, 1, NTX>); else if (kt0 == 2) rc_ = go(KERNEL<2, 1, NTX>); else if (kt0 == 3) rc_ = go(KERNEL<3, 1, NTX>); else rc_ = go(KERNEL<4, 1, NTX>); } \
        else if (n_feat == 2) { if (kt0 == 1) rc_ = go(KERNEL<1, 2, NTX>); else if (kt0 == 2) rc_ = go(KERNEL<2, 2, NTX>); else if (kt0 == 3) rc_ = go(KERNEL<3, 2, NTX>); else rc_ = go(KERNEL<4, 2, NTX>); } \
        else if (n_feat == 4) { if (kt0 == 1) rc_ = go(KERNEL<1, 4, NTX>); else if (kt0 == 2) rc_ = go(KERNEL<2, 4, NTX>); else if (kt0 == 3) rc_ = go(KERNEL<3, 4, NTX>); else rc_ = go(KERNEL<4, 4, NTX>); } \
        else { if (kt0 == 1) rc_ = go(KERNEL<1, 8, NTX>); else if (kt0 == 2) rc_ = go(KERNEL<2, 8, NTX>); else if (kt0 == 3) rc_ = go(KERNEL<3, 8, NTX>); else rc_ = go(KERNEL<4, 8, NTX>); } \
        return rc_;                                                                                            \
    } while (0)

// enc_lm [L][n][F] -> h1 = relu(enc W0^T + b0) [n][64] -> out = h1 W1^T + b1.
// n_out == 64 / 128: out0 [n][64] (features 0..63), out1 [n][64] (features 64..127), dens = exp(out[:,0] - 1) if non-null;
// n_out == 1: dens = exp(out - 1) only (proposal density MLP).  w0 [64][L*F], w1 [n_out][64] row-major.
extern "C" int emer_neck_fwd(const float *enc_lm, int32_t n_levels, int32_t n_feat, int64_t n, const float *w0, const float *b0,
                             const float *w1, const float *b1, int32_t n_out, float *h1, float *out0, float *out1, float *dens,
                             void *stream) {
    EMER_REQUIRE(n >= 0, "neck_fwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(emer_neck_supported(n_levels, n_feat, 64, n_out), "neck_fwd: unsupported shape L=%d F=%d n_out=%d", n_levels, n_feat, n_out);
    EMER_REQUIRE(enc_lm && w0 && w1, "neck_fwd: null pointer");
    EMER_REQUIRE(n_out == 1 ? dens != nullptr : (out0 != nullptr && (n_out == 64 || out1 != nullptr)), "neck_fwd: missing output buffer");
    const int k0 = n_levels * n_feat;
    NeckFwdArgs a;
    a.enc = enc_lm; a.n = n; a.n_levels = n_levels;
    a.w0 = WSrc{w0, k0, 1, 64, k0};
    a.w1 = WSrc{w1, 64, 1, n_out, 64};
    a.b0 = b0; a.b1 = b1; a.h1 = h1; a.out0 = out0; a.out1 = out1; a.dens = dens;
    hipStream_t st = as_stream(stream);
    const uint32_t grid = fused_grid(((n + 15) / 16 + kNeckChunk - 1) / kNeckChunk, kNThreads);
    if (n_out == 1) {
        auto lds = [](int kt0) { return (size_t)(64 * (kt0 * 16 + 4) + 16 * 68 + 64 + 16) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_fwd_kernel, 1, a, lds, "neck_fwd");
    } else if (n_out == 64) {
        auto lds = [](int kt0) { return (size_t)(64 * (kt0 * 16 + 4) + 64 * 68 + 64 + 64) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_fwd_kernel, 4, a, lds, "neck_fwd");
    } else {
        auto lds = [](int kt0) { return (size_t)(64 * (kt0 * 16 + 4) + 128 * 68 + 64 + 128) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_fwd_kernel, 8, a, lds, "neck_fwd");
    }
}

// Data-gradient chain of emer_neck_fwd.  d0 / d1: gradients of output features 0..63 / 64..127 (either may be NULL
// = zero; n_out == 1: both NULL).  ddens/dens: trunc_exp backward, joins feature 0.  Writes dpre0 [n][64] (gradient
// at the hidden pre-activation), denc_lm [L][n][F], for n_out == 1 dpre1 [n], and (optionally) dcol0 [n] = d0[:,0] + fix.
extern "C" int emer_neck_bwd(const float *d0, const float *d1, const float *ddens, const float *dens, const float *h1,
                             int32_t n_levels, int32_t n_feat, int64_t n, const float *w0, const float *w1, int32_t n_out,
                             float *dpre1, float *dcol0, float *dpre0, float *denc_lm, void *stream) {
    EMER_REQUIRE(n >= 0, "neck_bwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(emer_neck_supported(n_levels, n_feat, 64, n_out), "neck_bwd: unsupported shape L=%d F=%d n_out=%d", n_levels, n_feat, n_out);
    EMER_REQUIRE(h1 && w0 && w1 && dpre0 && denc_lm, "neck_bwd: null pointer");
    EMER_REQUIRE(!ddens || dens, "neck_bwd: ddens needs the saved density");
    EMER_REQUIRE(n_out != 1 || (ddens && !d0 && !d1), "neck_bwd: density mode takes ddens only");
    const int k0 = n_levels * n_feat;
    NeckBwdArgs a;
    a.d0 = d0; a.d1 = d1; a.ddens = ddens; a.dens = dens; a.h1 = h1; a.n = n; a.n_levels = n_levels;
    a.dpre1 = dpre1; a.dcol0 = dcol0; a.dpre0 = dpre0; a.denc = denc_lm;
    a.w0t = WSrc{w0, 1, k0, k0, 64};     // (n = input feature, k = hidden) = w0[k][n]
    hipStream_t st = as_stream(stream);
    const uint32_t grid = fused_grid(((n + 15) / 16 + kNeckChunk - 1) / kNeckChunk, kNThreads);
    if (n_out == 1) {
        a.w1t = WSrc{w1, 1, 64, 64, 1};  // (n = hidden, k = 0) = w1[0][n]
        auto lds = [](int kt0) { return (size_t)(64 * 20 + kt0 * 16 * 68) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_bwd_kernel, 0, a, lds, "neck_bwd");
    } else if (n_out == 64 || !d1) {
        a.w1t = WSrc{w1, 1, 64, 64, 64};  // only the first 64 outputs carry a gradient
        auto lds = [](int kt0) { return (size_t)(64 * 68 + kt0 * 16 * 68) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_bwd_kernel, 4, a, lds, "neck_bwd");
    } else {
        a.w1t = WSrc{w1, 1, 64, 64, 128};
        auto lds = [](int kt0) { return (size_t)(64 * 132 + kt0 * 16 * 68) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_bwd_kernel, 8, a, lds, "neck_bwd");
    }
}

// rgb head forward: a1 = relu(geo W0g^T + rb0[ray]); a2 = relu(a1 W1a^T + geo W1g^T + rb1[ray]); out = sigmoid(a2 W2^T + b2).
// w0 [64][kh + 64] = [W0h | W0g], w1 [64][64 + kh + 64] = [W1a | W1h | W1g], w2 [3][64] (torch Linear layouts of
// mlp.py:20-36 with the skip connection at layer 1).  rb0 = hray W0h^T + b0, rb1 = hray W1h^T + b1: [rays][64].
// Rows of ray r are r*S .. r*S + S - 1; S must be a multiple of 16.
extern "C" int emer_rgb_head_fwd(const float *geo, int64_t ld_geo, const float *rb0, const float *rb1, int64_t ld_rb, int64_t n_rays,
                                 int32_t samples_per_ray, int32_t kh, const float *w0, const float *w1, const float *w2,
                                 const float *b2, float *a1, float *a2, float *out, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && samples_per_ray >= 16 && samples_per_ray % 16 == 0 && kh >= 0, "rgb_head_fwd: bad sizes (S must be a multiple of 16)");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(geo && rb0 && rb1 && w0 && w1 && w2 && a1 && a2 && out && ld_geo >= 64 && ld_geo % 4 == 0 && ld_rb >= 64 && ld_rb % 4 == 0,
                 "rgb_head_fwd: bad arguments");
    RgbFwdArgs a;
    a.geo = geo; a.ld_geo = ld_geo; a.rb0 = rb0; a.rb1 = rb1; a.ld_rb = ld_rb; a.tiles_per_ray = samples_per_ray / 16; a.n_rays = n_rays;
    const int64_t k0 = kh + 64, k1 = 64 + k0;
    a.w0g = WSrc{w0 + kh, k0, 1, 64, 64};
    a.w1a = WSrc{w1, k1, 1, 64, 64};
    a.w1g = WSrc{w1 + 64 + kh, k1, 1, 64, 64};
    a.w2 = WSrc{w2, 64, 1, 3, 64};
    a.b2 = b2; a.a1 = a1; a.a2 = a2; a.out = out;
    const size_t lds = (size_t)(3 * 64 * 68 + 16 * 68 + 16) * sizeof(float);
    if (int rc = set_lds(rgb_fwd_kernel, lds, "rgb_head_fwd")) return rc;
    hipLaunchKernelGGL(rgb_fwd_kernel, dim3(fused_grid(n_rays, kNThreads)), dim3(kNThreads), lds, as_stream(stream), a);
    return check_launch("rgb_head_fwd");
}

// rgb head data-gradient chain.  Writes dpre2 [n][3], dpre1 / dpre0 / dgeo [n][64] and the per-ray sums s1 / s0
// [rays][64] of dpre1 / dpre0 (everything the per-ray operands hray, W0h, W1h, b0, b1 need).
extern "C" int emer_rgb_head_bwd(const float *dout, const float *out, const float *a1, const float *a2, int64_t n_rays,
                                 int32_t samples_per_ray, int32_t kh, const float *w0, const float *w1, const float *w2,
                                 float *dpre2, float *dpre1, float *dpre0, float *dgeo, float *s1, float *s0, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && samples_per_ray >= 16 && samples_per_ray % 16 == 0 && kh >= 0, "rgb_head_bwd: bad sizes (S must be a multiple of 16)");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(dout && out && a1 && a2 && w0 && w1 && w2 && dpre2 && dpre1 && dpre0 && dgeo && s1 && s0, "rgb_head_bwd: null pointer");
    RgbBwdArgs a;
    a.dout = dout; a.out = out; a.a1 = a1; a.a2 = a2; a.tiles_per_ray = samples_per_ray / 16; a.n_rays = n_rays;
    const int64_t k0 = kh + 64, k1 = 64 + k0;
    a.w2t = WSrc{w2, 1, 64, 64, 3};               // (n = hidden, k = channel) = w2[k][n]
    a.w1at = WSrc{w1, 1, k1, 64, 64};             // (n = a1 feature, k = layer-1 output) = w1[k][n]
    a.w1gt = WSrc{w1 + 64 + kh, 1, k1, 64, 64};
    a.w0gt = WSrc{w0 + kh, 1, k0, 64, 64};
    a.dpre2 = dpre2; a.dpre1 = dpre1; a.dpre0 = dpre0; a.dgeo = dgeo; a.s1 = s1; a.s0 = s0;
    const size_t lds = (size_t)(64 * 20 + 3 * 64 * 68) * sizeof(float);
    if (int rc = set_lds(rgb_bwd_kernel, lds, "rgb_head_bwd")) return rc;
    hipLaunchKernelGGL(rgb_bwd_kernel, dim3(fused_grid(n_rays, kNThreads)), dim3(kNThreads), lds, as_stream(stream), a);
    return check_launch("rgb_head_bwd");
}

// ---- plain 2- / 3-layer heads -------------------------------------------------------------------------------------
// 1 when emer_rmlp_fwd / emer_rmlp_bwd cover this stack: n_layers Linear layers (2 or 3) with hidden width 64, k0 <= 64
// inputs (row-major: n_feat == 0, k0 a multiple of 4; level-major grid encoding: n_feat == 4), n_out <= 64 outputs.
extern "C" int emer_rmlp_supported(int32_t n_layers, int32_t k0, int32_t n_feat, int32_t hidden, int32_t n_out) {
    const bool in_ok = (n_feat == 0 && k0 % 4 == 0) || (n_feat == 4 && k0 % 4 == 0);
    return ((n_layers == 2 || n_layers == 3) && in_ok && k0 >= 4 && k0 <= 64 && hidden == 64 && n_out >= 1 && n_out <= 64) ? 1 : 0;
}

#define EMER_RMLP_DISPATCH(KERNEL, ARGS, LDS, WHAT)                                                                      \
    do {                                                                                                               \
        int rc_ = EMER_E_INVALID;                                                                                      \
        auto go = [&](auto kern) {                                                                                     \
            if (int r = set_lds(kern, LDS, WHAT)) return r;                                                            \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kNThreads), LDS, st, ARGS);                                      \
            return check_launch(WHAT);                                                                                 \
        };                                                                                                             \
        if (n_feat == 0) {                                                                                             \
            if (n_layers == 2) { if (nto == 1) rc_ = go(KERNEL<4, 0, 2, 1>); else rc_ = go(KERNEL<4, 0, 2, 4>); }      \
            else               { if (nto == 1) rc_ = go(KERNEL<4, 0, 3, 1>); else rc_ = go(KERNEL<4, 0, 3, 4>); }      \
        } else if (kt0 <= 2) {                                                                                         \
            if (n_layers == 2) { if (nto == 1) rc_ = go(KERNEL<2, 4, 2, 1>); else rc_ = go(KERNEL<2, 4, 2, 4>); }      \
            else               { if (nto == 1) rc_ = go(KERNEL<2, 4, 3, 1>); else rc_ = go(KERNEL<2, 4, 3, 4>); }      \
        } else if (kt0 == 3) {                                                                                         \
            if (n_layers == 2) { if (nto == 1) rc_ = go(KERNEL<3, 4, 2, 1>); else rc_ = go(KERNEL<3, 4, 2, 4>); }      \
            else               { if (nto == 1) rc_ = go(KERNEL<3, 4, 3, 1>); else rc_ = go(KERNEL<3, 4, 3, 4>); }      \
        } else {                                                                                                       \
            if (n_layers == 2) { if (nto == 1) rc_ = go(KERNEL<4, 4, 2, 1>); else rc_ = go(KERNEL<4, 4, 2, 4>); }      \
            else               { if (nto == 1) rc_ = go(KERNEL<4, 4, 3, 1>); else rc_ = go(KERNEL<4, 4, 3, 4>); }      \
        }                                                                                                              \
        return rc_;                                                                                                    \
    } while (0)

// x: row-major [n][ldx] (n_feat == 0, n_levels ignored) or level-major [n_levels][n][n_feat] with k0 = n_levels * n_feat.
// Weights in torch Linear layout: w0 [64][k0], (w1 [64][64],) w_last [n_out][64]; for two layers pass w2 = b2 = NULL and
// the output layer as w1 / b1.  h1 / h2 [n][64]: saved activations for the backward (NULL: inference).
extern "C" int emer_rmlp_fwd(const float *x, int64_t ldx, int32_t n_levels, int32_t n_feat, int32_t k0, int64_t n, int32_t n_layers,
                             const float *w0, const float *b0, const float *w1, const float *b1, const float *w2, const float *b2,
                             int32_t n_out, int32_t final_act, float *h1, float *h2, float *out, int64_t ldo, void *stream) {
    EMER_REQUIRE(n >= 0, "rmlp_fwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(emer_rmlp_supported(n_layers, k0, n_feat, 64, n_out), "rmlp_fwd: unsupported stack (layers=%d k0=%d F=%d n_out=%d)", n_layers, k0, n_feat, n_out);
    EMER_REQUIRE(x && w0 && w1 && out && (n_layers == 2 || w2) && ldo >= n_out, "rmlp_fwd: bad arguments");
    EMER_REQUIRE(n_feat != 0 || (ldx >= k0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0), "rmlp_fwd: row-major input needs ldx %% 4 == 0 and 16-byte alignment");
    EMER_REQUIRE(n_feat == 0 || n_levels * n_feat == k0, "rmlp_fwd: k0 != n_levels * n_feat");
    EMER_REQUIRE(final_act == EMER_ACT_NONE || final_act == EMER_ACT_SIGMOID, "rmlp_fwd: final activation must be none or sigmoid");
    RMlpFwdArgs a;
    a.x = x; a.ldx = ldx; a.n = n; a.n_levels = n_levels; a.k0 = k0; a.n_out = n_out; a.final_act = final_act;
    a.w0 = WSrc{w0, k0, 1, 64, k0};
    a.w1 = WSrc{w1, 64, 1, n_layers == 2 ? n_out : 64, 64};
    a.w2 = WSrc{w2, 64, 1, n_out, 64};
    a.b0 = b0; a.b1 = b1; a.b2 = b2; a.h1 = h1; a.h2 = h2; a.out = out; a.ldo = ldo;
    const int kt0 = n_feat == 0 ? 4 : (k0 + 15) / 16, nto = n_out <= 16 ? 1 : 4;
    const int kt0i = n_feat == 0 ? 4 : (kt0 <= 2 ? 2 : kt0);
    const size_t lds = (size_t)(64 * (kt0i * 16 + 4) + (n_layers == 3 ? 64 : nto * 16) * 68 + (n_layers == 3 ? nto * 16 * 68 : 0) + 3 * 64) * sizeof(float);
    hipStream_t st = as_stream(stream);
    const uint32_t grid = fused_grid(((n + 15) / 16 + kNeckChunk - 1) / kNeckChunk, kNThreads);
    EMER_RMLP_DISPATCH(rmlp_fwd_kernel, a, lds, "rmlp_fwd");
}

// Data-gradient chain of emer_rmlp_fwd.  dlast [n][ldd]: gradient at the last pre-activation (the caller applies
// sigmoid').  Writes dpre0 (and dpre1 for three layers) [n][64] -- the operands of the weight gradients -- and, when dx
// is non-null, the input gradient in the input's own layout.
extern "C" int emer_rmlp_bwd(const float *dlast, int64_t ldd, const float *h1, const float *h2, int32_t n_levels, int32_t n_feat,
                             int32_t k0, int64_t n, int32_t n_layers, const float *w0, const float *w1, const float *w2, int32_t n_out,
                             float *dpre1, float *dpre0, float *dx, int64_t lddx, void *stream) {
    EMER_REQUIRE(n >= 0, "rmlp_bwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(emer_rmlp_supported(n_layers, k0, n_feat, 64, n_out), "rmlp_bwd: unsupported stack (layers=%d k0=%d F=%d n_out=%d)", n_layers, k0, n_feat, n_out);
    EMER_REQUIRE(dlast && h1 && w0 && w1 && dpre0 && ldd >= n_out && (n_layers == 2 || (w2 && h2 && dpre1)), "rmlp_bwd: bad arguments");
    EMER_REQUIRE(!dx || n_feat != 0 || (lddx >= k0 && lddx % 4 == 0 && ((uintptr_t)dx % 16) == 0), "rmlp_bwd: row-major dx needs lddx %% 4 == 0 and 16-byte alignment");
    RMlpBwdArgs a;
    a.dlast = dlast; a.ldd = ldd; a.h1 = h1; a.h2 = h2; a.n = n; a.n_levels = n_levels; a.k0 = k0; a.n_out = n_out;
    const float *wl = n_layers == 2 ? w1 : w2;
    a.wlt = WSrc{wl, 1, 64, 64, n_out};   // (n = hidden, k = output) = wl[k][n]
    a.w1t = WSrc{w1, 1, 64, 64, 64};
    a.w0t = WSrc{w0, 1, k0, k0, 64};      // (n = input feature, k = hidden) = w0[k][n]
    a.dpre1 = dpre1; a.dpre0 = dpre0; a.dx = dx; a.lddx = lddx;
    const int kt0 = n_feat == 0 ? 4 : (k0 + 15) / 16, nto = n_out <= 16 ? 1 : 4;
    const int kt0i = n_feat == 0 ? 4 : (kt0 <= 2 ? 2 : kt0);
    const size_t lds = (size_t)(64 * (nto * 16 + 4) + (n_layers == 3 ? 64 * 68 : 0) + kt0i * 16 * 68) * sizeof(float);
    hipStream_t st = as_stream(stream);
    const uint32_t grid = fused_grid((n + 15) / 16, kNThreads);
    EMER_RMLP_DISPATCH(rmlp_bwd_kernel, a, lds, "rmlp_bwd");
}
