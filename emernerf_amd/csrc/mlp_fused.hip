// Register-resident fused heads for gfx950 (hidden width 64): neck (grid encoding -> 64 -> 64/128 [+ density]),
// proposal density MLP (grid -> 64 -> 1 -> trunc_exp) and the rgb head (3-layer skip MLP + sigmoid), forward and
// data-gradient chains.  Replaces the nn.Sequential / mlp.MLP stacks of radiance_field.py:74-198,808-840 and
// mlp.py:7-46 on the per-sample hot path.
//
// Transposed chaining.  Every layer is computed as Y^T = W X^T with the WEIGHTS as the MFMA A operand and the
// ACTIVATIONS as the B operand.  A 16x16 result tile t' leaves lane (m = lane & 15, g = lane >> 4) holding features
// 16t' + 4g + i (i = 0..3) of row m -- which is a legal B operand of the next layer if its reduction index is
// enumerated in that order (any bijection of k is a valid GEMM as long as A uses the same one).  So a wave owns 16
// rows END TO END in registers: no activation ever touches LDS, there is no barrier after the weights are staged, and
// occupancy is bounded by VGPRs instead of a 16 KB-per-wave LDS row buffer.
//
// Arithmetic [r3]: fp32 results on the bf16 matrix pipe.  gfx950's f32-input MFMA runs at the fp32 VECTOR rate (157
// TFLOP/s, 1/16 of the bf16 rate) and was the wall of round 2 (heads at 42-58 % of that peak).  Every fp32 operand is
// split EXACTLY into three bf16 terms, x = x_h + x_m + x_l (|x_m| <= 2^-8 |x|, |x_l| <= 2^-16 |x|, remainder <= 2^-24 |x|:
// three 8-bit significands cover fp32's 24), and a product is evaluated as the six bf16 x bf16 partial products of
// order <= 2^-16 -- w_l x_h, w_h x_l, w_m x_m, w_m x_h, w_h x_m, w_h x_h, smallest first -- on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  bf16 x bf16 is exact in fp32; the dropped terms (w_m x_l, w_l x_m,
// w_l x_l) are <= 2^-23 relative per product, i.e. the result differs from an fp32 FMA chain by fp32-roundoff-sized
// errors (tests: same tolerances as the fp32 kernels of round 2, 1e-4 forward / 2e-4 gradients vs fp64).  Six K = 32
// instructions of 16 cycles replace eight K = 4 instructions of 32 cycles: 0.375x the matrix time.  Non-finite inputs
// give NaN (inf - inf in the split) where an fp32 product would give inf.
//
// LDS holds only the weights, pre-split and FRAGMENT-MAJOR: fragment (p, s) = the A operand of output tile p, k-step
// s is 64 lanes x 16 B contiguous, so one conflict-free ds_read_b128 per split yields an operand.  The k enumeration
// of k-step s: lane group g, element j = 0..7  <->  feature 32 s + 16 (j >> 2) + 4 g + (j & 3), i.e. two consecutive
// result tiles of the previous layer.
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

// ---- bf16x3 operands ---------------------------------------------------------------------------------------------
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// (a, b) -> packed bf16 pairs h, m, l with a = a_h + a_m + a_l (+ <= 2^-24 |a|); both subtractions are exact
__device__ __forceinline__ void split3(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    h = pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = pk_bf16(sa, sb);
}

// B operand (activations) of KS k-steps: lane (m, g) holds, for k-step s, features 32 s + 16 (j >> 2) + 4 g + (j & 3)
template <int KS> struct Opd { u32x4 h[KS], m[KS], l[KS]; };

template <int KT>
__device__ __forceinline__ void make_opd(const f32x4 (&in)[KT], Opd<(KT + 1) / 2> &o) {
#pragma unroll
    for (int s = 0; s < (KT + 1) / 2; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = 2 * s + (q >> 1), e = 2 * (q & 1);
            if (t < KT) {
                unsigned h, m, l;
                split3(in[t][e], in[t][e + 1], h, m, l);
                o.h[s][q] = h; o.m[s][q] = m; o.l[s][q] = l;
            } else {
                o.h[s][q] = 0u; o.m[s][q] = 0u; o.l[s][q] = 0u;
            }
        }
    }
}

// Weights in LDS: three split planes of np x ks fragments (64 lanes x 16 B each).
constexpr int w3_units(int np, int ks) { return 3 * np * ks * 64; }  // u32x4 units
struct W3 { const u32x4 *p; int ks, plane; };  // p: this lane's entry of fragment (0, 0) in the h plane; plane = np * ks * 64

// Stage the (zero padded) matrix of `s` as np output tiles x ks k-steps.
__device__ __forceinline__ void stage_w3(u32x4 *dst, int np, int ks, const WSrc s) {
    const int plane = np * ks * 64;
    for (int idx = threadIdx.x; idx < plane; idx += (int)blockDim.x) {
        const int lane = idx & 63, fs = idx >> 6, p = fs / ks, st = fs - p * ks;
        const int n = 16 * p + (lane & 15), g = lane >> 4;
        u32x4 h, m, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 32 * st + 16 * (q >> 1) + 4 * g + 2 * (q & 1);
            const float v0 = (n < s.n && c < s.k) ? s.w[n * s.sn + c * s.sk] : 0.0f;
            const float v1 = (n < s.n && c + 1 < s.k) ? s.w[n * s.sn + (c + 1) * s.sk] : 0.0f;
            unsigned hh, mm, ll;
            split3(v0, v1, hh, mm, ll);
            h[q] = hh; m[q] = mm; l[q] = ll;
        }
        dst[idx] = h; dst[plane + idx] = m; dst[2 * plane + idx] = l;
    }
}
__device__ __forceinline__ W3 w3_at(const u32x4 *base, int np, int ks, int lane) { return W3{base + lane, ks, np * ks * 64}; }
// the same matrix from output tile p0 on
__device__ __forceinline__ W3 w3_tile(const W3 w, int p0) { return W3{w.p + p0 * w.ks * 64, w.ks, w.plane}; }

__device__ __forceinline__ void stage_b(float *dst, int npad, const float *b, int n) {
    for (int i = threadIdx.x; i < npad; i += (int)blockDim.x) dst[i] = (b && i < n) ? b[i] : 0.0f;
}

#define EMER_MF(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

// acc[p] (output tile p) += W[16p .. 16p+15][:] . in, fp32-equivalent (six bf16 partial products, smallest first).
// Two output tiles are interleaved so that consecutive instructions hit different accumulators; the scheduling barrier
// keeps the compiler from hoisting all weight reads of a chain (it would trade ~100 VGPRs for latency that the other
// waves of the SIMD already hide).
template <int KS, int NT, bool PAIR = true>
__device__ __forceinline__ void tgemm(const W3 w, const Opd<KS> &b, f32x4 (&acc)[NT]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int p = 0; p < NT; p += (PAIR ? 2 : 1)) {
            const u32x4 *f0 = w.p + (p * w.ks + s) * 64;
            if (PAIR && p + 1 < NT) {
                const u32x4 *f1 = f0 + w.ks * 64;
                const u32x4 l0 = f0[2 * w.plane], l1 = f1[2 * w.plane], h0 = f0[0], h1 = f1[0], m0 = f0[w.plane], m1 = f1[w.plane];
                acc[p] = EMER_MF(l0, b.h[s], acc[p]); acc[p + 1] = EMER_MF(l1, b.h[s], acc[p + 1]);
                acc[p] = EMER_MF(h0, b.l[s], acc[p]); acc[p + 1] = EMER_MF(h1, b.l[s], acc[p + 1]);
                acc[p] = EMER_MF(m0, b.m[s], acc[p]); acc[p + 1] = EMER_MF(m1, b.m[s], acc[p + 1]);
                acc[p] = EMER_MF(m0, b.h[s], acc[p]); acc[p + 1] = EMER_MF(m1, b.h[s], acc[p + 1]);
                acc[p] = EMER_MF(h0, b.m[s], acc[p]); acc[p + 1] = EMER_MF(h1, b.m[s], acc[p + 1]);
                acc[p] = EMER_MF(h0, b.h[s], acc[p]); acc[p + 1] = EMER_MF(h1, b.h[s], acc[p + 1]);
            } else {
                const u32x4 l0 = f0[2 * w.plane], h0 = f0[0], m0 = f0[w.plane];
                acc[p] = EMER_MF(l0, b.h[s], acc[p]);
                acc[p] = EMER_MF(h0, b.l[s], acc[p]);
                acc[p] = EMER_MF(m0, b.m[s], acc[p]);
                acc[p] = EMER_MF(m0, b.h[s], acc[p]);
                acc[p] = EMER_MF(h0, b.m[s], acc[p]);
                acc[p] = EMER_MF(h0, b.h[s], acc[p]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
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

// The same accesses with a wave-uniform base (the tile's first row: enc + row0 * F) and 32-bit per-lane offsets: `mrow` is the
// lane's row inside the tile.  Requires n_total * n_levels * F < 2^30 (host check).
template <int KT, int F>
__device__ __forceinline__ void ld_lm_t(const float *base, unsigned n_total, int n_levels, unsigned mrow, int g, f32x4 (&v)[KT]) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        v[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (F == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lv = 16 * t + 4 * g + i;
                if (lv < n_levels) v[t][i] = base[(unsigned)lv * n_total + mrow];
            }
        } else if (F == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lv = 8 * t + 2 * g + j;
                if (lv < n_levels) {
                    const float2 x = *reinterpret_cast<const float2 *>(base + ((unsigned)lv * n_total + mrow) * 2u);
                    v[t][2 * j] = x.x; v[t][2 * j + 1] = x.y;
                }
            }
        } else if (F == 4) {
            const int lv = 4 * t + g;
            if (lv < n_levels) v[t] = *reinterpret_cast<const f32x4 *>(base + ((unsigned)lv * n_total + mrow) * 4u);
        } else {  // F == 8
            const int lv = 2 * t + (g >> 1);
            if (lv < n_levels) v[t] = *reinterpret_cast<const f32x4 *>(base + ((unsigned)lv * n_total + mrow) * 8u + 4u * (g & 1));
        }
    }
}
template <int KT, int F>
__device__ __forceinline__ void st_lm_t(float *base, unsigned n_total, int n_levels, unsigned mrow, bool ok, int g, const f32x4 (&v)[KT]) {
    if (!ok) return;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        if (F == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lv = 16 * t + 4 * g + i;
                if (lv < n_levels) base[(unsigned)lv * n_total + mrow] = v[t][i];
            }
        } else if (F == 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lv = 8 * t + 2 * g + j;
                if (lv < n_levels) *reinterpret_cast<float2 *>(base + ((unsigned)lv * n_total + mrow) * 2u) = make_float2(v[t][2 * j], v[t][2 * j + 1]);
            }
        } else if (F == 4) {
            const int lv = 4 * t + g;
            if (lv < n_levels) *reinterpret_cast<f32x4 *>(base + ((unsigned)lv * n_total + mrow) * 4u) = v[t];
        } else {
            const int lv = 2 * t + (g >> 1);
            if (lv < n_levels) *reinterpret_cast<f32x4 *>(base + ((unsigned)lv * n_total + mrow) * 8u + 4u * (g & 1)) = v[t];
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
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    constexpr int KS0 = (KT0 + 1) / 2;
    u32x4 *w0l = smem, *w1l = w0l + w3_units(4, KS0);
    float *b0l = reinterpret_cast<float *>(w1l + w3_units(NT1, 2)), *b1l = b0l + 64, *w1v = b1l + NT1 * 16;
    stage_w3(w0l, 4, KS0, a.w0);
    if constexpr (NT1 == 1) {   // density MLP: the single output row stays fp32 (a 64-term dot product per row on the VALU)
        for (int i = threadIdx.x; i < 64; i += (int)blockDim.x) w1v[i] = a.w1.w[i * a.w1.sk];
    } else {
        stage_w3(w1l, NT1, 2, a.w1);
    }
    stage_b(b0l, 64, a.b0, a.w0.n);
    stage_b(b1l, NT1 * 16, a.b1, a.w1.n);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w0p = w3_at(w0l, 4, KS0, lane), w1p = w3_at(w1l, NT1, 2, lane);
    f32x4 w1r[NT1 == 1 ? 4 : 1];   // density MLP: this lane's 16 weights of the output row (features 16 p + 4 g + i)
    if constexpr (NT1 == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) w1r[p] = *reinterpret_cast<const f32x4 *>(w1v + 16 * p + 4 * g);
    }
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
            Opd<KS0> xo;
            make_opd<KT0>(x, xo);
            f32x4 h[4];
            init_bias<4>(b0l, g, h);
            tgemm<KS0, 4>(w0p, xo, h);
            relu<4>(h);
            if (a.h1) st_rm<4>(a.h1 + row * 64, ok, g, h);
            if constexpr (NT1 == 1) {
                // 64 -> 1: sixteen fp32 FMAs per lane and a reduction over the four lane groups of the row -- no split, no
                // matrix instruction (the bf16x3 form spent 88 VALU on the split and 12 instructions on a 1-row output)
                float dot = 0.0f;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dot = fmaf(w1r[p][i], h[p][i], dot);
                dot += __shfl_xor(dot, 16, 64);
                dot += __shfl_xor(dot, 32, 64);
                if (ok && g == 0) a.dens[row] = expf(dot + b1l[0] - 1.0f);
            } else {
                Opd<2> ho;
                make_opd<4>(h, ho);
                // 64 output features at a time (16 live accumulators instead of 32)
                f32x4 o[4];
                init_bias<4>(b1l, g, o);
                tgemm<2, 4>(w1p, ho, o);
                st_rm<4>(a.out0 + row * 64, ok, g, o);
                if (a.dens && ok && g == 0) a.dens[row] = expf(o[0][0] - 1.0f);
                if constexpr (NT1 == 8) {
                    init_bias<4>(b1l + 64, g, o);
                    tgemm<2, 4>(w3_tile(w1p, 4), ho, o);
                    st_rm<4>(a.out1 + row * 64, ok, g, o);
                }
            }
        }
    }
}

// Density MLP of a proposal network when its input is narrow (L * F <= 16; the reference's proposal grids are L8 / F1):
// the first layer has only two (.. four) k-steps of the fp32 matrix instruction, which is cheaper than six bf16 products of a
// K = 32 step that is three quarters padding (8 x 32 cycles against 24 x 20 per 16 rows), needs no operand split, and takes the
// encoding in its own layout: lane (m, g) reads features 4 s + g of row m.  W0 / b0 / W1 live in registers.  The launch is
// latency-bound (one wave processes 16 tiles), so a chunk's eight tiles are loaded with all loads in flight (branch-free:
// clamped row and feature; the weight of a padded feature is zero) before any is consumed.
template <int KS4>
__global__ __launch_bounds__(kNThreads, 4) void density_fwd_kernel(const NeckFwdArgs a, int32_t F) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const int32_t K0 = a.w0.k;
    float aw[4][KS4];
    int64_t koff[KS4];  // element offset of feature 4 s + g inside the level-major encoding, without the row
#pragma unroll
    for (int s = 0; s < KS4; ++s) {
        const int32_t k = 4 * s + g, kc = k < K0 ? k : K0 - 1;
        koff[s] = (int64_t)(kc / F) * a.n * F + kc % F;
#pragma unroll
        for (int p = 0; p < 4; ++p) aw[p][s] = k < K0 ? a.w0.w[(int64_t)(16 * p + m) * a.w0.sn + (int64_t)k * a.w0.sk] : 0.0f;
    }
    f32x4 b0r[4], w1r[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b0r[p][i] = a.b0 ? a.b0[16 * p + 4 * g + i] : 0.0f;
            w1r[p][i] = a.w1.w[(int64_t)(16 * p + 4 * g + i) * a.w1.sk];
        }
    const float b1 = a.b1 ? a.b1[0] : 0.0f;
    const int64_t n_tiles = (a.n + 15) >> 4, n_chunks = (n_tiles + kNeckChunk - 1) / kNeckChunk;
    for (int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; c < n_chunks; c += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t t0 = c * kNeckChunk;
        float x[kNeckChunk][KS4];
#pragma unroll
        for (int j = 0; j < kNeckChunk; ++j) {
            const int64_t row = (t0 + j) * 16 + m, rc = row < a.n ? row : a.n - 1;
#pragma unroll
            for (int s = 0; s < KS4; ++s) x[j][s] = a.enc[koff[s] + rc * F];
        }
#pragma unroll
        for (int j = 0; j < kNeckChunk; ++j) {
            if (t0 + j >= n_tiles) break;
            const int64_t row = (t0 + j) * 16 + m;
            const bool ok = row < a.n;
            f32x4 h[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) h[p] = b0r[p];
#pragma unroll
            for (int s = 0; s < KS4; ++s)
#pragma unroll
                for (int p = 0; p < 4; ++p) h[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[p][s], x[j][s], h[p], 0, 0, 0);
            relu<4>(h);
            if (a.h1) st_rm<4>(a.h1 + row * 64, ok, g, h);
            float dot = 0.0f;
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) dot = fmaf(w1r[p][i], h[p][i], dot);
            dot += __shfl_xor(dot, 16, 64);
            dot += __shfl_xor(dot, 32, 64);
            if (ok && g == 0) a.dens[row] = expf(dot + b1 - 1.0f);
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
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    constexpr int KS1 = KT1 / 2;  // k-steps of the transposed second layer (0: density mode, W1 is one fp32 row)
    constexpr int KS1e = KS1 > 0 ? KS1 : 1;
    static_assert(KT1 == 0 || KT1 == 4 || KT1 == 8, "neck_bwd: 0, 64 or 128 gradient columns");
    u32x4 *w0l = smem, *w1l = w0l + w3_units(KT0, 2);
    float *w1v = reinterpret_cast<float *>(w1l);  // density mode: W1[0][0..63]
    stage_w3(w0l, KT0, 2, a.w0t);
    if constexpr (KT1 == 0) {
        for (int i = threadIdx.x; i < 64; i += (int)blockDim.x) w1v[i] = a.w1t.w[i * a.w1t.sn];
    } else {
        stage_w3(w1l, 4, KS1, a.w1t);
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w0p = w3_at(w0l, KT0, 2, lane), w1p = w3_at(w1l, 4, KS1e, lane);
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
                    for (int i = 0; i < 4; ++i) da[p][i] = w1v[16 * p + 4 * g + i] * fix;
            } else {
                // 64 gradient columns (two k-steps) at a time: one operand set live
                zero<4>(da);
                {
                    f32x4 lo[4];
                    if (a.d0) ld_rm<4>(a.d0 + row * 64, ok, g, lo); else zero<4>(lo);
                    if (g == 0) {
                        lo[0][0] += fix;
                        if (a.dcol0 && ok) a.dcol0[row] = lo[0][0];
                    }
                    Opd<2> dop;
                    make_opd<4>(lo, dop);
                    tgemm<2, 4>(w1p, dop, da);
                }
                if constexpr (KT1 == 8) {
                    f32x4 hi[4];
                    ld_rm<4>(a.d1 + row * 64, ok, g, hi);
                    Opd<2> dop;
                    make_opd<4>(hi, dop);
                    tgemm<2, 4>(W3{w1p.p + 2 * 64, w1p.ks, w1p.plane}, dop, da);  // k-steps 2, 3
                }
            }
            relu_mask<4>(da, mk);
            st_rm<4>(a.dpre0 + row * 64, ok, g, da);
            Opd<2> dao;
            make_opd<4>(da, dao);
            f32x4 de[KT0];
            zero<KT0>(de);
            tgemm<2, KT0>(w0p, dao, de);
            st_lm<KT0, F>(a.denc, a.n, a.n_levels, row, ok, g, de);
        }
    }
}

// ------------------------------------------------------------------------ neck backward with fused weight gradients
// [r3]  dW = dPre^T X needs the ROWS on the reduction index of the matrix core, while the transposed chain keeps a row on
// a lane.  Two facts make the fusion cheap once the matrix pipe is 16x faster than fp32:
//  * The matrix core is its own transposer.  An operand X in chain layout (lane (m, g): eight features of row m) times a
//    0/1 selection matrix E_p -- B[k][j] = [feature(k) == 16 p + j] -- gives D[i = row][j] = X[row][16 p + j] in the
//    accumulator layout: lane (j, g) holds rows 4 g .. 4 g + 3 of feature 16 p + j.  One instruction per bf16 term; the
//    result is exact (one product per output), so the three terms re-pack into bf16 without another split.
//  * That layout IS the operand layout of v_mfma_f32_16x16x16_bf16 with the reduction over a tile's 16 rows (lane (i, g)
//    supplies k = 4 g .. 4 g + 3), for A (dPre: feature i) and B (X: feature j) alike.
// The 16x16 tiles of dW (64 x 64 and 64 x K0: 24 tiles = 96 registers) stay in the wave's accumulators for the whole
// launch -- two waves per SIMD instead of four; an LDS copy per workgroup with ds_add_f32 was tried first and is
// hopeless on this chip (0.37 lane-adds per clock per CU, DESIGN 4.1: 2 ms for this kernel).  The waves of a workgroup are
// summed through LDS once at the end and the per-workgroup partials by linear_dw_reduce_kernel.  dPre never goes to HBM
// and h1 / enc / d are read once: the separate weight-gradient pass of the neck (two launches, 0.94 GB) is gone.
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;
struct SwT { u32x2 h, m, l; };  // one 16-row x 16-feature tile, rows on the reduction index, three bf16 terms

struct SelE { unsigned a0, a1; };  // the two non-zero registers of the selection fragments (built once per lane)
__device__ __forceinline__ SelE make_sel(int lane) {
    const bool on = (lane >> 4) == ((lane & 15) >> 2);
    const unsigned one = on ? (0x3F80u << (16 * (lane & 1))) : 0u;   // bf16 1.0 in the half this lane's feature sits in
    return SelE{((lane & 3) >> 1) == 0 ? one : 0u, ((lane & 3) >> 1) == 1 ? one : 0u};
}
// feature tile p of the operand `o` (k-step p >> 1, half p & 1) -> swapped layout; colsum += the lane's four rows (fp32)
template <int KS>
__device__ __forceinline__ SwT to_rows(const Opd<KS> &o, int p, const SelE e, float *colsum = nullptr) {
    const u32x4 sel = (p & 1) ? u32x4{0u, 0u, e.a0, e.a1} : u32x4{e.a0, e.a1, 0u, 0u};
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4 th = EMER_MF(o.h[p >> 1], sel, z), tm = EMER_MF(o.m[p >> 1], sel, z), tl = EMER_MF(o.l[p >> 1], sel, z);
    SwT t;
    t.h = u32x2{pk_bf16(th[0], th[1]), pk_bf16(th[2], th[3])};
    t.m = u32x2{pk_bf16(tm[0], tm[1]), pk_bf16(tm[2], tm[3])};
    t.l = u32x2{pk_bf16(tl[0], tl[1]), pk_bf16(tl[2], tl[3])};
    if (colsum) *colsum += ((tl[0] + tl[1]) + (tl[2] + tl[3])) + ((tm[0] + tm[1]) + (tm[2] + tm[3])) + ((th[0] + th[1]) + (th[2] + th[3]));
    return t;
}
#define EMER_MF16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, A), __builtin_bit_cast(s16x4, B), C, 0, 0, 0)
// acc[p][b] (lane (j, g), register r: dW[16 p + 4 g + r][16 b + j]) += sum over the tile's rows of a[p][row][.] * bt[b][row][.]
// Products outermost: consecutive instructions hit different accumulators.
template <int NA, int NB>
__device__ __forceinline__ void dw_tiles(f32x4 (&acc)[NA][NB], const SwT (&a)[NA], const SwT (&bt)[NB]) {
#define EMER_DW_PASS(X, Y)                                                                   \
    _Pragma("unroll") for (int p = 0; p < NA; ++p)                                           \
        _Pragma("unroll") for (int b = 0; b < NB; ++b) acc[p][b] = EMER_MF16(a[p].X, bt[b].Y, acc[p][b]);
    EMER_DW_PASS(l, h) EMER_DW_PASS(h, l) EMER_DW_PASS(m, m) EMER_DW_PASS(m, h) EMER_DW_PASS(h, m) EMER_DW_PASS(h, h)
#undef EMER_DW_PASS
}

struct SwP { u32x4 h, m, l; };  // eight reduction-index entries per lane and term: the operand of a K = 32 (16 x 16) or K = 16 (32 x 32) dW step
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define EMER_MF32(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)
// two 16-feature tiles (rows on the reduction index) -> the operand of their 32-feature block
__device__ __forceinline__ SwP block32(const SwT &ta, const SwT &tb) {
    SwP o;
#define EMER_SWAP(T)                                                                      \
    {                                                                                     \
        const auto r0 = __builtin_amdgcn_permlane16_swap(ta.T[0], tb.T[0], false, false); \
        const auto r1 = __builtin_amdgcn_permlane16_swap(ta.T[1], tb.T[1], false, false); \
        o.T = u32x4{r0[0], r1[0], r0[1], r1[1]};                                          \
    }
    EMER_SWAP(h) EMER_SWAP(m) EMER_SWAP(l)
#undef EMER_SWAP
    return o;
}
// acc[q] += A^T B[q] over the tile's 16 rows for NQ 32-feature blocks of B: six partial products, NQ independent accumulators per term
template <int NQ>
__device__ __forceinline__ void dw_blocks(f32x16 (&acc)[NQ], const SwP &a, const SwP (&b)[NQ]) {
#define EMER_DWB(X, Y) _Pragma("unroll") for (int q = 0; q < NQ; ++q) acc[q] = EMER_MF32(a.X, b[q].Y, acc[q]);
    EMER_DWB(l, h) EMER_DWB(h, l) EMER_DWB(m, m) EMER_DWB(m, h) EMER_DWB(h, m) EMER_DWB(h, h)
#undef EMER_DWB
}

constexpr int neckw_threads(int kt0, int no = 1) { return no == 2 ? 256 : 512; }
// (512 threads = 8 waves, ONE workgroup per CU = 2 waves per SIMD, <= 256 registers; the weights, 48-72 KB, are staged once per CU and
// leave room for 7-10 KB of per-wave staging)

struct NeckBwdWArgs {
    const float *d0;     // [n][64] gradient of output features 0..63 (null: zero)
    const float *d1;     // [n][64] gradient of output features 64..127 (128-output necks; null: zero)
    const float *ddens;  // [n] gradient of the density (null: none)
    const float *dens;   // [n] saved density
    const float *enc;    // level-major [L][n][F]: the forward's input (operand of dW0; the hidden layer is recomputed from it)
    const float *b0;     // [64] bias of the first layer
    int64_t n; int32_t n_levels, k0;
    WSrc w1t, w0t, w0;   // W1^T (64 x 64, or 64 x 128 for the 128-output neck), W0^T (K0 x 64), W0 (64 x K0, for the recomputation)
    float *denc;         // level-major [L][n][F]
    float *partials;     // [gridDim.x][stride]: dW1 [64 NO][64] | db1 [64 NO] | dW0 [64][k0] | db0 [64]
    int64_t stride;
};

// NO = 1: 64 outputs (geometry features).  NO = 2 [r5]: the 128-output neck of the feature configs (geometry | semantic features): dW1 [128][64]
// as eight 32 x 32 blocks -- 192 accumulator registers with dW0, so four waves, one per SIMD (neckw_threads).
template <int KT0, int F, int NO = 1>
__global__ __launch_bounds__((neckw_threads(KT0, NO)), (neckw_threads(KT0, NO) == 256 ? 1 : 2)) void neck_bwdw_kernel(const NeckBwdWArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    constexpr int K0P = 16 * KT0, KS0 = (KT0 + 1) / 2;
    u32x4 *w0l = smem, *w1l = w0l + w3_units(KT0, 2), *w0fl = w1l + w3_units(4, 2 * NO);
    float *b0l = reinterpret_cast<float *>(w0fl + w3_units(4, KS0));
    stage_w3(w0l, KT0, 2, a.w0t);
    stage_w3(w1l, 4, 2 * NO, a.w1t);
    stage_w3(w0fl, 4, KS0, a.w0);
    stage_b(b0l, 64, a.b0, 64);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w0p = w3_at(w0l, KT0, 2, lane), w1p = w3_at(w1l, 4, 2 * NO, lane), w0fp = w3_at(w0fl, 4, KS0, lane);
    const SelE sel = make_sel(lane);
    // [r4] dW1 [64][64] and dW0 [64][K0P] as 32 x 32 blocks on v_mfma_f32_32x32x16_bf16 (one instruction per block and partial product
    // where the 16 x 16 x 16 shape needed four at the same issue cost each; operands: two transposer outputs joined by v_permlane16_swap,
    // see block32): lane (j, h), register r = dW[32 P + 8 (r >> 2) + 4 h + (r & 3)][32 Q + j]
    constexpr int QB = (KT0 + 1) / 2;   // 32-feature blocks of the encoding (the last one half empty when KT0 is odd)
    f32x16 bw1[2 * NO][2], bw0[2][QB];
    float ab1[4 * NO], ab0[4];      // bias gradients: this lane's rows 4 g .. 4 g + 3 of feature 16 p + m
#pragma unroll
    for (int P = 0; P < 2 * NO; ++P)
#pragma unroll
        for (int Q = 0; Q < 2; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) bw1[P][Q][r] = 0.0f;
#pragma unroll
    for (int P = 0; P < 2; ++P)
#pragma unroll
        for (int Q = 0; Q < QB; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) bw0[P][Q][r] = 0.0f;
#pragma unroll
    for (int p = 0; p < 4 * NO; ++p) ab1[p] = 0.0f;
#pragma unroll
    for (int p = 0; p < 4; ++p) ab0[p] = 0.0f;
    // Each wave owns a contiguous range of 16-row tiles.  With two waves per SIMD nothing else hides the HBM latency, so the
    // inputs of tile t + 1 are in flight while tile t is in the matrix pipe -- WITHOUT holding them in registers (the
    // accumulators leave none: a register prefetch was spilled to scratch by the compiler, i.e. loaded, waited for and
    // stored again): d0 goes global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave instruction, lane-major, so
    // each lane reads back its own 16 bytes conflict-free); only the small enc tile and the density scalars ride in
    // registers.  Loads are unconditional (rows past the end re-read row n - 1 and are zeroed at use).
    // The hidden layer is NOT read back: h1 = relu(W0 x + b0) is recomputed from the enc tile this kernel needs anyway
    // (24-48 instructions on the matrix pipe instead of 268 MB written by the forward and read here; bitwise the forward's
    // values -- same fragments, same order).
    const int64_t n_tiles = (a.n + 15) >> 4, n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t per_wave = (n_tiles + n_waves - 1) / n_waves;
    const int64_t t_begin = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * per_wave;
    const int64_t t_end = t_begin + per_wave < n_tiles ? t_begin + per_wave : n_tiles;
    float *stg = b0l + 64 + wave * (1024 * NO + 384 * KT0);   // per wave: d0 (| d1) tile [4 NO][64][4] | parked enc operand tiles [KT0][3][64][2]
    u32x2 *park = reinterpret_cast<u32x2 *>(stg + 1024 * NO) + lane;
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    f32x4 xn[KT0];
    float ddn = 0.0f, den = 0.0f;
    auto issue = [&](int64_t tile) {   // addressing: the tile is wave-uniform -> one scalar base per tensor, 32-bit lane offsets
        const int64_t row0 = tile * 16;
        const unsigned mrow = row0 + m < a.n ? (unsigned)m : (unsigned)(a.n - 1 - row0);
        const unsigned o64 = mrow * 64u + 4u * g;
        if (a.d0) {
            const float *d0 = a.d0 + row0 * 64;
#pragma unroll
            for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((gptr)(d0 + (o64 + 16u * p)), (lptr)(stg + 256 * p), 16, 0, 0);
        }
        if (NO == 2 && a.d1) {
            const float *d1 = a.d1 + row0 * 64;
#pragma unroll
            for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((gptr)(d1 + (o64 + 16u * p)), (lptr)(stg + 1024 + 256 * p), 16, 0, 0);
        }
        ld_lm_t<KT0, F>(a.enc + row0 * F, (unsigned)a.n, a.n_levels, mrow, g, xn);
        ddn = a.ddens ? (a.ddens + row0)[mrow] : 0.0f;
        den = a.ddens ? (a.dens + row0)[mrow] : 0.0f;
    };
    struct Raw { f32x4 d[4 * NO], x[KT0]; float dd, de; };
    if (t_begin < t_end) issue(t_begin);
    // [r4] denc of a tile is stored at the START of the next tile, behind that tile's wait: stores count in vmcnt like loads here, so a
    // store issued mid-tile would still be in flight at the next `s_waitcnt vmcnt(0)` and the wave would sit out its acknowledgement
    f32x4 dep[KT0];
    int64_t tile_prev = -1;
    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        Raw cur;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's inputs have landed (issued one tile ago)
        if (tile_prev >= 0) st_lm_t<KT0, F>(a.denc + tile_prev * 16 * F, (unsigned)a.n, a.n_levels, (unsigned)m, tile_prev * 16 + m < a.n, g, dep);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            cur.d[p] = a.d0 ? *reinterpret_cast<const f32x4 *>(stg + 256 * p + 4 * lane) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (NO == 2) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                cur.d[4 + p] = a.d1 ? *reinterpret_cast<const f32x4 *>(stg + 1024 + 256 * p + 4 * lane) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int b = 0; b < KT0; ++b) cur.x[b] = xn[b];
        cur.dd = ddn; cur.de = den;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging buffer has been read: it may be overwritten
        issue(tile + 1 < t_end ? tile + 1 : tile);
        {
            const int64_t row = tile * 16 + m;
            const bool ok = row < a.n;
            const float fix = ok ? cur.dd * fminf(cur.de, 3269017.3724721107f) : 0.0f;
            // enc tile -> operand of dW0 (xs) and, through the first layer, h1: operand of dW1 (hs) and relu'(h1) as 16 bits
            SwP hq[2];   // h1 features 0-31, 32-63 with the rows on the reduction index
            unsigned relu_bits = 0u;
            {
#pragma unroll
                for (int b = 0; b < KT0; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur.x[b][i] = ok ? cur.x[b][i] : 0.0f;
                Opd<KS0> xo;
                make_opd<KT0>(cur.x, xo);
                f32x4 h[4];
                init_bias<4>(b0l, g, h);
                tgemm<KS0, 4, false>(w0fp, xo, h);
#pragma unroll
                for (int b = 0; b < KT0; ++b) {   // operand of dW0, needed at the end of the tile: parked in LDS, not in 6 KT0 registers
                    const SwT t = to_rows<KS0>(xo, b, sel);
                    park[(3 * b + 0) * 64] = t.h; park[(3 * b + 1) * 64] = t.m; park[(3 * b + 2) * 64] = t.l;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h[p][i] = (ok && h[p][i] > 0.0f) ? h[p][i] : 0.0f;   // relu; rows past the end contribute nothing
                        relu_bits |= (h[p][i] > 0.0f ? 1u : 0u) << (4 * p + i);
                    }
                Opd<2> ho;
                make_opd<4>(h, ho);
#pragma unroll
                for (int q = 0; q < 2; ++q) hq[q] = block32(to_rows<2>(ho, 2 * q, sel), to_rows<2>(ho, 2 * q + 1, sel));
            }
            f32x4 da[4];
            zero<4>(da);
            {
#pragma unroll
                for (int p = 0; p < 4 * NO; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur.d[p][i] = ok ? cur.d[p][i] : 0.0f;
                if (g == 0) cur.d[0][0] += fix;
                Opd<2 * NO> dop;
                make_opd<4 * NO>(cur.d, dop);
                tgemm<2 * NO, 4, false>(w1p, dop, da);
#pragma unroll
                for (int P = 0; P < 2 * NO; ++P) {   // dW1 rows 32 P .. += d^T h1
                    const SwT t0 = to_rows<2 * NO>(dop, 2 * P, sel, &ab1[2 * P]), t1 = to_rows<2 * NO>(dop, 2 * P + 1, sel, &ab1[2 * P + 1]);
                    dw_blocks<2>(bw1[P], block32(t0, t1), hq);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) da[p][i] = ((relu_bits >> (4 * p + i)) & 1u) ? da[p][i] : 0.0f;
            Opd<2> dao;
            make_opd<4>(da, dao);
            f32x4 de[KT0];
            zero<KT0>(de);
            tgemm<2, KT0, false>(w0p, dao, de);
#pragma unroll
            for (int b = 0; b < KT0; ++b) dep[b] = de[b];
            tile_prev = tile;
            {   // dW0 += dPre0^T enc
                SwP xq[QB];
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    SwT xa, xb;
                    xa.h = park[(3 * (2 * q) + 0) * 64]; xa.m = park[(3 * (2 * q) + 1) * 64]; xa.l = park[(3 * (2 * q) + 2) * 64];
                    if (2 * q + 1 < KT0) { xb.h = park[(3 * (2 * q + 1) + 0) * 64]; xb.m = park[(3 * (2 * q + 1) + 1) * 64]; xb.l = park[(3 * (2 * q + 1) + 2) * 64]; }
                    else { xb.h = u32x2{0u, 0u}; xb.m = u32x2{0u, 0u}; xb.l = u32x2{0u, 0u}; }
                    xq[q] = block32(xa, xb);
                }
#pragma unroll
                for (int P = 0; P < 2; ++P) {
                    const SwT t0 = to_rows<2>(dao, 2 * P, sel, &ab0[2 * P]), t1 = to_rows<2>(dao, 2 * P + 1, sel, &ab0[2 * P + 1]);
                    dw_blocks<QB>(bw0[P], block32(t0, t1), xq);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (tile_prev >= 0) st_lm_t<KT0, F>(a.denc + tile_prev * 16 * F, (unsigned)a.n, a.n_levels, (unsigned)m, tile_prev * 16 + m < a.n, g, dep);
    // ---- sum the waves through LDS (the weights are dead: the buffer takes their place), one coalesced partial per workgroup
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem);   // dW1 [64 NO][64] | db1 [64 NO] | dW0 [64][K0P] | db0 [64]
    float *r1 = red, *rb1 = r1 + 64 * NO * 64, *r0 = rb1 + 64 * NO, *rb0 = r0 + 64 * K0P;
#pragma unroll
    for (int p = 0; p < 4 * NO; ++p) { ab1[p] += __shfl_xor(ab1[p], 16, 64); ab1[p] += __shfl_xor(ab1[p], 32, 64); }   // bias: every lane holds the column sum
#pragma unroll
    for (int p = 0; p < 4; ++p) { ab0[p] += __shfl_xor(ab0[p], 16, 64); ab0[p] += __shfl_xor(ab0[p], 32, 64); }
    const int j32 = lane & 31, h32 = lane >> 5;
    for (int w = 0; w < neckw_threads(KT0, NO) / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int P = 0; P < 2 * NO; ++P)
#pragma unroll
                for (int Q = 0; Q < 2; ++Q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *q = r1 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * 64 + 32 * Q + j32;
                        *q = (w == 0) ? bw1[P][Q][r] : *q + bw1[P][Q][r];
                    }
#pragma unroll
            for (int P = 0; P < 2; ++P) {
#pragma unroll
                for (int Q = 0; Q < QB; ++Q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (32 * Q + j32 < K0P) {   // (the upper half of an odd last block has no column in the K0P-wide buffer)
                            float *q = r0 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * K0P + 32 * Q + j32;
                            *q = (w == 0) ? bw0[P][Q][r] : *q + bw0[P][Q][r];
                        }
                    }
            }
            if (g == 0) {
#pragma unroll
                for (int p = 0; p < 4 * NO; ++p) rb1[16 * p + m] = (w == 0) ? ab1[p] : rb1[16 * p + m] + ab1[p];
#pragma unroll
                for (int p = 0; p < 4; ++p) rb0[16 * p + m] = (w == 0) ? ab0[p] : rb0[16 * p + m] + ab0[p];
            }
        }
        __syncthreads();
    }
    float *part = a.partials + (int64_t)blockIdx.x * a.stride;
    for (int i = threadIdx.x; i < 64 * NO * 65; i += (int)blockDim.x) part[i] = red[i];
    float *part0 = part + 64 * NO * 65;
    for (int i = threadIdx.x; i < 64 * a.k0; i += (int)blockDim.x) { const int nn = i / a.k0, kk = i - nn * a.k0; part0[i] = r0[nn * K0P + kk]; }
    for (int i = threadIdx.x; i < 64; i += (int)blockDim.x) part0[64 * a.k0 + i] = rb0[i];
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
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    u32x4 *w0l = smem, *w1al = w0l + w3_units(4, 2), *w1gl = w1al + w3_units(4, 2), *w2l = w1gl + w3_units(4, 2);
    float *b2l = reinterpret_cast<float *>(w2l + w3_units(1, 2));
    stage_w3(w0l, 4, 2, a.w0g);
    stage_w3(w1al, 4, 2, a.w1a);
    stage_w3(w1gl, 4, 2, a.w1g);
    stage_w3(w2l, 1, 2, a.w2);
    stage_b(b2l, 16, a.b2, a.w2.n);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w0p = w3_at(w0l, 4, 2, lane), w1ap = w3_at(w1al, 4, 2, lane), w1gp = w3_at(w1gl, 4, 2, lane), w2p = w3_at(w2l, 1, 2, lane);
    const int tpr = a.tiles_per_ray;
    // Addressing: the ray is wave-uniform, so every tensor gets ONE scalar base per ray and the lanes share 32-bit
    // offsets (SGPR base + VGPR offset loads / stores) instead of a 64-bit pointer pair per tensor.
    const unsigned lo64 = (unsigned)(m * 64 + 4 * g), log = (unsigned)m * (unsigned)a.ld_geo + 4u * g;
    for (int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; ray < a.n_rays; ray += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        // the per-ray pre-activations are re-read for every tile (L1/L2 hits) instead of living in 32 VGPRs for the
        // whole ray: the kernel then fits 128 VGPRs = 4 waves per SIMD
        const float *rb0 = a.rb0 + ray * a.ld_rb + 4 * g, *rb1 = a.rb1 + ray * a.ld_rb + 4 * g;
        const int64_t row0 = ray * tpr * 16;
        const float *geo = a.geo + row0 * a.ld_geo;
        float *a1 = a.a1 + row0 * 64, *a2 = a.a2 + row0 * 64, *outp = a.out + row0 * 3;
        const bool keep = a.a1 != nullptr, keep2 = a.a2 != nullptr;  // inference: the hidden activations are not stored (2/3 of the kernel's bytes)
        for (int j = 0; j < tpr; ++j) {
            f32x4 x[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) x[p] = *reinterpret_cast<const f32x4 *>(geo + (log + (unsigned)j * 16u * (unsigned)a.ld_geo + 16u * p));
            Opd<2> xo;
            make_opd<4>(x, xo);
            f32x4 h[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) h[p] = *reinterpret_cast<const f32x4 *>(rb0 + 16 * p);
            tgemm<2, 4, false>(w0p, xo, h);
            relu<4>(h);
            if (keep)
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(a1 + (lo64 + (unsigned)j * 1024u + 16u * p)) = h[p];
            f32x4 h2[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) h2[p] = *reinterpret_cast<const f32x4 *>(rb1 + 16 * p);
            tgemm<2, 4, false>(w1gp, xo, h2);   // geo part first: its operand dies here
            Opd<2> ho;
            make_opd<4>(h, ho);
            tgemm<2, 4, false>(w1ap, ho, h2);
            relu<4>(h2);
            if (keep2)
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(a2 + (lo64 + (unsigned)j * 1024u + 16u * p)) = h2[p];
            make_opd<4>(h2, ho);
            f32x4 o[1];
            init_bias<1>(b2l, g, o);
            tgemm<2, 1>(w2p, ho, o);
            if (g == 0) {
                float *op = outp + ((unsigned)j * 48u + 3u * m);
#pragma unroll
                for (int i = 0; i < 3; ++i) op[i] = 1.0f / (1.0f + expf(-o[0][i]));
            }
        }
    }
}

// ------------------------------------------------------------------------------ neck + rgb head in one forward kernel
// The static field of a colour query: level-major grid encoding -> neck (geometry features, density) -> rgb head, a wave
// keeping its 16 rows in registers from the encoding to the colour.  Same arithmetic, instruction for instruction, as
// neck_fwd_kernel followed by rgb_fwd_kernel; what it saves is the rgb head's read of the 256 B / sample geometry features
// (they are still WRITTEN once: the backward needs them) and a launch.  The weights of both stages (114-126 KB) leave room for
// one workgroup per CU, so the workgroup is 1024 lanes = the same four waves per SIMD as the separate kernels.
constexpr int kFieldThreads = 1024;
struct FieldFwdArgs {
    const float *enc; int64_t n; int32_t n_levels;   // [L][n][F], n = n_rays * samples_per_ray
    WSrc nw0, nw1; const float *nb0, *nb1;           // neck: W0 [64][L F], W1 [64][64]
    float *geo;                                      // [n][64]
    float *dens;                                     // [n] exp(feature 0 - 1)
    RgbFwdArgs r;                                    // the rgb head (r.geo / r.ld_geo unused)
};

template <int KT0, int F>
__global__ __launch_bounds__(kFieldThreads) void field_fwd_kernel(const FieldFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    constexpr int KS0 = (KT0 + 1) / 2;
    u32x4 *nw0l = smem, *nw1l = nw0l + w3_units(4, KS0), *w0l = nw1l + w3_units(4, 2), *w1al = w0l + w3_units(4, 2),
          *w1gl = w1al + w3_units(4, 2), *w2l = w1gl + w3_units(4, 2);
    float *nb0l = reinterpret_cast<float *>(w2l + w3_units(1, 2)), *nb1l = nb0l + 64, *b2l = nb1l + 64;
    stage_w3(nw0l, 4, KS0, a.nw0);
    stage_w3(nw1l, 4, 2, a.nw1);
    stage_w3(w0l, 4, 2, a.r.w0g);
    stage_w3(w1al, 4, 2, a.r.w1a);
    stage_w3(w1gl, 4, 2, a.r.w1g);
    stage_w3(w2l, 1, 2, a.r.w2);
    stage_b(nb0l, 64, a.nb0, a.nw0.n);
    stage_b(nb1l, 64, a.nb1, a.nw1.n);
    stage_b(b2l, 16, a.r.b2, a.r.w2.n);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 nw0p = w3_at(nw0l, 4, KS0, lane), nw1p = w3_at(nw1l, 4, 2, lane);
    const W3 w0p = w3_at(w0l, 4, 2, lane), w1ap = w3_at(w1al, 4, 2, lane), w1gp = w3_at(w1gl, 4, 2, lane), w2p = w3_at(w2l, 1, 2, lane);
    const int tpr = a.r.tiles_per_ray;
    const unsigned lo64 = (unsigned)(m * 64 + 4 * g);
    const bool keep = a.r.a1 != nullptr, keep2 = a.r.a2 != nullptr;   // [r6] a2 (or both) may be left to a recomputing backward
    for (int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; ray < a.r.n_rays; ray += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const float *rb0 = a.r.rb0 + ray * a.r.ld_rb + 4 * g, *rb1 = a.r.rb1 + ray * a.r.ld_rb + 4 * g;
        const int64_t row0 = ray * tpr * 16;   // wave-uniform: one scalar base per tensor, 32-bit lane offsets
        const float *encp = a.enc + row0 * F;
        float *geo = a.geo + row0 * 64, *dens = a.dens + row0, *a1 = a.r.a1 + row0 * 64, *a2 = a.r.a2 + row0 * 64, *outp = a.r.out + row0 * 3;
        for (int j = 0; j < tpr; ++j) {
            f32x4 x[4];
            {   // neck: enc -> relu(W0 . + b0) -> W1 . + b1 = geometry features
                f32x4 e[KT0];
                ld_lm_t<KT0, F>(encp + (unsigned)j * 16u * (unsigned)F, (unsigned)a.n, a.n_levels, (unsigned)m, g, e);
                Opd<KS0> eo;
                make_opd<KT0>(e, eo);
                f32x4 hn[4];
                init_bias<4>(nb0l, g, hn);
                tgemm<KS0, 4>(nw0p, eo, hn);
                relu<4>(hn);
                Opd<2> hno;
                make_opd<4>(hn, hno);
                init_bias<4>(nb1l, g, x);
                tgemm<2, 4>(nw1p, hno, x);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(geo + (lo64 + (unsigned)j * 1024u + 16u * p)) = x[p];
            if (g == 0) dens[(unsigned)j * 16u + (unsigned)m] = expf(x[0][0] - 1.0f);
            // rgb head (rgb_fwd_kernel's tile body)
            Opd<2> xo;
            make_opd<4>(x, xo);
            f32x4 h[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) h[p] = *reinterpret_cast<const f32x4 *>(rb0 + 16 * p);
            tgemm<2, 4, false>(w0p, xo, h);
            relu<4>(h);
            if (keep)
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(a1 + (lo64 + (unsigned)j * 1024u + 16u * p)) = h[p];
            f32x4 h2[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) h2[p] = *reinterpret_cast<const f32x4 *>(rb1 + 16 * p);
            tgemm<2, 4, false>(w1gp, xo, h2);
            Opd<2> ho;
            make_opd<4>(h, ho);
            tgemm<2, 4, false>(w1ap, ho, h2);
            relu<4>(h2);
            if (keep2)
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(a2 + (lo64 + (unsigned)j * 1024u + 16u * p)) = h2[p];
            make_opd<4>(h2, ho);
            f32x4 o[1];
            init_bias<1>(b2l, g, o);
            tgemm<2, 1>(w2p, ho, o);
            if (g == 0) {
                float *op = outp + ((unsigned)j * 48u + 3u * m);
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
    float *w2part;                     // [workgroups][196] partial (dW2 [3][64] | db2 [3]) per workgroup, or null: not computed here
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes that share g (a DPP row); every lane gets it
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror: quad q <-> quad q ^ 1
    v += dpp_mov<0x140>(v);  // row_mirror: lower eight <-> upper eight
    return v;
}

// Sum over the 16 lanes of a DPP row of SIXTEEN quantities at once, lane m ending with the total of quantity m ("reduce-scatter":
// 8 + 4 + 2 + 1 exchange steps instead of 16 x 4).  Partners: m ^ 8 (row_ror:8), m ^ 7 (row_half_mirror), m ^ 2, m ^ 1 (quad
// permutes); at each step a lane keeps the half of its quantities whose index bit equals its own lane bit and sends the other.
// The quantities are scale * e[p][i] (index 4 p + i), formed inside the first step so that no 16-entry product array is ever live.
__device__ __forceinline__ float row16_reduce_scatter(const f32x4 (&e)[4], float scale, int m) {
    const bool b3 = m & 8, b2 = m & 4, b1 = m & 2, b0 = m & 1;
    float r1[8], r2[4], r3[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float lo = e[j >> 2][j & 3], hi = e[2 + (j >> 2)][j & 3];  // quantities j and 8 + j
        r1[j] = scale * (b3 ? hi : lo) + dpp_mov<0x128>(scale * (b3 ? lo : hi));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) r2[j] = (b2 ? r1[4 + j] : r1[j]) + dpp_mov<0x141>(b2 ? r1[j] : r1[4 + j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) r3[j] = (b1 ? r2[2 + j] : r2[j]) + dpp_mov<0x4E>(b1 ? r2[j] : r2[2 + j]);
    return (b0 ? r3[1] : r3[0]) + dpp_mov<0xB1>(b0 ? r3[0] : r3[1]);
}

__global__ __launch_bounds__(kNThreads, 4) void rgb_bwd_kernel(const RgbBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    u32x4 *w1al = smem, *w1gl = w1al + w3_units(4, 2), *w0l = w1gl + w3_units(4, 2);
    stage_w3(w1al, 4, 2, a.w1at);
    stage_w3(w1gl, 4, 2, a.w1gt);
    stage_w3(w0l, 4, 2, a.w0gt);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w1ap = w3_at(w1al, 4, 2, lane), w1gp = w3_at(w1gl, 4, 2, lane), w0p = w3_at(w0l, 4, 2, lane);
    // W2^T (64 x 3) stays in four registers: the K = 4 step of v_mfma_f32_16x16x4_f32 covers the three colour channels
    // exactly (lane (n, g) supplies W2[g][16 p + n]), so the first backward layer is four exact-fp32 instructions per
    // tile and costs no LDS.
    float w2a[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) w2a[p] = (g < a.w2t.k && 16 * p + m < a.w2t.n) ? a.w2t.w[(16 * p + m) * a.w2t.sn + g * a.w2t.sk] : 0.0f;
    const int tpr = a.tiles_per_ray;
    const unsigned lo64 = (unsigned)(m * 64 + 4 * g), lo3 = (unsigned)(3 * m + g);
    // dW2 / db2 ride along (a.w2part): lane (m, g) owns dW2[c][16 (m >> 2) + 4 g + (m & 3)] for the three channels c and, for
    // g < 3, its rows' share of db2[g] -- three + one accumulators for the whole kernel instead of a 280 MB pass over a2.
    float w2acc[3] = {0.0f, 0.0f, 0.0f}, b2acc = 0.0f;
    for (int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; ray < a.n_rays; ray += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        f32x4 s1[4], s0[4];
        zero<4>(s1); zero<4>(s0);
        const int64_t row0 = ray * tpr * 16;   // wave-uniform: one scalar base per tensor, 32-bit lane offsets
        const float *a1 = a.a1 + row0 * 64, *a2 = a.a2 + row0 * 64, *outp = a.out + row0 * 3, *doutp = a.dout + row0 * 3;
        float *dpre2 = a.dpre2 + row0 * 3, *dpre1 = a.dpre1 + row0 * 64, *dpre0 = a.dpre0 + row0 * 64, *dgeo = a.dgeo + row0 * 64;
        for (int j = 0; j < tpr; ++j) {
            const unsigned o64 = lo64 + (unsigned)j * 1024u, o3 = lo3 + (unsigned)j * 48u;
            float d2 = 0.0f;  // lane (m, g): channel g of row m
            if (g < 3) {
                const float y = outp[o3];
                d2 = doutp[o3] * y * (1.0f - y);  // sigmoid'
                dpre2[o3] = d2;
            }
            f32x4 m2[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) m2[p] = *reinterpret_cast<const f32x4 *>(a2 + (o64 + 16u * p));
            if (a.w2part) {  // wave-uniform; before d1 exists: only the mask, the ray sums and d2 are live here
                b2acc += d2;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float dc = __shfl(d2, 16 * c + m, 64);  // channel c of this lane's row
                    w2acc[c] += row16_reduce_scatter(m2, dc, m);
                    __builtin_amdgcn_sched_barrier(0);  // one channel at a time
                }
            }
            f32x4 d1[4];
            zero<4>(d1);
#pragma unroll
            for (int p = 0; p < 4; ++p) d1[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[p], d2, d1[p], 0, 0, 0);
            relu_mask<4>(d1, m2);
#pragma unroll
            for (int p = 0; p < 4; ++p) { *reinterpret_cast<f32x4 *>(dpre1 + (o64 + 16u * p)) = d1[p]; s1[p] += d1[p]; }
            Opd<2> d1o;
            make_opd<4>(d1, d1o);
            f32x4 d0[4];
            zero<4>(d0);
            tgemm<2, 4, false>(w1ap, d1o, d0);
            {
                f32x4 m1[4];  // loaded behind tgemm's scheduling barrier: the mask is not live across the GEMM (other waves hide the latency)
#pragma unroll
                for (int p = 0; p < 4; ++p) m1[p] = *reinterpret_cast<const f32x4 *>(a1 + (o64 + 16u * p));
                relu_mask<4>(d0, m1);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) { *reinterpret_cast<f32x4 *>(dpre0 + (o64 + 16u * p)) = d0[p]; s0[p] += d0[p]; }
            f32x4 dg[4];
            zero<4>(dg);
            tgemm<2, 4, false>(w1gp, d1o, dg);
            make_opd<4>(d0, d1o);
            tgemm<2, 4, false>(w0p, d1o, dg);
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(dgeo + (o64 + 16u * p)) = dg[p];
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
    if (a.w2part) {  // the eight waves' sums through LDS (the weights are dead), one 196-float partial per workgroup
        __syncthreads();
        float *part = reinterpret_cast<float *>(smem);
        const int nw = (int)(blockDim.x >> 6);
#pragma unroll
        for (int c = 0; c < 3; ++c) part[wave * 196 + c * 64 + 16 * (m >> 2) + 4 * g + (m & 3)] = w2acc[c];
        b2acc = row16_sum(b2acc);
        if (m == 0 && g < 3) part[wave * 196 + 192 + g] = b2acc;
        __syncthreads();
        if ((int)threadIdx.x < 195) {
            float t = 0.0f;
            for (int w = 0; w < nw; ++w) t += part[w * 196 + threadIdx.x];
            a.w2part[(int64_t)blockIdx.x * 196 + threadIdx.x] = t;
        }
    }
}

// ----------------------------------------------------------------- rgb backward with the layer-0 / layer-1 weight gradients [r4]
// rgb_bwd_kernel's data-gradient chain INCLUDING dW1[:, a1 | geo] = dpre1^T [a1 | geo] and dW0[:, geo] = dpre0^T geo: dpre1 and dpre0 never
// reach memory (the round-3 step wrote 512 B per sample of them and streamed 1280 B per sample back through two weight-gradient
// launches: 1.9 of the step's 8.4 GB).  What made this "not feasible" at four waves per SIMD is the accumulator count -- 48 tiles of
// 16 x 16 = 192 registers -- so the kernel runs ONE wave per SIMD (256-lane workgroups, one per CU, up to 512 registers per lane) and
// does by hand what the other waves of a SIMD do for the four-wave kernels:
//   * every input of the next tile is in flight while this one is in the matrix pipe: global -> LDS directly (global_load_lds_dwordx4,
//     lane-major staging as in neck_bwdw_kernel), no register holds a prefetched value;
//   * the chain GEMMs run as stages of twelve matrix instructions on two accumulators whose weight fragments were read one stage ahead
//     (Frag6); the dW accumulators are independent by construction;
//   * rows come onto the reduction index through the matrix core (to_rows) for what the chain produced (dpre1, dpre0) and by transposed
//     LDS reads for the staged inputs (a1, geo); two 16-feature tiles make one operand of a 32 x 32 x 16 product (block32).
// (A variant that paired two row tiles per dW step on the K = 32 form of the 16 x 16 instruction spilled 33 accumulator moves per tile and
// ran at 0.57 ms against 0.37: deleted in round 6, the record is DESIGN.md 4.3 [r4] and profiles/r04_ab_step.txt.)
constexpr int kRWThreads = 256;

struct RgbBwdWArgs {
    const float *dout, *out;            // [n][3]
    const float *a1, *a2;               // [n][64] saved activations
    const float *geo; int64_t ld_geo;   // [n][>= 64] the head's per-sample input
    int32_t tiles_per_ray; int64_t n_rays;
    WSrc w2t, w1at, w1gt, w0gt;         // transposed views, as RgbBwdArgs
    float *dgeo;                        // [n][64]
    float *s1, *s0;                     // [rays][64] sums of dpre1 / dpre0 over the samples of each ray
    float *partials; int64_t stride;    // per workgroup: dW1 [64][128] (columns: a1 0..63 | geo 64..127) | dW0 [64][64] (geo) | dW2 [3][64] | db2 [3] | pad
};

// One software-pipeline stage of a chain GEMM for the one-wave kernels: the A fragments of two output tiles at one k-step (six
// ds_read_b128), loaded ONE STAGE AHEAD of the twelve matrix instructions that use them -- with a single wave per SIMD nothing else
// covers the LDS latency of a read issued right in front of its use (measured: ~2500 of 14000 cycles per tile sat in s_waitcnt).
struct Frag6 { u32x4 l[2], h[2], m[2]; };
__device__ __forceinline__ void ld_frag6(Frag6 &f, const W3 w, int s, int pp) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const u32x4 *q = w.p + ((2 * pp + i) * w.ks + s) * 64;
        f.l[i] = q[2 * w.plane]; f.h[i] = q[0]; f.m[i] = q[w.plane];
    }
}
template <int KS>
__device__ __forceinline__ void mma_frag6(const Frag6 &f, const Opd<KS> &b, int s, f32x4 &a0, f32x4 &a1) {
    a0 = EMER_MF(f.l[0], b.h[s], a0); a1 = EMER_MF(f.l[1], b.h[s], a1);
    a0 = EMER_MF(f.h[0], b.l[s], a0); a1 = EMER_MF(f.h[1], b.l[s], a1);
    a0 = EMER_MF(f.m[0], b.m[s], a0); a1 = EMER_MF(f.m[1], b.m[s], a1);
    a0 = EMER_MF(f.m[0], b.h[s], a0); a1 = EMER_MF(f.m[1], b.h[s], a1);
    a0 = EMER_MF(f.h[0], b.m[s], a0); a1 = EMER_MF(f.h[1], b.m[s], a1);
    a0 = EMER_MF(f.h[0], b.h[s], a0); a1 = EMER_MF(f.h[1], b.h[s], a1);
}

// One 16-row tile at a time (no state carried between tiles), with the dW
// products on v_mfma_f32_32x32x16_bf16: its reduction index is 16 long -- one row tile -- and one instruction covers a 32 x 32 block of
// dW (four of the 16 x 16 tiles) in 32.5 cycles, where four K = 16 instructions of the 16 x 16 shape cost 79.  The operand of a 32-feature
// block -- lane (i, kg): feature i of the block, eight rows -- is two transposer outputs (lane (j, g): rows 4 g .. 4 g + 3 of feature j)
// after ONE v_permlane16_swap_b32 per register: lanes 16-31 / 48-63 take the second tile's rows 0-3 / 8-11 from lanes 0-15 / 32-47 and
// give the first tile's rows 4-7 / 12-15 back, which leaves lane (i, kg) with rows 8 kg .. 8 kg + 7 of its feature.
// Per tile: 148 chain + 48 transposer (16 x 16 x 32) + 72 dW (32 x 32 x 16) instructions.  Any S % 16 == 0.
#define EMER_RGBW_SB() __builtin_amdgcn_sched_barrier(0)   // pins the order of the stages of the one-wave kernels; inside a stage the scheduler is free
__global__ __launch_bounds__(kRWThreads, 1) void rgb_bwdw16_kernel(const RgbBwdWArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    u32x4 *w1al = smem, *w1gl = w1al + w3_units(4, 2), *w0l = w1gl + w3_units(4, 2);
    stage_w3(w1al, 4, 2, a.w1at);
    stage_w3(w1gl, 4, 2, a.w1gt);
    stage_w3(w0l, 4, 2, a.w0gt);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w1ap = w3_at(w1al, 4, 2, lane), w1gp = w3_at(w1gl, 4, 2, lane), w0p = w3_at(w0l, 4, 2, lane);
    const SelE sel = make_sel(lane);
    float w2a[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) w2a[p] = (g < a.w2t.k && 16 * p + m < a.w2t.n) ? a.w2t.w[(16 * p + m) * a.w2t.sn + g * a.w2t.sk] : 0.0f;
    // per-wave staging, filled global -> LDS (lane-major: lane (m, g) of piece p holds columns 16 p + 4 g .. + 3 of row m): the next tile's
    // a2 (1024 floats, single buffer: consumed at the start of a tile) and, DOUBLE buffered, its a1 | geo (2 x 2048 floats: they are
    // consumed at the END of a tile -- as the relu mask of d0 and, read back TRANSPOSED, as the B operands of the dW products)
    float *stg = reinterpret_cast<float *>(w0l + w3_units(4, 2)) + wave * (2048 + 3072);
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    // dW1 [64][128] and dW0 [64][64] as 32 x 32 blocks: lane (j, h), register r = dW[32 P + 8 (r >> 2) + 4 h + (r & 3)][32 Q + j]
    f32x16 acc1[2][4], acc0[2][2];
#pragma unroll
    for (int P = 0; P < 2; ++P) {
#pragma unroll
        for (int Q = 0; Q < 4; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[P][Q][r] = 0.0f;
#pragma unroll
        for (int Q = 0; Q < 2; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[P][Q][r] = 0.0f;
    }
    float w2acc[3] = {0.0f, 0.0f, 0.0f}, b2acc = 0.0f;
    const int tpr = a.tiles_per_ray;
    const int64_t wave_id = (int64_t)blockIdx.x * (kRWThreads / 64) + wave, n_waves = (int64_t)gridDim.x * (kRWThreads / 64);
    const unsigned lo64 = (unsigned)(m * 64 + 4 * g), log = (unsigned)m * (unsigned)a.ld_geo + 4u * g;
    const unsigned lo3c = (unsigned)(3 * m + (g < 3 ? g : 2));
    float yn = 0.0f, dn = 0.0f;
    auto issue = [&](int64_t ray, int j, int buf) {
        const int64_t row0 = (ray * tpr + j) * 16;
        const float *p2 = a.a2 + row0 * 64, *p1 = a.a1 + row0 * 64, *pg = a.geo + row0 * a.ld_geo;
        float *sb = stg + 1024 + 2048 * buf;
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((gptr)(p2 + (lo64 + 16u * p)), (lptr)(stg + 256 * p), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((gptr)(p1 + (lo64 + 16u * p)), (lptr)(sb + 256 * p), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds((gptr)(pg + (log + 16u * p)), (lptr)(sb + 1024 + 256 * p), 16, 0, 0);
        yn = (a.out + row0 * 3)[lo3c];
        dn = (a.dout + row0 * 3)[lo3c];
    };
    // B operand of a dW product straight from the staged tile: lane (j, gg) of a transposer output holds rows 4 gg .. 4 gg + 3 of feature
    // 16 p + j -- which sits at float 256 p + 4 (row + 16 (j >> 2)) + (j & 3) of the lane-major tile.  Four strided LDS reads and two
    // exact splits per 16-feature tile replace (per tensor) one operand split of the chain layout, four transposer passes through the
    // matrix pipe and their 48 accumulator read-backs: the matrix core only has to transpose what it produced itself (dpre1, dpre0).
    auto b_tile = [&](const float *tile, int p) -> SwT {
        const float *q = tile + 256 * p + 64 * (m >> 2) + (m & 3) + 16 * g;   // row 4 g (+ i below: 4 floats further per row)
        const float v0 = q[0], v1 = q[4], v2 = q[8], v3 = q[12];
        SwT t;
        unsigned h0, m0, l0, h1, m1, l1;
        split3(v0, v1, h0, m0, l0);
        split3(v2, v3, h1, m1, l1);
        t.h = u32x2{h0, h1}; t.m = u32x2{m0, m1}; t.l = u32x2{l0, l1};
        return t;
    };
    int buf = 0;
    auto advance = [&](int64_t ray, int j, int by, int64_t &nr, int &nj) {   // tile `by` positions further in this wave's sequence (clamped at its end)
        nr = ray; nj = j;
        for (int i = 0; i < by; ++i) {
            int64_t r2 = nr; int j2 = nj + 1;
            if (j2 == tpr) { j2 = 0; r2 = nr + n_waves; }
            if (r2 >= a.n_rays) break;
            nr = r2; nj = j2;
        }
    };
    if (wave_id < a.n_rays) issue(wave_id, 0, 0);
    // dgeo of a tile is STORED AT THE START OF THE NEXT TILE, right behind that tile's wait for its inputs: on this part stores count in
    // vmcnt like loads, so a store issued mid-tile is still in flight at the next `s_waitcnt vmcnt(0)` and the wave sits out its write
    // acknowledgement (measured: 90 of 420 us); issued behind the wait it has a whole tile to drain.  The values stay where the matrix
    // pipe left them (accumulator registers) in the meantime.
    f32x4 dgp[4];
    float *dgeo_prev = nullptr;
    Frag6 fa, fb;
    ld_frag6(fa, w1ap, 0, 0);
    for (int64_t ray = wave_id; ray < a.n_rays; ray += n_waves) {
        float s1c[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s0c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < tpr; ++j) {
            float *dgeo = a.dgeo + ((ray * tpr + j) * 16) * 64;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (dgeo_prev) {
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(dgeo_prev + (lo64 + 16u * p)) = dgp[p];
            }
            f32x4 m2[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) m2[p] = *reinterpret_cast<const f32x4 *>(stg + 256 * p + 4 * lane);
            const float d2 = g < 3 ? dn * yn * (1.0f - yn) : 0.0f;
            const float *sb = stg + 1024 + 2048 * buf;   // this tile's a1 | geo
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                int64_t nr; int nj;
                advance(ray, j, 1, nr, nj);
                issue(nr, nj, buf ^ 1);
            }
            buf ^= 1;
            // Order of the tile body.  One wave per SIMD: whatever overlaps, overlaps inside this instruction stream.  The three GEMMs of the
            // chain run as twelve stages of twelve matrix instructions (two output tiles x one k-step x six partial products); every stage
            // first issues the NEXT stage's weight-fragment reads (the last one of a tile: the next tile's first), then its matrix
            // instructions, then a share of the vector work that does not depend on them -- the output layer's weight gradient next to the
            // first GEMM, the masking / splitting of d0 next to dgeo's d1 half, the split + transposition of the B operands (a1, geo) next to
            // its d0 half.  EMER_RGBW_SB() pins the stage order; inside a stage the scheduler is free.
            Opd<2> d1o, d0o;
            {
                f32x4 d1[4];
                zero<4>(d1);
#pragma unroll
                for (int p = 0; p < 4; ++p) d1[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[p], d2, d1[p], 0, 0, 0);
                relu_mask<4>(d1, m2);
                make_opd<4>(d1, d1o);
            }
            EMER_RGBW_SB();
            f32x4 dg[4], d0[4];
            zero<4>(dg); zero<4>(d0);
            SwP Bq[4];   // a1 features 0-31, 32-63; geo features 0-31, 32-63 (rows on the reduction index, 32-feature blocks)
            // ---- GEMM 1: d0 = W1a^T d1 (stages 0-3) next to dW2 / db2
            ld_frag6(fb, w1ap, 0, 1); mma_frag6<2>(fa, d1o, 0, d0[0], d0[1]);
            b2acc += d2;
            { const float dc = __shfl(d2, m, 64); w2acc[0] += row16_reduce_scatter(m2, dc, m); }
            EMER_RGBW_SB();
            ld_frag6(fa, w1ap, 1, 0); mma_frag6<2>(fb, d1o, 0, d0[2], d0[3]);
            { const float dc = __shfl(d2, 16 + m, 64); w2acc[1] += row16_reduce_scatter(m2, dc, m); }
            EMER_RGBW_SB();
            ld_frag6(fb, w1ap, 1, 1); mma_frag6<2>(fa, d1o, 1, d0[0], d0[1]);
            { const float dc = __shfl(d2, 32 + m, 64); w2acc[2] += row16_reduce_scatter(m2, dc, m); }
            EMER_RGBW_SB();
            ld_frag6(fa, w1gp, 0, 0); mma_frag6<2>(fb, d1o, 1, d0[2], d0[3]);
            EMER_RGBW_SB();
            // ---- GEMM 2: dgeo += W1g^T d1 (stages 4-7) next to the mask and split of d0
            ld_frag6(fb, w1gp, 0, 1); mma_frag6<2>(fa, d1o, 0, dg[0], dg[1]);
            {
                f32x4 m1[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) m1[p] = *reinterpret_cast<const f32x4 *>(sb + 256 * p + 4 * lane);
                relu_mask<4>(d0, m1);
            }
            EMER_RGBW_SB();
            ld_frag6(fa, w1gp, 1, 0); mma_frag6<2>(fb, d1o, 0, dg[2], dg[3]);
            {   // k-step 0 of the split operand = tiles 0, 1; k-step 1 = tiles 2, 3
                f32x4 v[2] = {d0[0], d0[1]};
                Opd<1> o1;
                make_opd<2>(v, o1);
                d0o.h[0] = o1.h[0]; d0o.m[0] = o1.m[0]; d0o.l[0] = o1.l[0];
            }
            EMER_RGBW_SB();
            ld_frag6(fb, w1gp, 1, 1); mma_frag6<2>(fa, d1o, 1, dg[0], dg[1]);
            {
                f32x4 v[2] = {d0[2], d0[3]};
                Opd<1> o1;
                make_opd<2>(v, o1);
                d0o.h[1] = o1.h[0]; d0o.m[1] = o1.m[0]; d0o.l[1] = o1.l[0];
            }
            EMER_RGBW_SB();
            ld_frag6(fa, w0p, 0, 0); mma_frag6<2>(fb, d1o, 1, dg[2], dg[3]);
            EMER_RGBW_SB();
            // ---- GEMM 3: dgeo += W0g^T d0 (stages 8-11) next to the B operands of the dW products
            ld_frag6(fb, w0p, 0, 1); mma_frag6<2>(fa, d0o, 0, dg[0], dg[1]);
            Bq[0] = block32(b_tile(sb, 0), b_tile(sb, 1));
            EMER_RGBW_SB();
            ld_frag6(fa, w0p, 1, 0); mma_frag6<2>(fb, d0o, 0, dg[2], dg[3]);
            Bq[1] = block32(b_tile(sb, 2), b_tile(sb, 3));
            EMER_RGBW_SB();
            ld_frag6(fb, w0p, 1, 1); mma_frag6<2>(fa, d0o, 1, dg[0], dg[1]);
            Bq[2] = block32(b_tile(sb + 1024, 0), b_tile(sb + 1024, 1));
            EMER_RGBW_SB();
            ld_frag6(fa, w1ap, 0, 0); mma_frag6<2>(fb, d0o, 1, dg[2], dg[3]);   // (fa: the next tile's first stage)
            Bq[3] = block32(b_tile(sb + 1024, 2), b_tile(sb + 1024, 3));
            EMER_RGBW_SB();
#pragma unroll
            for (int p = 0; p < 4; ++p) dgp[p] = dg[p];
            dgeo_prev = dgeo;
            EMER_RGBW_SB();
            // ---- dW1 += dpre1^T [a1 | geo], dW0 += dpre0^T geo on this tile's 16 rows, as 32 x 32 blocks
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const SwT t0 = to_rows<2>(d1o, 2 * P, sel, &s1c[2 * P]), t1 = to_rows<2>(d1o, 2 * P + 1, sel, &s1c[2 * P + 1]);
                const SwP A = block32(t0, t1);
                dw_blocks<4>(acc1[P], A, Bq);
            }
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const SwT t0 = to_rows<2>(d0o, 2 * P, sel, &s0c[2 * P]), t1 = to_rows<2>(d0o, 2 * P + 1, sel, &s0c[2 * P + 1]);
                const SwP A = block32(t0, t1);
                dw_blocks<2>(acc0[P], A, *reinterpret_cast<const SwP (*)[2]>(&Bq[2]));
            }
            EMER_RGBW_SB();
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            s1c[p] += __shfl_xor(s1c[p], 16, 64); s1c[p] += __shfl_xor(s1c[p], 32, 64);
            s0c[p] += __shfl_xor(s0c[p], 16, 64); s0c[p] += __shfl_xor(s0c[p], 32, 64);
            if (g == 0) { a.s1[ray * 64 + 16 * p + m] = s1c[p]; a.s0[ray * 64 + 16 * p + m] = s0c[p]; }
        }
    }
    if (dgeo_prev) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(dgeo_prev + (lo64 + 16u * p)) = dgp[p];
    }
    // ---- sum the four waves through LDS (the weights are dead), one coalesced partial per workgroup (the paired kernel's layout)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr int P1 = 132, P0 = 68;
    float *r1 = reinterpret_cast<float *>(smem), *r0 = r1 + 64 * P1, *r2 = r0 + 64 * P0;
    const int j32 = lane & 31, h32 = lane >> 5;
    for (int w = 0; w < kRWThreads / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int P = 0; P < 2; ++P) {
#pragma unroll
                for (int Q = 0; Q < 4; ++Q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *q = r1 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * P1 + 32 * Q + j32;
                        *q = (w == 0) ? acc1[P][Q][r] : *q + acc1[P][Q][r];
                    }
#pragma unroll
                for (int Q = 0; Q < 2; ++Q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *q = r0 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * P0 + 32 * Q + j32;
                        *q = (w == 0) ? acc0[P][Q][r] : *q + acc0[P][Q][r];
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) r2[wave * 196 + c * 64 + 16 * (m >> 2) + 4 * g + (m & 3)] = w2acc[c];
    b2acc = row16_sum(b2acc);
    if (m == 0 && g < 3) r2[wave * 196 + 192 + g] = b2acc;
    __syncthreads();
    float *part = a.partials + (int64_t)blockIdx.x * a.stride;
    for (int i = threadIdx.x; i < 64 * 128; i += kRWThreads) part[i] = r1[(i >> 7) * P1 + (i & 127)];
    for (int i = threadIdx.x; i < 64 * 64; i += kRWThreads) part[64 * 128 + i] = r0[(i >> 6) * P0 + (i & 63)];
    if ((int)threadIdx.x < 195) {
        float t = 0.0f;
        for (int w = 0; w < kRWThreads / 64; ++w) t += r2[w * 196 + threadIdx.x];
        part[64 * 128 + 64 * 64 + threadIdx.x] = t;
    }
}

// ---- [r6] the same backward with a1 / a2 RECOMPUTED from geo and the per-ray pre-activations: the forward (field_fwd_kernel / rgb_fwd_kernel
// with a1 = a2 = NULL) then stores nothing but geo and the colours, and this kernel reads geo (256 B / sample) instead of geo + a1 + a2
// (768 B / sample): 1.07 GB of HBM traffic per million samples and step gone.  a1 = relu(W0g geo + rb0), a2 = relu(W1g geo + W1a a1 + rb1)
// are evaluated with the forward's fragments in the forward's order per accumulator (rb, then k-step 0, then k-step 1; for a2: the geo
// part first), so they are BITWISE the forward's values and every result of this kernel is bitwise rgb_bwdw16_kernel's.
//   * LDS: six 64 x 64 weight matrices as bf16x3 fragments (W0g, W1a, W1g for the recomputation, their transposes for the chain) = 144 KB,
//     plus 1 KB per wave for the per-ray pre-activations of the current and of the next ray (global -> LDS, 4 B per lane).  No room is
//     left for staging tiles, and none is needed: geo is the only per-sample input; the next tile's 16 floats per lane are prefetched in
//     REGISTERS one tile ahead (an unconditional load at the top of the tile body, consumed behind the next tile's wait).
//   * The B operands of the dW products (a1, geo with the rows on the reduction index) come from the chain-layout operands the
//     recomputation has split anyway, through the matrix-core transposer (to_rows: 3 instructions per 16-feature tile, exact) -- the
//     same bits b_tile() reads back from the staged tile in rgb_bwdw16_kernel (the 3-term split is elementwise).
//   * Per tile: 144 (recomputation) + 148 (chain) + 48 (transposer: dpre1, dpre0, a1, geo) 16 x 16 x 32 instructions + 72 of 32 x 32 x 16.
struct RgbBwdRArgs {
    const float *dout, *out;            // [n][3]
    const float *geo; int64_t ld_geo;   // [n][>= 64] the head's per-sample input
    const float *rb0, *rb1; int64_t ld_rb;   // [rays][64] per-ray pre-activations of layers 0 / 1 (bias included), as the forward's
    int32_t tiles_per_ray; int64_t n_rays;
    const float *a1;                    // [n][64] the forward's stored a1 (A1_STORED) or null
    WSrc w0g, w1a, w1g;                 // forward views (RgbFwdArgs)
    WSrc w2t, w1at, w1gt, w0gt;         // transposed views (RgbBwdArgs)
    float *dgeo;                        // [n][64]
    float *s1, *s0;                     // [rays][64]
    float *partials; int64_t stride;    // as RgbBwdWArgs
};

// A1_STORED: the cheaper half -- a1 comes from the forward's store (prefetched in registers like geo), only a2 is recomputed (96 instead of
// 144 extra chain instructions; the forward then stores a1 but not a2).
template <bool A1_STORED>
__global__ __launch_bounds__(kRWThreads, 1) void rgb_bwdwr_kernel(const RgbBwdRArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    u32x4 *w0fl = smem, *w1afl = w0fl + w3_units(4, 2), *w1gfl = w1afl + w3_units(4, 2);
    u32x4 *w1al = w1gfl + w3_units(4, 2), *w1gl = w1al + w3_units(4, 2), *w0l = w1gl + w3_units(4, 2);
    if (!A1_STORED) stage_w3(w0fl, 4, 2, a.w0g);
    stage_w3(w1afl, 4, 2, a.w1a);
    stage_w3(w1gfl, 4, 2, a.w1g);
    stage_w3(w1al, 4, 2, a.w1at);
    stage_w3(w1gl, 4, 2, a.w1gt);
    stage_w3(w0l, 4, 2, a.w0gt);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w0f = w3_at(w0fl, 4, 2, lane), w1af = w3_at(w1afl, 4, 2, lane), w1gf = w3_at(w1gfl, 4, 2, lane);
    const W3 w1ap = w3_at(w1al, 4, 2, lane), w1gp = w3_at(w1gl, 4, 2, lane), w0p = w3_at(w0l, 4, 2, lane);
    const SelE sel = make_sel(lane);
    float w2a[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) w2a[p] = (g < a.w2t.k && 16 * p + m < a.w2t.n) ? a.w2t.w[(16 * p + m) * a.w2t.sn + g * a.w2t.sk] : 0.0f;
    // per wave: [parity][rb0 (64) | rb1 (64)] floats
    float *rbl = reinterpret_cast<float *>(w0l + w3_units(4, 2)) + wave * 256;
    using gptr = const __attribute__((address_space(1))) void *;
    using lptr = __attribute__((address_space(3))) void *;
    f32x16 acc1[2][4], acc0[2][2];
#pragma unroll
    for (int P = 0; P < 2; ++P) {
#pragma unroll
        for (int Q = 0; Q < 4; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[P][Q][r] = 0.0f;
#pragma unroll
        for (int Q = 0; Q < 2; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[P][Q][r] = 0.0f;
    }
    float w2acc[3] = {0.0f, 0.0f, 0.0f}, b2acc = 0.0f;
    const int tpr = a.tiles_per_ray;
    const int64_t wave_id = (int64_t)blockIdx.x * (kRWThreads / 64) + wave, n_waves = (int64_t)gridDim.x * (kRWThreads / 64);
    const unsigned lo64 = (unsigned)(m * 64 + 4 * g), log = (unsigned)m * (unsigned)a.ld_geo + 4u * g;
    const unsigned lo3c = (unsigned)(3 * m + (g < 3 ? g : 2));
    float yn = 0.0f, dn = 0.0f;
    f32x4 xg[4];   // the NEXT tile's geo in chain layout (lane (m, g): columns 16 p + 4 g .. + 3 of row m)
    f32x4 ag[4];   // ... and its stored a1 (A1_STORED)
    int rpar = 0;  // parity of the per-ray buffer the CURRENT tile reads
    auto issue = [&](int64_t ray, int j, int par) {
        const int64_t row0 = (ray * tpr + j) * 16;
        const float *pg = a.geo + row0 * a.ld_geo;
#pragma unroll
        for (int p = 0; p < 4; ++p) xg[p] = *reinterpret_cast<const f32x4 *>(pg + (log + 16u * p));
        if (A1_STORED) {
            const float *pa = a.a1 + row0 * 64;
#pragma unroll
            for (int p = 0; p < 4; ++p) ag[p] = *reinterpret_cast<const f32x4 *>(pa + (lo64 + 16u * p));
        }
        yn = (a.out + row0 * 3)[lo3c];
        dn = (a.dout + row0 * 3)[lo3c];
        if (j == 0) {   // (wave-uniform) first tile of a ray: its pre-activations, 64 + 64 floats, one per lane each, into buffer `par`
            if (!A1_STORED) __builtin_amdgcn_global_load_lds((gptr)(a.rb0 + ray * a.ld_rb + lane), (lptr)(rbl + 128 * par), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(a.rb1 + ray * a.ld_rb + lane), (lptr)(rbl + 128 * par + 64), 4, 0, 0);
        }
    };
    auto advance = [&](int64_t ray, int j, int64_t &nr, int &nj) {   // the next tile of this wave's sequence (clamped at its end)
        nr = ray; nj = j + 1;
        if (nj == tpr) { nj = 0; nr = ray + n_waves; }
        if (nr >= a.n_rays) { nr = ray; nj = j; }
    };
    if (wave_id < a.n_rays) issue(wave_id, 0, 0);
    f32x4 dgp[4];
    float *dgeo_prev = nullptr;
    Frag6 fa, fb;
    const W3 wfirst = A1_STORED ? w1gf : w0f;   // the matrix of a tile's first stage
    ld_frag6(fa, wfirst, 0, 0);
    for (int64_t ray = wave_id; ray < a.n_rays; ray += n_waves) {
        float s1c[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s0c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < tpr; ++j) {
            float *dgeo = a.dgeo + ((ray * tpr + j) * 16) * 64;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (dgeo_prev) {
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(dgeo_prev + (lo64 + 16u * p)) = dgp[p];
            }
            f32x4 x[4], a1[4], a2[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) { x[p] = xg[p]; if (A1_STORED) a1[p] = ag[p]; }
            const float d2 = g < 3 ? dn * yn * (1.0f - yn) : 0.0f;
            const float *rb = rbl + 128 * rpar;   // this ray's rb0 | rb1
            {
                int64_t nr; int nj;
                advance(ray, j, nr, nj);
                issue(nr, nj, (nj == 0 && (nr != ray || tpr == 1)) ? (rpar ^ 1) : rpar);
                // (a clamped repeat of the last tile re-reads with j != 0 unless tpr == 1: then it reloads its own ray into the other buffer, harmless)
            }
            // ---- recomputation: a1 = relu(W0g x + rb0), a2 = relu(W1g x + W1a a1 + rb1).  Eight stages on x (k-step 0 of all four
            // accumulator pairs first, so that the split of k-step 1 runs next to them), four on a1.
            Opd<2> xo, ho;
            SwP Bq[4];   // a1 features 0-31, 32-63; geo features 0-31, 32-63 (rows on the reduction index, 32-feature blocks)
            if constexpr (A1_STORED) {
                // ---- a2 = relu(W1g x + W1a a1 + rb1) with a1 from the forward's store: four stages on x, four on a1
                {
                    f32x4 v[2] = {x[0], x[1]};
                    Opd<1> o1;
                    make_opd<2>(v, o1);
                    xo.h[0] = o1.h[0]; xo.m[0] = o1.m[0]; xo.l[0] = o1.l[0];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) a2[p] = *reinterpret_cast<const f32x4 *>(rb + 64 + 16 * p + 4 * g);
                EMER_RGBW_SB();
                ld_frag6(fb, w1gf, 0, 1); mma_frag6<2>(fa, xo, 0, a2[0], a2[1]);
                {
                    f32x4 v[2] = {x[2], x[3]};
                    Opd<1> o1;
                    make_opd<2>(v, o1);
                    xo.h[1] = o1.h[0]; xo.m[1] = o1.m[0]; xo.l[1] = o1.l[0];
                }
                EMER_RGBW_SB();
                ld_frag6(fa, w1gf, 1, 0); mma_frag6<2>(fb, xo, 0, a2[2], a2[3]);
                make_opd<4>(a1, ho);
                EMER_RGBW_SB();
                ld_frag6(fb, w1gf, 1, 1); mma_frag6<2>(fa, xo, 1, a2[0], a2[1]);
                Bq[2] = block32(to_rows<2>(xo, 0, sel), to_rows<2>(xo, 1, sel));
                EMER_RGBW_SB();
                ld_frag6(fa, w1af, 0, 0); mma_frag6<2>(fb, xo, 1, a2[2], a2[3]);
                Bq[3] = block32(to_rows<2>(xo, 2, sel), to_rows<2>(xo, 3, sel));
                EMER_RGBW_SB();
                ld_frag6(fb, w1af, 0, 1); mma_frag6<2>(fa, ho, 0, a2[0], a2[1]);
                Bq[0] = block32(to_rows<2>(ho, 0, sel), to_rows<2>(ho, 1, sel));
                EMER_RGBW_SB();
                ld_frag6(fa, w1af, 1, 0); mma_frag6<2>(fb, ho, 0, a2[2], a2[3]);
                Bq[1] = block32(to_rows<2>(ho, 2, sel), to_rows<2>(ho, 3, sel));
                EMER_RGBW_SB();
                ld_frag6(fb, w1af, 1, 1); mma_frag6<2>(fa, ho, 1, a2[0], a2[1]);
                EMER_RGBW_SB();
                ld_frag6(fa, w1ap, 0, 0); mma_frag6<2>(fb, ho, 1, a2[2], a2[3]);   // (fa: the chain's first stage)
                EMER_RGBW_SB();
            } else {
                {
                    f32x4 v[2] = {x[0], x[1]};
                    Opd<1> o1;
                    make_opd<2>(v, o1);
                    xo.h[0] = o1.h[0]; xo.m[0] = o1.m[0]; xo.l[0] = o1.l[0];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    a1[p] = *reinterpret_cast<const f32x4 *>(rb + 16 * p + 4 * g);
                    a2[p] = *reinterpret_cast<const f32x4 *>(rb + 64 + 16 * p + 4 * g);
                }
                EMER_RGBW_SB();
                ld_frag6(fb, w1gf, 0, 0); mma_frag6<2>(fa, xo, 0, a1[0], a1[1]);
                {
                    f32x4 v[2] = {x[2], x[3]};
                    Opd<1> o1;
                    make_opd<2>(v, o1);
                    xo.h[1] = o1.h[0]; xo.m[1] = o1.m[0]; xo.l[1] = o1.l[0];
                }
                EMER_RGBW_SB();
                ld_frag6(fa, w0f, 0, 1); mma_frag6<2>(fb, xo, 0, a2[0], a2[1]);
                EMER_RGBW_SB();
                ld_frag6(fb, w1gf, 0, 1); mma_frag6<2>(fa, xo, 0, a1[2], a1[3]);
                EMER_RGBW_SB();
                ld_frag6(fa, w0f, 1, 0); mma_frag6<2>(fb, xo, 0, a2[2], a2[3]);
                EMER_RGBW_SB();
                ld_frag6(fb, w1gf, 1, 0); mma_frag6<2>(fa, xo, 1, a1[0], a1[1]);
                EMER_RGBW_SB();
                ld_frag6(fa, w0f, 1, 1); mma_frag6<2>(fb, xo, 1, a2[0], a2[1]);
                {   // a1 tiles 0, 1 are final: ReLU, split -> k-step 0 of the next operand
                    f32x4 v[2] = {a1[0], a1[1]};
                    relu<2>(v);
                    a1[0] = v[0]; a1[1] = v[1];
                    Opd<1> o1;
                    make_opd<2>(v, o1);
                    ho.h[0] = o1.h[0]; ho.m[0] = o1.m[0]; ho.l[0] = o1.l[0];
                }
                EMER_RGBW_SB();
                ld_frag6(fb, w1gf, 1, 1); mma_frag6<2>(fa, xo, 1, a1[2], a1[3]);
                EMER_RGBW_SB();
                ld_frag6(fa, w1af, 0, 0); mma_frag6<2>(fb, xo, 1, a2[2], a2[3]);
                {
                    f32x4 v[2] = {a1[2], a1[3]};
                    relu<2>(v);
                    a1[2] = v[0]; a1[3] = v[1];
                    Opd<1> o1;
                    make_opd<2>(v, o1);
                    ho.h[1] = o1.h[0]; ho.m[1] = o1.m[0]; ho.l[1] = o1.l[0];
                }
                EMER_RGBW_SB();
                ld_frag6(fb, w1af, 0, 1); mma_frag6<2>(fa, ho, 0, a2[0], a2[1]);
                Bq[2] = block32(to_rows<2>(xo, 0, sel), to_rows<2>(xo, 1, sel));
                EMER_RGBW_SB();
                ld_frag6(fa, w1af, 1, 0); mma_frag6<2>(fb, ho, 0, a2[2], a2[3]);
                Bq[3] = block32(to_rows<2>(xo, 2, sel), to_rows<2>(xo, 3, sel));
                EMER_RGBW_SB();
                ld_frag6(fb, w1af, 1, 1); mma_frag6<2>(fa, ho, 1, a2[0], a2[1]);
                Bq[0] = block32(to_rows<2>(ho, 0, sel), to_rows<2>(ho, 1, sel));
                EMER_RGBW_SB();
                ld_frag6(fa, w1ap, 0, 0); mma_frag6<2>(fb, ho, 1, a2[2], a2[3]);   // (fa: the chain's first stage)
                Bq[1] = block32(to_rows<2>(ho, 2, sel), to_rows<2>(ho, 3, sel));
                EMER_RGBW_SB();
            }
            relu<4>(a2);
            // ---- the chain, as rgb_bwdw16_kernel (a2 / a1 from registers instead of the staged tile)
            Opd<2> d1o, d0o;
            {
                f32x4 d1[4];
                zero<4>(d1);
#pragma unroll
                for (int p = 0; p < 4; ++p) d1[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[p], d2, d1[p], 0, 0, 0);
                relu_mask<4>(d1, a2);
                make_opd<4>(d1, d1o);
            }
            EMER_RGBW_SB();
            f32x4 dg[4], d0[4];
            zero<4>(dg); zero<4>(d0);
            // ---- GEMM 1: d0 = W1a^T d1 (stages 0-3) next to dW2 / db2
            ld_frag6(fb, w1ap, 0, 1); mma_frag6<2>(fa, d1o, 0, d0[0], d0[1]);
            b2acc += d2;
            { const float dc = __shfl(d2, m, 64); w2acc[0] += row16_reduce_scatter(a2, dc, m); }
            EMER_RGBW_SB();
            ld_frag6(fa, w1ap, 1, 0); mma_frag6<2>(fb, d1o, 0, d0[2], d0[3]);
            { const float dc = __shfl(d2, 16 + m, 64); w2acc[1] += row16_reduce_scatter(a2, dc, m); }
            EMER_RGBW_SB();
            ld_frag6(fb, w1ap, 1, 1); mma_frag6<2>(fa, d1o, 1, d0[0], d0[1]);
            { const float dc = __shfl(d2, 32 + m, 64); w2acc[2] += row16_reduce_scatter(a2, dc, m); }
            EMER_RGBW_SB();
            ld_frag6(fa, w1gp, 0, 0); mma_frag6<2>(fb, d1o, 1, d0[2], d0[3]);
            EMER_RGBW_SB();
            // ---- GEMM 2: dgeo += W1g^T d1 (stages 4-7) next to the mask and split of d0
            ld_frag6(fb, w1gp, 0, 1); mma_frag6<2>(fa, d1o, 0, dg[0], dg[1]);
            relu_mask<4>(d0, a1);
            EMER_RGBW_SB();
            ld_frag6(fa, w1gp, 1, 0); mma_frag6<2>(fb, d1o, 0, dg[2], dg[3]);
            {
                f32x4 v[2] = {d0[0], d0[1]};
                Opd<1> o1;
                make_opd<2>(v, o1);
                d0o.h[0] = o1.h[0]; d0o.m[0] = o1.m[0]; d0o.l[0] = o1.l[0];
            }
            EMER_RGBW_SB();
            ld_frag6(fb, w1gp, 1, 1); mma_frag6<2>(fa, d1o, 1, dg[0], dg[1]);
            {
                f32x4 v[2] = {d0[2], d0[3]};
                Opd<1> o1;
                make_opd<2>(v, o1);
                d0o.h[1] = o1.h[0]; d0o.m[1] = o1.m[0]; d0o.l[1] = o1.l[0];
            }
            EMER_RGBW_SB();
            ld_frag6(fa, w0p, 0, 0); mma_frag6<2>(fb, d1o, 1, dg[2], dg[3]);
            EMER_RGBW_SB();
            // ---- GEMM 3: dgeo += W0g^T d0 (stages 8-11)
            ld_frag6(fb, w0p, 0, 1); mma_frag6<2>(fa, d0o, 0, dg[0], dg[1]);
            EMER_RGBW_SB();
            ld_frag6(fa, w0p, 1, 0); mma_frag6<2>(fb, d0o, 0, dg[2], dg[3]);
            EMER_RGBW_SB();
            ld_frag6(fb, w0p, 1, 1); mma_frag6<2>(fa, d0o, 1, dg[0], dg[1]);
            EMER_RGBW_SB();
            ld_frag6(fa, wfirst, 0, 0); mma_frag6<2>(fb, d0o, 1, dg[2], dg[3]);   // (fa: the next tile's first stage)
            EMER_RGBW_SB();
#pragma unroll
            for (int p = 0; p < 4; ++p) dgp[p] = dg[p];
            dgeo_prev = dgeo;
            EMER_RGBW_SB();
            // ---- dW1 += dpre1^T [a1 | geo], dW0 += dpre0^T geo on this tile's 16 rows, as 32 x 32 blocks
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const SwT t0 = to_rows<2>(d1o, 2 * P, sel, &s1c[2 * P]), t1 = to_rows<2>(d1o, 2 * P + 1, sel, &s1c[2 * P + 1]);
                const SwP A = block32(t0, t1);
                dw_blocks<4>(acc1[P], A, Bq);
            }
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const SwT t0 = to_rows<2>(d0o, 2 * P, sel, &s0c[2 * P]), t1 = to_rows<2>(d0o, 2 * P + 1, sel, &s0c[2 * P + 1]);
                const SwP A = block32(t0, t1);
                dw_blocks<2>(acc0[P], A, *reinterpret_cast<const SwP (*)[2]>(&Bq[2]));
            }
            EMER_RGBW_SB();
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            s1c[p] += __shfl_xor(s1c[p], 16, 64); s1c[p] += __shfl_xor(s1c[p], 32, 64);
            s0c[p] += __shfl_xor(s0c[p], 16, 64); s0c[p] += __shfl_xor(s0c[p], 32, 64);
            if (g == 0) { a.s1[ray * 64 + 16 * p + m] = s1c[p]; a.s0[ray * 64 + 16 * p + m] = s0c[p]; }
        }
        rpar ^= 1;
    }
    if (dgeo_prev) {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(dgeo_prev + (lo64 + 16u * p)) = dgp[p];
    }
    // ---- sum the four waves through LDS (the weights are dead), one coalesced partial per workgroup (rgb_bwdw16_kernel's layout)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr int P1 = 132, P0 = 68;
    float *r1 = reinterpret_cast<float *>(smem), *r0 = r1 + 64 * P1, *r2 = r0 + 64 * P0;
    const int j32 = lane & 31, h32 = lane >> 5;
    for (int w = 0; w < kRWThreads / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int P = 0; P < 2; ++P) {
#pragma unroll
                for (int Q = 0; Q < 4; ++Q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *q = r1 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * P1 + 32 * Q + j32;
                        *q = (w == 0) ? acc1[P][Q][r] : *q + acc1[P][Q][r];
                    }
#pragma unroll
                for (int Q = 0; Q < 2; ++Q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *q = r0 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * P0 + 32 * Q + j32;
                        *q = (w == 0) ? acc0[P][Q][r] : *q + acc0[P][Q][r];
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) r2[wave * 196 + c * 64 + 16 * (m >> 2) + 4 * g + (m & 3)] = w2acc[c];
    b2acc = row16_sum(b2acc);
    if (m == 0 && g < 3) r2[wave * 196 + 192 + g] = b2acc;
    __syncthreads();
    float *part = a.partials + (int64_t)blockIdx.x * a.stride;
    for (int i = threadIdx.x; i < 64 * 128; i += kRWThreads) part[i] = r1[(i >> 7) * P1 + (i & 127)];
    for (int i = threadIdx.x; i < 64 * 64; i += kRWThreads) part[64 * 128 + i] = r0[(i >> 6) * P0 + (i & 63)];
    if ((int)threadIdx.x < 195) {
        float t = 0.0f;
        for (int w = 0; w < kRWThreads / 64; ++w) t += r2[w * 196 + threadIdx.x];
        part[64 * 128 + 64 * 64 + threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------ density MLP backward with its weight gradients
// Backward of density_fwd_kernel (narrow input, L F <= 16) INCLUDING dW0, db0, dW1, db1 -- the proposal network's step.  The
// hidden layer is recomputed from the encoding (two k-steps of the fp32 matrix instruction), the gradient of the single
// output is rank-1 (dh = (h > 0) w1 dpre1), the input gradient is sixteen more fp32 matrix instructions, and every weight
// gradient is a sum over ROWS of per-lane products -- 10 + L F of the 16-quantity DPP reduce-scatters of the rgb backward per
// tile, 11 + L F accumulators per lane for the whole kernel.  Against the unfused path (hidden layer stored by the forward and
// re-read twice, dpre0 written and re-read: 1.1 GB per million rows) this moves 72 MB.
struct DensBwdWArgs {
    const float *ddens, *dens, *enc; int64_t n; int32_t n_levels;
    WSrc w0; const float *b0; const float *w1;   // W0 [64][K0], b0 [64], W1 [64] (one output row)
    float *denc;                                  // level-major [L][n][F]
    float *partials; int64_t stride;              // per workgroup: dW0 [64][K0] | db0 [64] | dW1 [64] | db1 [1]
};
constexpr int kDensPartMax = 64 * 16 + 64 + 64 + 4;

template <int KS4>
__global__ __launch_bounds__(kNThreads, 2) void density_bwdw_kernel(const DensBwdWArgs a, int32_t F) {
    __shared__ float part[kNThreads / 64][kDensPartMax];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const int32_t K0 = a.w0.k;
    float aw[4][KS4];
    int64_t koff[KS4];
#pragma unroll
    for (int s = 0; s < KS4; ++s) {
        const int32_t k = 4 * s + g, kc = k < K0 ? k : K0 - 1;
        koff[s] = (int64_t)(kc / F) * a.n * F + kc % F;
#pragma unroll
        for (int p = 0; p < 4; ++p) aw[p][s] = k < K0 ? a.w0.w[(int64_t)(16 * p + m) * a.w0.sn + (int64_t)k * a.w0.sk] : 0.0f;
    }
    f32x4 b0r[4], w1r[4], awt[4];  // awt[p][i] = W0[16 p + 4 g + i][m]: the A operand of the input gradient (rows = input feature m)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int32_t nf = 16 * p + 4 * g + i;
            b0r[p][i] = a.b0 ? a.b0[nf] : 0.0f;
            w1r[p][i] = a.w1[nf];
            awt[p][i] = m < K0 ? a.w0.w[(int64_t)nf * a.w0.sn + (int64_t)m * a.w0.sk] : 0.0f;
        }
    float w0acc[4 * KS4], w1acc = 0.0f, b0acc = 0.0f, b1acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4 * KS4; ++k) w0acc[k] = 0.0f;
    const int64_t n_tiles = (a.n + 15) >> 4, n_chunks = (n_tiles + kNeckChunk - 1) / kNeckChunk;
    for (int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; c < n_chunks; c += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t t0 = c * kNeckChunk;
        float x[kNeckChunk][KS4], dd[kNeckChunk], de[kNeckChunk];
#pragma unroll
        for (int j = 0; j < kNeckChunk; ++j) {  // the chunk's inputs, all loads in flight (clamped rows)
            const int64_t row = (t0 + j) * 16 + m, rc = row < a.n ? row : a.n - 1;
#pragma unroll
            for (int s = 0; s < KS4; ++s) x[j][s] = a.enc[koff[s] + rc * F];
            dd[j] = a.ddens[rc];
            de[j] = a.dens[rc];
        }
#pragma unroll
        for (int j = 0; j < kNeckChunk; ++j) {
            if (t0 + j >= n_tiles) break;
            const int64_t row = (t0 + j) * 16 + m;
            const bool ok = row < a.n;
            f32x4 h[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) h[p] = b0r[p];
#pragma unroll
            for (int s = 0; s < KS4; ++s)
#pragma unroll
                for (int p = 0; p < 4; ++p) h[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[p][s], x[j][s], h[p], 0, 0, 0);
            relu<4>(h);
            const float fix = ok ? dd[j] * fminf(de[j], 3269017.3724721107f) : 0.0f;  // trunc_exp': d(pre-activation of the output)
            f32x4 dh[4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[p][i] = h[p][i] > 0.0f ? w1r[p][i] * fix : 0.0f;
            // input gradient: dx[k][row] = sum_n W0[n][k] dh[n][row]; lane (m, g) receives k = 4 g .. 4 g + 3 of row m
            f32x4 dx = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) dx = __builtin_amdgcn_mfma_f32_16x16x4f32(awt[p][i], dh[p][i], dx, 0, 0, 0);
            if (ok)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int32_t k = 4 * g + r;
                    if (k < K0) a.denc[(int64_t)(k / F) * a.n * F + row * F + k % F] = dx[r];
                }
            // weight gradients: sums over the tile's rows, lane m keeping feature 16 (m >> 2) + 4 g + (m & 3)
            w1acc += row16_reduce_scatter(h, fix, m);
            b0acc += row16_reduce_scatter(dh, 1.0f, m);
            if (g == 0) b1acc += fix;
#pragma unroll
            for (int k = 0; k < 4 * KS4; ++k) {
                const float ek = __shfl(x[j][k >> 2], 16 * (k & 3) + m, 64);  // input feature k of this lane's row
                w0acc[k] += row16_reduce_scatter(dh, ek, m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // the waves' sums through LDS, one partial per workgroup
    const int nf = 16 * (m >> 2) + 4 * g + (m & 3);
#pragma unroll
    for (int k = 0; k < 4 * KS4; ++k)
        if (k < K0) part[wave][nf * K0 + k] = w0acc[k];
    part[wave][64 * K0 + nf] = b0acc;
    part[wave][64 * K0 + 64 + nf] = w1acc;
    b1acc = row16_sum(b1acc);
    if (lane == 0) part[wave][64 * K0 + 128] = b1acc;
    __syncthreads();
    const int total = 64 * K0 + 129;
    for (int i = threadIdx.x; i < total; i += (int)blockDim.x) {
        float t = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w][i];
        a.partials[(int64_t)blockIdx.x * a.stride + i] = t;
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
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    constexpr int KS0 = (KT0 + 1) / 2, NP1 = NL == 3 ? 4 : NTO;
    u32x4 *w0l = smem, *w1l = w0l + w3_units(4, KS0), *w2l = w1l + w3_units(NP1, 2);
    float *b0l = reinterpret_cast<float *>(w2l + (NL == 3 ? w3_units(NTO, 2) : 0)), *b1l = b0l + 64, *b2l = b1l + 64;
    stage_w3(w0l, 4, KS0, a.w0);
    stage_w3(w1l, NP1, 2, a.w1);
    if (NL == 3) stage_w3(w2l, NTO, 2, a.w2);
    stage_b(b0l, 64, a.b0, a.w0.n);
    stage_b(b1l, 64, a.b1, a.w1.n);
    if (NL == 3) stage_b(b2l, 64, a.b2, a.w2.n);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w0p = w3_at(w0l, 4, KS0, lane), w1p = w3_at(w1l, NP1, 2, lane), w2p = w3_at(w2l, NTO, 2, lane);
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
            Opd<KS0> xo;
            make_opd<KT0>(x, xo);
            f32x4 h[4];
            init_bias<4>(b0l, g, h);
            tgemm<KS0, 4>(w0p, xo, h);
            relu<4>(h);
            if (a.h1) st_rm<4>(a.h1 + row * 64, ok, g, h);
            Opd<2> ho;
            make_opd<4>(h, ho);
            if constexpr (NL == 3) {
                f32x4 h2[4];
                init_bias<4>(b1l, g, h2);
                tgemm<2, 4>(w1p, ho, h2);
                relu<4>(h2);
                if (a.h2) st_rm<4>(a.h2 + row * 64, ok, g, h2);
                make_opd<4>(h2, ho);
            }
            f32x4 o[NTO];
            init_bias<NTO>(NL == 3 ? b2l : b1l, g, o);
            tgemm<2, NTO>(NL == 3 ? w2p : w1p, ho, o);
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
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    constexpr int KSL = (NTO + 1) / 2;
    u32x4 *wll = smem, *w1l = wll + w3_units(4, KSL), *w0l = w1l + (NL == 3 ? w3_units(4, 2) : 0);
    stage_w3(wll, 4, KSL, a.wlt);
    if (NL == 3) stage_w3(w1l, 4, 2, a.w1t);
    stage_w3(w0l, KT0, 2, a.w0t);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 wlp = w3_at(wll, 4, KSL, lane), w1p = w3_at(w1l, 4, 2, lane), w0p = w3_at(w0l, KT0, 2, lane);
    const bool wide_in = (a.n_out & 3) == 0 && (a.ldd & 3) == 0;
    const int64_t n_tiles = (a.n + 15) >> 4;
    for (int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; t < n_tiles; t += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int64_t row = t * 16 + m;
        const bool ok = row < a.n;
        f32x4 d[NTO];
        if (wide_in) ld_rm_k<NTO>(a.dlast + row * a.ldd, ok, g, a.n_out, d);
        else ld_narrow<NTO>(a.dlast + row * a.ldd, ok, g, a.n_out, d);
        Opd<KSL> dop;
        make_opd<NTO>(d, dop);
        f32x4 da[4];
        zero<4>(da);
        tgemm<KSL, 4>(wlp, dop, da);
        if constexpr (NL == 3) {
            f32x4 mk2[4];
            ld_rm<4>(a.h2 + row * 64, ok, g, mk2);
            relu_mask<4>(da, mk2);
            st_rm<4>(a.dpre1 + row * 64, ok, g, da);
            Opd<2> dao;
            make_opd<4>(da, dao);
            f32x4 db[4];
            zero<4>(db);
            tgemm<2, 4>(w1p, dao, db);
#pragma unroll
            for (int p = 0; p < 4; ++p) da[p] = db[p];
        }
        f32x4 mk1[4];
        ld_rm<4>(a.h1 + row * 64, ok, g, mk1);
        relu_mask<4>(da, mk1);
        st_rm<4>(a.dpre0 + row * 64, ok, g, da);
        if (a.dx) {
            Opd<2> dao;
            make_opd<4>(da, dao);
            f32x4 de[KT0];
            zero<KT0>(de);
            tgemm<2, KT0>(w0p, dao, de);
            if constexpr (F == 0) st_rm_k<KT0>(a.dx + row * a.lddx, ok, g, a.k0, de);
            else st_lm<KT0, F>(a.dx, a.n, a.n_levels, row, ok, g, de);
        }
    }
}

// ------------------------------------------- plain 2- / 3-layer heads: backward with the weight gradients fused [r5]
// The flow MLP (xyzt encoding 40 -> 64 -> 64 -> 6) and the shadow head (64 -> 64 -> 1, sigmoid) ran their backward as rmlp_bwd + one
// streamed weight-gradient launch and one reduction per layer: 327 us for the flow MLP's 524 288-row call of a flow step, 414 us for the
// shadow head at 1 M rows, most of it h1 / h2 / dpre tensors making a round trip through HBM.  Here the whole backward is one kernel in
// the manner of neck_bwdw_kernel: the hidden layers are RECOMPUTED from the input tile (the forward stores no activations), sigmoid'
// is applied to the incoming gradient in registers, the rows are moved onto the reduction index by the matrix core (to_rows) and
// dW_last / dW1 / dW0 / the bias gradients stay in accumulator registers for the whole launch.  Outputs up to 16 wide (n_out <= 16).
// NL == 2: 8 waves, two per SIMD; NL == 3: 144 accumulator registers -> 4 waves, one per SIMD (as rgb_bwdw16_kernel).
struct RMlpBwdWArgs {
    const float *dout; int64_t ldd;   // [n][ldd >= n_out] gradient of the OUTPUT (after the final activation)
    const float *out; int64_t ldo;    // [n][ldo >= n_out] saved output (sigmoid: d_last = dout * out * (1 - out)); unused otherwise
    const float *x; int64_t ldx;      // the forward's input: row-major [n][ldx] (F == 0) or level-major [L][n][F]
    int64_t n; int32_t n_levels, k0, n_out, final_act;
    WSrc w0, w1;                      // forward weights for the recomputation: W0 (64 x k0), W1 (64 x 64, NL == 3)
    const float *b0, *b1;
    WSrc wlt, w1t, w0t;               // W_last^T (64 x n_out), W1^T (64 x 64, NL == 3), W0^T (k0 x 64)
    float *dx; int64_t lddx;          // gradient of the input in the input's own layout (null: not needed)
    float *partials; int64_t stride;  // [gridDim.x][stride]: dW_last [n_out][64] | db_last [n_out] | (dW1 [64][64] | db1 [64]) | dW0 [64][k0] | db0 [64]
};

template <int NL, int NTO> constexpr int rmlp_w_threads() { return (NL == 3 || NTO == 4) ? 256 : 512; }

// NTO = 1: outputs up to 16 wide (dW_last as four 16 x 16 tiles); NTO = 4 [r5]: outputs up to 64 wide (the feature heads' 64 -> 64 -> 64 -> E:
// dW_last as 32 x 32 blocks like dW1 -- 192 accumulator registers, one wave per SIMD)
template <int KT0, int F, int NL, int NTO>
__global__ __launch_bounds__((rmlp_w_threads<NL, NTO>()), (rmlp_w_threads<NL, NTO>() == 256 ? 1 : 2)) void rmlp_bwdw_kernel(const RMlpBwdWArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
    constexpr int K0P = 16 * KT0, KS0 = (KT0 + 1) / 2, QB = (KT0 + 1) / 2, NW = rmlp_w_threads<NL, NTO>() / 64, KSL = (NTO + 1) / 2;
    u32x4 *w0fl = smem, *w1fl = w0fl + w3_units(4, KS0), *wltl = w1fl + (NL == 3 ? w3_units(4, 2) : 0), *w1tl = wltl + w3_units(4, KSL),
          *w0tl = w1tl + (NL == 3 ? w3_units(4, 2) : 0);
    float *b0l = reinterpret_cast<float *>(w0tl + w3_units(KT0, 2)), *b1l = b0l + 64;
    stage_w3(w0fl, 4, KS0, a.w0);
    if (NL == 3) stage_w3(w1fl, 4, 2, a.w1);
    stage_w3(wltl, 4, KSL, a.wlt);
    if (NL == 3) stage_w3(w1tl, 4, 2, a.w1t);
    stage_w3(w0tl, KT0, 2, a.w0t);
    stage_b(b0l, 64, a.b0, 64);
    if (NL == 3) stage_b(b1l, 64, a.b1, 64);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, m = lane & 15, g = lane >> 4;
    const W3 w0fp = w3_at(w0fl, 4, KS0, lane), w1fp = w3_at(w1fl, 4, 2, lane), wlp = w3_at(wltl, 4, KSL, lane), w1tp = w3_at(w1tl, 4, 2, lane),
             w0tp = w3_at(w0tl, KT0, 2, lane);
    const SelE sel = make_sel(lane);
    // accumulators: dW_last [16][64] as 16 x 16 tiles (NTO = 1: lane (j, g), register r: row 4 g + r, column 16 b + j) or [64][64] as 32 x 32
    // blocks (NTO = 4); dW1 [64][64] and dW0 [64][K0P] as 32 x 32 blocks (lane (j, h), register r: row 32 P + 8 (r >> 2) + 4 h + (r & 3),
    // column 32 Q + j)
    f32x4 bwl[1][4];
    f32x16 bwlb[NTO == 4 ? 2 : 1][2], bw1[NL == 3 ? 2 : 1][2], bw0[2][QB];
    float abl[NTO], ab1[4], ab0[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bwl[0][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int P = 0; P < 2; ++P) {
#pragma unroll
        for (int Q = 0; Q < 2; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (NL == 3) bw1[NL == 3 ? P : 0][Q][r] = 0.0f;
                if (NTO == 4) bwlb[NTO == 4 ? P : 0][Q][r] = 0.0f;
            }
#pragma unroll
        for (int Q = 0; Q < QB; ++Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) bw0[P][Q][r] = 0.0f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) { ab1[p] = 0.0f; ab0[p] = 0.0f; }
#pragma unroll
    for (int p = 0; p < NTO; ++p) abl[p] = 0.0f;
    const int64_t n_tiles = (a.n + 15) >> 4, n_waves = (int64_t)gridDim.x * NW;
    const int64_t per_wave = (n_tiles + n_waves - 1) / n_waves;
    const int64_t t_begin = ((int64_t)blockIdx.x * NW + wave) * per_wave;
    const int64_t t_end = t_begin + per_wave < n_tiles ? t_begin + per_wave : n_tiles;
    float *stg = b1l + 64 + wave * (384 * KT0);   // per wave: the x operand tiles with the rows on the reduction index [KT0][3][64][2]
    u32x2 *park = reinterpret_cast<u32x2 *>(stg) + lane;
    const bool sig = a.final_act == EMER_ACT_SIGMOID;
    // next tile's inputs ride in registers (x: KT0 x 16 bytes, dout / out: the lane's columns of its row; lanes whose columns lie beyond
    // n_out load nothing); loads are unconditional in the row (rows past the end re-read row n - 1 and are zeroed at use)
    f32x4 xn[KT0], dn[NTO], on[NTO];
    auto issue = [&](int64_t tile) {
        const int64_t row0 = tile * 16;
        const int64_t row = row0 + m < a.n ? row0 + m : a.n - 1;
        if constexpr (F == 0) ld_rm_k<KT0>(a.x + row * a.ldx, true, g, a.k0, xn);
        else ld_lm_t<KT0, F>(a.x + row0 * F, (unsigned)a.n, a.n_levels, (unsigned)(row - row0), g, xn);
        const float *dp = a.dout + row * a.ldd, *op = a.out + row * a.ldo;
        if constexpr (NTO == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool col = 4 * g + i < a.n_out;
                dn[0][i] = col ? dp[4 * g + i] : 0.0f;
                on[0][i] = (col && sig) ? op[4 * g + i] : 0.0f;
            }
        } else {   // wide outputs: n_out and the leading dimensions are multiples of 4 (host check): 16-byte loads
            ld_rm_k<NTO>(dp, true, g, a.n_out, dn);
            if (sig) ld_rm_k<NTO>(op, true, g, a.n_out, on);
            else {
#pragma unroll
                for (int p = 0; p < NTO; ++p) on[p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    };
    if (t_begin < t_end) issue(t_begin);
    f32x4 dep[KT0];
    int64_t tile_prev = -1;
    auto store_dx = [&](int64_t tile) {
        const int64_t row = tile * 16 + m;
        if constexpr (F == 0) st_rm_k<KT0>(a.dx + row * a.lddx, row < a.n, g, a.k0, dep);
        else st_lm_t<KT0, F>(a.dx + tile * 16 * F, (unsigned)a.n, a.n_levels, (unsigned)m, row < a.n, g, dep);
    };
    for (int64_t tile = t_begin; tile < t_end; ++tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's inputs have landed (issued one tile ago)
        if (a.dx && tile_prev >= 0) store_dx(tile_prev);   // (stores count in vmcnt: issued right behind the wait, see neck_bwdw_kernel)
        f32x4 x[KT0], d[NTO], o[NTO];
#pragma unroll
        for (int b = 0; b < KT0; ++b) x[b] = xn[b];
#pragma unroll
        for (int p = 0; p < NTO; ++p) { d[p] = dn[p]; o[p] = on[p]; }
        issue(tile + 1 < t_end ? tile + 1 : tile);
        const bool ok = tile * 16 + m < a.n;
        // ---- forward recomputation: h1 (and h2), their relu bits, and the operands with the rows on the reduction index
        unsigned bits1 = 0u, bits2 = 0u;
        SwP hq1[2];   // h1 as two 32-feature blocks (B operand of dW1; NL == 3)
        SwT hl[4];    // the last hidden layer as four 16-feature tiles (B operand of dW_last, NTO == 1)
        SwP hql[2];   // ... or as two 32-feature blocks (NTO == 4)
        {
#pragma unroll
            for (int b = 0; b < KT0; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) x[b][i] = ok ? x[b][i] : 0.0f;
            Opd<KS0> xo;
            make_opd<KT0>(x, xo);
            f32x4 h[4];
            init_bias<4>(b0l, g, h);
            tgemm<KS0, 4, false>(w0fp, xo, h);
#pragma unroll
            for (int b = 0; b < KT0; ++b) {   // operand of dW0, needed at the end of the tile: parked in LDS
                const SwT t = to_rows<KS0>(xo, b, sel);
                park[(3 * b + 0) * 64] = t.h; park[(3 * b + 1) * 64] = t.m; park[(3 * b + 2) * 64] = t.l;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h[p][i] = (ok && h[p][i] > 0.0f) ? h[p][i] : 0.0f;   // relu; rows past the end contribute nothing
                    bits1 |= (h[p][i] > 0.0f ? 1u : 0u) << (4 * p + i);
                }
            Opd<2> ho;
            make_opd<4>(h, ho);
            if constexpr (NL == 3) {
#pragma unroll
                for (int q = 0; q < 2; ++q) hq1[q] = block32(to_rows<2>(ho, 2 * q, sel), to_rows<2>(ho, 2 * q + 1, sel));
                f32x4 h2[4];
                init_bias<4>(b1l, g, h2);
                tgemm<2, 4, false>(w1fp, ho, h2);
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h2[p][i] = (ok && h2[p][i] > 0.0f) ? h2[p][i] : 0.0f;
                        bits2 |= (h2[p][i] > 0.0f ? 1u : 0u) << (4 * p + i);
                    }
                make_opd<4>(h2, ho);
            }
            if constexpr (NTO == 1) {
#pragma unroll
                for (int b = 0; b < 4; ++b) hl[b] = to_rows<2>(ho, b, sel);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) hql[q] = block32(to_rows<2>(ho, 2 * q, sel), to_rows<2>(ho, 2 * q + 1, sel));
            }
        }
        // ---- gradient at the last pre-activation, dW_last, and down the chain
#pragma unroll
        for (int p = 0; p < NTO; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = ok ? d[p][i] : 0.0f;
                if (sig) v = v * o[p][i] * (1.0f - o[p][i]);
                d[p][i] = v;
            }
        f32x4 da[4];
        zero<4>(da);
        {
            Opd<KSL> dop;
            make_opd<NTO>(d, dop);
            if constexpr (NTO == 1) {
                const SwT dt[1] = {to_rows<KSL>(dop, 0, sel, &abl[0])};
                dw_tiles<1, 4>(bwl, dt, hl);
            } else {
#pragma unroll
                for (int P = 0; P < 2; ++P) {   // dW_last rows 32 P .. += d^T h_last
                    const SwT t0 = to_rows<KSL>(dop, 2 * P, sel, &abl[(2 * P) % NTO]), t1 = to_rows<KSL>(dop, 2 * P + 1, sel, &abl[(2 * P + 1) % NTO]);
                    dw_blocks<2>(bwlb[NTO == 4 ? P : 0], block32(t0, t1), hql);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tgemm<KSL, 4, false>(wlp, dop, da);
        }
        if constexpr (NL == 3) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) da[p][i] = ((bits2 >> (4 * p + i)) & 1u) ? da[p][i] : 0.0f;
            Opd<2> dao1;
            make_opd<4>(da, dao1);
#pragma unroll
            for (int P = 0; P < 2; ++P) {   // dW1 rows 32 P .. += dPre1^T h1
                const SwT t0 = to_rows<2>(dao1, 2 * P, sel, &ab1[2 * P]), t1 = to_rows<2>(dao1, 2 * P + 1, sel, &ab1[2 * P + 1]);
                dw_blocks<2>(bw1[NL == 3 ? P : 0], block32(t0, t1), hq1);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 db[4];
            zero<4>(db);
            tgemm<2, 4, false>(w1tp, dao1, db);
#pragma unroll
            for (int p = 0; p < 4; ++p) da[p] = db[p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) da[p][i] = ((bits1 >> (4 * p + i)) & 1u) ? da[p][i] : 0.0f;
        Opd<2> dao0;
        make_opd<4>(da, dao0);
        if (a.dx) {
            f32x4 de[KT0];
            zero<KT0>(de);
            tgemm<2, KT0, false>(w0tp, dao0, de);
#pragma unroll
            for (int b = 0; b < KT0; ++b) dep[b] = de[b];
        }
        tile_prev = tile;
        {   // dW0 += dPre0^T x
            SwP xq[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                SwT xa, xb;
                xa.h = park[(3 * (2 * q) + 0) * 64]; xa.m = park[(3 * (2 * q) + 1) * 64]; xa.l = park[(3 * (2 * q) + 2) * 64];
                if (2 * q + 1 < KT0) { xb.h = park[(3 * (2 * q + 1) + 0) * 64]; xb.m = park[(3 * (2 * q + 1) + 1) * 64]; xb.l = park[(3 * (2 * q + 1) + 2) * 64]; }
                else { xb.h = u32x2{0u, 0u}; xb.m = u32x2{0u, 0u}; xb.l = u32x2{0u, 0u}; }
                xq[q] = block32(xa, xb);
            }
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                const SwT t0 = to_rows<2>(dao0, 2 * P, sel, &ab0[2 * P]), t1 = to_rows<2>(dao0, 2 * P + 1, sel, &ab0[2 * P + 1]);
                dw_blocks<QB>(bw0[P], block32(t0, t1), xq);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.dx && tile_prev >= 0) store_dx(tile_prev);
    // ---- sum the waves through LDS (the weights are dead), one coalesced partial per workgroup
    __syncthreads();
    constexpr int NLR = 16 * NTO;   // rows of dW_last in the reduction buffer
    float *red = reinterpret_cast<float *>(smem);   // dWl [NLR][64] | dbl [NLR] | dW1 [64][64] | db1 [64] | dW0 [64][K0P] | db0 [64]
    float *rl = red, *rbl = rl + NLR * 64, *r1 = rbl + NLR, *rb1 = r1 + (NL == 3 ? 64 * 64 : 0), *r0 = rb1 + (NL == 3 ? 64 : 0), *rb0 = r0 + 64 * K0P;
#pragma unroll
    for (int p = 0; p < NTO; ++p) { abl[p] += __shfl_xor(abl[p], 16, 64); abl[p] += __shfl_xor(abl[p], 32, 64); }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ab1[p] += __shfl_xor(ab1[p], 16, 64); ab1[p] += __shfl_xor(ab1[p], 32, 64);
        ab0[p] += __shfl_xor(ab0[p], 16, 64); ab0[p] += __shfl_xor(ab0[p], 32, 64);
    }
    const int j32 = lane & 31, h32 = lane >> 5;
    for (int w = 0; w < NW; ++w) {
        if (wave == w) {
            if constexpr (NTO == 1) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *q = rl + (4 * g + r) * 64 + 16 * b + m;
                        *q = (w == 0) ? bwl[0][b][r] : *q + bwl[0][b][r];
                    }
            }
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                if constexpr (NTO == 4) {
#pragma unroll
                    for (int Q = 0; Q < 2; ++Q)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float *q = rl + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * 64 + 32 * Q + j32;
                            *q = (w == 0) ? bwlb[NTO == 4 ? P : 0][Q][r] : *q + bwlb[NTO == 4 ? P : 0][Q][r];
                        }
                }
                if constexpr (NL == 3) {
#pragma unroll
                    for (int Q = 0; Q < 2; ++Q)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float *q = r1 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * 64 + 32 * Q + j32;
                            *q = (w == 0) ? bw1[NL == 3 ? P : 0][Q][r] : *q + bw1[NL == 3 ? P : 0][Q][r];
                        }
                }
#pragma unroll
                for (int Q = 0; Q < QB; ++Q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (32 * Q + j32 < K0P) {
                            float *q = r0 + (32 * P + 8 * (r >> 2) + 4 * h32 + (r & 3)) * K0P + 32 * Q + j32;
                            *q = (w == 0) ? bw0[P][Q][r] : *q + bw0[P][Q][r];
                        }
                    }
            }
            if (g == 0) {
#pragma unroll
                for (int p = 0; p < NTO; ++p) rbl[16 * p + m] = (w == 0) ? abl[p] : rbl[16 * p + m] + abl[p];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (NL == 3) rb1[16 * p + m] = (w == 0) ? ab1[p] : rb1[16 * p + m] + ab1[p];
                    rb0[16 * p + m] = (w == 0) ? ab0[p] : rb0[16 * p + m] + ab0[p];
                }
            }
        }
        __syncthreads();
    }
    float *part = a.partials + (int64_t)blockIdx.x * a.stride;
    const int no = a.n_out;
    for (int i = threadIdx.x; i < no * 64; i += (int)blockDim.x) part[i] = rl[i];
    for (int i = threadIdx.x; i < no; i += (int)blockDim.x) part[no * 64 + i] = rbl[i];
    float *p1 = part + no * 65;
    if (NL == 3) {
        for (int i = threadIdx.x; i < 64 * 64 + 64; i += (int)blockDim.x) p1[i] = r1[i];   // (db1 follows dW1 in `red` as in the partial)
        p1 += 64 * 65;
    }
    for (int i = threadIdx.x; i < 64 * a.k0; i += (int)blockDim.x) { const int nn = i / a.k0, kk = i - nn * a.k0; p1[i] = r0[nn * K0P + kk]; }
    for (int i = threadIdx.x; i < 64; i += (int)blockDim.x) p1[64 * a.k0 + i] = rb0[i];
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
    if (n_out == 1 && k0 <= 16) {  // narrow input (the proposal networks): fp32 matrix instruction, weights in registers
        const int ks4 = (k0 + 3) / 4;
#define EMER_DENS(KS) hipLaunchKernelGGL(density_fwd_kernel<KS>, dim3(grid), dim3(kNThreads), 0, st, a, n_feat)
        if (ks4 == 1) EMER_DENS(1); else if (ks4 == 2) EMER_DENS(2); else if (ks4 == 3) EMER_DENS(3); else EMER_DENS(4);
#undef EMER_DENS
        return check_launch("neck_fwd");
    } else if (n_out == 1) {
        auto lds = [](int kt0) { return (size_t)(w3_units(4, (kt0 + 1) / 2) + w3_units(1, 2)) * 16 + (64 + 16 + 64) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_fwd_kernel, 1, a, lds, "neck_fwd");
    } else if (n_out == 64) {
        auto lds = [](int kt0) { return (size_t)(w3_units(4, (kt0 + 1) / 2) + w3_units(4, 2)) * 16 + (64 + 64 + 64) * sizeof(float); };
        EMER_NECK_DISPATCH(neck_fwd_kernel, 4, a, lds, "neck_fwd");
    } else {
        auto lds = [](int kt0) { return (size_t)(w3_units(4, (kt0 + 1) / 2) + w3_units(8, 2)) * 16 + (64 + 128 + 64) * sizeof(float); };
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
        auto lds = [](int kt0) { return (size_t)w3_units(kt0, 2) * 16 + 64 * sizeof(float); };
        EMER_NECK_DISPATCH(neck_bwd_kernel, 0, a, lds, "neck_bwd");
    } else if (n_out == 64 || !d1) {
        a.w1t = WSrc{w1, 1, 64, 64, 64};  // only the first 64 outputs carry a gradient
        auto lds = [](int kt0) { return (size_t)(w3_units(kt0, 2) + w3_units(4, 2)) * 16; };
        EMER_NECK_DISPATCH(neck_bwd_kernel, 4, a, lds, "neck_bwd");
    } else {
        a.w1t = WSrc{w1, 1, 64, 64, 128};
        auto lds = [](int kt0) { return (size_t)(w3_units(kt0, 2) + w3_units(4, 4)) * 16; };
        EMER_NECK_DISPATCH(neck_bwd_kernel, 8, a, lds, "neck_bwd");
    }
}

// ---- neck backward with the weight gradients fused [r3] --------------------------------------------------------------------
static inline uint32_t neck_bwdw_grid(int64_t n, int kt0, int no) {
    const int64_t tiles = (n + 15) / 16;
    const int nw = neckw_threads(kt0, no) / 64;
    int64_t blocks = (tiles + nw - 1) / nw;
    if (blocks > 256) blocks = 256;   // persistent: one workgroup per CU
    return (uint32_t)(blocks < 1 ? 1 : blocks);
}
static inline size_t neck_bwdw_lds(int kt0, int no) {   // weights + bias + a staging buffer of 4-8 KB per wave; the final reduction reuses the front
    const size_t w = (size_t)(w3_units(kt0, 2) + w3_units(4, 2 * no) + w3_units(4, (kt0 + 1) / 2)) * 16
                     + (size_t)(64 + (neckw_threads(kt0, no) / 64) * (1024 * no + 384 * kt0)) * sizeof(float);
    const size_t r = (size_t)(64 * no * 65 + 64 * 16 * kt0 + 64) * sizeof(float);
    return w > r ? w : r;
}
static inline int64_t neck_bwdw_stride(int k0, int n_out) { return (int64_t)n_out * 65 + 64 * (int64_t)k0 + 64; }

// 1 when emer_neck_bwd_fused covers this neck: 64 outputs (the geometry features) or [r5] 128 (geometry | semantic features of the feature
// configs); the density MLP, n_out == 1, has emer_density_bwd_fused
extern "C" int emer_neck_bwd_fused_supported(int32_t n_levels, int32_t n_feat, int32_t hidden, int32_t n_out) {
    return (emer_neck_supported(n_levels, n_feat, hidden, n_out) && (n_out == 64 || n_out == 128)) ? 1 : 0;
}
static inline bool neck_bwdw_fits(int32_t n_levels, int32_t n_feat, int64_t n) { return n * n_levels * n_feat < (1ll << 30); }  // 32-bit lane offsets
// floats of workspace emer_neck_bwd_fused needs (per-workgroup partial weight gradients)
extern "C" int64_t emer_neck_bwd_fused_workspace(int32_t n_levels, int32_t n_feat, int64_t n, int32_t n_out) {
    if (n <= 0 || !emer_neck_bwd_fused_supported(n_levels, n_feat, 64, n_out) || !neck_bwdw_fits(n_levels, n_feat, n)) return 0;
    return (int64_t)neck_bwdw_grid(n, (n_levels * n_feat + 15) / 16, n_out / 64) * neck_bwdw_stride(n_levels * n_feat, n_out);
}
// Backward of emer_neck_fwd (n_out == 64 or 128) INCLUDING the weight gradients: writes denc_lm [L][n][F]; ACCUMULATES (+=) dw0
// [64][ld_dw0 >= L*F], db0 [64], dw1 [n_out][ld_dw1 >= 64], db1 [n_out] (torch Linear layouts; what autograd's AccumulateGrad would
// add).  d0 / d1 [n][64]: gradients of output features 0..63 / 64..127 (either may be NULL = zero; d1 only with n_out == 128); ddens as
// in emer_neck_bwd.  enc_lm: the forward's input; the hidden activations are recomputed from it (the forward need not store them: pass
// h1 = NULL to emer_neck_fwd).
extern "C" int emer_neck_bwd_fused(const float *d0, const float *d1, const float *ddens, const float *dens, const float *enc_lm,
                                   int32_t n_levels, int32_t n_feat, int64_t n, const float *w0, const float *b0, const float *w1, int32_t n_out,
                                   float *denc_lm, float *workspace, float *dw0, int64_t ld_dw0, float *db0, float *dw1, int64_t ld_dw1,
                                   float *db1, void *stream) {
    EMER_REQUIRE(n >= 0, "neck_bwd_fused: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(emer_neck_bwd_fused_supported(n_levels, n_feat, 64, n_out), "neck_bwd_fused: unsupported shape L=%d F=%d n_out=%d", n_levels, n_feat, n_out);
    EMER_REQUIRE(neck_bwdw_fits(n_levels, n_feat, n), "neck_bwd_fused: n * L * F must stay below 2^30 (32-bit lane offsets); split the batch or use emer_neck_bwd");
    EMER_REQUIRE(enc_lm && w0 && b0 && w1 && denc_lm && workspace && dw0 && db0 && dw1 && db1, "neck_bwd_fused: null pointer");
    EMER_REQUIRE(!ddens || dens, "neck_bwd_fused: ddens needs the saved density");
    EMER_REQUIRE(!d1 || n_out == 128, "neck_bwd_fused: d1 is the gradient of outputs 64..127 of a 128-output neck");
    const int k0 = n_levels * n_feat;
    EMER_REQUIRE(ld_dw0 >= k0 && ld_dw1 >= 64, "neck_bwd_fused: leading dimension smaller than the row");
    NeckBwdWArgs a;
    a.d0 = d0; a.d1 = d1; a.ddens = ddens; a.dens = dens; a.enc = enc_lm; a.b0 = b0; a.n = n; a.n_levels = n_levels; a.k0 = k0;
    a.w0 = WSrc{w0, k0, 1, 64, k0};
    a.w0t = WSrc{w0, 1, k0, k0, 64};          // (n = input feature, k = hidden) = w0[k][n]
    a.w1t = WSrc{w1, 1, 64, 64, n_out};       // (n = hidden, k = output) = w1[k][n]
    a.denc = denc_lm; a.partials = workspace; a.stride = neck_bwdw_stride(k0, n_out);
    hipStream_t st = as_stream(stream);
    const int kt0 = (k0 + 15) / 16, no = n_out / 64;
    const uint32_t grid = neck_bwdw_grid(n, kt0, no);
    const size_t lds = neck_bwdw_lds(kt0, no);
    int rc = EMER_E_INVALID;
    auto go = [&](auto kern) {
        if (int r = set_lds(kern, lds, "neck_bwd_fused")) return r;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(neckw_threads(kt0, no)), lds, st, a);
        return check_launch("neck_bwd_fused");
    };
#define EMER_NBW(FF, NN) (kt0 == 1 ? go(neck_bwdw_kernel<1, FF, NN>) : kt0 == 2 ? go(neck_bwdw_kernel<2, FF, NN>) : kt0 == 3 ? go(neck_bwdw_kernel<3, FF, NN>) : go(neck_bwdw_kernel<4, FF, NN>))
    if (no == 1) { if (n_feat == 1) rc = EMER_NBW(1, 1); else if (n_feat == 2) rc = EMER_NBW(2, 1); else if (n_feat == 4) rc = EMER_NBW(4, 1); else rc = EMER_NBW(8, 1); }
    else { if (n_feat == 1) rc = EMER_NBW(1, 2); else if (n_feat == 2) rc = EMER_NBW(2, 2); else if (n_feat == 4) rc = EMER_NBW(4, 2); else rc = EMER_NBW(8, 2); }
#undef EMER_NBW
    if (rc) return rc;
    const DwReduceJob jb[2] = {{0, n_out, 64, dw1, ld_dw1, db1, 0, {0}, {0}, {0}}, {(int64_t)n_out * 65, 64, k0, dw0, ld_dw0, db0, 0, {0}, {0}, {0}}};
    return launch_dw_reduce_multi(workspace, (int32_t)grid, a.stride, 2, jb, st);
}

// Backward of the density MLP (emer_neck_fwd with n_out = 1) INCLUDING the weight gradients, for narrow inputs (L F <= 16: the
// proposal networks).  Writes denc_lm [L][n][F]; ACCUMULATES dw0 [64][ld_dw0 >= L F], db0 [64], dw1 [1][64], db1 [1].  The hidden
// layer is recomputed from enc_lm (pass h1 = NULL to emer_neck_fwd).
static inline int64_t dens_bwdw_stride(int k0) { return ((int64_t)64 * k0 + 129 + 3) / 4 * 4; }
extern "C" int64_t emer_density_bwd_fused_workspace(int32_t n_levels, int32_t n_feat, int64_t n) {
    const int k0 = n_levels * n_feat;
    if (n <= 0 || k0 < 1 || k0 > 16 || !emer_neck_supported(n_levels, n_feat, 64, 1)) return 0;
    return (int64_t)fused_grid(((n + 15) / 16 + kNeckChunk - 1) / kNeckChunk, kNThreads) * dens_bwdw_stride(k0);
}
extern "C" int emer_density_bwd_fused(const float *ddens, const float *dens, const float *enc_lm, int32_t n_levels, int32_t n_feat, int64_t n,
                                      const float *w0, const float *b0, const float *w1, float *denc_lm, float *workspace, float *dw0,
                                      int64_t ld_dw0, float *db0, float *dw1, float *db1, void *stream) {
    EMER_REQUIRE(n >= 0, "density_bwd_fused: negative n");
    if (n == 0) return EMER_OK;
    const int k0 = n_levels * n_feat;
    EMER_REQUIRE(emer_density_bwd_fused_workspace(n_levels, n_feat, n) > 0, "density_bwd_fused: unsupported shape L=%d F=%d (needs L F <= 16)", n_levels, n_feat);
    EMER_REQUIRE(ddens && dens && enc_lm && w0 && w1 && denc_lm && workspace && dw0 && db0 && dw1 && db1 && ld_dw0 >= k0, "density_bwd_fused: bad arguments");
    DensBwdWArgs a;
    a.ddens = ddens; a.dens = dens; a.enc = enc_lm; a.n = n; a.n_levels = n_levels;
    a.w0 = WSrc{w0, k0, 1, 64, k0};
    a.b0 = b0; a.w1 = w1; a.denc = denc_lm; a.partials = workspace; a.stride = dens_bwdw_stride(k0);
    hipStream_t st = as_stream(stream);
    const uint32_t grid = fused_grid(((n + 15) / 16 + kNeckChunk - 1) / kNeckChunk, kNThreads);
    const int ks4 = (k0 + 3) / 4;
#define EMER_DBW(KS) hipLaunchKernelGGL(density_bwdw_kernel<KS>, dim3(grid), dim3(kNThreads), 0, st, a, n_feat)
    if (ks4 == 1) EMER_DBW(1); else if (ks4 == 2) EMER_DBW(2); else if (ks4 == 3) EMER_DBW(3); else EMER_DBW(4);
#undef EMER_DBW
    if (int rc = check_launch("density_bwd_fused")) return rc;
    const DwReduceJob jb[2] = {{0, 64, k0, dw0, ld_dw0, db0, 0, {0}, {0}, {0}}, {64 * k0 + 64, 1, 64, dw1, 64, db1, 0, {0}, {0}, {0}}};
    return launch_dw_reduce_multi(workspace, (int32_t)grid, a.stride, 2, jb, st);
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
    EMER_REQUIRE(geo && rb0 && rb1 && w0 && w1 && w2 && (a2 == nullptr || a1 != nullptr) && out && ld_geo >= 64 && ld_geo % 4 == 0 && ld_rb >= 64 && ld_rb % 4 == 0,
                 "rgb_head_fwd: bad arguments");
    RgbFwdArgs a;
    a.geo = geo; a.ld_geo = ld_geo; a.rb0 = rb0; a.rb1 = rb1; a.ld_rb = ld_rb; a.tiles_per_ray = samples_per_ray / 16; a.n_rays = n_rays;
    const int64_t k0 = kh + 64, k1 = 64 + k0;
    a.w0g = WSrc{w0 + kh, k0, 1, 64, 64};
    a.w1a = WSrc{w1, k1, 1, 64, 64};
    a.w1g = WSrc{w1 + 64 + kh, k1, 1, 64, 64};
    a.w2 = WSrc{w2, 64, 1, 3, 64};
    a.b2 = b2; a.a1 = a1; a.a2 = a2; a.out = out;
    const size_t lds = (size_t)(3 * w3_units(4, 2) + w3_units(1, 2)) * 16 + 16 * sizeof(float);
    if (int rc = set_lds(rgb_fwd_kernel, lds, "rgb_head_fwd")) return rc;
    hipLaunchKernelGGL(rgb_fwd_kernel, dim3(fused_grid(n_rays, kNThreads)), dim3(kNThreads), lds, as_stream(stream), a);
    return check_launch("rgb_head_fwd");
}

// 1 when emer_field_fwd covers a neck of this input shape (64 hidden, 64 geometry features) followed by the rgb head
extern "C" int emer_field_fwd_supported(int32_t n_levels, int32_t n_feat) {
    const int k0 = n_levels * n_feat;
    return ((n_feat == 2 || n_feat == 4) && k0 >= 1 && k0 <= 64) ? 1 : 0;
}

// neck (emer_neck_fwd with n_out = 64, no hidden layer stored) + rgb head (emer_rgb_head_fwd) in one launch; rows of ray r are
// r*S .. r*S+S-1.  Outputs: geo [n][64], dens [n], a1 / a2 [n][64] (both NULL: inference), out [n][3].
extern "C" int emer_field_fwd(const float *enc_lm, int32_t n_levels, int32_t n_feat, int64_t n_rays, int32_t samples_per_ray,
                              const float *nw0, const float *nb0, const float *nw1, const float *nb1, const float *rb0, const float *rb1,
                              int64_t ld_rb, int32_t kh, const float *w0, const float *w1, const float *w2, const float *b2, float *geo,
                              float *dens, float *a1, float *a2, float *out, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && samples_per_ray >= 16 && samples_per_ray % 16 == 0 && kh >= 0, "field_fwd: bad sizes (S must be a multiple of 16)");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(emer_field_fwd_supported(n_levels, n_feat), "field_fwd: unsupported encoding L=%d F=%d", n_levels, n_feat);
    EMER_REQUIRE(enc_lm && nw0 && nw1 && rb0 && rb1 && w0 && w1 && w2 && geo && dens && out && (a2 == nullptr || a1 != nullptr) &&
                 ld_rb >= 64 && ld_rb % 4 == 0, "field_fwd: bad arguments");
    const int64_t n = n_rays * samples_per_ray;
    const int k0 = n_levels * n_feat;
    EMER_REQUIRE(n * k0 < ((int64_t)1 << 30), "field_fwd: n * L * F must stay below 2^30 (32-bit lane offsets)");
    FieldFwdArgs a;
    a.enc = enc_lm; a.n = n; a.n_levels = n_levels;
    a.nw0 = WSrc{nw0, k0, 1, 64, k0};
    a.nw1 = WSrc{nw1, 64, 1, 64, 64};
    a.nb0 = nb0; a.nb1 = nb1; a.geo = geo; a.dens = dens;
    a.r.geo = nullptr; a.r.ld_geo = 64; a.r.rb0 = rb0; a.r.rb1 = rb1; a.r.ld_rb = ld_rb; a.r.tiles_per_ray = samples_per_ray / 16; a.r.n_rays = n_rays;
    const int64_t kk0 = kh + 64, kk1 = 64 + kk0;
    a.r.w0g = WSrc{w0 + kh, kk0, 1, 64, 64};
    a.r.w1a = WSrc{w1, kk1, 1, 64, 64};
    a.r.w1g = WSrc{w1 + 64 + kh, kk1, 1, 64, 64};
    a.r.w2 = WSrc{w2, 64, 1, 3, 64};
    a.r.b2 = b2; a.r.a1 = a1; a.r.a2 = a2; a.r.out = out;
    const int kt0 = (k0 + 15) / 16;
    const size_t lds = (size_t)(w3_units(4, (kt0 + 1) / 2) + 4 * w3_units(4, 2) + w3_units(1, 2)) * 16 + (64 + 64 + 16) * sizeof(float);
    hipStream_t st = as_stream(stream);
    int64_t blocks = (n_rays + kFieldThreads / 64 - 1) / (kFieldThreads / 64);
    const uint32_t grid = (uint32_t)(blocks > 256 ? 256 : blocks);  // persistent: one workgroup per CU
    int rc = EMER_E_INVALID;
    auto go = [&](auto kern) {
        if (int r = set_lds(kern, lds, "field_fwd")) return r;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kFieldThreads), lds, st, a);
        return check_launch("field_fwd");
    };
    if (n_feat == 2) { if (kt0 == 1) rc = go(field_fwd_kernel<1, 2>); else if (kt0 == 2) rc = go(field_fwd_kernel<2, 2>); else if (kt0 == 3) rc = go(field_fwd_kernel<3, 2>); else rc = go(field_fwd_kernel<4, 2>); }
    else { if (kt0 == 1) rc = go(field_fwd_kernel<1, 4>); else if (kt0 == 2) rc = go(field_fwd_kernel<2, 4>); else if (kt0 == 3) rc = go(field_fwd_kernel<3, 4>); else rc = go(field_fwd_kernel<4, 4>); }
    return rc;
}

// rgb head data-gradient chain.  Writes dpre2 [n][3], dpre1 / dpre0 / dgeo [n][64] and the per-ray sums s1 / s0
// [rays][64] of dpre1 / dpre0 (everything the per-ray operands hray, W0h, W1h, b0, b1 need).
extern "C" int emer_rgb_head_bwd(const float *dout, const float *out, const float *a1, const float *a2, int64_t n_rays,
                                 int32_t samples_per_ray, int32_t kh, const float *w0, const float *w1, const float *w2,
                                 float *dpre2, float *dpre1, float *dpre0, float *dgeo, float *s1, float *s0, float *workspace,
                                 float *dw2, int64_t ld_dw2, float *db2, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && samples_per_ray >= 16 && samples_per_ray % 16 == 0 && kh >= 0, "rgb_head_bwd: bad sizes (S must be a multiple of 16)");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(dout && out && a1 && a2 && w0 && w1 && w2 && dpre2 && dpre1 && dpre0 && dgeo && s1 && s0, "rgb_head_bwd: null pointer");
    EMER_REQUIRE(!dw2 || (workspace && ld_dw2 >= 64), "rgb_head_bwd: dw2 needs emer_rgb_head_bwd_workspace floats of workspace and ld_dw2 >= 64");
    RgbBwdArgs a;
    a.dout = dout; a.out = out; a.a1 = a1; a.a2 = a2; a.tiles_per_ray = samples_per_ray / 16; a.n_rays = n_rays;
    const int64_t k0 = kh + 64, k1 = 64 + k0;
    a.w2t = WSrc{w2, 1, 64, 64, 3};               // (n = hidden, k = channel) = w2[k][n]
    a.w1at = WSrc{w1, 1, k1, 64, 64};             // (n = a1 feature, k = layer-1 output) = w1[k][n]
    a.w1gt = WSrc{w1 + 64 + kh, 1, k1, 64, 64};
    a.w0gt = WSrc{w0 + kh, 1, k0, 64, 64};
    a.dpre2 = dpre2; a.dpre1 = dpre1; a.dpre0 = dpre0; a.dgeo = dgeo; a.s1 = s1; a.s0 = s0;
    a.w2part = dw2 ? workspace : nullptr;
    const size_t lds = (size_t)(3 * w3_units(4, 2)) * 16;
    if (int rc = set_lds(rgb_bwd_kernel, lds, "rgb_head_bwd")) return rc;
    const uint32_t grid = fused_grid(n_rays, kNThreads);
    hipLaunchKernelGGL(rgb_bwd_kernel, dim3(grid), dim3(kNThreads), lds, as_stream(stream), a);
    if (int rc = check_launch("rgb_head_bwd")) return rc;
    if (dw2) return launch_dw_reduce(workspace, (int32_t)grid, 196, 3, 64, dw2, ld_dw2, db2, as_stream(stream));
    return EMER_OK;
}

// ---- rgb head backward with the weight gradients of layers 0 / 1 (per-sample column blocks) and 2 fused [r4] -------------------------
static inline uint32_t rgb_bwdw_grid(int64_t n_rays) {
    int64_t blocks = (n_rays + kRWThreads / 64 - 1) / (kRWThreads / 64);
    if (blocks > 256) blocks = 256;   // persistent: one 4-wave workgroup per CU (one wave per SIMD)
    return (uint32_t)(blocks < 1 ? 1 : blocks);
}
static constexpr int64_t kRgbBwdWStride = 64 * 128 + 64 * 64 + 196;
// 1 when emer_rgb_head_bwd_fused covers the shape (whole 16-row tiles per ray, as emer_rgb_head_bwd)
extern "C" int emer_rgb_head_bwd_fused_supported(int32_t samples_per_ray) {
    return (samples_per_ray >= 16 && samples_per_ray % 16 == 0) ? 1 : 0;
}

extern "C" int64_t emer_rgb_head_bwd_fused_workspace(int64_t n_rays, int32_t samples_per_ray) {
    if (n_rays <= 0 || !emer_rgb_head_bwd_fused_supported(samples_per_ray)) return 0;
    return (int64_t)rgb_bwdw_grid(n_rays) * kRgbBwdWStride;
}
// Backward of emer_rgb_head_fwd INCLUDING the weight gradients of the per-sample column blocks: writes dgeo [n][64] and the per-ray
// sums s1 / s0 [rays][64] of dpre1 / dpre0 (what the per-ray operands need: emer_ray_wgrad, emer_ray_pre_bwd); ACCUMULATES (+=)
//   dw1[:, 0:64] += dpre1^T a1,  dw1[:, 64 + kh : 128 + kh] += dpre1^T geo   (dw1 [64][ld_dw1 >= 128 + kh], torch layout of layers.1.weight),
//   dw0[:, kh : kh + 64] += dpre0^T geo                                        (dw0 [64][ld_dw0 >= 64 + kh]),
//   dw2 [3][ld_dw2 >= 64] += dpre2^T a2,  db2 [3] += column sums of dpre2.
// Neither dpre1 nor dpre0 (nor dpre2) is written.  geo [n][ld_geo] is the forward's input.
extern "C" int emer_rgb_head_bwd_fused(const float *dout, const float *out, const float *a1, const float *a2, const float *geo, int64_t ld_geo,
                                       int64_t n_rays, int32_t samples_per_ray, int32_t kh, const float *w0, const float *w1, const float *w2,
                                       float *dgeo, float *s1, float *s0, float *workspace, float *dw0, int64_t ld_dw0, float *dw1,
                                       int64_t ld_dw1, float *dw2, int64_t ld_dw2, float *db2, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && kh >= 0, "rgb_head_bwd_fused: bad sizes");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(emer_rgb_head_bwd_fused_supported(samples_per_ray), "rgb_head_bwd_fused: samples_per_ray must be a multiple of 16 (got %d)", samples_per_ray);
    EMER_REQUIRE(dout && out && a1 && a2 && geo && w0 && w1 && w2 && dgeo && s1 && s0 && workspace && dw0 && dw1 && dw2 && db2,
                 "rgb_head_bwd_fused: null pointer");
    EMER_REQUIRE(ld_geo >= 64 && ld_geo % 4 == 0 && ld_dw0 >= 64 + kh && ld_dw1 >= 128 + kh && ld_dw2 >= 64, "rgb_head_bwd_fused: bad leading dimension");
    EMER_REQUIRE(n_rays * samples_per_ray * ld_geo < ((int64_t)1 << 40), "rgb_head_bwd_fused: batch too large");
    RgbBwdWArgs a;
    a.dout = dout; a.out = out; a.a1 = a1; a.a2 = a2; a.geo = geo; a.ld_geo = ld_geo; a.tiles_per_ray = samples_per_ray / 16; a.n_rays = n_rays;
    const int64_t k0 = kh + 64, k1 = 64 + k0;
    a.w2t = WSrc{w2, 1, 64, 64, 3};
    a.w1at = WSrc{w1, 1, k1, 64, 64};
    a.w1gt = WSrc{w1 + 64 + kh, 1, k1, 64, 64};
    a.w0gt = WSrc{w0 + kh, 1, k0, 64, 64};
    a.dgeo = dgeo; a.s1 = s1; a.s0 = s0; a.partials = workspace; a.stride = kRgbBwdWStride;
    const size_t lds = (size_t)(3 * w3_units(4, 2)) * 16 + (size_t)(kRWThreads / 64) * (2048 + 3072) * sizeof(float);
    hipStream_t st = as_stream(stream);
    const uint32_t grid = rgb_bwdw_grid(n_rays);
    if (int rc = set_lds(rgb_bwdw16_kernel, lds, "rgb_head_bwd_fused")) return rc;
    hipLaunchKernelGGL(rgb_bwdw16_kernel, dim3(grid), dim3(kRWThreads), lds, st, a);
    if (int rc = check_launch("rgb_head_bwd_fused")) return rc;
    // the workgroups' partials -> the parameters' gradients (+=): dW1's two column blocks land 0.. and 64 + kh.., dW0's at kh..
    // (one launch for the three)
    const DwReduceJob jb[3] = {{0, 64, 128, dw1, ld_dw1, nullptr, 2, {0, 64}, {64, 64}, {0, 64 + kh}},
                               {64 * 128, 64, 64, dw0, ld_dw0, nullptr, 1, {0}, {64}, {kh}},
                               {64 * 128 + 64 * 64, 3, 64, dw2, ld_dw2, db2, 0, {0}, {0}, {0}}};
    return launch_dw_reduce_multi(workspace, (int32_t)grid, kRgbBwdWStride, 3, jb, st);
}

// [r6] emer_rgb_head_bwd_fused WITHOUT saved activations: a1 / a2 are recomputed from geo and the per-ray pre-activations rb0 / rb1
// ([rays][64], row stride ld_rb: what emer_ray_pre_fwd wrote for the forward), bitwise the forward's values; every output is bitwise
// emer_rgb_head_bwd_fused's.  The forward can then be called with a1 = a2 = NULL.  Workspace: emer_rgb_head_bwd_fused_workspace.
extern "C" int emer_rgb_head_bwd_recompute(const float *dout, const float *out, const float *a1, const float *geo, int64_t ld_geo, const float *rb0,
                                           const float *rb1, int64_t ld_rb, int64_t n_rays, int32_t samples_per_ray, int32_t kh,
                                           const float *w0, const float *w1, const float *w2, float *dgeo, float *s1, float *s0,
                                           float *workspace, float *dw0, int64_t ld_dw0, float *dw1, int64_t ld_dw1, float *dw2,
                                           int64_t ld_dw2, float *db2, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && kh >= 0, "rgb_head_bwd_recompute: bad sizes");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(emer_rgb_head_bwd_fused_supported(samples_per_ray), "rgb_head_bwd_recompute: samples_per_ray must be a multiple of 16 (got %d)", samples_per_ray);
    EMER_REQUIRE(dout && out && geo && rb0 && rb1 && w0 && w1 && w2 && dgeo && s1 && s0 && workspace && dw0 && dw1 && dw2 && db2,
                 "rgb_head_bwd_recompute: null pointer");
    EMER_REQUIRE(ld_geo >= 64 && ld_geo % 4 == 0 && ld_rb >= 64 && ld_dw0 >= 64 + kh && ld_dw1 >= 128 + kh && ld_dw2 >= 64,
                 "rgb_head_bwd_recompute: bad leading dimension");
    EMER_REQUIRE(n_rays * samples_per_ray * ld_geo < ((int64_t)1 << 40), "rgb_head_bwd_recompute: batch too large");
    RgbBwdRArgs a;
    a.dout = dout; a.out = out; a.a1 = a1; a.geo = geo; a.ld_geo = ld_geo; a.rb0 = rb0; a.rb1 = rb1; a.ld_rb = ld_rb;
    a.tiles_per_ray = samples_per_ray / 16; a.n_rays = n_rays;
    const int64_t k0 = kh + 64, k1 = 64 + k0;
    a.w0g = WSrc{w0 + kh, k0, 1, 64, 64};
    a.w1a = WSrc{w1, k1, 1, 64, 64};
    a.w1g = WSrc{w1 + 64 + kh, k1, 1, 64, 64};
    a.w2t = WSrc{w2, 1, 64, 64, 3};
    a.w1at = WSrc{w1, 1, k1, 64, 64};
    a.w1gt = WSrc{w1 + 64 + kh, 1, k1, 64, 64};
    a.w0gt = WSrc{w0 + kh, 1, k0, 64, 64};
    a.dgeo = dgeo; a.s1 = s1; a.s0 = s0; a.partials = workspace; a.stride = kRgbBwdWStride;
    const size_t lds = (size_t)(6 * w3_units(4, 2)) * 16 + (size_t)(kRWThreads / 64) * 256 * sizeof(float);
    hipStream_t st = as_stream(stream);
    const uint32_t grid = rgb_bwdw_grid(n_rays);
    if (a1) {   // the cheaper half: a1 stored by the forward, a2 recomputed
        if (int rc = set_lds(rgb_bwdwr_kernel<true>, lds, "rgb_head_bwd_recompute")) return rc;
        hipLaunchKernelGGL(rgb_bwdwr_kernel<true>, dim3(grid), dim3(kRWThreads), lds, st, a);
    } else {
        if (int rc = set_lds(rgb_bwdwr_kernel<false>, lds, "rgb_head_bwd_recompute")) return rc;
        hipLaunchKernelGGL(rgb_bwdwr_kernel<false>, dim3(grid), dim3(kRWThreads), lds, st, a);
    }
    if (int rc = check_launch("rgb_head_bwd_recompute")) return rc;
    const DwReduceJob jb[3] = {{0, 64, 128, dw1, ld_dw1, nullptr, 2, {0, 64}, {64, 64}, {0, 64 + kh}},
                               {64 * 128, 64, 64, dw0, ld_dw0, nullptr, 1, {0}, {64}, {kh}},
                               {64 * 128 + 64 * 64, 3, 64, dw2, ld_dw2, db2, 0, {0}, {0}, {0}}};
    return launch_dw_reduce_multi(workspace, (int32_t)grid, kRgbBwdWStride, 3, jb, st);
}

// floats of workspace emer_rgb_head_bwd needs when it also produces dw2 / db2
extern "C" int64_t emer_rgb_head_bwd_workspace(int64_t n_rays) {
    return n_rays <= 0 ? 0 : (int64_t)fused_grid(n_rays, kNThreads) * 196;
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
    const size_t lds = (size_t)(w3_units(4, (kt0i + 1) / 2) + w3_units(n_layers == 3 ? 4 : nto, 2) + (n_layers == 3 ? w3_units(nto, 2) : 0)) * 16 + 3 * 64 * sizeof(float);
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
    const size_t lds = (size_t)(w3_units(4, (nto + 1) / 2) + (n_layers == 3 ? w3_units(4, 2) : 0) + w3_units(kt0i, 2)) * 16;
    hipStream_t st = as_stream(stream);
    const uint32_t grid = fused_grid((n + 15) / 16, kNThreads);
    EMER_RMLP_DISPATCH(rmlp_bwd_kernel, a, lds, "rmlp_bwd");
}

// ---- plain 2- / 3-layer heads: backward with the weight gradients fused [r5] ------------------------------------------------
static inline int rmlp_bwdw_nto(int n_out) { return n_out <= 16 ? 1 : 4; }
static inline int rmlp_bwdw_threads(int n_layers, int n_out) { return (n_layers == 3 || rmlp_bwdw_nto(n_out) == 4) ? 256 : 512; }
static inline uint32_t rmlp_bwdw_grid(int64_t n, int n_layers, int n_out) {
    const int nw = rmlp_bwdw_threads(n_layers, n_out) / 64;
    const int64_t tiles = (n + 15) / 16;
    int64_t blocks = (tiles + nw - 1) / nw;
    if (blocks > 256) blocks = 256;   // persistent: one workgroup per CU
    return (uint32_t)(blocks < 1 ? 1 : blocks);
}
static inline int64_t rmlp_bwdw_stride(int n_layers, int k0, int n_out) {
    return ((int64_t)n_out * 65 + (n_layers == 3 ? 64 * 65 : 0) + 64 * (int64_t)k0 + 64 + 3) / 4 * 4;
}
static inline size_t rmlp_bwdw_lds(int n_layers, int kt0, int n_out) {
    const int nw = rmlp_bwdw_threads(n_layers, n_out) / 64, nto = rmlp_bwdw_nto(n_out);
    const size_t w = (size_t)(w3_units(4, (kt0 + 1) / 2) + w3_units(4, (nto + 1) / 2) + w3_units(kt0, 2) + (n_layers == 3 ? 2 * w3_units(4, 2) : 0)) * 16
                     + (size_t)(128 + nw * 384 * kt0) * sizeof(float);
    const size_t r = (size_t)(16 * nto * 65 + (n_layers == 3 ? 64 * 64 + 64 : 0) + 64 * 16 * kt0 + 64) * sizeof(float);
    return w > r ? w : r;
}
// 1 when emer_rmlp_bwd_fused covers this stack: what emer_rmlp_fwd covers with at most 16 outputs (the flow MLP's 6, the shadow head's 1),
// or three layers on a row-major input with up to 64 outputs in multiples of 4 (the feature heads 64 -> 64 -> 64 -> E)
extern "C" int emer_rmlp_bwd_fused_supported(int32_t n_layers, int32_t k0, int32_t n_feat, int32_t hidden, int32_t n_out) {
    if (!emer_rmlp_supported(n_layers, k0, n_feat, hidden, n_out)) return 0;
    if (n_out <= 16) return 1;
    return (n_layers == 3 && n_feat == 0 && n_out % 4 == 0 && n_out <= 64) ? 1 : 0;
}
// floats of workspace emer_rmlp_bwd_fused needs (per-workgroup partial weight gradients); 0: not supported for this call
extern "C" int64_t emer_rmlp_bwd_fused_workspace(int32_t n_layers, int32_t k0, int32_t n_feat, int64_t n, int32_t n_out) {
    if (n <= 0 || !emer_rmlp_bwd_fused_supported(n_layers, k0, n_feat, 64, n_out)) return 0;
    if (n_feat != 0 && n * k0 >= (1ll << 30)) return 0;   // level-major input: 32-bit lane offsets
    return (int64_t)rmlp_bwdw_grid(n, n_layers, n_out) * rmlp_bwdw_stride(n_layers, k0, n_out);
}
// Backward of emer_rmlp_fwd INCLUDING the weight gradients.  dout [n][ldd]: gradient of the OUTPUT (sigmoid' is applied here from the
// saved `out` [n][ldo]; final_act none: `out` may be NULL).  x: the forward's input (the hidden layers are recomputed from it: pass
// h1 = h2 = NULL to emer_rmlp_fwd).  Writes dx in x's layout when non-null; ACCUMULATES (+=) dw0 [64][ld_dw0 >= k0], db0 [64],
// dw1 / db1 (three layers: [64][ld_dw1 >= 64], [64]; two layers: the output layer [n_out][ld_dw1 >= 64], [n_out]) and, for three layers,
// dw2 [n_out][ld_dw2 >= 64], db2 [n_out] -- torch Linear layouts.  Bias pointers may be NULL (layer without a bias gradient).
extern "C" int emer_rmlp_bwd_fused(const float *dout, int64_t ldd, const float *out, int64_t ldo, const float *x, int64_t ldx, int32_t n_levels,
                                   int32_t n_feat, int32_t k0, int64_t n, int32_t n_layers, const float *w0, const float *b0, const float *w1,
                                   const float *b1, const float *w2, int32_t n_out, int32_t final_act, float *dx, int64_t lddx, float *workspace,
                                   float *dw0, int64_t ld_dw0, float *db0, float *dw1, int64_t ld_dw1, float *db1, float *dw2, int64_t ld_dw2,
                                   float *db2, void *stream) {
    EMER_REQUIRE(n >= 0, "rmlp_bwd_fused: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(emer_rmlp_bwd_fused_supported(n_layers, k0, n_feat, 64, n_out), "rmlp_bwd_fused: unsupported stack (layers=%d k0=%d F=%d n_out=%d)", n_layers, k0, n_feat, n_out);
    EMER_REQUIRE(dout && x && w0 && b0 && w1 && workspace && dw0 && dw1 && ldd >= n_out && (n_layers == 2 || (w2 && b1 && dw2)), "rmlp_bwd_fused: bad arguments");
    EMER_REQUIRE(final_act == EMER_ACT_NONE || (final_act == EMER_ACT_SIGMOID && out && ldo >= n_out), "rmlp_bwd_fused: final activation must be none, or sigmoid with the saved output");
    EMER_REQUIRE(n_feat != 0 || (ldx >= k0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0), "rmlp_bwd_fused: row-major input needs ldx %% 4 == 0 and 16-byte alignment");
    EMER_REQUIRE(n_feat == 0 || (n_levels * n_feat == k0 && n * k0 < (1ll << 30)), "rmlp_bwd_fused: level-major input needs k0 == n_levels * n_feat and n * k0 < 2^30");
    EMER_REQUIRE(!dx || n_feat != 0 || (lddx >= k0 && lddx % 4 == 0 && ((uintptr_t)dx % 16) == 0), "rmlp_bwd_fused: row-major dx needs lddx %% 4 == 0 and 16-byte alignment");
    EMER_REQUIRE(ld_dw0 >= k0 && ld_dw1 >= 64 && (n_layers == 2 || ld_dw2 >= 64), "rmlp_bwd_fused: leading dimension smaller than the row");
    const int nto = rmlp_bwdw_nto(n_out);
    EMER_REQUIRE(nto == 1 || (ldd % 4 == 0 && ((uintptr_t)dout % 16) == 0 && (final_act == EMER_ACT_NONE || (ldo % 4 == 0 && ((uintptr_t)out % 16) == 0))),
                 "rmlp_bwd_fused: wide outputs need 16-byte aligned rows of dout (and of out for a sigmoid)");
    RMlpBwdWArgs a;
    a.dout = dout; a.ldd = ldd; a.out = out; a.ldo = ldo; a.x = x; a.ldx = ldx; a.n = n; a.n_levels = n_levels; a.k0 = k0; a.n_out = n_out;
    a.final_act = final_act;
    const float *wl = n_layers == 2 ? w1 : w2;
    a.w0 = WSrc{w0, k0, 1, 64, k0};
    a.w1 = WSrc{w1, 64, 1, 64, 64};
    a.b0 = b0; a.b1 = b1;
    a.wlt = WSrc{wl, 1, 64, 64, n_out};   // (n = hidden, k = output) = wl[k][n]
    a.w1t = WSrc{w1, 1, 64, 64, 64};
    a.w0t = WSrc{w0, 1, k0, k0, 64};      // (n = input feature, k = hidden) = w0[k][n]
    a.dx = dx; a.lddx = lddx; a.partials = workspace; a.stride = rmlp_bwdw_stride(n_layers, k0, n_out);
    const int kt0 = n_feat == 0 ? 4 : ((k0 + 15) / 16 <= 2 ? 2 : (k0 + 15) / 16);
    const uint32_t grid = rmlp_bwdw_grid(n, n_layers, n_out);
    const size_t lds = rmlp_bwdw_lds(n_layers, kt0, n_out);
    const int threads = rmlp_bwdw_threads(n_layers, n_out);
    hipStream_t st = as_stream(stream);
    int rc = EMER_E_INVALID;
    auto go = [&](auto kern) {
        if (int r = set_lds(kern, lds, "rmlp_bwd_fused")) return r;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a);
        return check_launch("rmlp_bwd_fused");
    };
    if (nto == 4) rc = go(rmlp_bwdw_kernel<4, 0, 3, 4>);
    else if (n_feat == 0) rc = n_layers == 2 ? go(rmlp_bwdw_kernel<4, 0, 2, 1>) : go(rmlp_bwdw_kernel<4, 0, 3, 1>);
    else if (kt0 == 2) rc = n_layers == 2 ? go(rmlp_bwdw_kernel<2, 4, 2, 1>) : go(rmlp_bwdw_kernel<2, 4, 3, 1>);
    else if (kt0 == 3) rc = n_layers == 2 ? go(rmlp_bwdw_kernel<3, 4, 2, 1>) : go(rmlp_bwdw_kernel<3, 4, 3, 1>);
    else rc = n_layers == 2 ? go(rmlp_bwdw_kernel<4, 4, 2, 1>) : go(rmlp_bwdw_kernel<4, 4, 3, 1>);
    if (rc) return rc;
    DwReduceJob jb[3];
    int nj = 0;
    int64_t off = 0;
    if (n_layers == 2) {
        jb[nj++] = DwReduceJob{off, n_out, 64, dw1, ld_dw1, db1, 0, {0}, {0}, {0}};
        off += (int64_t)n_out * 65;
    } else {
        jb[nj++] = DwReduceJob{off, n_out, 64, dw2, ld_dw2, db2, 0, {0}, {0}, {0}};
        off += (int64_t)n_out * 65;
        jb[nj++] = DwReduceJob{off, 64, 64, dw1, ld_dw1, db1, 0, {0}, {0}, {0}};
        off += 64 * 65;
    }
    jb[nj++] = DwReduceJob{off, 64, k0, dw0, ld_dw0, db0, 0, {0}, {0}, {0}};
    return launch_dw_reduce_multi(workspace, (int32_t)grid, a.stride, nj, jb, st);
}
