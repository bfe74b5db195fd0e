// Small-MLP head kernels for gfx950 on the fp32-input matrix cores.
//
// Replaces the torch.nn.Linear (+ReLU / Sigmoid / trunc_exp) chains of the reference's heads
// (radiance_fields/radiance_field.py:74-198, radiance_fields/mlp.py:7-46), i.e. one cuBLAS GEMM plus
// one or two elementwise launches per layer with [N,64] fp32 activations bouncing through HBM.
//
// Precision: v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 -- f32 in, f32 accumulate, bit-identical
// to a k-ordered fmaf chain (MI355X guide, "FP32-input MFMA"), so the heads keep the reference's
// fp32 semantics (north-star tolerance 1e-4 on composited RGB/depth) while running on the matrix
// pipe at the full fp32 rate and leaving the VALU free for bias/activation epilogues.
//
// Kernels
//   linear_fwd  : Y = act(X W^T + b).  128-row workgroup tile, 4 waves x 32 rows; 64-wide (32x32x2)
//                 or 16-wide (16x16x4, for 1/3/6-channel outputs) column tiles; X/W K-chunks of 32
//                 staged in LDS with an odd (33) / even-offset (34) row pitch so every ds_read_b32
//                 lane group is bank-conflict free.  Generic B strides serve dX = dPre W as well.
//   act_bwd     : dPre = dY * act'(Y)  (act' from the saved output only).
//   linear_dw   : dW += dPre^T X, db += colsum(dPre): split over rows, 32-row LDS tiles, per-wave
//                 32x32 output tiles held in accumulators for the whole row range, one fp32 atomic
//                 pass per workgroup at the end.
#include "common.h"

#include <type_traits>

namespace emer {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float apply_act(int act, float x) {
    switch (act) {
        case EMER_ACT_RELU: return x > 0.0f ? x : 0.0f;
        case EMER_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case EMER_ACT_TRUNC_EXP: return expf(x - 1.0f);
        default: return x;
    }
}
// derivative expressed through the saved OUTPUT y
__device__ __forceinline__ float act_grad_from_y(int act, float y) {
    switch (act) {
        case EMER_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case EMER_ACT_SIGMOID: return y * (1.0f - y);
        case EMER_ACT_TRUNC_EXP: return fminf(y, 3269017.3724721107f);  // exp(min(x-1, 15)) = min(y, e^15)
        default: return 1.0f;
    }
}

constexpr int kBM = 128;  // rows per workgroup
constexpr int kBK = 32;   // K chunk staged in LDS

// b(j, kk) = wmat[j * sbj + kk * sbk]:  fwd: W[N,K] row-major -> (K, 1);  dX: W^T -> (1, K)
template <int BN>  // 64 -> mfma 32x32x2, 16 -> mfma 16x16x4
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ wmat,
                                                         int64_t sbj, int64_t sbk, const float *__restrict__ bias,
                                                         float *__restrict__ y, int64_t ldy, int64_t M, int32_t N, int32_t K,
                                                         int act, float *__restrict__ aux,
                                                         // A-operand prologue (backward only): x <- x * act'(ya) (+ d_aux term on column 0)
                                                         const float *__restrict__ ya, int64_t ldya, int act_a,
                                                         const float *__restrict__ d_aux, const float *__restrict__ aux_y) {
    constexpr int PITCH = (BN == 64) ? 33 : 34;
    __shared__ float xs[kBM * PITCH];
    __shared__ float ws[BN * PITCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t row0 = (int64_t)blockIdx.x * kBM;
    const int32_t n0 = blockIdx.y * BN;

    f32x16 acc0 = {0}, acc1 = {0};  // BN == 64: two 32x32 column tiles
    f32x4 c0 = {0}, c1 = {0};       // BN == 16: two 16-row tiles

    for (int32_t k0 = 0; k0 < K; k0 += kBK) {
        // stage X[row0 : row0+128, k0 : k0+32] (zero padded), coalesced along k
        for (int idx = tid; idx < kBM * kBK; idx += 256) {
            const int r = idx >> 5, c = idx & 31;
            const int64_t gr = row0 + r;
            const int32_t gk = k0 + c;
            float xv = 0.0f;
            if (gr < M && gk < K) {
                xv = x ? x[gr * ldx + gk] : 0.0f;
                if (ya) xv *= act_grad_from_y(act_a, ya[gr * ldya + gk]);
                if (d_aux && gk == 0) xv += d_aux[gr] * act_grad_from_y(EMER_ACT_TRUNC_EXP, aux_y[gr]);
            }
            xs[r * PITCH + c] = xv;
        }
        for (int idx = tid; idx < BN * kBK; idx += 256) {
            const int j = idx >> 5, c = idx & 31;
            const int32_t gn = n0 + j, gk = k0 + c;
            ws[j * PITCH + c] = (gn < N && gk < K) ? wmat[gn * sbj + gk * sbk] : 0.0f;
        }
        __syncthreads();
        if constexpr (BN == 64) {
            const float *xa = xs + (wave * 32 + (lane & 31)) * PITCH + (lane >> 5);
            const float *wb = ws + (lane & 31) * PITCH + (lane >> 5);
#pragma unroll
            for (int s = 0; s < kBK / 2; ++s) {
                const float a = xa[2 * s];
                const float b0 = wb[2 * s], b1 = wb[32 * PITCH + 2 * s];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            }
        } else {
            const float *xa = xs + (wave * 32 + (lane & 15)) * PITCH + (lane >> 4);
            const float *wb = ws + (lane & 15) * PITCH + (lane >> 4);
#pragma unroll
            for (int s = 0; s < kBK / 4; ++s) {
                const float b = wb[4 * s];
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * s], b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[16 * PITCH + 4 * s], b, c1, 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if constexpr (BN == 64) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int32_t col = n0 + t * 32 + (lane & 31);
            if (col >= N) continue;
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M) {
                    const float pre = (t == 0 ? acc0[r] : acc1[r]) + bv;
                    y[row * ldy + col] = apply_act(act, pre);
                    if (aux && col == 0) aux[row] = expf(pre - 1.0f);  // density side output (radiance_field.py:422)
                }
            }
        }
    } else {
        const int32_t col = n0 + (lane & 15);
        if (col < N) {
            const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + wave * 32 + t * 16 + (lane >> 4) * 4 + r;
                    if (row < M) {
                        const float pre = (t == 0 ? c0[r] : c1[r]) + bv;
                        y[row * ldy + col] = apply_act(act, pre);
                        if (aux && col == 0) aux[row] = expf(pre - 1.0f);
                    }
                }
            }
        }
    }
}

// dW[N,K] += dPre[M,N]^T X[M,K];  db[N] += colsum(dPre), with dPre = dY * act'(Y) formed while staging.
// grid = (row blocks, K groups, N groups).  A workgroup reduces `rows_per_block` rows: 32-row tiles are
// prefetched into registers while the previous tile feeds the MFMAs out of LDS; its (NG/32)*(KG/32)
// output tiles are dealt round-robin to the 4 waves (TPW per wave) and live in accumulators for the whole
// row range.  Partial results go to a workspace [row block][N*K + N] with plain stores; a second kernel
// sums the row blocks (L2 float atomics retire only ~21 G/s on this chip, see tools/atomic_probe.hip).
template <int NGT, int KGT>  // NG = 32*NGT output rows (n), KG = 32*KGT output columns (k) per workgroup
__global__ __launch_bounds__(256) void linear_dw_kernel(const float *__restrict__ dy, int64_t lddy, const float *__restrict__ ysave,
                                                        int64_t ldy, int act, const float *__restrict__ d_aux,
                                                        const float *__restrict__ aux_y, const float *__restrict__ x, int64_t ldx,
                                                        float *__restrict__ partials, int64_t M, int32_t N, int32_t K,
                                                        int32_t rows_per_block, int want_bias) {
    constexpr int NG = 32 * NGT, KG = 32 * KGT, TILES = NGT * KGT, TPW = (TILES + 3) / 4;
    constexpr int ND = (32 * NG) / 256, NX = (32 * KG) / 256;  // floats staged per thread per 32-row tile
    __shared__ float ds[32 * NG];
    __shared__ float xs[32 * KG];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int32_t n_base = blockIdx.z * NG, k_base = blockIdx.y * KG;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < M) ? r_begin + rows_per_block : M;

    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f32x16{0};
    float bsum = 0.0f;  // wave 0: column sum for n = lane
    float dreg[ND], xreg[NX];

    auto fetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            const int r = idx / NG, c = idx % NG;
            const int64_t gr = r0 + r;
            const int32_t gn = n_base + c;
            float v = 0.0f;
            if (gr < r_end && gn < N) {
                v = dy ? dy[gr * lddy + gn] * act_grad_from_y(act, ysave ? ysave[gr * ldy + gn] : 0.0f) : 0.0f;
                if (d_aux && gn == 0) v += d_aux[gr] * act_grad_from_y(EMER_ACT_TRUNC_EXP, aux_y[gr]);
            }
            dreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + i * 256;
            const int r = idx / KG, c = idx % KG;
            const int64_t gr = r0 + r;
            const int32_t gk = k_base + c;
            xreg[i] = (gr < r_end && gk < K) ? x[gr * ldx + gk] : 0.0f;
        }
    };

    fetch(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
#pragma unroll
        for (int i = 0; i < ND; ++i) ds[tid + i * 256] = dreg[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[tid + i * 256] = xreg[i];
        __syncthreads();
        if (r0 + 32 < r_end) fetch(r0 + 32);  // next tile's global loads fly while the MFMAs below run
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int t = wave + 4 * j;
            if (t < TILES) {  // wave-uniform
                const int nt = t / KGT, kt = t % KGT;
                const float *ap = ds + (lane >> 5) * NG + nt * 32 + (lane & 31);
                const float *bp = xs + (lane >> 5) * KG + kt * 32 + (lane & 31);
#pragma unroll 4
                for (int s = 0; s < 16; ++s)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * NG], bp[2 * s * KG], acc[j], 0, 0, 0);
            }
        }
        if (want_bias && blockIdx.y == 0 && wave == 0 && lane < NG) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) bsum += ds[r * NG + lane];
        }
        __syncthreads();
    }

    float *__restrict__ part = partials + (int64_t)blockIdx.x * ((int64_t)N * K + N);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = wave + 4 * j;
        if (t >= TILES) continue;
        const int nt = t / KGT, kt = t % KGT;
        const int32_t k = k_base + kt * 32 + (lane & 31);
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int32_t n = n_base + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < N) part[(int64_t)n * K + k] = acc[j][r];
        }
    }
    if (want_bias && blockIdx.y == 0 && wave == 0 && lane < NG && n_base + lane < N) part[(int64_t)N * K + n_base + lane] = bsum;
}

// dw[i] += sum_b partials[b][i]  (i < N*K),  dbias[i - N*K] += ...  (i >= N*K).
// The partial blocks are cut into gridDim.y ranges so that the (small) N*K + N extent still fills the chip; each
// thread sums its range with four independent accumulators (loads in flight) and merges with one f32 atomic.
struct DwDst {  // where column k of the (virtually concatenated) operand lands in dw: dw[n * ld + dst + (k - col)]
    int32_t col[EMER_CHAIN_MAX_SEGS], width[EMER_CHAIN_MAX_SEGS], dst[EMER_CHAIN_MAX_SEGS];
    int32_t n, K;
    int64_t ld;
};
static inline DwDst dw_dst_identity(int32_t k) {
    DwDst d;
    for (int i = 0; i < EMER_CHAIN_MAX_SEGS; ++i) { d.col[i] = 0; d.width[i] = 0; d.dst[i] = 0; }
    d.col[0] = 0; d.width[0] = k; d.dst[0] = 0; d.n = 1; d.K = k; d.ld = k;
    return d;
}

__global__ __launch_bounds__(256) void linear_dw_reduce_kernel(const float *__restrict__ partials, int32_t n_blocks, int64_t stride,
                                                               int64_t nk, float *__restrict__ dw, float *__restrict__ dbias, const DwDst dst,
                                                               int64_t extent = -1) {
    // extent: floats of a partial that belong to this (dW | dbias) pair; -1: the whole partial (= stride)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (extent < 0 ? stride : extent)) return;
    const int32_t per = (n_blocks + (int32_t)gridDim.y - 1) / (int32_t)gridDim.y;
    const int32_t b0 = (int32_t)blockIdx.y * per;
    const int32_t b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    const float *__restrict__ p = partials + i;
    int32_t b = b0;
    for (; b + 4 <= b1; b += 4) {
        a0 += p[(int64_t)b * stride];
        a1 += p[(int64_t)(b + 1) * stride];
        a2 += p[(int64_t)(b + 2) * stride];
        a3 += p[(int64_t)(b + 3) * stride];
    }
    for (; b < b1; ++b) a0 += p[(int64_t)b * stride];
    const float a = (a0 + a1) + (a2 + a3);
    if (b0 >= b1) return;
    if (i < nk) {
        const int32_t n = (int32_t)(i / dst.K), k = (int32_t)(i - (int64_t)n * dst.K);
        int64_t o = -1;
#pragma unroll
        for (int sg = 0; sg < EMER_CHAIN_MAX_SEGS; ++sg)
            if (sg < dst.n && k >= dst.col[sg] && k < dst.col[sg] + dst.width[sg]) o = (int64_t)n * dst.ld + dst.dst[sg] + (k - dst.col[sg]);
        if (o >= 0) __hip_atomic_fetch_add(dw + o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (dbias) __hip_atomic_fetch_add(dbias + (i - nk), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// [r4] Several (dW | dbias) pairs of ONE partial buffer in one launch (blockIdx.z = job): the fused backward kernels leave two or three
// gradients in each workgroup's partial, and three 5-us launches in a row cost more than the sums themselves.
struct DwJobs {
    int32_t n_jobs;
    int64_t off[EMER_DW_MAX_JOBS], nk[EMER_DW_MAX_JOBS], extent[EMER_DW_MAX_JOBS];   // start inside a partial, floats of dW, floats of dW + dbias
    float *dw[EMER_DW_MAX_JOBS], *dbias[EMER_DW_MAX_JOBS];
    DwDst dst[EMER_DW_MAX_JOBS];
};
__global__ __launch_bounds__(256) void linear_dw_reduce_multi_kernel(const float *__restrict__ partials, int32_t n_blocks, int64_t stride, const DwJobs jobs) {
    const int j = (int)blockIdx.z;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= jobs.extent[j]) return;
    const int32_t per = (n_blocks + (int32_t)gridDim.y - 1) / (int32_t)gridDim.y;
    const int32_t b0 = (int32_t)blockIdx.y * per;
    const int32_t b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
    if (b0 >= b1) return;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    const float *__restrict__ p = partials + jobs.off[j] + i;
    int32_t b = b0;
    for (; b + 4 <= b1; b += 4) {
        a0 += p[(int64_t)b * stride];
        a1 += p[(int64_t)(b + 1) * stride];
        a2 += p[(int64_t)(b + 2) * stride];
        a3 += p[(int64_t)(b + 3) * stride];
    }
    for (; b < b1; ++b) a0 += p[(int64_t)b * stride];
    const float a = (a0 + a1) + (a2 + a3);
    const DwDst &dst = jobs.dst[j];
    if (i < jobs.nk[j]) {
        const int32_t n = (int32_t)(i / dst.K), k = (int32_t)(i - (int64_t)n * dst.K);
        int64_t o = -1;
#pragma unroll
        for (int sg = 0; sg < EMER_CHAIN_MAX_SEGS; ++sg)
            if (sg < dst.n && k >= dst.col[sg] && k < dst.col[sg] + dst.width[sg]) o = (int64_t)n * dst.ld + dst.dst[sg] + (k - dst.col[sg]);
        if (o >= 0) __hip_atomic_fetch_add(jobs.dw[j] + o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (jobs.dbias[j]) __hip_atomic_fetch_add(jobs.dbias[j] + (i - jobs.nk[j]), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static inline uint32_t dw_reduce_splits(int32_t n_blocks, int64_t stride) {
    // ~2048 workgroups in total, at least 8 partial blocks per range
    int64_t s = 2048 / ((stride + 255) / 256);
    if (s > n_blocks / 8) s = n_blocks / 8;
    return (uint32_t)(s < 1 ? 1 : s);
}

// Rows per workgroup of the dW kernels.  Big tiles hold few waves per SIMD (their accumulators fill the register
// file), so they want long row ranges (fewer partials, longer streams): 64x128 -> 4096 rows, 64 x (<=64) -> 2048, n <= 32 ->
// 1024.  Never fewer than ~512 workgroups' worth of parallelism for per-ray heads (8192 rows would otherwise occupy 8
// CUs).  emer_linear_bwd_workspace uses the same rule.
static inline int32_t dw_rows_per_block(int64_t m, int32_t n, int32_t k) {
    const int tiles = ((n + 31) / 32) * ((k + 31) / 32);
    const int64_t cap = tiles >= 6 ? 4096 : (n > 32 ? 2048 : 1024);
    int64_t r = (m + 511) / 512;
    r = (r + 63) / 64 * 64;  // 16-row blocks per wave (wgrad_stream), 32-row LDS tiles (wgrad_seg)
    if (r < 64) r = 64;
    if (r > cap) r = cap;
    return (int32_t)r;
}

// Sum `n_blocks` partials (`stride` floats apart, each holding dW [n][k] | dbias [n] at its start) into dw (+=, leading dimension
// ld_dw) and dbias (+=, may be null).  Used by the fused backward kernels of csrc/mlp_fused.hip.
int launch_dw_reduce(const float *partials, int32_t n_blocks, int64_t stride, int32_t n, int32_t k, float *dw, int64_t ld_dw,
                     float *dbias, hipStream_t st) {
    DwDst d = dw_dst_identity(k);
    d.ld = ld_dw;
    const int64_t extent = (int64_t)n * k + n;
    hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3((uint32_t)ceil_div(extent, 256), dw_reduce_splits(n_blocks, extent)), dim3(256), 0, st, partials,
                       n_blocks, stride, (int64_t)n * k, dw, dbias, d, extent);
    return check_launch("dw_reduce");
}

// The same for a partial whose column blocks go to different places: block s = columns col[s] .. col[s] + width[s] - 1 of the [n][k]
// partial lands at dw[row * ld_dw + dst[s] ..]; no bias part.
int launch_dw_reduce_cols(const float *partials, int32_t n_blocks, int64_t stride, int32_t n, int32_t k, float *dw, int64_t ld_dw,
                          int32_t n_segs, const int32_t *col, const int32_t *width, const int32_t *dst, hipStream_t st) {
    DwDst d = dw_dst_identity(k);
    d.ld = ld_dw;
    d.n = n_segs;
    for (int i = 0; i < n_segs && i < EMER_CHAIN_MAX_SEGS; ++i) { d.col[i] = col[i]; d.width[i] = width[i]; d.dst[i] = dst[i]; }
    const int64_t extent = (int64_t)n * k;
    hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3((uint32_t)ceil_div(extent, 256), dw_reduce_splits(n_blocks, extent)), dim3(256), 0, st, partials,
                       n_blocks, stride, (int64_t)n * k, dw, (float *)nullptr, d, extent);
    return check_launch("dw_reduce_cols");
}

// Up to EMER_DW_MAX_JOBS gradients of one partial buffer in one launch.  Job j: dW [n][k] (+ dbias [n] when db != null) at float `off`
// of every partial; `n_segs` == 0: dW goes to dw[row * ld + column]; else its column blocks are scattered as in launch_dw_reduce_cols.
int launch_dw_reduce_multi(const float *partials, int32_t n_blocks, int64_t stride, int n_jobs, const DwReduceJob *jb, hipStream_t st) {
    DwJobs jobs;
    jobs.n_jobs = n_jobs;
    int64_t max_extent = 0;
    for (int j = 0; j < EMER_DW_MAX_JOBS; ++j) {
        const bool on = j < n_jobs;
        const DwReduceJob &q = jb[on ? j : 0];
        DwDst d = dw_dst_identity(q.k);
        d.ld = q.ld_dw;
        if (q.n_segs > 0) {
            d.n = q.n_segs;
            for (int i = 0; i < q.n_segs && i < EMER_CHAIN_MAX_SEGS; ++i) { d.col[i] = q.col[i]; d.width[i] = q.width[i]; d.dst[i] = q.dst[i]; }
        }
        jobs.dst[j] = d;
        jobs.off[j] = q.off; jobs.nk[j] = (int64_t)q.n * q.k; jobs.extent[j] = on ? (int64_t)q.n * q.k + (q.db ? q.n : 0) : 0;
        jobs.dw[j] = q.dw; jobs.dbias[j] = q.db;
        if (jobs.extent[j] > max_extent) max_extent = jobs.extent[j];
    }
    hipLaunchKernelGGL(linear_dw_reduce_multi_kernel, dim3((uint32_t)ceil_div(max_extent, 256), dw_reduce_splits(n_blocks, max_extent), (uint32_t)n_jobs),
                       dim3(256), 0, st, partials, n_blocks, stride, jobs);
    return check_launch("dw_reduce_multi");
}

static int launch_linear(const float *x, int64_t ldx, const float *w, int64_t sbj, int64_t sbk, const float *bias, float *y,
                         int64_t ldy, int64_t M, int32_t N, int32_t K, int act, float *aux, hipStream_t st,
                         const float *ya = nullptr, int64_t ldya = 0, int act_a = 0, const float *d_aux = nullptr,
                         const float *aux_y = nullptr) {
    const uint32_t gx = (uint32_t)ceil_div(M, kBM);
    if (N <= 16) {
        hipLaunchKernelGGL(linear_fwd_kernel<16>, dim3(gx, (uint32_t)ceil_div(N, 16)), dim3(256), 0, st, x, ldx, w, sbj, sbk, bias, y,
                           ldy, M, N, K, act, aux, ya, ldya, act_a, d_aux, aux_y);
    } else {
        hipLaunchKernelGGL(linear_fwd_kernel<64>, dim3(gx, (uint32_t)ceil_div(N, 64)), dim3(256), 0, st, x, ldx, w, sbj, sbk, bias, y,
                           ldy, M, N, K, act, aux, ya, ldya, act_a, d_aux, aux_y);
    }
    return check_launch("linear");
}

}  // namespace emer

using namespace emer;

extern "C" int emer_linear_fwd(const float *x, int64_t ldx, const float *w, const float *bias, float *y, int64_t ldy,
                               int64_t m, int32_t n, int32_t k, int act, float *aux_density, void *stream) {
    EMER_REQUIRE(m >= 0 && n >= 1 && k >= 1, "linear_fwd: bad sizes m=%lld n=%d k=%d", (long long)m, n, k);
    if (m == 0) return EMER_OK;
    EMER_REQUIRE(x && w && y, "linear_fwd: null pointer");
    EMER_REQUIRE(ldx >= k && ldy >= n, "linear_fwd: leading dimension smaller than the row");
    EMER_REQUIRE(act >= EMER_ACT_NONE && act <= EMER_ACT_TRUNC_EXP, "linear_fwd: unknown activation %d", act);
    return launch_linear(x, ldx, w, k, 1, bias, y, ldy, m, n, k, act, aux_density, as_stream(stream));
}

// floats of workspace emer_linear_bwd needs for the dW / dbias partial sums (0 when dw is not requested)
extern "C" int64_t emer_linear_bwd_workspace(int64_t m, int32_t n, int32_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    return ceil_div(m, dw_rows_per_block(m, n, k)) * ((int64_t)n * k + n);
}

extern "C" int emer_linear_bwd(const float *dy, int64_t lddy, const float *y, int64_t ldy, const float *x, int64_t ldx,
                               const float *w, float *workspace, float *dx, int64_t lddx, float *dw, float *dbias, int64_t m,
                               int32_t n, int32_t k, int act, const float *d_aux_density, const float *aux_density,
                               void *stream) {
    EMER_REQUIRE(m >= 0 && n >= 1 && k >= 1, "linear_bwd: bad sizes m=%lld n=%d k=%d", (long long)m, n, k);
    if (m == 0) return EMER_OK;
    EMER_REQUIRE(dy || d_aux_density, "linear_bwd: no incoming gradient");
    EMER_REQUIRE(!d_aux_density || aux_density, "linear_bwd: d_aux_density needs the saved aux_density");
    EMER_REQUIRE(act >= EMER_ACT_NONE && act <= EMER_ACT_TRUNC_EXP, "linear_bwd: unknown activation %d", act);
    EMER_REQUIRE(act == EMER_ACT_NONE || y || !dy, "linear_bwd: saved output y required for a non-linear activation");
    hipStream_t st = as_stream(stream);
    const float *ya = (act == EMER_ACT_NONE) ? nullptr : y;
    if (dx) {
        EMER_REQUIRE(w && lddx >= k, "linear_bwd: dx requested but w missing or lddx too small");
        // dX[M,K] = (dY * act'(Y))[M,N] @ W[N,K]: a linear with reduction dim N and b(j=k, kk=n) = W[n*K + k];
        // act' (and the density side-gradient) are applied while the A tile is staged: dPre never touches HBM
        if (int rc = launch_linear(dy, lddy, w, 1, k, nullptr, dx, lddx, m, k, n, EMER_ACT_NONE, nullptr, st, ya, ldy, act,
                                   d_aux_density, aux_density)) return rc;
    }
    if (dw) {
        EMER_REQUIRE(x && ldx >= k && workspace, "linear_bwd: dw requested but x / workspace missing or ldx too small");
        const int32_t NG = n <= 32 ? 32 : 64;
        const int32_t KG = k <= 32 ? 32 : (k <= 64 ? 64 : (k <= 128 ? 128 : 256));
        const int32_t rpb = dw_rows_per_block(m, n, k);
        const int32_t n_row_blocks = (int32_t)ceil_div(m, rpb);
        const dim3 grid((uint32_t)n_row_blocks, (uint32_t)ceil_div(k, KG), (uint32_t)ceil_div(n, NG));
#define EMER_DW(A, B) hipLaunchKernelGGL((linear_dw_kernel<A, B>), grid, dim3(256), 0, st, dy, lddy, ya, ldy, act, d_aux_density, \
                                         aux_density, x, ldx, workspace, m, n, k, rpb, dbias ? 1 : 0)
        if (NG == 32) { if (KG == 32) EMER_DW(1, 1); else if (KG == 64) EMER_DW(1, 2); else if (KG == 128) EMER_DW(1, 4); else EMER_DW(1, 8); }
        else          { if (KG == 32) EMER_DW(2, 1); else if (KG == 64) EMER_DW(2, 2); else if (KG == 128) EMER_DW(2, 4); else EMER_DW(2, 8); }
#undef EMER_DW
        if (int rc = check_launch("linear_dw")) return rc;
        const int64_t stride = (int64_t)n * k + n;
        const DwDst ddst = dw_dst_identity(k);
        hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3((uint32_t)ceil_div(stride, 256), dw_reduce_splits(n_row_blocks, stride)), dim3(256), 0, st, workspace, n_row_blocks,
                           stride, (int64_t)n * k, dw, dbias, ddst);
        if (int rc = check_launch("linear_dw_reduce")) return rc;
    }
    return EMER_OK;
}

// ================================================================================ fused MLP chains
// Row-split design for gfx950: a wave owns 16-row tiles end to end, so there is NO workgroup barrier inside
// the tile loop; all layer weights of the chain sit in LDS (loaded once per persistent workgroup, zero-padded
// to K%4 == 0 / N%16 == 0); activations live in a wave-private LDS row buffer whose pitch P satisfies
// (P/2) odd, which makes the mfma_16x16x4 A-fragment read (lanes 0-15: 16 rows of one k, lanes 16-31: k+1)
// hit 32 distinct banks.  Column groups of 64 outputs give 4 independent accumulators per k-step, enough to
// keep the 32-cycle fp32 MFMA pipe issuing back to back (dependent latency 40 cycles).
namespace emer {

struct ChainLds {
    int32_t w_off[EMER_CHAIN_MAX_LAYERS], b_off[EMER_CHAIN_MAX_LAYERS], w_pitch[EMER_CHAIN_MAX_LAYERS];
    int32_t kpad[EMER_CHAIN_MAX_LAYERS], npad[EMER_CHAIN_MAX_LAYERS];
    int32_t w_total, P;
};

static inline int32_t round_up_i(int32_t a, int32_t b) { return (a + b - 1) / b * b; }

static ChainLds chain_lds_plan(const emer_chain_desc *d) {
    ChainLds p;
    int32_t off = 0;
    for (int l = 0; l < EMER_CHAIN_MAX_LAYERS; ++l) { p.w_off[l] = p.b_off[l] = p.w_pitch[l] = p.kpad[l] = p.npad[l] = 0; }
    for (int l = 0; l < d->n_layers; ++l) {
        p.kpad[l] = round_up_i(d->layers[l].K, 8);   // 4 lane groups x an even number of k each (ds_read_b64 pairs)
        p.npad[l] = round_up_i(d->layers[l].N, 16);
        p.w_pitch[l] = p.kpad[l] + 2;  // kpad % 4 == 0  ->  (pitch / 2) odd
        p.w_off[l] = off; off += p.npad[l] * p.w_pitch[l];
        p.b_off[l] = off; off += p.npad[l];
    }
    p.w_total = round_up_i(off, 4);
    p.P = round_up_i(d->buf_cols + 8, 4) + 2;
    return p;
}


// One column group (NT tiles of 16 outputs) of one layer for a 16-row tile.
// The reduction index is PERMUTED: lane group g = lane >> 4 owns the contiguous range k in [g*kq, (g+1)*kq),
// kq = kpad / 4, and MFMA step s uses k = g*kq + s for both operands (any bijection of k is a valid GEMM).
// Each lane therefore reads CONSECUTIVE floats of its activation row / weight row: one ds_read_b64 feeds
// two MFMA steps, and with the 4-step unroll ten 8-byte reads are in flight behind sixteen MFMAs.
template <int NT>
__device__ __forceinline__ void chain_gemm(const float *__restrict__ ap, const float *__restrict__ bp, int wp, int kq,
                                           f32x4 (&acc)[4]) {
    const float2 *a2 = reinterpret_cast<const float2 *>(ap);
    const float2 *b2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b2[t] = reinterpret_cast<const float2 *>(bp + t * 16 * wp);
    const int n2 = kq >> 1;
    float2 a = a2[0], b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = b2[t][0];
    for (int s2 = 0; s2 < n2; ++s2) {  // software pipelined: the next pair's LDS reads are issued before this pair's MFMAs
        const int nx = (s2 + 1 < n2) ? s2 + 1 : s2;
        const float2 an = a2[nx];
        float2 bn[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bn[t] = b2[t][nx];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
        a = an;
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = bn[t];
    }
}

__global__ __launch_bounds__(1024) void mlp_chain_kernel(const emer_chain_desc d, const ChainLds lp, int64_t n_rows, int64_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // ---- stage every layer's weights (zero padded) and bias once per workgroup
    for (int l = 0; l < d.n_layers; ++l) {
        const emer_chain_layer &L = d.layers[l];
        const int32_t kp = lp.kpad[l], np = lp.npad[l];
        for (int idx = tid; idx < np * kp; idx += (int)blockDim.x) {
            const int n = idx / kp, k = idx - n * kp;
            smem[lp.w_off[l] + n * lp.w_pitch[l] + k] = (n < L.N && k < L.K) ? L.w[n * L.w_sn + k * L.w_sk] : 0.0f;
        }
        for (int n = tid; n < np; n += (int)blockDim.x) smem[lp.b_off[l] + n] = (L.bias && n < L.N) ? L.bias[n] : 0.0f;
    }
    const int32_t P = lp.P;
    float *buf = smem + lp.w_total + wave * 16 * P;
    for (int i = lane; i < 16 * P; i += 64) buf[i] = 0.0f;  // padding columns must stay finite (they meet zero weights)
    __syncthreads();

    const int n_waves = (int)blockDim.x >> 6;
    for (int64_t tile = (int64_t)blockIdx.x * n_waves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * n_waves) {
        const int64_t row0 = tile * 16;
        // ---- fill the input segments: rows outer, lanes walk columns (coalesced rows, no integer divisions)
        for (int s = 0; s < d.n_segs; ++s) {
            const emer_chain_seg S = d.segs[s];
            if (S.mode == 1) {
                // level-major [L][N][f]: lane = (row, level group) so 16 lanes read 16 consecutive rows of ONE level
                // (16*f*4 contiguous bytes); four levels per load instruction
                const int r = lane & 15, nl = S.width / S.f;
                const bool ok = row0 + r < n_rows;
                for (int lv = lane >> 4; lv < nl; lv += 4) {
                    const float *src = S.ptr + ((int64_t)lv * S.n_total + row0 + r) * S.f;
                    float *dst = buf + r * P + S.col + lv * S.f;
                    if (S.f == 2) { const float2 v = ok ? *reinterpret_cast<const float2 *>(src) : make_float2(0.f, 0.f); dst[0] = v.x; dst[1] = v.y; }
                    else if (S.f == 4) { const float4 v = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f); dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
                    else { for (int f = 0; f < S.f; ++f) dst[f] = ok ? src[f] : 0.0f; }
                }
            } else if (S.row_div == 1) {
                for (int c = lane; c < S.width; c += 64) {
                    const float *src = S.ptr + row0 * S.ld + c;
#pragma unroll 4
                    for (int r = 0; r < 16; ++r) buf[r * P + S.col + c] = (row0 + r < n_rows) ? src[(int64_t)r * S.ld] : 0.0f;
                }
            } else {  // per-ray data: one division per tile; the 16 rows usually share one source row
                const int64_t ray0 = row0 / S.row_div;
                const int32_t rem0 = (int32_t)(row0 - ray0 * S.row_div);
                for (int c = lane; c < S.width; c += 64) {
                    if (rem0 + 15 < S.row_div) {
                        const float v = S.ptr[ray0 * S.ld + c];
#pragma unroll 4
                        for (int r = 0; r < 16; ++r) buf[r * P + S.col + c] = (row0 + r < n_rows) ? v : 0.0f;
                    } else {
                        for (int r = 0; r < 16; ++r) {
                            const int64_t ray = ray0 + (rem0 + r) / S.row_div;
                            buf[r * P + S.col + c] = (row0 + r < n_rows) ? S.ptr[ray * S.ld + c] : 0.0f;
                        }
                    }
                }
            }
            if (S.fix_a && lane < 16 && row0 + lane < n_rows)
                buf[lane * P + S.col] += S.fix_a[row0 + lane] * fminf(S.fix_b[row0 + lane], 3269017.3724721107f);
        }
        // ---- layers
        for (int l = 0; l < d.n_layers; ++l) {
            const emer_chain_layer L = d.layers[l];
            const float *wl = smem + lp.w_off[l];
            const float *bl = smem + lp.b_off[l];
            const int32_t wp = lp.w_pitch[l], kq = lp.kpad[l] >> 2, ntiles = lp.npad[l] >> 4;
            const float *ap = buf + (lane & 15) * P + L.in_col + (lane >> 4) * kq;
            for (int cg = 0; cg * 4 < ntiles; ++cg) {
                const int nt = (ntiles - cg * 4) < 4 ? (ntiles - cg * 4) : 4;  // wave-uniform
                const float *bp = wl + (cg * 64 + (lane & 15)) * wp + (lane >> 4) * kq;
                // relu' masks of this column group are fetched ahead of the MFMA loop
                float mk[4][4];
                if (L.mask) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int col = cg * 64 + t * 16 + (lane & 15);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int64_t row = row0 + (lane >> 4) * 4 + i;
                            mk[t][i] = (t < nt && col < L.N && row < n_rows) ? L.mask[row * L.mask_ld + col] : 0.0f;
                        }
                    }
                }
                f32x4 acc[4] = {f32x4{0}, f32x4{0}, f32x4{0}, f32x4{0}};
                if (nt == 4) chain_gemm<4>(ap, bp, wp, kq, acc);
                else if (nt == 3) chain_gemm<3>(ap, bp, wp, kq, acc);
                else if (nt == 2) chain_gemm<2>(ap, bp, wp, kq, acc);
                else chain_gemm<1>(ap, bp, wp, kq, acc);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int col = cg * 64 + t * 16 + (lane & 15);
                    if (t < nt && col < L.N) {
                        const float bv = bl[col];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int r = (lane >> 4) * 4 + i;
                            float v = apply_act(L.act, acc[t][i] + bv);
                            float *o = buf + r * P + L.out_col + col;
                            if (L.accumulate) v += *o;
                            if (L.mask) v = mk[t][i] > 0.0f ? v : 0.0f;
                            *o = v;
                        }
                    }
                }
            }
            // ---- coalesced copy-out of (part of) the layer output
            if (L.store) {
                const float *ob = buf + L.out_col + L.store_col;
                if (L.store_mode == 1) {  // level-major: 16 lanes write 16 consecutive rows of one level
                    const int r = lane & 15, nl = L.store_n / L.store_f;
                    if (row0 + r < n_rows) {
                        for (int lv = lane >> 4; lv < nl; lv += 4) {
                            float *dst = L.store + ((int64_t)lv * L.store_ntotal + row0 + r) * L.store_f;
                            const float *src = ob + r * P + lv * L.store_f;
                            if (L.store_f == 2) *reinterpret_cast<float2 *>(dst) = make_float2(src[0], src[1]);
                            else if (L.store_f == 4) *reinterpret_cast<float4 *>(dst) = make_float4(src[0], src[1], src[2], src[3]);
                            else { for (int f = 0; f < L.store_f; ++f) dst[f] = src[f]; }
                        }
                    }
                } else {
                    for (int c = lane; c < L.store_n; c += 64) {
                        float *dst = L.store + row0 * L.store_ld + c;
#pragma unroll 4
                        for (int r = 0; r < 16; ++r) if (row0 + r < n_rows) dst[(int64_t)r * L.store_ld] = ob[r * P + c];
                    }
                }
            }
            if (L.store_exp0 && lane < 16 && row0 + lane < n_rows) L.store_exp0[row0 + lane] = expf(buf[lane * P + L.out_col] - 1.0f);
        }
    }
}

// dW kernel with a segmented (virtual concat) X operand; dPre is given materialised.
struct SegX { emer_chain_seg s[EMER_CHAIN_MAX_SEGS]; int32_t n; const float *col0; };  // col0: replaces column 0 of dPre when non-null

template <int NGT, int KGT>
__global__ __launch_bounds__(256) void wgrad_seg_kernel(const float *__restrict__ dpre, int64_t ldd, const SegX sx,
                                                        float *__restrict__ partials, int64_t M, int32_t N, int32_t K,
                                                        int32_t rows_per_block, int want_bias) {
    constexpr int NG = 32 * NGT, KG = 32 * KGT, TILES = NGT * KGT, TPW = (TILES + 3) / 4;
    constexpr int ND = (32 * NG) / 256, NX = (32 * KG) / 256;
    __shared__ float ds[32 * NG];
    __shared__ float xs[32 * KG];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int32_t n_base = blockIdx.z * NG, k_base = blockIdx.y * KG;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < M) ? r_begin + rows_per_block : M;
    f32x16 acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f32x16{0};
    float bsum = 0.0f;
    float dreg[ND], xreg[NX];
    int fl[EMER_CHAIN_MAX_SEGS];
#pragma unroll
    for (int s = 0; s < EMER_CHAIN_MAX_SEGS; ++s) fl[s] = __ffs(sx.s[s].f > 0 ? sx.s[s].f : 1) - 1;
    auto fetch = [&](int64_t r0) {
        int64_t ray0[EMER_CHAIN_MAX_SEGS];
        int32_t rem0[EMER_CHAIN_MAX_SEGS];
#pragma unroll
        for (int s = 0; s < EMER_CHAIN_MAX_SEGS; ++s) {  // per-ray operands: ONE scalar division per segment per 32-row tile
            const int32_t rd = sx.s[s].row_div > 0 ? sx.s[s].row_div : 1;
            ray0[s] = (rd == 1) ? r0 : r0 / rd;
            rem0[s] = (int32_t)(r0 - ray0[s] * rd);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int idx = tid + i * 256;
            const int r = idx / NG, c = idx % NG;
            const int64_t gr = r0 + r;
            const int32_t gn = n_base + c;
            dreg[i] = (gr < r_end && gn < N) ? ((sx.col0 && gn == 0) ? sx.col0[gr] : dpre[gr * ldd + gn]) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = tid + i * 256;
            const int r = idx / KG, c = idx % KG;
            const int64_t gr = r0 + r;
            const int32_t gk = k_base + c;
            float v = 0.0f;
            if (gr < r_end && gk < K) {
#pragma unroll
                for (int s = 0; s < EMER_CHAIN_MAX_SEGS; ++s) {
                    if (s < sx.n && gk >= sx.s[s].col && gk < sx.s[s].col + sx.s[s].width) {
                        const int32_t c = gk - sx.s[s].col;
                        if (sx.s[s].mode == 1) {
                            v = sx.s[s].ptr[((int64_t)(c >> fl[s]) * sx.s[s].n_total + gr) * sx.s[s].f + (c & (sx.s[s].f - 1))];
                        } else if (sx.s[s].row_div == 1) {
                            v = sx.s[s].ptr[gr * sx.s[s].ld + c];
                        } else {  // per-ray operand (32 <= row_div in practice: at most one ray boundary inside a tile)
                            const int32_t rem = rem0[s] + r;
                            const int64_t ray = rem < sx.s[s].row_div ? ray0[s] : ray0[s] + rem / sx.s[s].row_div;
                            v = sx.s[s].ptr[ray * sx.s[s].ld + c];
                        }
                    }
                }
            }
            xreg[i] = v;
        }
    };
    fetch(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
#pragma unroll
        for (int i = 0; i < ND; ++i) ds[tid + i * 256] = dreg[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[tid + i * 256] = xreg[i];
        __syncthreads();
        if (r0 + 32 < r_end) fetch(r0 + 32);
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int t = wave + 4 * j;
            if (t < TILES) {
                const int nt = t / KGT, kt = t % KGT;
                const float *ap = ds + (lane >> 5) * NG + nt * 32 + (lane & 31);
                const float *bp = xs + (lane >> 5) * KG + kt * 32 + (lane & 31);
#pragma unroll 4
                for (int s = 0; s < 16; ++s)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * NG], bp[2 * s * KG], acc[j], 0, 0, 0);
            }
        }
        if (want_bias && blockIdx.y == 0 && wave == 0 && lane < NG) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) bsum += ds[r * NG + lane];
        }
        __syncthreads();
    }
    float *__restrict__ part = partials + (int64_t)blockIdx.x * ((int64_t)N * K + N);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = wave + 4 * j;
        if (t >= TILES) continue;
        const int nt = t / KGT, kt = t % KGT;
        const int32_t k = k_base + kt * 32 + (lane & 31);
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int32_t n = n_base + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n < N) part[(int64_t)n * K + k] = acc[j][r];
        }
    }
    if (want_bias && blockIdx.y == 0 && wave == 0 && lane < NG && n_base + lane < N) part[(int64_t)N * K + n_base + lane] = bsum;
}

// Streaming dW on the bf16 matrix pipe with fp32-equivalent results [r3].
// dW[n][k] = sum_m dPre[m][n] X[m][k] as v_mfma_f32_32x32x16_bf16: i = n, j = k, the reduction index is the ROW.  Lane
// (j = lane & 31, kg = lane >> 5) supplies the eight rows m0 + 8 kg .. + 7 of ONE column of each operand: a load
// instruction covers two 128-byte (x tiles) row pieces, i.e. row-major operands stream from global memory straight into
// the matrix-core layout -- no LDS staging, no transposes, no barrier in the row loop.  Each fp32 value is split exactly
// into three bf16 terms (row pairs share a register) and a product is the six partial products of order <= 2^-16, smallest
// first, accumulated in fp32 (csrc/mlp_fused.hip has the error argument: the result differs from an fp32 FMA chain by
// fp32-roundoff-sized terms).  Six K = 16 instructions of 32 cycles replace eight K = 2 instructions of 64 cycles per
// 16 rows and tile: 0.375x the matrix time, which takes the matrix pipe off the critical path -- the kernel is a pure
// HBM stream.  Each wave owns a quarter of the workgroup's rows and the whole N x K tile (NT x KT accumulators of 32x32);
// two 16-row blocks of loads are in flight ahead of the block being multiplied.  The four waves are summed through LDS
// once at the end.  Column k of the virtual concatenation maps to "base + m * stride" for row-major and level-major
// segments alike, so the per-lane addressing is hoisted out of the loop.

template <int W> struct VecF;
template <> struct VecF<1> { using T = float; };
template <> struct VecF<2> { using T = float2; };
template <> struct VecF<4> { using T = float4; };
template <int W> __device__ __forceinline__ void vec_load(const float *p, float (&v)[W]) {
    typename VecF<W>::T t = *reinterpret_cast<const typename VecF<W>::T *>(p);
    const float *f = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = f[i];
}

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
#define EMER_MF32(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

// VEC = false: tile t of an operand holds columns 32 t + j (one dword per tile per row).
// VEC = true : tile t holds columns T j + t (T = tiles of that operand) -- the assignment of columns to MFMA tiles is
//              free, and with this one a lane's T columns of a row are CONTIGUOUS: one 4*T-byte load per operand per row
//              instead of T dword loads.  Needs every segment boundary and leading dimension to be a multiple of T.
// C0 = true: column 0 of dPre is replaced by its own array (sx.col0, VEC layouts; the scalar layout redirects lane 0's pointer).
// FULL = true: every lane's column exists (N = 32 NT, K = 32 KT): no column masks.
// BIAS = false: no bias-gradient accumulators (the widest tile has no registers to spare for them).
template <int NT, int KT, bool VEC, bool C0, bool FULL, bool BIAS = true>
__global__ __launch_bounds__(256) void wgrad_stream_kernel(const float *__restrict__ dpre, int64_t ldd, const SegX sx,
                                                           float *__restrict__ partials, int64_t M, int32_t N, int32_t K,
                                                           int32_t rows_per_block, int want_bias) {
    constexpr int KP = KT * 32;
    constexpr int AG = VEC ? 1 : NT, AW = VEC ? NT : 1;  // A operand: AG loads of AW floats per row
    constexpr int BG = VEC ? 1 : KT, BW = VEC ? KT : 1;
    __shared__ float red[NT * 32 * KP + NT * 32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_end = (r_begin + rows_per_block < M) ? r_begin + rows_per_block : M;
    const int32_t rpw = rows_per_block >> 2;  // rows_per_block is a multiple of 32
    const int64_t w_begin = r_begin + (int64_t)wave * rpw;
    const int64_t w_end = (w_begin + rpw < r_end) ? w_begin + rpw : r_end;

    // first column of load group q: VEC: q = 0, columns T j .. T j + T - 1;  scalar: column 32 q + j.
    // Lanes whose column does not exist read element 0 with stride 0 and multiply by 0: no predicate (= no branch, and
    // no generic-address-space pointer: a flat_load would force s_waitcnt vmcnt(0) and serialise the stream) on any load.
    const float *ap[AG];
    int64_t ast[AG];
    float amul[AG];
#pragma unroll
    for (int q = 0; q < AG; ++q) {
        const int32_t n0 = VEC ? NT * j : q * 32 + j;
        const bool ok = n0 < N;  // VEC: N is a multiple of NT (host check)
        ap[q] = dpre + (ok ? n0 : 0);
        ast[q] = ok ? ldd : 0;
        amul[q] = ok ? 1.0f : 0.0f;
    }
    const bool c0 = sx.col0 != nullptr && j == 0;  // column 0 of dPre comes from its own array (trunc_exp side gradient merged)
    if (!VEC && c0) { ap[0] = sx.col0; ast[0] = 1; }
    const float *bp[BG];
    int64_t bst[BG];
    float bmul[BG];
#pragma unroll
    for (int q = 0; q < BG; ++q) {
        const int32_t kk = VEC ? KT * j : q * 32 + j;
        bp[q] = sx.s[0].ptr; bst[q] = 0; bmul[q] = 0.0f;
#pragma unroll
        for (int sg = 0; sg < EMER_CHAIN_MAX_SEGS; ++sg) {
            if (sg < sx.n && kk < K && kk >= sx.s[sg].col && kk < sx.s[sg].col + sx.s[sg].width) {
                const int32_t c = kk - sx.s[sg].col;
                bmul[q] = 1.0f;
                if (sx.s[sg].mode == 1) {
                    const int32_t f = sx.s[sg].f, lv = c / f;
                    bp[q] = sx.s[sg].ptr + (int64_t)lv * sx.s[sg].n_total * f + (c - lv * f);
                    bst[q] = f;
                } else {
                    bp[q] = sx.s[sg].ptr + c;
                    bst[q] = sx.s[sg].ld;
                }
            }
        }
    }

    f32x16 acc[NT][KT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) acc[a][b] = f32x16{0};
    // bias gradient = dPre^T 1: one more B tile, all ones (every column of the result holds the column sums of dPre).  Three
    // instructions per A tile and block on the pipe that is already there, instead of a per-lane fp32 side sum.
    f32x16 accb[BIAS ? NT : 1];
#pragma unroll
    for (int a = 0; a < (BIAS ? NT : 1); ++a) accb[a] = f32x16{0};
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    // per-lane pointers to row 8 kg of the current block; the block loop advances them by 16 rows
#pragma unroll
    for (int q = 0; q < AG; ++q) ap[q] += (w_begin + 8 * kg) * ast[q];
#pragma unroll
    for (int q = 0; q < BG; ++q) bp[q] += (w_begin + 8 * kg) * bst[q];
    const float *c0p = C0 ? sx.col0 + (c0 ? w_begin + 8 * kg : 0) : nullptr;  // other lanes re-read element 0 (unused)
    const int64_t c0st = c0 ? 1 : 0;
    // one 16-row block of both operands: rows m0 + 8 kg + r
    struct Blk { float a[8][NT], b[8][KT], c[C0 ? 8 : 1]; };
    // `ahead` blocks past the pointers.  tail = true_type (last block of a ragged range only): rows past w_end re-read
    // the last valid row and are multiplied by 0.
    auto load = [&](auto tail, int64_t m0, int ahead, Blk &v) {
        constexpr bool TAIL = decltype(tail)::value;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const bool ok = !TAIL || m0 + 8 * kg + r < w_end;
            const int64_t dr = ok ? (int64_t)(16 * ahead + r) : w_end - 1 - (m0 + 8 * kg);
            const float rmul = ok ? 1.0f : 0.0f;
#pragma unroll
            for (int q = 0; q < AG; ++q) {
                float t[AW];
                vec_load<AW>(ap[q] + dr * ast[q], t);
#pragma unroll
                for (int i = 0; i < AW; ++i) {
                    if (!FULL) t[i] *= amul[q];
                    if (TAIL) t[i] *= rmul;
                    v.a[r][q * AW + i] = t[i];
                }
            }
            if constexpr (VEC && C0) v.c[r] = c0p[dr * c0st] * (TAIL ? rmul : 1.0f);  // merged at use (no wait inside the load burst)
#pragma unroll
            for (int q = 0; q < BG; ++q) {
                float t[BW];
                vec_load<BW>(bp[q] + dr * bst[q], t);
#pragma unroll
                for (int i = 0; i < BW; ++i) {
                    if (!FULL) t[i] *= bmul[q];
                    if (TAIL) t[i] *= rmul;
                    v.b[r][q * BW + i] = t[i];
                }
            }
        }
    };
    auto advance = [&](int blocks) {
#pragma unroll
        for (int q = 0; q < AG; ++q) ap[q] += 16 * blocks * ast[q];
#pragma unroll
        for (int q = 0; q < BG; ++q) bp[q] += 16 * blocks * bst[q];
        c0p += 16 * blocks * c0st;
    };
    auto mult = [&](const Blk &v) {
        u32x4 ah[NT], am[NT], al[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x0 = v.a[2 * q][a], x1 = v.a[2 * q + 1][a];
                if constexpr (VEC && C0) {
                    if (a == 0 && c0) { x0 = v.c[2 * q]; x1 = v.c[2 * q + 1]; }
                }
                unsigned h, m, l;
                split3(x0, x1, h, m, l);
                ah[a][q] = h; am[a][q] = m; al[a][q] = l;
            }
        }
        if (BIAS && want_bias) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                accb[a] = EMER_MF32(al[a], ones, accb[a]);
                accb[a] = EMER_MF32(am[a], ones, accb[a]);
                accb[a] = EMER_MF32(ah[a], ones, accb[a]);
            }
        }
#pragma unroll
        for (int b = 0; b < KT; ++b) {
            u32x4 bh, bm, bl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned h, m, l;
                split3(v.b[2 * q][b], v.b[2 * q + 1][b], h, m, l);
                bh[q] = h; bm[q] = m; bl[q] = l;
            }
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                acc[a][b] = EMER_MF32(al[a], bh, acc[a][b]);
                acc[a][b] = EMER_MF32(ah[a], bl, acc[a][b]);
                acc[a][b] = EMER_MF32(am[a], bm, acc[a][b]);
                acc[a][b] = EMER_MF32(am[a], bh, acc[a][b]);
                acc[a][b] = EMER_MF32(ah[a], bm, acc[a][b]);
                acc[a][b] = EMER_MF32(ah[a], bh, acc[a][b]);
            }
            __builtin_amdgcn_sched_barrier(0);  // one operand tile of splits live at a time
        }
    };
    // Full blocks stream with one block of loads in flight ahead of the one in the matrix pipe (two for small tiles,
    // whose accumulators leave room); the ragged remainder (< 16 rows, last workgroup only) takes the predicated path.
    // Every load of the steady state is UNCONDITIONAL (past the end the last block is simply read again): a load under a
    // branch merges with the buffer's previous contents, and hipcc then copies the loaded registers right behind the
    // load burst and waits for it -- the stream degenerates to load, wait, multiply.
    const int64_t n_full = w_end > w_begin ? (w_end - w_begin) >> 4 : 0;
    const int64_t m_tail = w_begin + 16 * n_full;
    constexpr std::false_type kFull{};
    constexpr std::true_type kTail{};
    if (n_full > 0) {
        const int64_t last = n_full - 1;
        auto ahead = [&](int64_t i, int d) { return (int)((i + d <= last ? i + d : last) - i); };  // blocks past the pointers (clamped)
        int64_t i = 0;
        {
            Blk v0, v1, v2;   // three 16-row blocks rotate: two blocks of loads in flight ahead of the one being multiplied
            load(kFull, 0, 0, v0);
            load(kFull, 0, ahead(0, 1), v1);
            for (; i + 2 < n_full; i += 3) {
                load(kFull, 0, 2, v2);
                mult(v0);
                load(kFull, 0, ahead(i, 3), v0);
                mult(v1);
                load(kFull, 0, ahead(i, 4), v1);
                mult(v2);
                advance(3);
            }
            if (i < n_full) mult(v0);
            if (i + 1 < n_full) mult(v1);
            advance((int)(n_full - i));
        }
    }
    if (m_tail < w_end) {
        Blk vt;
        load(kTail, m_tail, 0, vt);
        mult(vt);
    }
    // ---- sum the four waves through LDS (wave 0 writes, 1..3 add in turn), then one coalesced store of the partial.
    // red is indexed by ACTUAL (n, k): tile (a, b), element (i, j)  ->  VEC: n = NT i + a, k = KT j + b;  else n = 32 a + i, k = 32 b + j
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
#pragma unroll
                for (int b = 0; b < KT; ++b) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int i = (v & 3) + 8 * (v >> 2) + 4 * kg;
                        const int n = VEC ? NT * i + a : a * 32 + i, k = VEC ? KT * j + b : b * 32 + j;
                        float *q = red + n * KP + k;
                        *q = (w == 0) ? acc[a][b][v] : *q + acc[a][b][v];
                    }
                }
                if (BIAS && j == 0) {   // column 0 of the ones product: dbias[n(a, i)]
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int i = (v & 3) + 8 * (v >> 2) + 4 * kg;
                        float *q = red + NT * 32 * KP + (VEC ? NT * i + a : a * 32 + i);
                        *q = (w == 0) ? accb[a][v] : *q + accb[a][v];
                    }
                }
            }
        }
        __syncthreads();
    }
    float *__restrict__ part = partials + (int64_t)blockIdx.x * ((int64_t)N * K + N);
    for (int idx = tid; idx < N * K; idx += 256) {
        const int n = idx / K, k = idx - n * K;
        part[idx] = red[n * KP + k];
    }
    if (want_bias)
        for (int n = tid; n < N; n += 256) part[(int64_t)N * K + n] = red[NT * 32 * KP + n];
}

}  // namespace emer

extern "C" int emer_mlp_chain(const emer_chain_desc *d, int64_t n_rows, void *stream) {
    EMER_REQUIRE(d != nullptr && n_rows >= 0, "mlp_chain: bad arguments");
    if (n_rows == 0) return EMER_OK;
    EMER_REQUIRE(d->n_layers >= 1 && d->n_layers <= EMER_CHAIN_MAX_LAYERS && d->n_segs >= 1 && d->n_segs <= EMER_CHAIN_MAX_SEGS,
                 "mlp_chain: n_layers=%d / n_segs=%d out of range", d->n_layers, d->n_segs);
    EMER_REQUIRE(d->buf_cols >= 4 && d->buf_cols <= 512, "mlp_chain: buf_cols=%d out of range", d->buf_cols);
    for (int s = 0; s < d->n_segs; ++s) {
        const emer_chain_seg &S = d->segs[s];
        EMER_REQUIRE(S.ptr && S.width >= 1 && S.col >= 0 && S.col + S.width <= d->buf_cols, "mlp_chain: bad segment %d", s);
        EMER_REQUIRE((S.mode == 0 && S.row_div >= 1) || (S.mode == 1 && S.f >= 1 && S.width % S.f == 0), "mlp_chain: bad segment mode %d", s);
        EMER_REQUIRE(!S.fix_a || S.fix_b, "mlp_chain: fix_a needs fix_b");
    }
    for (int l = 0; l < d->n_layers; ++l) {
        const emer_chain_layer &L = d->layers[l];
        EMER_REQUIRE(L.w && L.K >= 1 && L.N >= 1 && L.in_col >= 0 && L.out_col >= 0, "mlp_chain: bad layer %d", l);
        EMER_REQUIRE((L.in_col & 1) == 0, "mlp_chain: layer %d: in_col must be even (8-byte LDS reads)", l);
        EMER_REQUIRE(L.in_col + ((L.K + 7) / 8) * 8 <= d->buf_cols + 8 && L.out_col + L.N <= d->buf_cols, "mlp_chain: layer %d leaves the row buffer", l);
        EMER_REQUIRE(L.act >= EMER_ACT_NONE && L.act <= EMER_ACT_TRUNC_EXP, "mlp_chain: layer %d: unknown activation", l);
        EMER_REQUIRE(!L.store || (L.store_n >= 1 && L.store_col + L.store_n <= L.N && (L.store_mode == 0 || (L.store_mode == 1 && L.store_f >= 1))),
                     "mlp_chain: layer %d: bad store", l);
    }
    const ChainLds lp = chain_lds_plan(d);
    // as many waves per workgroup as the LDS allows (each owns a 16-row buffer), at least 4, at most 16:
    // with small weight sets this gives 2-4 waves per SIMD to hide LDS / HBM latency
    int n_waves = 16;
    while (n_waves > 4 && ((size_t)lp.w_total + (size_t)n_waves * 16 * lp.P) * sizeof(float) > 160 * 1024) n_waves -= 2;
    const size_t lds = ((size_t)lp.w_total + (size_t)n_waves * 16 * lp.P) * sizeof(float);
    EMER_REQUIRE(lds <= 160 * 1024, "mlp_chain: chain needs %zu B of LDS (> 160 KiB); split it", lds);
    if (int rc = reserve_lds(reinterpret_cast<const void *>(mlp_chain_kernel), lds, "mlp_chain")) return rc;
    const int64_t n_tiles = ceil_div(n_rows, 16);
    int64_t grid = 256;  // persistent: one workgroup per CU
    if (grid > ceil_div(n_tiles, n_waves)) grid = ceil_div(n_tiles, n_waves);
    hipLaunchKernelGGL(mlp_chain_kernel, dim3((uint32_t)grid), dim3(64 * n_waves), lds, as_stream(stream), *d, lp, n_rows, n_tiles);
    return check_launch("mlp_chain");
}

extern "C" int emer_wgrad_segmented(const float *dpre, int64_t ldd, const float *col0, const emer_chain_seg *segs,
                                    int32_t n_segs, float *workspace, float *dw, int64_t ld_dw, float *dbias, int64_t m, int32_t n,
                                    int32_t k, void *stream) {
    EMER_REQUIRE(m >= 0 && n >= 1 && k >= 1, "wgrad_segmented: bad sizes");
    if (m == 0) return EMER_OK;
    EMER_REQUIRE(dpre && segs && workspace && dw && n_segs >= 1 && n_segs <= EMER_CHAIN_MAX_SEGS, "wgrad_segmented: bad arguments");
    SegX sx;
    sx.n = n_segs; sx.col0 = col0;
    DwDst ddst = dw_dst_identity(k);
    ddst.n = n_segs; ddst.ld = ld_dw;
    int32_t covered = 0;
    for (int s = 0; s < n_segs; ++s) {
        EMER_REQUIRE(segs[s].dst_col >= 0 && segs[s].dst_col + segs[s].width <= ld_dw, "wgrad_segmented: segment %d lands outside dw (ld_dw=%lld)", s, (long long)ld_dw);
        ddst.col[s] = segs[s].col; ddst.width[s] = segs[s].width; ddst.dst[s] = segs[s].dst_col;
        EMER_REQUIRE(segs[s].ptr && segs[s].col == covered && ((segs[s].mode == 0 && segs[s].row_div >= 1) || (segs[s].mode == 1 && segs[s].f >= 1)),
                     "wgrad_segmented: segments must be contiguous, mode 0 or 1");
        sx.s[s] = segs[s];
        covered += segs[s].width;
    }
    for (int s = n_segs; s < EMER_CHAIN_MAX_SEGS; ++s) sx.s[s] = segs[0];
    EMER_REQUIRE(covered == k, "wgrad_segmented: segments cover %d columns, k = %d", covered, k);
    hipStream_t st = as_stream(stream);
    const int32_t NG = n <= 32 ? 32 : 64;
    const int32_t KG = k <= 32 ? 32 : (k <= 64 ? 64 : (k <= 128 ? 128 : 256));
    const int32_t rpb = dw_rows_per_block(m, n, k);
    const int32_t n_row_blocks = (int32_t)ceil_div(m, rpb);
    bool stream_ok = n <= 64 && k <= 128;
    for (int s = 0; s < n_segs; ++s) stream_ok = stream_ok && (segs[s].mode == 1 || segs[s].row_div == 1);
    const int NT = n <= 32 ? 1 : 2;
    int KT = (k + 31) / 32;
    // vector loads (one per operand per row) when every boundary is a multiple of the vector width
    const int KTv = KT == 3 ? 4 : KT;
    bool vec = stream_ok && (n % NT == 0) && (ldd % NT == 0) && ((uintptr_t)dpre % (4 * NT) == 0) && (k % KTv == 0) && (NT * KTv > 1);
    for (int s = 0; s < n_segs && vec; ++s) {
        const emer_chain_seg &S = segs[s];
        vec = (S.col % KTv == 0) && (S.width % KTv == 0) && ((uintptr_t)S.ptr % (4 * KTv) == 0) &&
              (S.mode == 1 ? (S.f % KTv == 0) : (S.ld % KTv == 0));
    }
    if (vec) KT = KTv;
    const bool full = vec && n == 32 * NT && k == 32 * KT;  // every lane owns real columns: no masks
    // the widest tile (64 x 128) streams only in its mask-free vector form: the masked / scalar forms of that tile do not fit the
    // register file (hundreds of spilled registers); such shapes (no shipped head) take the LDS-staged kernel below
    if (NT * KT >= 8 && !full) stream_ok = false;
    if (stream_ok) {  // no per-ray operand: operands stream straight into the MFMA layout
        const dim3 sgrid((uint32_t)n_row_blocks);
#define EMER_WSK(A, B, V, C, F, BI) hipLaunchKernelGGL((wgrad_stream_kernel<A, B, V, C, F, BI>), sgrid, dim3(256), 0, st, dpre, ldd, sx, workspace, m, n, k, rpb, dbias ? 1 : 0)
#define EMER_WSL(A, B, V, C, F) do { if (A * B >= 8 && !dbias) EMER_WSK(A, B, V, C, F, (A * B < 8)); else EMER_WSK(A, B, V, C, F, true); } while (0)
#define EMER_WS(A, B) do { if (vec && col0 && full) EMER_WSL(A, B, true, true, true); else if (vec && col0) EMER_WSL(A, B, true, true, false); \
                           else if (vec && full) EMER_WSL(A, B, true, false, true); else if (vec) EMER_WSL(A, B, true, false, false); \
                           else EMER_WSL(A, B, false, false, false); } while (0)
#define EMER_WS3(A) EMER_WSL(A, 3, false, false, false)
        if (NT == 1) { if (KT == 1) EMER_WS(1, 1); else if (KT == 2) EMER_WS(1, 2); else if (KT == 3) EMER_WS3(1); else EMER_WS(1, 4); }
        else         { if (KT == 1) EMER_WS(2, 1); else if (KT == 2) EMER_WS(2, 2); else if (KT == 3) EMER_WS3(2); else EMER_WS(2, 4); }
#undef EMER_WS
#undef EMER_WS3
#undef EMER_WSL
#undef EMER_WSK
        if (int rc = check_launch("wgrad_stream")) return rc;
        const int64_t stride = (int64_t)n * k + n;
        hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3((uint32_t)ceil_div(stride, 256), dw_reduce_splits(n_row_blocks, stride)), dim3(256), 0, st,
                           workspace, n_row_blocks, stride, (int64_t)n * k, dw, dbias, ddst);
        return check_launch("wgrad_reduce");
    }
    const dim3 grid((uint32_t)n_row_blocks, (uint32_t)ceil_div(k, KG), (uint32_t)ceil_div(n, NG));
#define EMER_WG(A, B) hipLaunchKernelGGL((wgrad_seg_kernel<A, B>), grid, dim3(256), 0, st, dpre, ldd, sx, workspace, m, n, k, rpb, dbias ? 1 : 0)
    if (NG == 32) { if (KG == 32) EMER_WG(1, 1); else if (KG == 64) EMER_WG(1, 2); else if (KG == 128) EMER_WG(1, 4); else EMER_WG(1, 8); }
    else          { if (KG == 32) EMER_WG(2, 1); else if (KG == 64) EMER_WG(2, 2); else if (KG == 128) EMER_WG(2, 4); else EMER_WG(2, 8); }
#undef EMER_WG
    if (int rc = check_launch("wgrad_segmented")) return rc;
    const int64_t stride = (int64_t)n * k + n;
    hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3((uint32_t)ceil_div(stride, 256), dw_reduce_splits(n_row_blocks, stride)), dim3(256), 0, st, workspace, n_row_blocks, stride,
                       (int64_t)n * k, dw, dbias, ddst);
    return check_launch("wgrad_reduce");
}
