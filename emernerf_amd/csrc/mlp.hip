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

// dw[i] += sum_b partials[b][i]  (i < N*K),  dbias[i - N*K] += ...  (i >= N*K)
__global__ __launch_bounds__(256) void linear_dw_reduce_kernel(const float *__restrict__ partials, int32_t n_blocks, int64_t stride,
                                                               int64_t nk, float *__restrict__ dw, float *__restrict__ dbias) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= stride) return;
    float a = 0.0f;
    for (int32_t b = 0; b < n_blocks; ++b) a += partials[(int64_t)b * stride + i];
    if (i < nk) dw[i] += a;
    else if (dbias) dbias[i - nk] += a;
}

constexpr int32_t kDwRowsPerBlock = 1024;

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
    return ceil_div(m, kDwRowsPerBlock) * ((int64_t)n * k + n);
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
        const int32_t n_row_blocks = (int32_t)ceil_div(m, kDwRowsPerBlock);
        const dim3 grid((uint32_t)n_row_blocks, (uint32_t)ceil_div(k, KG), (uint32_t)ceil_div(n, NG));
#define EMER_DW(A, B) hipLaunchKernelGGL((linear_dw_kernel<A, B>), grid, dim3(256), 0, st, dy, lddy, ya, ldy, act, d_aux_density, \
                                         aux_density, x, ldx, workspace, m, n, k, kDwRowsPerBlock, dbias ? 1 : 0)
        if (NG == 32) { if (KG == 32) EMER_DW(1, 1); else if (KG == 64) EMER_DW(1, 2); else if (KG == 128) EMER_DW(1, 4); else EMER_DW(1, 8); }
        else          { if (KG == 32) EMER_DW(2, 1); else if (KG == 64) EMER_DW(2, 2); else if (KG == 128) EMER_DW(2, 4); else EMER_DW(2, 8); }
#undef EMER_DW
        if (int rc = check_launch("linear_dw")) return rc;
        const int64_t stride = (int64_t)n * k + n;
        hipLaunchKernelGGL(linear_dw_reduce_kernel, dim3((uint32_t)ceil_div(stride, 256)), dim3(256), 0, st, workspace, n_row_blocks,
                           stride, (int64_t)n * k, dw, dbias);
        if (int rc = check_launch("linear_dw_reduce")) return rc;
    }
    return EMER_OK;
}
