// Per-ray side of the colour heads: everything that is evaluated once per RAY (8192 rows) instead of once per sample.
//
//   emer_ray_inputs_fwd     [dir-PE | appearance embedding] rows of the rgb head (remapped directions) and of the sky head
//                           (raw directions) in one launch        (radiance_field.py:622-643, 660-674; encodings.py:86-104)
//   emer_embed_grad         gradient of the embedding table: a deterministic per-row segment sum of both consumers' input
//                           gradients (replaces autograd's add + zero fill + atomic index_add)
//   emer_ray_pre_fwd/bwd    the per-ray operand's share of the rgb head's layers 0 and 1 (mlp.py:38-46 with
//                           skip_connections=[1]): pre-activation offsets rb = [W0[:, :Kh] h + b0 | W1[:, H:H+Kh] h + b1]
//                           and d h = s0 W0[:, :Kh] + s1 W1[:, H:H+Kh], reading the weight blocks in place (row strides);
//                           also the input's share of the sky head's layers 0 and 1
//   emer_ray_head_fwd/bwd   the rest of the sky head (mlp.MLP, 3 layers, skip at 1, hidden 64) and its data gradients
//   emer_ray_wgrad          weight / bias gradients of per-ray layers, several layers per launch
//
// A few MFLOP each: what they cost is launches and memory LATENCY, so each is ONE launch (the torch / generic
// formulation was 3-6 launches apiece) whose loads are issued in large independent batches.
#include "common.h"

namespace emer {

// These kernels run on a few thousand rows: there is one workgroup per CU at most, so what they cost is LATENCY, not
// throughput.  Every global read is therefore issued in batches of independent loads (EMER_BATCH) before anything consumes
// them, and the workgroups are 1024 threads wide so that a CU has four waves per SIMD to overlap.
#define EMER_BATCH 8

// one element of a direction row: [x (3) | sin(2^k x_d) (k-major) | sin(2^k x_d + pi/2)]   (same expressions as
// dir_encode_kernel of elementwise.hip, so the rows are bit-identical)
__device__ __forceinline__ float encode_elem(const float x[3], int32_t j, int32_t n_deg) {
    if (j < 3) return x[j];
    const float half_pi = 0.5f * 3.14159265358979323846f;
    int32_t t = j - 3;
    const bool shifted = t >= 3 * n_deg;
    if (shifted) t -= 3 * n_deg;
    const int32_t k = t / 3, d = t - 3 * k;
    const float xb = x[d] * ldexpf(1.0f, k);  // the reference's scale doubles per degree: exact powers of two
    return shifted ? sinf(xb + half_pi) : sinf(xb);
}

// thread per (ray, output, column)
__global__ __launch_bounds__(256) void ray_inputs_kernel(const float *__restrict__ dirs, int64_t ld_dirs, const int64_t *__restrict__ idx,
                                                         int64_t idx_stride, const float *__restrict__ emb, int32_t n_emb, int32_t E,
                                                         int32_t max_deg, int64_t R, float *__restrict__ out_rgb, int64_t ld_rgb,
                                                         float *__restrict__ out_sky, int64_t ld_sky) {
    const int32_t n_deg = max_deg + 1;
    const int32_t P = max_deg == 0 ? 3 : 3 * (1 + 2 * n_deg), W = P + E;
    const int64_t total = R * 2 * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (2 * W);
        const int32_t rem = (int32_t)(i - r * 2 * W);
        const int32_t which = rem >= W, j = rem - which * W;
        float *__restrict__ out = which ? out_sky : out_rgb;
        if (!out) continue;
        float v;
        if (j < P) {
            float x[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float raw = dirs[r * ld_dirs + d];
                x[d] = which ? raw : (raw + 1.0f) / 2.0f;
            }
            v = encode_elem(x, j, n_deg);
        } else {
            const int64_t e = idx ? idx[r * idx_stride] : 0;
            v = (e >= 0 && e < n_emb) ? emb[e * E + (j - P)] : __builtin_nanf("");  // torch raises a device assert here
        }
        out[r * (which ? ld_sky : ld_rgb) + j] = v;
    }
}

// One 1024-lane workgroup per embedding row.  Pass 1: the sixteen waves scan the ray indices (eight independent loads per
// lane per 8192 rays) and compact the matching rays into per-wave lists in scan order (ballot + prefix count: deterministic
// positions).  Pass 2: lane l of a wave takes list entries l, l+64, ... two at a time and loads their gradient rows (all loads
// in flight together).  A butterfly over the lanes and a fixed-order sum over the waves finish: deterministic, no atomics, no
// workspace, and ~3 dependent memory round trips at 8192 rays however few rows the table has.
constexpr int kEmbedThreads = 1024;
constexpr int kEmbedSuper = kEmbedThreads * EMER_BATCH;  // rays per scan round (512 list slots per wave)
template <int EC>
__global__ __launch_bounds__(kEmbedThreads) void embed_grad_kernel(const float *__restrict__ ga, int64_t ld_a, const float *__restrict__ gb,
                                                                   int64_t ld_b, const int64_t *__restrict__ idx, int64_t idx_stride,
                                                                   int64_t R, int32_t E, float *__restrict__ dw) {
    __shared__ int32_t list[kEmbedThreads / 64][kEmbedSuper / (kEmbedThreads / 64)];
    __shared__ float wred[kEmbedThreads / 64][EC + 1];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int32_t c0 = 0; c0 < E; c0 += EC) {
        float acc[EC];
#pragma unroll
        for (int c = 0; c < EC; ++c) acc[c] = 0.f;
        for (int64_t sbase = 0; sbase < R; sbase += kEmbedSuper) {
            int64_t e[EMER_BATCH];
#pragma unroll
            for (int u = 0; u < EMER_BATCH; ++u) {
                const int64_t r = sbase + u * kEmbedThreads + tid;
                e[u] = r < R ? idx[r * idx_stride] : -1;
            }
            int32_t cnt = 0;  // wave-uniform
#pragma unroll
            for (int u = 0; u < EMER_BATCH; ++u) {
                const bool m = e[u] == row;
                const uint64_t mask = __ballot(m);
                if (m) list[wave][cnt + __popcll(mask & ((1ull << lane) - 1ull))] = u * kEmbedThreads + tid;
                cnt += __popcll(mask);
            }
            __syncthreads();
            for (int32_t i = lane; i < cnt; i += 128) {
                const bool two = i + 64 < cnt;
                const int64_t ra = sbase + list[wave][i], rb = sbase + list[wave][two ? i + 64 : i];
                float va[EC], vb[EC], wa[EC], wb[EC];
#pragma unroll
                for (int c = 0; c < EC; ++c) {
                    const bool in = c0 + c < E;
                    va[c] = (ga && in) ? ga[ra * ld_a + c0 + c] : 0.f;
                    vb[c] = (gb && in) ? gb[ra * ld_b + c0 + c] : 0.f;
                    wa[c] = (ga && in && two) ? ga[rb * ld_a + c0 + c] : 0.f;
                    wb[c] = (gb && in && two) ? gb[rb * ld_b + c0 + c] : 0.f;
                }
#pragma unroll
                for (int c = 0; c < EC; ++c) acc[c] += (va[c] + vb[c]) + (wa[c] + wb[c]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int c = 0; c < EC; ++c) acc[c] += __shfl_xor(acc[c], off);
        if (lane == 0)
#pragma unroll
            for (int c = 0; c < EC; ++c) wred[wave][c] = acc[c];
        __syncthreads();
        if (tid < EC && c0 + tid < E) {
            float sum = 0.f;
            for (int w = 0; w < kEmbedThreads / 64; ++w) sum += wred[w][tid];
            dw[row * E + c0 + tid] += sum;
        }
        __syncthreads();
    }
}

constexpr int kRayTile = 32;     // rows per workgroup
constexpr int kRayThreads = 1024;

// dst[i] = src(i) for i < total, all loads of a batch in flight together
template <typename Src, typename Dst>
__device__ __forceinline__ void stage_batched(int32_t total, Src src, Dst dst) {
    for (int32_t base = 0; base < total; base += kRayThreads * EMER_BATCH) {
        float v[EMER_BATCH];
#pragma unroll
        for (int u = 0; u < EMER_BATCH; ++u) {
            const int32_t i = base + u * kRayThreads + (int32_t)threadIdx.x;
            v[u] = i < total ? src(i) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < EMER_BATCH; ++u) {
            const int32_t i = base + u * kRayThreads + (int32_t)threadIdx.x;
            if (i < total) dst(i, v[u]);
        }
    }
}

// rb[r][0:H] = wa[H][Kh] h[r] + ba ; rb[r][H:2H] = wb[H][Kh] h[r] + bb      (H <= 64, Kh <= 64; 32 rays per workgroup)
// lane (c = tid & 127, q = tid >> 7) owns output column c of rays 4q..4q+3
__global__ __launch_bounds__(kRayThreads) void ray_pre_fwd_kernel(const float *__restrict__ h, int64_t ld_h, int64_t R, int32_t Kh, int32_t H,
                                                                  const float *__restrict__ wa, int64_t ld_wa, const float *__restrict__ ba,
                                                                  const float *__restrict__ wb, int64_t ld_wb, const float *__restrict__ bb,
                                                                  float *__restrict__ rb, int64_t ld_rb) {
    extern __shared__ float lds_f[];
    const int32_t ldw = Kh | 1, N = 2 * H;
    float *ws = lds_f, *hs = lds_f + 128 * ldw;
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * kRayTile;
    stage_batched(N * Kh,
                  [&](int32_t i) { const int32_t c = i / Kh, k = i - c * Kh; return c < H ? wa[c * ld_wa + k] : wb[(c - H) * ld_wb + k]; },
                  [&](int32_t i, float v) { const int32_t c = i / Kh, k = i - c * Kh; ws[c * ldw + k] = v; });
    stage_batched(kRayTile * Kh,
                  [&](int32_t i) { const int32_t rr = i / Kh, k = i - rr * Kh; return r0 + rr < R ? h[(r0 + rr) * ld_h + k] : 0.f; },
                  [&](int32_t i, float v) { const int32_t rr = i / Kh, k = i - rr * Kh; hs[rr * ldw + k] = v; });
    __syncthreads();
    const int32_t c = tid & 127, q = tid >> 7;
    if (c >= N) return;
    float acc[4];
    const float bias = c < H ? (ba ? ba[c] : 0.f) : (bb ? bb[c - H] : 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = bias;
#pragma unroll 4
    for (int32_t k = 0; k < Kh; ++k) {
        const float w = ws[c * ldw + k];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(w, hs[(q * 4 + i) * ldw + k], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + q * 4 + i;
        if (r < R) rb[r * ld_rb + c] = acc[i];
    }
}

// dh[r][k] = sum_c s0[r][c] wa[c][k] + s1[r][c] wb[c][k];   lane (k = tid & 63, q = tid >> 6) owns column k of rays 2q, 2q+1
__global__ __launch_bounds__(kRayThreads) void ray_pre_bwd_kernel(const float *__restrict__ s0, const float *__restrict__ s1, int64_t ld_s,
                                                                  int64_t R, int32_t Kh, int32_t H, const float *__restrict__ wa, int64_t ld_wa,
                                                                  const float *__restrict__ wb, int64_t ld_wb, float *__restrict__ dh,
                                                                  int64_t ld_dh) {
    extern __shared__ float lds_f[];
    const int32_t N = 2 * H, lds = N + 1;
    float *ws = lds_f, *ss = lds_f + N * Kh;
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * kRayTile;
    stage_batched(N * Kh,
                  [&](int32_t i) { const int32_t c = i / Kh, k = i - c * Kh; return c < H ? wa[c * ld_wa + k] : wb[(c - H) * ld_wb + k]; },
                  [&](int32_t i, float v) { ws[i] = v; });
    stage_batched(kRayTile * N,
                  [&](int32_t i) {
                      const int32_t rr = i / N, c = i - rr * N;
                      if (r0 + rr >= R) return 0.f;
                      return c < H ? s0[(r0 + rr) * ld_s + c] : s1[(r0 + rr) * ld_s + c - H];
                  },
                  [&](int32_t i, float v) { const int32_t rr = i / N, c = i - rr * N; ss[rr * lds + c] = v; });
    __syncthreads();
    const int32_t k = tid & 63, q = tid >> 6;
    if (k >= Kh) return;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 4
    for (int32_t c = 0; c < N; ++c) {
        const float w = ws[c * Kh + k];
        acc0 = fmaf(ss[(q * 2) * lds + c], w, acc0);
        acc1 = fmaf(ss[(q * 2 + 1) * lds + c], w, acc1);
    }
    if (r0 + q * 2 < R) dh[(r0 + q * 2) * ld_dh + k] = acc0;
    if (r0 + q * 2 + 1 < R) dh[(r0 + q * 2 + 1) * ld_dh + k] = acc1;
}

// Per-ray skip MLP (the sky head: mlp.MLP(num_layers=3, skip_connections=[1]), hidden width 64) after emer_ray_pre_fwd has
// produced rb = [W0 x + b0 | W1[:, 64:] x + b1]:   a1 = relu(rb0);  a2 = relu(W1[:, :64] a1 + rb1);  out = act(W2 a2 + b2).
// 32 rows per workgroup, weights in LDS, lane (c = tid & 63, q = tid >> 6) owns column c of rows 2q, 2q+1.
__global__ __launch_bounds__(kRayThreads) void ray_head_fwd_kernel(const float *__restrict__ rb, int64_t ld_rb, int64_t R,
                                                                   const float *__restrict__ w1, int64_t ld_w1, const float *__restrict__ w2,
                                                                   const float *__restrict__ b2, int32_t C, int act, float *__restrict__ a1,
                                                                   float *__restrict__ a2, float *__restrict__ out) {
    __shared__ float w1s[64][65], w2s[16][65], a1s[kRayTile][65], a2s[kRayTile][65];
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * kRayTile;
    const int64_t ra = r0 + q * 2, rbw = ra + 1;
    // this lane's four rb values and the weights: all loads in flight together
    const float p0 = ra < R ? rb[ra * ld_rb + c] : 0.f, p1 = rbw < R ? rb[rbw * ld_rb + c] : 0.f;
    float acc0 = ra < R ? rb[ra * ld_rb + 64 + c] : 0.f, acc1 = rbw < R ? rb[rbw * ld_rb + 64 + c] : 0.f;
    stage_batched(64 * 64, [&](int32_t i) { return w1[(int64_t)(i >> 6) * ld_w1 + (i & 63)]; }, [&](int32_t i, float v) { w1s[i >> 6][i & 63] = v; });
    stage_batched(C * 64, [&](int32_t i) { return w2[i]; }, [&](int32_t i, float v) { w2s[i >> 6][i & 63] = v; });
    const float v0 = fmaxf(p0, 0.f), v1 = fmaxf(p1, 0.f);
    a1s[q * 2][c] = v0; a1s[q * 2 + 1][c] = v1;
    if (ra < R) a1[ra * 64 + c] = v0;
    if (rbw < R) a1[rbw * 64 + c] = v1;
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        const float w = w1s[c][k];
        acc0 = fmaf(w, a1s[q * 2][k], acc0);
        acc1 = fmaf(w, a1s[q * 2 + 1][k], acc1);
    }
    acc0 = fmaxf(acc0, 0.f); acc1 = fmaxf(acc1, 0.f);
    a2s[q * 2][c] = acc0; a2s[q * 2 + 1][c] = acc1;
    if (ra < R) a2[ra * 64 + c] = acc0;
    if (rbw < R) a2[rbw * 64 + c] = acc1;
    __syncthreads();
    if (tid < kRayTile * C) {
        const int rr = tid / C, j = tid - rr * C;
        const int64_t r = r0 + rr;
        if (r < R) {
            float s = b2 ? b2[j] : 0.f;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) s = fmaf(w2s[j][k], a2s[rr][k], s);
            out[r * C + j] = act == EMER_ACT_SIGMOID ? 1.0f / (1.0f + expf(-s)) : s;
        }
    }
}

// dpre2 = dout * act'(out);  dpre1 = (a2 > 0) * (dpre2 W2);  dpre0 = (a1 > 0) * (dpre1 W1[:, :64])
__global__ __launch_bounds__(kRayThreads) void ray_head_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ out,
                                                                   const float *__restrict__ a1, const float *__restrict__ a2, int64_t R,
                                                                   const float *__restrict__ w1, int64_t ld_w1, const float *__restrict__ w2,
                                                                   int32_t C, int act, float *__restrict__ dpre2, float *__restrict__ dpre1,
                                                                   float *__restrict__ dpre0) {
    __shared__ float w1s[64][65], w2s[16][65], d2s[kRayTile][17], d1s[kRayTile][65];
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * kRayTile;
    const int64_t ra = r0 + q * 2, rbw = ra + 1;
    // the activation masks of this lane's two rows, the upstream gradient and the weights: all loads in flight together
    const float m2a = ra < R ? a2[ra * 64 + c] : 0.f, m2b = rbw < R ? a2[rbw * 64 + c] : 0.f;
    const float m1a = ra < R ? a1[ra * 64 + c] : 0.f, m1b = rbw < R ? a1[rbw * 64 + c] : 0.f;
    float d = 0.f, y = 0.f;
    const int rr2 = tid / C, j2 = tid - rr2 * C;
    const bool has2 = tid < kRayTile * C && r0 + rr2 < R;
    if (has2) {
        d = dout[(r0 + rr2) * C + j2];
        y = out[(r0 + rr2) * C + j2];
    }
    stage_batched(64 * 64, [&](int32_t i) { return w1[(int64_t)(i >> 6) * ld_w1 + (i & 63)]; }, [&](int32_t i, float v) { w1s[i >> 6][i & 63] = v; });
    stage_batched(C * 64, [&](int32_t i) { return w2[i]; }, [&](int32_t i, float v) { w2s[i >> 6][i & 63] = v; });
    if (tid < kRayTile * C) {
        if (act == EMER_ACT_SIGMOID) d = d * y * (1.0f - y);
        if (has2) dpre2[(r0 + rr2) * C + j2] = d;
        d2s[rr2][j2] = has2 ? d : 0.f;
    }
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
    for (int j = 0; j < C; ++j) {
        const float w = w2s[j][c];
        s0 = fmaf(d2s[q * 2][j], w, s0);
        s1 = fmaf(d2s[q * 2 + 1][j], w, s1);
    }
    s0 = m2a > 0.f ? s0 : 0.f; s1 = m2b > 0.f ? s1 : 0.f;
    d1s[q * 2][c] = s0; d1s[q * 2 + 1][c] = s1;
    if (ra < R) dpre1[ra * 64 + c] = s0;
    if (rbw < R) dpre1[rbw * 64 + c] = s1;
    __syncthreads();
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 8
    for (int n = 0; n < 64; ++n) {
        const float w = w1s[n][c];
        acc0 = fmaf(d1s[q * 2][n], w, acc0);
        acc1 = fmaf(d1s[q * 2 + 1][n], w, acc1);
    }
    if (ra < R) dpre0[ra * 64 + c] = m1a > 0.f ? acc0 : 0.f;
    if (rbw < R) dpre0[rbw * 64 + c] = m1b > 0.f ? acc1 : 0.f;
}

// Weight gradients of per-ray layers, several layers per launch: job j computes dw_j[n][dst + k] += sum_rows dy_j[row][n] x_j[row][k]
// (x given as one or two column blocks: the virtual concat of a skip connection) and dbias_j[n] += sum_rows dy_j[row][n] (an extra
// all-ones operand column).  Workgroup (chunk of 256 rows, job) of sixteen waves.  What bounds these launches is memory LATENCY
// (~2 us per dependent round trip, one workgroup per CU), so the chunk is fetched in two rounds of 128 rows with every load of a
// round in flight at once (up to 24 per lane, coalesced, branch-free) and parked in LDS; wave (q, t) then accumulates output
// rows 16t..16t+15 over rows 32q..32q+31 of the round with v_mfma_f32_16x16x4_f32 fed from LDS.  The four row quarters are summed
// through LDS in a fixed order and the chunk's sums leave with relaxed float atomics (0.2 M at 8192 rows; 64-row chunks were
// atomic-bound, operands loaded per MFMA step latency-bound).
constexpr int kWgChunk = 256, kWgRound = 128, kWgThreads = 1024, kDyLd = 72, kXLd = 136;
struct RayWgradJobs { emer_ray_wgrad_job j[EMER_RAY_WGRAD_MAX_JOBS]; };
using f32x4r = __attribute__((__vector_size__(4 * sizeof(float)))) float;

__global__ __launch_bounds__(kWgThreads) void ray_wgrad_kernel(const RayWgradJobs jobs, int64_t M) {
    extern __shared__ float lds_f[];
    float *dys = lds_f;                       // [128][kDyLd]
    float *xs = lds_f + kWgRound * kDyLd;     // [128][kXLd]
    const emer_ray_wgrad_job &J = jobs.j[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = wave & 3, q = wave >> 2;
    const int c = lane & 15, kk = lane >> 4;
    const int32_t N = J.n, w0 = J.width[0], w1 = J.n_segs > 1 ? J.width[1] : 0, Kx = w0 + w1;
    const int32_t Kt = Kx + (J.dbias ? 1 : 0);  // operand columns incl. the ones column
    const int32_t KT = (Kt + 15) >> 4;          // k-tiles in use (<= 8)
    const bool active = 16 * t < N;              // this wave's output rows exist
    f32x4r acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4r{0.f, 0.f, 0.f, 0.f};
    for (int round = 0; round < kWgChunk / kWgRound; ++round) {
        const int64_t r0 = (int64_t)blockIdx.x * kWgChunk + round * kWgRound;
        if (r0 >= M) break;
        // uniform base pointers + 32-bit lane offsets (a 64-bit address per load in flight would not fit the register file);
        // rows past the end are clamped to the last row and their dy is zeroed, columns past a block's width are clamped and
        // land in LDS columns whose outputs are never written
        float vd[8], va[8], vb[8];
        const float *__restrict__ bdy = J.dy, *__restrict__ bx0 = J.x[0], *__restrict__ bx1 = J.n_segs > 1 ? J.x[1] : J.x[0];
        const uint32_t ldd = (uint32_t)J.ld_dy, ld0 = (uint32_t)J.ld_x[0], ld1 = (uint32_t)(J.n_segs > 1 ? J.ld_x[1] : J.ld_x[0]);
        const uint32_t last = (uint32_t)(M - 1 - r0);  // last live row of the round (may exceed 127)
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // 128 rows x 64 slots per block
            const uint32_t i = u * kWgThreads + tid, rr = i >> 6, cc = i & 63;
            const uint32_t row = (uint32_t)r0 + (rr < last ? rr : last);
            vd[u] = bdy[row * ldd + (cc < (uint32_t)N ? cc : N - 1)];
            va[u] = bx0[row * ld0 + (cc < (uint32_t)w0 ? cc : w0 - 1)];
            if (w1 > 0) vb[u] = bx1[row * ld1 + (cc < (uint32_t)w1 ? cc : w1 - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = u * kWgThreads + tid, rr = i >> 6, cc = i & 63;
            const bool live = rr <= last;
            dys[rr * kDyLd + cc] = live ? vd[u] : 0.f;
            if (cc < (uint32_t)w0) xs[rr * kXLd + cc] = va[u];
            if (w1 > 0 && cc < (uint32_t)w1) xs[rr * kXLd + w0 + cc] = vb[u];
        }
        if (tid < kWgRound) xs[tid * kXLd + Kx] = 1.0f;  // the ones column (read only when the job has a bias)
        __syncthreads();
        if (active) {
#pragma unroll 2
            for (int s = 0; s < 8; ++s) {  // this wave's 32 rows of the round, 4 per MFMA step
                const int rr = q * 32 + 4 * s + kk;
                const float av = dys[rr * kDyLd + 16 * t + c];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < KT) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xs[rr * kXLd + 16 * j + c], acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // quarters 2, 3 -> LDS; quarters 0, 1 add; quarter 1 -> LDS; quarter 0 adds and owns the chunk's sums
    f32x4r *slot = reinterpret_cast<f32x4r *>(lds_f);  // [8 slots][8 k-tiles][64 lanes]: 64 KiB over the staging area
    if (q >= 2)
#pragma unroll
        for (int j = 0; j < 8; ++j) slot[(((q - 2) * 4 + t) * 8 + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (q < 2)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += slot[((q * 4 + t) * 8 + j) * 64 + lane];
    __syncthreads();
    if (q == 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) slot[(t * 8 + j) * 64 + lane] = acc[j];
    __syncthreads();
    if (q != 0 || !active) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j >= KT) continue;
        const f32x4r o = acc[j] + slot[(t * 8 + j) * 64 + lane];
        const int32_t k = 16 * j + c;  // D layout: row 4 kk + i of the tile, column c
        if (k >= Kt) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int32_t n = 16 * t + 4 * kk + i;
            if (n >= N) continue;
            float *dst;
            if (k < w0) dst = J.dw + (int64_t)n * J.ld_dw + J.dst_col[0] + k;
            else if (k < Kx) dst = J.dw + (int64_t)n * J.ld_dw + J.dst_col[1] + (k - w0);
            else dst = J.dbias + n;
            __hip_atomic_fetch_add(dst, o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static inline uint32_t small_blocks(int64_t n) { return (uint32_t)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096); }

}  // namespace emer

using namespace emer;

extern "C" int emer_ray_inputs_fwd(const float *dirs, int64_t ld_dirs, const int64_t *idx, int64_t idx_stride, const float *emb,
                                   int32_t n_emb, int32_t emb_dim, int32_t max_deg, int64_t n_rays, float *out_rgb, int64_t ld_rgb,
                                   float *out_sky, int64_t ld_sky, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && max_deg >= 0 && max_deg <= 16 && emb_dim >= 0 && ld_dirs >= 3, "ray_inputs_fwd: bad arguments");
    if (n_rays == 0) return EMER_OK;
    const int32_t width = (max_deg == 0 ? 3 : 3 * (1 + 2 * (max_deg + 1))) + emb_dim;
    EMER_REQUIRE(dirs && (out_rgb || out_sky), "ray_inputs_fwd: null pointer");
    EMER_REQUIRE((!out_rgb || ld_rgb >= width) && (!out_sky || ld_sky >= width), "ray_inputs_fwd: output row stride below the row width");
    EMER_REQUIRE(emb_dim == 0 || (emb && n_emb >= 1 && (idx || n_emb == 1)), "ray_inputs_fwd: embedding table / indices missing");
    hipLaunchKernelGGL(ray_inputs_kernel, dim3(small_blocks(n_rays * 2 * width)), dim3(256), 0, as_stream(stream), dirs, ld_dirs, idx, idx_stride, emb,
                       n_emb, emb_dim, max_deg, n_rays, out_rgb, ld_rgb, out_sky, ld_sky);
    return check_launch("ray_inputs_fwd");
}

extern "C" int emer_embed_grad(const float *g_a, int64_t ld_a, const float *g_b, int64_t ld_b, const int64_t *idx, int64_t idx_stride,
                               int64_t n_rays, int32_t n_emb, int32_t emb_dim, float *dw, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && n_emb >= 1 && emb_dim >= 1, "embed_grad: bad arguments");
    EMER_REQUIRE(idx && dw && (g_a || g_b), "embed_grad: null pointer");
    if (n_rays == 0) return EMER_OK;
    hipLaunchKernelGGL(embed_grad_kernel<16>, dim3((uint32_t)n_emb), dim3(kEmbedThreads), 0, as_stream(stream), g_a, ld_a, g_b, ld_b, idx, idx_stride,
                       n_rays, emb_dim, dw);
    return check_launch("embed_grad");
}

extern "C" int emer_ray_pre_fwd(const float *h, int64_t ld_h, int64_t n_rays, int32_t kh, int32_t n_hidden, const float *wa, int64_t ld_wa,
                                const float *ba, const float *wb, int64_t ld_wb, const float *bb, float *rb, int64_t ld_rb,
                                void *stream) {
    EMER_REQUIRE(n_rays >= 0 && kh >= 1 && kh <= 64 && n_hidden >= 1 && n_hidden <= 64, "ray_pre_fwd: kh and n_hidden must be in [1, 64]");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(h && wa && wb && rb && ld_h >= kh && ld_wa >= kh && ld_wb >= kh && ld_rb >= 2 * n_hidden, "ray_pre_fwd: null pointer or short row stride");
    const size_t lds = (size_t)(128 + kRayTile) * (kh | 1) * sizeof(float);
    hipLaunchKernelGGL(ray_pre_fwd_kernel, dim3((uint32_t)ceil_div(n_rays, kRayTile)), dim3(kRayThreads), lds, as_stream(stream), h, ld_h, n_rays, kh,
                       n_hidden, wa, ld_wa, ba, wb, ld_wb, bb, rb, ld_rb);
    return check_launch("ray_pre_fwd");
}

extern "C" int emer_ray_pre_bwd(const float *s0, const float *s1, int64_t ld_s, int64_t n_rays, int32_t kh, int32_t n_hidden,
                                const float *wa, int64_t ld_wa, const float *wb, int64_t ld_wb, float *dh, int64_t ld_dh, void *stream) {
    EMER_REQUIRE(n_rays >= 0 && kh >= 1 && kh <= 64 && n_hidden >= 1 && n_hidden <= 64, "ray_pre_bwd: kh and n_hidden must be in [1, 64]");
    if (n_rays == 0) return EMER_OK;
    EMER_REQUIRE(s0 && s1 && wa && wb && dh && ld_s >= n_hidden && ld_wa >= kh && ld_wb >= kh && ld_dh >= kh, "ray_pre_bwd: null pointer or short row stride");
    const size_t lds = (size_t)(2 * n_hidden * kh + kRayTile * (2 * n_hidden + 1)) * sizeof(float);
    if (int rc = reserve_lds(reinterpret_cast<const void *>(ray_pre_bwd_kernel), lds, "ray_pre_bwd")) return rc;
    hipLaunchKernelGGL(ray_pre_bwd_kernel, dim3((uint32_t)ceil_div(n_rays, kRayTile)), dim3(kRayThreads), lds, as_stream(stream), s0, s1, ld_s, n_rays,
                       kh, n_hidden, wa, ld_wa, wb, ld_wb, dh, ld_dh);
    return check_launch("ray_pre_bwd");
}

extern "C" int emer_ray_head_fwd(const float *rb, int64_t ld_rb, int64_t n_rows, const float *w1, int64_t ld_w1, const float *w2,
                                 const float *b2, int32_t n_out, int act, float *a1, float *a2, float *out, void *stream) {
    EMER_REQUIRE(n_rows >= 0 && n_out >= 1 && n_out <= 16, "ray_head_fwd: n_out must be in [1, 16]");
    EMER_REQUIRE(act == EMER_ACT_NONE || act == EMER_ACT_SIGMOID, "ray_head_fwd: final activation must be none or sigmoid");
    if (n_rows == 0) return EMER_OK;
    EMER_REQUIRE(rb && w1 && w2 && a1 && a2 && out && ld_rb >= 128 && ld_w1 >= 64, "ray_head_fwd: null pointer or short row stride");
    hipLaunchKernelGGL(ray_head_fwd_kernel, dim3((uint32_t)ceil_div(n_rows, kRayTile)), dim3(kRayThreads), 0, as_stream(stream), rb, ld_rb, n_rows, w1, ld_w1,
                       w2, b2, n_out, act, a1, a2, out);
    return check_launch("ray_head_fwd");
}

extern "C" int emer_ray_head_bwd(const float *dout, const float *out, const float *a1, const float *a2, int64_t n_rows, const float *w1,
                                 int64_t ld_w1, const float *w2, int32_t n_out, int act, float *dpre2, float *dpre1, float *dpre0,
                                 void *stream) {
    EMER_REQUIRE(n_rows >= 0 && n_out >= 1 && n_out <= 16, "ray_head_bwd: n_out must be in [1, 16]");
    EMER_REQUIRE(act == EMER_ACT_NONE || act == EMER_ACT_SIGMOID, "ray_head_bwd: final activation must be none or sigmoid");
    if (n_rows == 0) return EMER_OK;
    EMER_REQUIRE(dout && out && a1 && a2 && w1 && w2 && dpre2 && dpre1 && dpre0 && ld_w1 >= 64, "ray_head_bwd: null pointer or short row stride");
    hipLaunchKernelGGL(ray_head_bwd_kernel, dim3((uint32_t)ceil_div(n_rows, kRayTile)), dim3(kRayThreads), 0, as_stream(stream), dout, out, a1, a2, n_rows, w1,
                       ld_w1, w2, n_out, act, dpre2, dpre1, dpre0);
    return check_launch("ray_head_bwd");
}

extern "C" int emer_ray_wgrad(const emer_ray_wgrad_job *jobs, int32_t n_jobs, int64_t m, void *stream) {
    EMER_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= EMER_RAY_WGRAD_MAX_JOBS && m >= 0, "ray_wgrad: 1..%d jobs", EMER_RAY_WGRAD_MAX_JOBS);
    if (m == 0) return EMER_OK;
    RayWgradJobs J;
    for (int32_t i = 0; i < n_jobs; ++i) {
        const emer_ray_wgrad_job &j = jobs[i];
        EMER_REQUIRE(j.dy && j.dw && j.n >= 1 && j.n <= 64 && j.ld_dy >= j.n && (j.n_segs == 1 || j.n_segs == 2), "ray_wgrad: job %d: n in [1, 64], one or two operand blocks", i);
        int32_t k = 0;
        for (int32_t s = 0; s < j.n_segs; ++s) {
            EMER_REQUIRE(j.x[s] && j.width[s] >= 1 && j.ld_x[s] >= j.width[s] && j.dst_col[s] >= 0 && j.dst_col[s] + j.width[s] <= j.ld_dw,
                         "ray_wgrad: job %d: operand block %d", i, s);
            k += j.width[s];
        }
        EMER_REQUIRE(k + 1 <= 128 && j.width[0] <= 64 && (j.n_segs == 1 || j.width[1] <= 64), "ray_wgrad: job %d: operand blocks of at most 64 columns, 127 in total", i);
        EMER_REQUIRE(m * (j.ld_dy > j.ld_x[0] ? j.ld_dy : j.ld_x[0]) < ((int64_t)1 << 31) && (j.n_segs == 1 || m * j.ld_x[1] < ((int64_t)1 << 31)),
                     "ray_wgrad: job %d: rows * row stride must stay below 2^31", i);
        J.j[i] = j;
    }
    const size_t lds = (size_t)kWgRound * (kDyLd + kXLd) * sizeof(float);  // 104 KiB (the 64 KiB of reduction slots alias it)
    if (int rc = reserve_lds(reinterpret_cast<const void *>(ray_wgrad_kernel), lds, "ray_wgrad")) return rc;
    hipLaunchKernelGGL(ray_wgrad_kernel, dim3((uint32_t)ceil_div(m, kWgChunk), (uint32_t)n_jobs), dim3(kWgThreads), lds, as_stream(stream), J, m);
    return check_launch("ray_wgrad");
}
