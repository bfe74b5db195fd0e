/*
 * emer_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic on EmerNeRF's volumetric-rendering hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (emernerf_amd/) never does and fails loudly
 * when its HIP extension is missing.
 *
 * PARITY STATUS: "parity unpinned" for the two un-vendored native libraries
 * (tiny-cuda-nn @ unpinned master, nerfacc @ 8340e19): their sources are not in
 * /root/reference and the reference ships no tests or golden vectors, so the
 * functions below restate their *published* algorithms and
 * are anchored on the reference's own call sites, cited per function.  The Python
 * layer above them (RadianceField / render_rays / PropNetEstimator) IS pinned:
 * tests/golden/make_golden.py runs the reference's Python verbatim on top of this
 * oracle and commits the vectors.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: the sampler must be bit-exact against the HIP kernel,
 * which spells every multiply/add explicitly (no FMA contraction).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_LEVELS 32

typedef struct {
    uint32_t n_dims, n_levels, n_features, log2_hashmap_size, base_resolution;
    float per_level_scale;
    float scale[ORC_MAX_LEVELS];
    uint32_t res[ORC_MAX_LEVELS];
    uint32_t size[ORC_MAX_LEVELS];   /* entries in the level            */
    uint32_t offset[ORC_MAX_LEVELS]; /* first entry of the level        */
    uint32_t hashed[ORC_MAX_LEVELS]; /* 1 -> coherent-prime hash        */
    uint32_t n_entries;              /* sum(size); n_params = n_entries*F */
} orc_grid;

/* ---- tiny-cuda-nn HashGrid level table ------------------------------------
 * Call site: radiance_fields/encodings.py:133-146 (encoding_config) ->
 * third_party/tcnn_modules.py:420-423 (_C.create_encoding).  Algorithm: SURVEY
 * Appendix A.1 [UPSTREAM-RECALL]: scale_l = exp2(l*log2(b))*base - 1,
 * res_l = ceil(scale_l)+1, size_l = min(round_up(res^D, 8), 2^T); a level is
 * hashed iff the dense stride product exceeds size_l. */
int orc_grid_init(orc_grid *g, uint32_t D, uint32_t L, uint32_t F, uint32_t log2_T,
                  uint32_t base_res, float per_level_scale) {
    if (D < 2 || D > 4 || L < 1 || L > ORC_MAX_LEVELS || F < 1 || F > 8) return -1;
    memset(g, 0, sizeof(*g));
    g->n_dims = D; g->n_levels = L; g->n_features = F;
    g->log2_hashmap_size = log2_T; g->base_resolution = base_res;
    g->per_level_scale = per_level_scale;
    const float log2_pls = log2f(per_level_scale);
    uint32_t offset = 0;
    for (uint32_t l = 0; l < L; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)base_res - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        uint32_t dense;
        if (powf((float)res, (float)D) > (float)max_params) {
            dense = max_params;
        } else {
            uint64_t p = 1;
            for (uint32_t d = 0; d < D; ++d) p *= res;
            dense = (uint32_t)p;
        }
        dense = (dense + 7u) / 8u * 8u;
        uint32_t size = dense;
        const uint32_t cap = 1u << log2_T;
        if (size > cap) size = cap;
        /* hashed iff the stride product of grid_index() overtakes size */
        uint64_t stride = 1;
        for (uint32_t d = 0; d < D && stride <= size; ++d) stride *= res;
        g->scale[l] = scale; g->res[l] = res; g->size[l] = size; g->offset[l] = offset;
        g->hashed[l] = (size < stride) ? 1u : 0u;
        offset += size;
    }
    g->n_entries = offset;
    return 0;
}

static inline uint32_t orc_grid_index(const orc_grid *g, uint32_t l, const uint32_t *c) {
    const uint32_t D = g->n_dims, size = g->size[l], res = g->res[l];
    uint32_t idx;
    if (g->hashed[l]) {
        static const uint32_t primes[4] = {1u, 2654435761u, 805459861u, 3674653429u};
        idx = 0;
        for (uint32_t d = 0; d < D; ++d) idx ^= c[d] * primes[d];
    } else {
        uint32_t stride = 1; idx = 0;
        for (uint32_t d = 0; d < D && stride <= size; ++d) { idx += c[d] * stride; stride *= res; }
    }
    return idx % size;
}

/* ---- tcnn grid forward (kernel_grid) ---------------------------------------
 * Call site: radiance_fields/encodings.py:159-160 -> third_party/tcnn_modules.py:122
 * (native.fwd).  x [N,D] in [0,1], params flat level-major/entry-major/F-contig,
 * out [N, L*F] row-major, fp32 accumulate.  SURVEY A.1. */
void orc_hashgrid_fwd(const orc_grid *g, const float *x, const float *params, float *out,
                      int64_t N) {
    const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (uint32_t l = 0; l < L; ++l) {
            float w[4]; uint32_t gi[4];
            for (uint32_t d = 0; d < D; ++d) {
                float pos = fmaf(g->scale[l], x[n * D + d], 0.5f);
                float fl = floorf(pos);
                gi[d] = (uint32_t)(int32_t)fl;
                w[d] = pos - fl;
            }
            float acc[8] = {0};
            for (uint32_t m = 0; m < (1u << D); ++m) {
                float wt = 1.0f; uint32_t c[4];
                for (uint32_t d = 0; d < D; ++d) {
                    if (m & (1u << d)) { wt *= w[d]; c[d] = gi[d] + 1u; }
                    else { wt *= 1.0f - w[d]; c[d] = gi[d]; }
                }
                const uint32_t idx = orc_grid_index(g, l, c);
                const float *e = params + ((size_t)g->offset[l] + idx) * F;
                for (uint32_t f = 0; f < F; ++f) acc[f] += wt * e[f];
            }
            for (uint32_t f = 0; f < F; ++f) out[n * (int64_t)(L * F) + l * F + f] = acc[f];
        }
    }
}

/* ---- tcnn grid backward w.r.t. params (kernel_grid_backward) ----------------
 * Call site: third_party/tcnn_modules.py:154-174 (native.bwd).  Upstream scatters
 * with float atomics in a nondeterministic order; the oracle accumulates in double
 * in sample order and rounds once -- the HIP kernel is compared to it by tolerance. */
void orc_hashgrid_bwd_params(const orc_grid *g, const float *x, const float *dout,
                             float *grad, int64_t N) {
    const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features;
    const size_t n_params = (size_t)g->n_entries * F;
    double *acc = (double *)calloc(n_params, sizeof(double));
#pragma omp parallel for schedule(static)
    for (uint32_t l = 0; l < L; ++l) { /* levels own disjoint slices: race-free */
        for (int64_t n = 0; n < N; ++n) {
            float w[4]; uint32_t gi[4];
            for (uint32_t d = 0; d < D; ++d) {
                float pos = fmaf(g->scale[l], x[n * D + d], 0.5f);
                float fl = floorf(pos);
                gi[d] = (uint32_t)(int32_t)fl;
                w[d] = pos - fl;
            }
            const float *go = dout + n * (int64_t)(L * F) + l * F;
            for (uint32_t m = 0; m < (1u << D); ++m) {
                float wt = 1.0f; uint32_t c[4];
                for (uint32_t d = 0; d < D; ++d) {
                    if (m & (1u << d)) { wt *= w[d]; c[d] = gi[d] + 1u; }
                    else { wt *= 1.0f - w[d]; c[d] = gi[d]; }
                }
                const uint32_t idx = orc_grid_index(g, l, c);
                double *e = acc + ((size_t)g->offset[l] + idx) * F;
                for (uint32_t f = 0; f < F; ++f) e[f] += (double)(wt * go[f]);
            }
        }
    }
    for (size_t i = 0; i < n_params; ++i) grad[i] = (float)acc[i];
    free(acc);
}

/* ---- tcnn grid backward w.r.t. input (kernel_grid_backward_input) -----------
 * Needed by the flow configs: warped positions carry grad into the xyzt encoders
 * (radiance_fields/radiance_field.py:572-608).  dX_d = sum_l scale_l sum_f dOut *
 * sum_{corners of the other dims} prod w * (table[d=1] - table[d=0]).  SURVEY A.1. */
void orc_hashgrid_bwd_input(const orc_grid *g, const float *x, const float *params,
                            const float *dout, float *dx, int64_t N) {
    const uint32_t D = g->n_dims, L = g->n_levels, F = g->n_features;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float gx[4] = {0, 0, 0, 0};
        for (uint32_t l = 0; l < L; ++l) {
            float w[4]; uint32_t gi[4];
            for (uint32_t d = 0; d < D; ++d) {
                float pos = fmaf(g->scale[l], x[n * D + d], 0.5f);
                float fl = floorf(pos);
                gi[d] = (uint32_t)(int32_t)fl;
                w[d] = pos - fl;
            }
            const float *go = dout + n * (int64_t)(L * F) + l * F;
            for (uint32_t gd = 0; gd < D; ++gd) {
                float acc = 0.0f;
                for (uint32_t m = 0; m < (1u << (D - 1)); ++m) {
                    float wt = g->scale[l]; uint32_t c[4]; uint32_t bit = 0;
                    for (uint32_t d = 0; d < D; ++d) {
                        if (d == gd) { c[d] = gi[d]; continue; }
                        if (m & (1u << bit)) { wt *= w[d]; c[d] = gi[d] + 1u; }
                        else { wt *= 1.0f - w[d]; c[d] = gi[d]; }
                        ++bit;
                    }
                    const float *e0 = params + ((size_t)g->offset[l] + orc_grid_index(g, l, c)) * F;
                    c[gd] = gi[gd] + 1u;
                    const float *e1 = params + ((size_t)g->offset[l] + orc_grid_index(g, l, c)) * F;
                    for (uint32_t f = 0; f < F; ++f) acc += wt * go[f] * (e1[f] - e0[f]);
                }
                gx[gd] += acc;
            }
        }
        for (uint32_t d = 0; d < D; ++d) dx[n * D + d] = gx[d];
    }
}

/* ---- scene contraction ------------------------------------------------------
 * radiance_fields/nerf_utils.py:13-28 (contract, ord=inf) +
 * radiance_fields/radiance_field.py:278-300 (selector zeroing); the same sequence is
 * inlined in DensityField.forward (radiance_field.py:828-835).  Operation order
 * follows the torch expression exactly (true divisions, no reciprocals). */
void orc_contract(const float *pos, const float *aabb, int unbounded, float *out, int64_t N) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float v[3];
        for (int d = 0; d < 3; ++d) v[d] = (pos[n * 3 + d] - aabb[d]) / (aabb[3 + d] - aabb[d]);
        if (unbounded) {
            float mag = 0.0f;
            for (int d = 0; d < 3; ++d) { v[d] = v[d] * 2.0f - 1.0f; mag = fmaxf(mag, fabsf(v[d])); }
            if (!(mag < 1.0f)) {
                const float s = 2.0f - 1.0f / mag;
                for (int d = 0; d < 3; ++d) v[d] = s * (v[d] / mag);
            }
            for (int d = 0; d < 3; ++d) v[d] = v[d] / 4.0f + 0.5f;
        }
        int inside = 1;
        for (int d = 0; d < 3; ++d) inside &= (v[d] > 0.0f) & (v[d] < 1.0f);
        for (int d = 0; d < 3; ++d) out[n * 3 + d] = inside ? v[d] : v[d] * 0.0f;
    }
}

/* ---- nerfacc.pdf.importance_sampling (batched) ------------------------------
 * Call sites: third_party/nerfacc_prop_net.py:153,172.  FROZEN SPEC (SURVEY A.2,
 * "choose & freeze"; upstream source unavailable):
 *   step = (cdf_last - cdf_first) / (n+1)          (float division)
 *   u_k  = cdf_first + (k + beta) * step, k = 0..n (separate mul, add)
 *   beta = 0.5 if jitter == NULL else jitter[ray]  (one U(0,1) per ray, an INPUT)
 *   p    = #{j : cdf_j <= u_k}                      (upper-bound search, 0..m)
 *   p0   = clamp(p - 1, 0, m-1), p1 = clamp(p, 0, m-1)   (the two edges clamped SEPARATELY, as nerfacc's
 *          pdf.cu does [UPSTREAM-RECALL]: u_k >= cdf_last brackets (m-1, m-1) and returns v[m-1])
 *   d    = cdf[p1] - cdf[p0]
 *   out  = d < 1e-10 ? (v[p0]+v[p1])*0.5 : (u_k - cdf[p0]) * ((v[p1]-v[p0]) / d) + v[p0]
 * Outputs are sorted by construction.  Must stay bit-exact with sampler.hip. */
void orc_importance_sample(const float *vals, const float *cdfs, int64_t R, int32_t m,
                           int32_t n, const float *jitter, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        const float *v = vals + r * m, *c = cdfs + r * m;
        const float c0 = c[0], cl = c[m - 1];
        const float step = (cl - c0) / (float)(n + 1);
        const float beta = jitter ? jitter[r] : 0.5f;
        for (int32_t k = 0; k <= n; ++k) {
            const float u = c0 + ((float)k + beta) * step;
            int32_t lo = 0, hi = m; /* first j with c[j] > u */
            while (lo < hi) { int32_t mid = (lo + hi) >> 1; if (c[mid] <= u) lo = mid + 1; else hi = mid; }
            const int32_t p0 = lo > 0 ? lo - 1 : 0, p1 = lo < m ? lo : m - 1;
            const float d = c[p1] - c[p0];
            float t;
            if (d < 1e-10f) t = (v[p0] + v[p1]) * 0.5f;
            else t = (u - c[p0]) * ((v[p1] - v[p0]) / d) + v[p0];
            out[r * (int64_t)(n + 1) + k] = t;
        }
    }
}

/* ---- s -> t transform -------------------------------------------------------
 * third_party/nerfacc_prop_net.py:299-339 (_transform_stot / TRANSFROM_DICT).
 * type 0 = "uniform", 1 = "uniform_lindisp" (linear to 200 m, disparity beyond),
 * 2 = "lindisp", 3 = "sqrt", 4 = "log", 5 = "uniform_lindisp_0".  Operation order follows the torch lambdas. */
static inline float orc_fwd_map(int type, float t) {
    if (type == 1) return t < 200.0f ? t / 400.0f : 1.0f - 1.0f / (2.0f * t / 200.0f);
    if (type == 2) return 1.0f / t;
    if (type == 3) return sqrtf(t);
    if (type == 4) return logf(t);
    if (type == 5) return t < 1.0f ? t / 2.0f : 1.0f - 1.0f / (2.0f * t);
    return t;
}
static inline float orc_inv_map(int type, float s) {
    if (type == 1) return s < 0.5f ? s * 400.0f : (1.0f / (2.0f - 2.0f * s)) * 200.0f; /* torch: scalar / tensor == tensor.reciprocal() * scalar */
    if (type == 2) return 1.0f / s;
    if (type == 3) return s * s;
    if (type == 4) return expf(s);
    if (type == 5) return s < 0.5f ? 2.0f * s : 1.0f / (2.0f - 2.0f * s);
    return s;
}
void orc_stot(const float *s, int64_t n, float t_min, float t_max, int type, float *t) {
    const float s_min = orc_fwd_map(type, t_min), s_max = orc_fwd_map(type, t_max);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) t[i] = orc_inv_map(type, s[i] * s_max + (1.0f - s[i]) * s_min);
}

/* ---- nerfacc dense volrend --------------------------------------------------
 * render_transmittance_from_density / render_weight_from_density on (R,S) tensors
 * (call sites radiance_fields/render_utils.py:35,73; nerfacc_prop_net.py:165).
 * sigma_dt = sigma*(t_end-t_start); alpha = 1-exp(-sigma_dt);
 * trans = exp(-exclusive_cumsum(sigma_dt)); weights = trans*alpha. */
void orc_render_weights(const float *t_starts, const float *t_ends, const float *sigma,
                        int64_t R, int32_t S, float *weights, float *trans, float *alphas) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        float cum = 0.0f;
        for (int32_t s = 0; s < S; ++s) {
            const int64_t i = r * S + s;
            const float sdt = sigma[i] * (t_ends[i] - t_starts[i]);
            const float T = expf(-cum), a = 1.0f - expf(-sdt);
            if (trans) trans[i] = T;
            if (alphas) alphas[i] = a;
            if (weights) weights[i] = T * a;
            cum += sdt;
        }
    }
}

/* accumulate_along_rays(weights, values): out[r,c] = sum_s w[r,s]*v[r,s,c]
 * (radiance_fields/render_utils.py:103-105,159-282). values==NULL -> sum of w. */
void orc_accumulate(const float *w, const float *values, int64_t R, int32_t S, int32_t C,
                    float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        if (!values) { float a = 0; for (int32_t s = 0; s < S; ++s) a += w[r * S + s]; out[r] = a; continue; }
        for (int32_t c = 0; c < C; ++c) {
            float a = 0;
            for (int32_t s = 0; s < S; ++s) a += w[r * S + s] * values[(r * S + s) * (int64_t)C + c];
            out[r * (int64_t)C + c] = a;
        }
    }
}

uint32_t orc_sizeof_grid(void) { return (uint32_t)sizeof(orc_grid); }
