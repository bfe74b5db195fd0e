// Training-ray generation for gfx950: pixel selection (uniform and error-buffer importance sampling) and the
// pixel -> ray gather, all on device-resident dataset tensors.  SURVEY.md section 8f row N2: the step immediately before
// the hot path (datasets/base/pixel_source.py:39-76 get_rays, :564-731 sample_important_rays / sample_uniform_rays /
// get_train_rays).
//
//   * emer_gen_rays: ONE kernel does everything get_train_rays does after the (img, y, x) selection: pinhole ray through
//     pixel (x + 0.5, y + 0.5), rotation by the camera-to-world matrix, normalisation, and the gathers of colour, sky
//     mask, timestamp and camera id.  The reference issues ~25 indexing / elementwise launches for this.
//   * emer_sample_uniform: torch.randint x3 + an index gather -> one launch on a counter-based generator
//     (splitmix64 of (seed, counter): every sample is a pure function of its index, so a captured hipGraph replays
//     different rays by bumping the seed word in device memory).
//   * emer_sample_importance: torch.multinomial(error_map, n, replacement=False) over several million weights.  Sampling
//     WITHOUT replacement with probabilities proportional to w is the Efraimidis-Spirakis race: key_i = -log(u_i) / w_i,
//     take the n smallest keys.  The n-th smallest key is found by a 3-pass radix select (11 + 11 + 10 bits, LDS
//     histograms, state in device memory: no host round trip), then one compaction pass emits the winners.  Keys are
//     recomputed from the generator in every pass instead of being stored (the weights are read 4 times = 4 x 31 MB for
//     200 images at 160 x 240: ~25 us at HBM speed).
#include "common.h"

namespace emer {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t rnd32(uint64_t seed, uint64_t ctr) { return (uint32_t)(mix64(seed + ctr * 0x9E3779B97F4A7C15ull) >> 32); }
__device__ __forceinline__ uint32_t rnd_below(uint64_t seed, uint64_t ctr, uint32_t n) { return (uint32_t)(((uint64_t)rnd32(seed, ctr) * n) >> 32); }

// ------------------------------------------------------------------------------------------------- ray gather
struct GenRaysArgs {
    const int64_t *img_idx, *y, *x;          // [n]
    const float *c2w;                        // [n_imgs][4][4]
    const float *intrinsics;                 // [n_imgs][3][3]
    const float *images;                     // [n_imgs][H][W][3] or null
    const float *sky_masks;                  // [n_imgs][H][W] or null
    const float *timestamps;                 // [n_imgs] or null
    const int64_t *cam_ids;                  // [n_imgs] or null
    int64_t n; int32_t H, W;
    float *origins, *viewdirs, *direction_norms, *pixel_coords, *pixels, *sky, *ray_t;
    int64_t *ray_cam;
};

__global__ __launch_bounds__(256) void gen_rays_kernel(const GenRaysArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int64_t im = a.img_idx[i], yy = a.y[i], xx = a.x[i];
    const float *K = a.intrinsics + im * 9, *M = a.c2w + im * 16;
    // pixel_source.py:57-66: camera_dirs = ((x - cx + 0.5) / fx, (y - cy + 0.5) / fy, 1)
    const float cd[3] = {((float)xx - K[2] + 0.5f) / K[0], ((float)yy - K[5] + 0.5f) / K[4], 1.0f};
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = (cd[0] * M[r * 4 + 0] + cd[1] * M[r * 4 + 1]) + cd[2] * M[r * 4 + 2];  // (camera_dirs * R).sum(-1)
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        a.origins[i * 3 + r] = M[r * 4 + 3];
        a.viewdirs[i * 3 + r] = d[r] / (nrm + 1e-8f);
    }
    a.direction_norms[i] = nrm;
    if (a.pixel_coords) { a.pixel_coords[i * 2 + 0] = (float)yy / (float)a.H; a.pixel_coords[i * 2 + 1] = (float)xx / (float)a.W; }
    const int64_t pix = (im * a.H + yy) * a.W + xx;
    if (a.images) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.pixels[i * 3 + c] = a.images[pix * 3 + c];
    }
    if (a.sky_masks) a.sky[i] = a.sky_masks[pix];
    if (a.timestamps) a.ray_t[i] = a.timestamps[im];
    if (a.cam_ids) a.ray_cam[i] = a.cam_ids[im];
}

// ---------------------------------------------------------------------------------------------- uniform pixels
__global__ __launch_bounds__(256) void sample_uniform_kernel(const uint64_t *__restrict__ seed_word, uint64_t salt, int64_t n,
                                                             const int64_t *__restrict__ cand, int32_t n_cand, int32_t H, int32_t W,
                                                             int64_t *__restrict__ img_idx, int64_t *__restrict__ y, int64_t *__restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t seed = seed_word[0] ^ salt;
    const uint32_t c = rnd_below(seed, 3ull * (uint64_t)i + 0, (uint32_t)n_cand);
    img_idx[i] = cand ? cand[c] : (int64_t)c;
    x[i] = (int64_t)rnd_below(seed, 3ull * (uint64_t)i + 1, (uint32_t)W);
    y[i] = (int64_t)rnd_below(seed, 3ull * (uint64_t)i + 2, (uint32_t)H);
}

// ------------------------------------------------------------------------------------------------ lidar rays
// datasets/base/lidar_source.py:223-308: sample_uniform_rays draws torch.randint(0, len(cached), n) and get_train_rays gathers origins,
// directions, ranges and normalised timestamps of the cached scans with it (five launches); one launch here.  idx_in != null: gather only
// (the draw was made elsewhere, e.g. a recording).
__global__ __launch_bounds__(256) void lidar_sample_kernel(const uint64_t *__restrict__ seed_word, uint64_t salt, int64_t n, int64_t n_points,
                                                           const int64_t *__restrict__ idx_in, const float *__restrict__ origins,
                                                           const float *__restrict__ dirs, const float *__restrict__ ranges,
                                                           const float *__restrict__ ts, int64_t *__restrict__ idx_out, float *__restrict__ o,
                                                           float *__restrict__ d, float *__restrict__ r, float *__restrict__ t) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t p;
    if (idx_in) {
        p = idx_in[i];
    } else {   // 64-bit product: scans of a whole log may hold more than 2^32 / 2 points
        const uint64_t seed = seed_word[0] ^ salt;
        const uint64_t u = mix64(seed + (uint64_t)i * 0x9E3779B97F4A7C15ull);
        p = (int64_t)__umul64hi(u, (uint64_t)n_points);
    }
    if (idx_out) idx_out[i] = p;
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[i * 3 + c] = origins[p * 3 + c]; d[i * 3 + c] = dirs[p * 3 + c]; }
    r[i] = ranges[p];
    if (ts) t[i] = ts[p];
}

// ------------------------------------------------------------------------------- importance sampling (race)
// race key of element i as an order-preserving u32 (positive floats compare like their bit patterns); w <= 0 -> +inf
__device__ __forceinline__ uint32_t race_key(uint64_t seed, int64_t i, float w) {
    if (!(w > 0.0f)) return 0x7F800000u;
    const float u = ((float)(rnd32(seed, (uint64_t)i) >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float key = -logf(u) / w;
    return __float_as_uint(key) & 0x7FFFFFFFu;  // u = 1 gives -log(1) / w = -0.0: the sign bit must not reach the radix digits
}

// state (device): [0] prefix value of the bits fixed so far, [1] winners still to be found inside the prefix bucket,
// [2] compaction cursor for keys below the threshold, [3] cursor for keys equal to it
constexpr int kSelBins = 2048;

__global__ __launch_bounds__(256) void race_hist_kernel(const float *__restrict__ w, int64_t n, const uint64_t *__restrict__ seed_word,
                                                        uint64_t salt, const uint32_t *__restrict__ state, uint32_t prefix_mask,
                                                        int shift, uint32_t bin_mask, uint32_t *__restrict__ hist) {
    __shared__ uint32_t lh[kSelBins];
    for (int b = threadIdx.x; b < kSelBins; b += 256) lh[b] = 0;
    __syncthreads();
    const uint64_t seed = seed_word[0] ^ salt;
    const uint32_t prefix = state[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t k = race_key(seed, i, w[i]);
        if ((k & prefix_mask) == prefix) atomicAdd(lh + ((k >> shift) & bin_mask), 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kSelBins; b += 256)
        if (lh[b]) atomicAdd(hist + b, lh[b]);
}

// one workgroup: find the bin holding the state[1]-th smallest key of the bucket, extend the prefix, clear the histogram
__global__ __launch_bounds__(1024) void race_select_kernel(uint32_t *__restrict__ hist, int n_bins, int shift, uint32_t *__restrict__ state) {
    __shared__ uint32_t cum[kSelBins];
    for (int b = threadIdx.x; b < kSelBins; b += 1024) cum[b] = b < n_bins ? hist[b] : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t need = state[1], acc = 0;
        int b = 0;
        for (; b < n_bins - 1; ++b) {
            if (acc + cum[b] >= need) break;
            acc += cum[b];
        }
        state[0] |= (uint32_t)b << shift;
        state[1] = need - acc;  // winners to take from bin b (>= 1)
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n_bins; b += 1024) hist[b] = 0u;
}

__global__ __launch_bounds__(256) void race_emit_kernel(const float *__restrict__ w, int64_t n, const uint64_t *__restrict__ seed_word,
                                                        uint64_t salt, uint32_t *__restrict__ state, uint32_t k_total,
                                                        int64_t *__restrict__ out) {
    const uint64_t seed = seed_word[0] ^ salt;
    const uint32_t thr = state[0], n_eq = state[1], n_lt = k_total - n_eq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t k = race_key(seed, i, w[i]);
        if (k < thr) {
            const uint32_t p = atomicAdd(state + 2, 1u);
            if (p < n_lt) out[p] = i;  // (always true when the histograms are consistent; never write out of bounds)
        } else if (k == thr) {
            const uint32_t p = atomicAdd(state + 3, 1u);
            if (p < n_eq) out[n_lt + p] = i;
        }
    }
}

__global__ void race_init_kernel(uint32_t *__restrict__ state, uint32_t k, uint32_t *__restrict__ hist) {
    if (threadIdx.x == 0) { state[0] = 0; state[1] = k; state[2] = 0; state[3] = 0; }
    for (int b = threadIdx.x; b < kSelBins; b += blockDim.x) hist[b] = 0u;
}

// flat index into [n_cand][Hb][Wb] -> (image, y, x) at full resolution with a random sub-cell offset (:600-620)
__global__ __launch_bounds__(256) void buffer_to_pixels_kernel(const int64_t *__restrict__ flat, int64_t n, int32_t Hb, int32_t Wb,
                                                               int32_t downscale, const int64_t *__restrict__ cand, int32_t H,
                                                               int32_t W, const uint64_t *__restrict__ seed_word, uint64_t salt,
                                                               int64_t *__restrict__ img_idx, int64_t *__restrict__ y, int64_t *__restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t seed = seed_word[0] ^ salt;
    const int64_t f = flat[i], plane = (int64_t)Hb * Wb;
    const int64_t c = f / plane, rem = f - c * plane;
    int64_t yy = (rem / Wb) * downscale + (int64_t)rnd_below(seed, 2ull * (uint64_t)i, (uint32_t)downscale);
    int64_t xx = (rem % Wb) * downscale + (int64_t)rnd_below(seed, 2ull * (uint64_t)i + 1, (uint32_t)downscale);
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    xx = xx < 0 ? 0 : (xx > W - 1 ? W - 1 : xx);
    img_idx[i] = cand ? cand[c] : c;
    y[i] = yy; x[i] = xx;
}

}  // namespace emer

using namespace emer;

extern "C" int emer_gen_rays(const int64_t *img_idx, const int64_t *y, const int64_t *x, const float *cam_to_worlds, const float *intrinsics,
                             const float *images, const float *sky_masks, const float *timestamps, const int64_t *cam_ids, int64_t n,
                             int32_t height, int32_t width, float *origins, float *viewdirs, float *direction_norms, float *pixel_coords,
                             float *pixels, float *sky, float *ray_timestamps, int64_t *ray_cam_ids, void *stream) {
    EMER_REQUIRE(n >= 0 && height >= 1 && width >= 1, "gen_rays: bad sizes");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(img_idx && y && x && cam_to_worlds && intrinsics && origins && viewdirs && direction_norms, "gen_rays: null pointer");
    EMER_REQUIRE((!images || pixels) && (!sky_masks || sky) && (!timestamps || ray_timestamps) && (!cam_ids || ray_cam_ids),
                 "gen_rays: a dataset tensor was given without its output buffer");
    GenRaysArgs a{img_idx, y, x, cam_to_worlds, intrinsics, images, sky_masks, timestamps, cam_ids, n, height, width,
                  origins, viewdirs, direction_norms, pixel_coords, pixels, sky, ray_timestamps, ray_cam_ids};
    hipLaunchKernelGGL(gen_rays_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), a);
    return check_launch("gen_rays");
}

extern "C" int emer_sample_uniform(const uint64_t *seed_word, uint64_t salt, int64_t n, const int64_t *candidates, int32_t n_candidates,
                                   int32_t height, int32_t width, int64_t *img_idx, int64_t *y, int64_t *x, void *stream) {
    EMER_REQUIRE(n >= 0 && n_candidates >= 1 && height >= 1 && width >= 1, "sample_uniform: bad sizes");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(seed_word && img_idx && y && x, "sample_uniform: null pointer");
    hipLaunchKernelGGL(sample_uniform_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), seed_word, salt, n,
                       candidates, n_candidates, height, width, img_idx, y, x);
    return check_launch("sample_uniform");
}

extern "C" int emer_lidar_sample_rays(const uint64_t *seed_word, uint64_t salt, int64_t n, int64_t n_points, const int64_t *idx_in,
                                      const float *origins, const float *directions, const float *ranges, const float *timestamps,
                                      int64_t *idx_out, float *out_origins, float *out_directions, float *out_ranges, float *out_timestamps,
                                      void *stream) {
    EMER_REQUIRE(n >= 0 && n_points >= 1, "lidar_sample_rays: need n >= 0 and at least one cached point");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE((seed_word || idx_in) && origins && directions && ranges && out_origins && out_directions && out_ranges,
                 "lidar_sample_rays: null pointer");
    EMER_REQUIRE((timestamps == nullptr) == (out_timestamps == nullptr), "lidar_sample_rays: timestamps and their output go together");
    hipLaunchKernelGGL(lidar_sample_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), seed_word, salt, n, n_points, idx_in,
                       origins, directions, ranges, timestamps, idx_out, out_origins, out_directions, out_ranges, out_timestamps);
    return check_launch("lidar_sample_rays");
}

// workspace: 4 + 2048 u32 words
extern "C" int emer_sample_importance(const float *weights, int64_t n_weights, const uint64_t *seed_word, uint64_t salt, int64_t k,
                                      uint32_t *workspace, int64_t *flat_out, void *stream) {
    EMER_REQUIRE(n_weights >= 1 && k >= 0 && k <= n_weights && k < (1ll << 31), "sample_importance: need 0 <= k <= n_weights");
    if (k == 0) return EMER_OK;
    EMER_REQUIRE(weights && seed_word && workspace && flat_out, "sample_importance: null pointer");
    hipStream_t st = as_stream(stream);
    uint32_t *state = workspace, *hist = workspace + 4;
    hipLaunchKernelGGL(race_init_kernel, dim3(1), dim3(1024), 0, st, state, (uint32_t)k, hist);
    int64_t blocks = ceil_div(n_weights, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    // keys are non-negative floats: bit 31 is 0, so the three digits are bits 30..21 | 20..10 | 9..0 -> 10 + 11 + 10 bits
    const int shifts[3] = {21, 10, 0};
    const int bins[3] = {1024, 2048, 1024};
    uint32_t prefix_mask = 0;
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(race_hist_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, weights, n_weights, seed_word, salt, state, prefix_mask,
                           shifts[p], (uint32_t)(bins[p] - 1), hist);
        hipLaunchKernelGGL(race_select_kernel, dim3(1), dim3(1024), 0, st, hist, bins[p], shifts[p], state);
        prefix_mask |= (uint32_t)(bins[p] - 1) << shifts[p];
    }
    hipLaunchKernelGGL(race_emit_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, weights, n_weights, seed_word, salt, state, (uint32_t)k, flat_out);
    return check_launch("sample_importance");
}

extern "C" int emer_buffer_to_pixels(const int64_t *flat, int64_t n, int32_t buffer_height, int32_t buffer_width, int32_t downscale,
                                     const int64_t *candidates, int32_t height, int32_t width, const uint64_t *seed_word, uint64_t salt,
                                     int64_t *img_idx, int64_t *y, int64_t *x, void *stream) {
    EMER_REQUIRE(n >= 0 && buffer_height >= 1 && buffer_width >= 1 && downscale >= 1, "buffer_to_pixels: bad sizes");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(flat && seed_word && img_idx && y && x, "buffer_to_pixels: null pointer");
    hipLaunchKernelGGL(buffer_to_pixels_kernel, dim3((uint32_t)ceil_div(n, 256)), dim3(256), 0, as_stream(stream), flat, n, buffer_height,
                       buffer_width, downscale, candidates, height, width, seed_word, salt, img_idx, y, x);
    return check_launch("buffer_to_pixels");
}
