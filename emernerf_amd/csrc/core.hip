// Host-side pieces of the C ABI: error string, version, hash-grid level table.
#include "common.h"

#include <math.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

namespace emer {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int reserve_lds(const void *kernel, size_t bytes, const char *what) {
    if (bytes <= 48 * 1024) return EMER_OK;
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> done;  // (kernel, device) -> largest size reserved so far
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = done[{kernel, dev}];
    if (have >= bytes) return EMER_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { set_error("%s: cannot reserve %zu B of LDS: %s", what, bytes, hipGetErrorString(e)); return EMER_E_LAUNCH; }
    have = bytes;
    return EMER_OK;
}
static thread_local ProfileEvents g_profile = {nullptr, nullptr};
ProfileEvents take_profile_events() {
    const ProfileEvents e = g_profile;
    g_profile = ProfileEvents{nullptr, nullptr};
    return e;
}
void arm_profile_events(hipEvent_t a, hipEvent_t b) { g_profile = ProfileEvents{a, b}; }
}  // namespace emer

extern "C" int emer_profile_next(void *start_event, void *stop_event) {
    emer::arm_profile_events(reinterpret_cast<hipEvent_t>(start_event), reinterpret_cast<hipEvent_t>(stop_event));
    return EMER_OK;
}

extern "C" const char *emer_last_error(void) { return emer::g_err; }
extern "C" int emer_version(void) { return 1; }

// Level table of tiny-cuda-nn's HashGrid (reference: radiance_fields/encodings.py:130-146 builds the
// config, third_party/tcnn_modules.py:420-423 hands it to _C.create_encoding).  Rules: SURVEY.md A.1:
//   scale_l = exp2(l * log2(per_level_scale)) * base - 1 (fp32), res_l = ceil(scale_l) + 1,
//   size_l  = min(round_up(res_l^D, 8), 2^T); hashed iff the dense stride product overtakes size_l.
extern "C" int emer_grid_desc_init(emer_grid_desc *g, uint32_t n_dims, uint32_t n_levels, uint32_t n_features,
                                   uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale) {
    EMER_REQUIRE(g != nullptr, "grid_desc_init: null descriptor");
    EMER_REQUIRE(n_dims >= 2 && n_dims <= 4, "grid_desc_init: n_dims=%u not in 2..4", n_dims);
    EMER_REQUIRE(n_levels >= 1 && n_levels <= EMER_MAX_LEVELS, "grid_desc_init: n_levels=%u not in 1..%d", n_levels, EMER_MAX_LEVELS);
    EMER_REQUIRE(n_features == 1 || n_features == 2 || n_features == 4 || n_features == 8, "grid_desc_init: n_features=%u not in {1,2,4,8}", n_features);
    EMER_REQUIRE(log2_hashmap_size >= 3 && log2_hashmap_size <= 28, "grid_desc_init: log2_hashmap_size=%u out of range", log2_hashmap_size);
    EMER_REQUIRE(base_resolution >= 1 && per_level_scale > 0.0f, "grid_desc_init: bad base_resolution / per_level_scale");
    memset(g, 0, sizeof(*g));
    g->n_dims = n_dims; g->n_levels = n_levels; g->n_features = n_features;
    g->log2_hashmap_size = log2_hashmap_size; g->base_resolution = base_resolution;
    g->per_level_scale = per_level_scale;
    const float log2_pls = log2f(per_level_scale);
    uint64_t offset = 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        uint32_t dense;
        if (powf((float)res, (float)n_dims) > (float)max_params) {
            dense = max_params;
        } else {
            uint64_t p = 1;
            for (uint32_t d = 0; d < n_dims; ++d) p *= res;
            dense = (uint32_t)p;
        }
        dense = (dense + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        const uint32_t size = dense < cap ? dense : cap;
        uint64_t stride = 1;
        for (uint32_t d = 0; d < n_dims && stride <= size; ++d) stride *= res;
        g->scale[l] = scale; g->res[l] = res; g->size[l] = size; g->offset[l] = (uint32_t)offset;
        g->hashed[l] = (size < stride) ? 1u : 0u;
        offset += size;
        EMER_REQUIRE(offset < (1ull << 32), "grid_desc_init: table too large");
    }
    g->n_entries = (uint32_t)offset;
    return EMER_OK;
}
