// Multiresolution hash-grid encode / backward for gfx950 (MI355X).
//
// Replaces tiny-cuda-nn's GridEncoding kernels behind tcnn.Encoding (reference call sites:
// radiance_fields/encodings.py:159-160 -> third_party/tcnn_modules.py:122 (fwd), :161-163 (bwd)).
// Semantics: SURVEY.md Appendix A.1; CPU restatement: oracle/emer_oracle.c.
//
// MI355X design notes
//   * one thread = one (sample, level); a workgroup = 256 consecutive samples of ONE level, so all
//     per-level constants live in SGPRs and the 64 lanes of a wave walk spatially adjacent samples
//     of one ray (coarse-level corners coalesce in the texture-address unit / L1).
//   * block -> (level, chunk) mapping is XCD-aware: MI355X dispatches block b to XCD b % 8 and every
//     XCD has a private 4 MiB L2.  Levels are dealt to XCDs round-robin in groups of 8 and each XCD
//     finishes one level before starting its next one, so a level's table (<= 4 MiB for T=2^19, F=2,
//     fp32; 2 MiB in fp16) is gathered / scattered out of ONE L2 instead of thrashing all eight.
//     This is a speed heuristic only: results never depend on placement.
//   * output / dOut use caller-supplied (stride_n, stride_l) so the fused heads can ask for the
//     level-major [L][N][F] layout (fully coalesced 64 x F*4 B stores per wave) while the drop-in
//     HashEncoder.forward can still get the reference's row-major [N, L*F].
#include "common.h"

namespace emer {

template <int F, typename PT>
__device__ __forceinline__ void load_feats(const PT *__restrict__ p, float (&v)[F]);

template <> __device__ __forceinline__ void load_feats<1, float>(const float *__restrict__ p, float (&v)[1]) { v[0] = p[0]; }
template <> __device__ __forceinline__ void load_feats<2, float>(const float *__restrict__ p, float (&v)[2]) {
    float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y;
}
template <> __device__ __forceinline__ void load_feats<4, float>(const float *__restrict__ p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load_feats<8, float>(const float *__restrict__ p, float (&v)[8]) {
    float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load_feats<1, __half>(const __half *__restrict__ p, float (&v)[1]) { v[0] = __half2float(p[0]); }
template <> __device__ __forceinline__ void load_feats<2, __half>(const __half *__restrict__ p, float (&v)[2]) {
    float2 t = __half22float2(*reinterpret_cast<const __half2 *>(p)); v[0] = t.x; v[1] = t.y;
}
template <> __device__ __forceinline__ void load_feats<4, __half>(const __half *__restrict__ p, float (&v)[4]) {
    uint2 raw = *reinterpret_cast<const uint2 *>(p);
    float2 a = __half22float2(*reinterpret_cast<__half2 *>(&raw.x)), b = __half22float2(*reinterpret_cast<__half2 *>(&raw.y));
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
template <> __device__ __forceinline__ void load_feats<8, __half>(const __half *__restrict__ p, float (&v)[8]) {
    uint4 raw = *reinterpret_cast<const uint4 *>(p);
    float2 a = __half22float2(*reinterpret_cast<__half2 *>(&raw.x)), b = __half22float2(*reinterpret_cast<__half2 *>(&raw.y));
    float2 c = __half22float2(*reinterpret_cast<__half2 *>(&raw.z)), d = __half22float2(*reinterpret_cast<__half2 *>(&raw.w));
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
}

template <int F>
__device__ __forceinline__ void atomic_add_feats(float *p, const float (&v)[F]) {
#pragma unroll
    for (int f = 0; f < F; ++f) __hip_atomic_fetch_add(p + f, v[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int F>
__device__ __forceinline__ void atomic_add_feats(__half *p, const float (&v)[F]) {
    static_assert(F % 2 == 0, "fp16 gradient tables need an even feature count");
#pragma unroll
    for (int f = 0; f < F; f += 2) unsafeAtomicAdd(reinterpret_cast<__half2 *>(p + f), __floats2half2_rn(v[f], v[f + 1]));
}

__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct LevelInfo {
    float scale;
    uint32_t res, size, offset, hashed;
};

__device__ __forceinline__ LevelInfo level_info(const emer_grid_desc &g, uint32_t l) {
    return LevelInfo{g.scale[l], g.res[l], g.size[l], g.offset[l], g.hashed[l]};
}

// XCD-aware block -> (level, chunk) map (see file header).  The (level, chunk) space is walked level-major and cut into
// eight CONTIGUOUS, equal-COST segments, one per XCD (block b runs on XCD b % 8): an XCD touches two or three levels at
// most, so a level's table stays in one or two L2s, and the finer levels -- whose gathers miss L1 more often -- get
// proportionally fewer samples per XCD.  Returns false for padding blocks.
struct LevelMap {
    uint32_t start[8], count[8];
};
__device__ __forceinline__ bool map_block(uint32_t bid, const LevelMap &lm, uint32_t n_chunks, uint32_t &level, uint32_t &chunk) {
    const uint32_t x = bid & 7u, j = bid >> 3;
    if (j >= lm.count[x]) return false;
    const uint32_t idx = lm.start[x] + j;
    level = idx / n_chunks;
    chunk = idx - level * n_chunks;
    return true;
}
// relative gather cost of a level per sample, fitted to tools/kbench.py --per-level on MI355X: flat up to res ~150,
// then ~+35 % per doubling of the resolution (adjacent ray samples stop sharing cache lines)
static inline float fwd_level_cost(const emer_grid_desc *g, uint32_t l) {
    const float r = (float)g->res[l];
    float c = 1.0f + 0.35f * log2f(r / 150.0f);
    return c < 1.0f ? 1.0f : (c > 2.1f ? 2.1f : c);
}
static LevelMap make_level_map(const emer_grid_desc *g, uint32_t n_chunks, uint32_t *blocks_out) {
    LevelMap lm;
    const uint32_t L = g->n_levels;
    double total = 0.0;
    for (uint32_t l = 0; l < L; ++l) total += fwd_level_cost(g, l);
    // boundary of segment x: smallest position whose cumulative cost reaches x/8 of the total
    uint64_t bound[9];
    bound[0] = 0; bound[8] = (uint64_t)L * n_chunks;
    for (int x = 1; x < 8; ++x) {
        const double target = total * x / 8.0;
        double cum = 0.0;
        uint64_t pos = bound[8];
        for (uint32_t l = 0; l < L; ++l) {
            const double c = fwd_level_cost(g, l);
            if (cum + c >= target) { pos = (uint64_t)l * n_chunks + (uint64_t)((target - cum) / c * n_chunks); break; }
            cum += c;
        }
        bound[x] = pos < bound[x - 1] ? bound[x - 1] : pos;
    }
    uint32_t mx = 0;
    for (int x = 0; x < 8; ++x) {
        lm.start[x] = (uint32_t)bound[x];
        lm.count[x] = (uint32_t)(bound[x + 1] - bound[x]);
        if (lm.count[x] > mx) mx = lm.count[x];
    }
    *blocks_out = mx * 8u;
    return lm;
}

template <int D>
__device__ __forceinline__ uint32_t grid_index(const LevelInfo &li, const uint32_t (&c)[D]) {
    uint32_t idx;
    if (li.hashed) {  // level-uniform branch
        idx = c[0];
        idx ^= c[1] * 2654435761u;
        if (D > 2) idx ^= c[2 < D ? 2 : 0] * 805459861u;
        if (D > 3) idx ^= c[3 < D ? 3 : 0] * 3674653429u;
    } else {
        uint32_t stride = 1;
        idx = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (stride <= li.size) { idx += c[d] * stride; stride *= li.res; }
        }
    }
    // idx % size without the integer divide on the common paths (size is level-uniform)
    if ((li.size & (li.size - 1u)) == 0u) return idx & (li.size - 1u);
    if (idx >= li.size) { idx -= li.size; if (idx >= li.size) idx %= li.size; }
    return idx;
}

template <int D>
__device__ __forceinline__ void load_x(const float *__restrict__ x, int64_t n, float (&v)[D]) {
    if (D == 4) {
        float4 t = *reinterpret_cast<const float4 *>(x + n * 4);
        v[0] = t.x; v[1] = t.y; v[2 < D ? 2 : 0] = t.z; v[3 < D ? 3 : 0] = t.w;
    } else if (D == 2) {
        float2 t = *reinterpret_cast<const float2 *>(x + n * 2);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = x[n * D + d];
    }
}

template <int D>
__device__ __forceinline__ void cell_of(const LevelInfo &li, const float (&xv)[D], uint32_t (&gi)[D], float (&w)[D]) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float pos = fmaf(li.scale, xv[d], 0.5f);
        const float fl = floorf(pos);
        gi[d] = (uint32_t)(int32_t)fl;
        w[d] = pos - fl;
    }
}

// slice plan of the owner-computes backward (described further down); the forward kernel emits the masks
struct SlicePlan {
    uint32_t shift[EMER_MAX_LEVELS];     // slice = idx >> shift (contiguous index ranges that fit the LDS)
    uint32_t n_slices[EMER_MAX_LEVELS];
    uint32_t gsub[EMER_MAX_LEVELS];      // log2(slices per bitmap group): bitmap row = slice >> gsub (<= 64 rows per level)
    uint32_t n_ranges[EMER_MAX_LEVELS];  // dense levels: the sample stream is also cut in ranges (2-D decomposition)
    uint32_t mask_q;                     // 64-row groups of bitmap rows per level (1 or 4): bitmaps[(level * 64 * mask_q + row) * n_words + word]
    uint32_t max_local;                  // largest slice (entries)
    uint32_t ok;                         // 0 when some level would need more than 64 bitmap groups of 64 slices
    // (32-bit on purpose: a uint8_t array in this by-value kernel argument, indexed in a loop, was read back wrong by
    // the device code -- hipcc 7.2)
    uint32_t xcd_of[EMER_MAX_LEVELS];    // backward: the XCD (0..7) that owns each level (cost-balanced)
    uint32_t order[EMER_MAX_LEVELS];     // backward: levels in the order an XCD's list walks them (heaviest items first)
    uint32_t items_per_xcd[8];           // backward work items ((level, slice, range) triples) on each XCD's list
    uint32_t total_items;
    uint32_t sched_shift;                // log2 of the scheduling block: consecutive work items dealt to one XCD
};

                           // with a 64 KiB slice (two per CU: 128 slices per hashed level, 256-row bitmaps)
constexpr uint32_t kSchedBlock = 32;  // consecutive work items dealt to one XCD (= its resident owners: one round)
// [r5] Grids WITHOUT dense levels (the xyzt tables: 10 hashed levels x 64 slices = 640 items of 300-450 us for 256 owners) have no
// small items to fill the tail with: tools/trace_sliced.py showed four XCDs working three rounds and four XCDs two -- 1041 us of work
// per owner in a 1400 us kernel.  For such grids (a) the levels at the END of the order, whose items make up the last, incomplete
// round, are cut in EMER_TAIL_SPLIT sample ranges (merged with atomics like the dense levels: half-size items fill the round), and
// (b) EMER_SCHED_BLOCK_NOFILL = 16 / 8 deals the items to the XCDs in smaller blocks, so that every XCD's list holds a part of MORE
// levels and the lists' totals even out -- measured and LOST: a level's x / dout lines are then fetched into four L2s instead of two
// (xyzt table at 1 M samples, same session: 1390 us before; wide pairs 1165; + tail split 2, blocks of 32: 1000; blocks of 16: 1090,
// of 8: 1120; tail split 4 with blocks of 16: 1025; profiles/r05_grid_schedule.txt).  Default 32 = no change.
#ifndef EMER_TAIL_SPLIT
#define EMER_TAIL_SPLIT 2
#endif
#ifndef EMER_SCHED_BLOCK_NOFILL
#define EMER_SCHED_BLOCK_NOFILL 32
#endif
static float level_cost(const emer_grid_desc *g, const SlicePlan &p, uint32_t l) {
    // fitted to tools/kbench.py --per-level on MI355X (1M samples, ms on one XCD): coarse levels pay for
    // same-address LDS adds (many samples per cell), dense levels for the ordered scan + run reduction
    const float rescans = (float)(1u << p.gsub[l]);
    if (g->hashed[l]) return (0.28f + 36.0f / (float)g->res[l]) * rescans;
    return (0.22f + 0.13f * log2f(1.0f + (float)p.n_slices[l])) * rescans;
}

// worst-case work of ONE item of a level, in units of the sample count: a hashed slice sees ~4/64 of the samples
// (more on coarse levels, see level_cost); a dense slab-range item may see ALL samples of its range (flat scenes
// concentrate in two or three slabs)
static float item_cost(const emer_grid_desc *g, const SlicePlan &p, uint32_t l) {
    if (g->hashed[l]) return (4.0f / 64.0f) * (0.28f + 36.0f / (float)g->res[l]) / 0.3f * (float)(1u << p.gsub[l]) / (float)p.n_ranges[l];
    return 2.0f / (float)p.n_ranges[l];
}

#ifndef EMER_DENSE_ITEMS
#define EMER_DENSE_ITEMS 512  // work items per dense level (slabs x sample ranges): 256 -> 512 -3 %, 1024 +8 % (merge atomics) on MI355X
#endif
static SlicePlan make_slice_plan(const emer_grid_desc *g) {
    SlicePlan p;
    const uint32_t F = g->n_features;
    const uint32_t max_entries = ((128u) * 1024u) / (F * 8u);  // 128 KiB of the CU's 160 KiB LDS, double accumulators
    p.max_local = 0; p.ok = 1;
    for (uint32_t l = 0; l < EMER_MAX_LEVELS; ++l) { p.shift[l] = 0; p.n_slices[l] = 0; p.n_ranges[l] = 1; p.gsub[l] = 0; }
    for (uint32_t l = 0; l < g->n_levels; ++l) {
        const uint32_t size = g->size[l];
        uint32_t k = 6;
        if (g->hashed[l]) {
            // the hash spreads samples evenly: 64 slices (more when 1/64 of the table does not fit the LDS), one pass
            // over all samples each
            while ((1ull << k) * 64ull < size) ++k;
            while ((1u << k) > max_entries) --k;
            // Coarse hashed levels: rays share cells, so some slices hold hot entries whose LDS adds serialise -- single
            // items of level 5 (res 81) take 2.5x the level's mean (tools/trace_sliced.py) and, started in a second round,
            // set the kernel's tail.  Cutting THEIR sample stream in ranges (merged with atomics, like the dense levels)
            // bounds the longest item.
            p.n_ranges[l] = 1u;   // (sample ranges on hashed levels: measured, merge atomics cost what the balance gains; only the tail split below cuts them)
        } else {
            // dense level: a slice is a contiguous z-slab and a flat scene lands in two or three of them, so
            // use as FEW slices as the LDS allows and cut the sample stream instead
            while ((1u << (k + 1)) <= max_entries && (1u << k) < size) ++k;
            const uint32_t ns = (uint32_t)ceil_div(size, 1ll << k);
            // ~256 work items per dense level (at most 128 ranges): a flat scene puts most samples into two or three
            // slabs, and ONE slab-range item must not become the critical path of the whole kernel (with 128 items the
            // heaviest item of level 4 alone took as long as the kernel: tools/probe_levels_train.py)
            uint32_t nr = (uint32_t)EMER_DENSE_ITEMS / (ns ? ns : 1u);
            if (nr > 128u) nr = 128u;
            p.n_ranges[l] = nr < 1u ? 1u : nr;
        }
        p.shift[l] = k;
        p.n_slices[l] = (uint32_t)ceil_div(size, 1ll << k);
        const uint32_t local = 1u << k;
        if (local > max_entries) p.ok = 0;
        if (local > p.max_local) p.max_local = local;
    }
    // [r5] tail filler (see EMER_TAIL_SPLIT above).  The hashed items (one per level and slice, ~equal cost) are taken in rounds of
    // `owners`; the last round is incomplete when their count is not a multiple of it, and the idle owners can only be fed with the
    // small items of the dense levels (~0.12 of a hashed item each, tools/trace_sliced.py).  Where that filler covers less than half of
    // the hole, the levels whose items make up the incomplete round -- the finest ones: they sort last -- are cut in R sample
    // ranges, R in {2, 4} chosen to minimise the tail ceil(rem R / owners) / R (dynamic xyzt table: 640 items -> rem 128 -> R = 2;
    // flow xyzt table: 576 hashed items + one dense level -> rem 64 -> R = 4; cfg-2 main grid: 1900 dense items -> no cut).
    p.sched_shift = kSchedBlock == 32u ? 5u : 6u;
    {
        uint32_t hashed_items = 0, dense_items = 0;
        for (uint32_t l = 0; l < g->n_levels; ++l) {
            if (g->hashed[l]) hashed_items += p.n_slices[l];
            else dense_items += p.n_slices[l] * p.n_ranges[l];
        }
        const uint32_t owners = 256u;
        const uint32_t rem = hashed_items % owners;
        const float hole = (float)(owners - rem), filler = 0.12f * (float)dense_items;
        if (EMER_TAIL_SPLIT > 1 && hashed_items > owners && rem != 0u && filler < 0.5f * hole) {
            if (dense_items == 0u) {   // (scheduling block experiment: only ever measured on a grid without dense levels)
                if (EMER_SCHED_BLOCK_NOFILL == 16) p.sched_shift = 4u;
                else if (EMER_SCHED_BLOCK_NOFILL == 8) p.sched_shift = 3u;
            }
            uint32_t best_r = 1; float best_t = 1.0f;
            for (uint32_t r = 2; r <= 4u; r *= 2u) {
                const float t = (float)((rem * r + owners - 1u) / owners) / (float)r;
                if (t < best_t - 1e-6f) { best_t = t; best_r = r; }
            }
            if (EMER_TAIL_SPLIT != 2) best_r = (uint32_t)EMER_TAIL_SPLIT;   // (A/B builds force a factor; 2 = automatic)
            uint32_t covered = 0;
            for (uint32_t l = g->n_levels; l-- > 0u && covered < rem;) {
                if (!g->hashed[l]) continue;
                p.n_ranges[l] = best_r;
                covered += p.n_slices[l];
            }
        }
    }
    // Bitmap rows.  The forward emits 64 rows per level, or 256 when some level has more slices (T = 2^20 with F = 4:
    // 256 slices): every slice then still has its OWN bitmap.  (Round 1 / first half of round 2 shared one 64-row bitmap
    // among 2^gsub neighbouring slices, whose owners each scanned -- gathered, hashed and mostly discarded -- the hits of
    // all of them: 4x the work on the default static grid.)  Beyond 256 slices the sharing remains.
    p.mask_q = 1;
    for (uint32_t l = 0; l < g->n_levels; ++l)
        if (p.n_slices[l] > 64u) p.mask_q = 4;
    for (uint32_t l = 0; l < g->n_levels; ++l) {
        while (((p.n_slices[l] + (1u << p.gsub[l]) - 1u) >> p.gsub[l]) > 64u * p.mask_q) ++p.gsub[l];
        if (p.gsub[l] > 6u) p.ok = 0;
    }
    // Scheduling.  ONE global order of work items -- levels sorted by the cost of a single item, heaviest first (below) --
    // dealt to the eight XCD lists in blocks of kSchedBlock consecutive items (block b -> XCD b % 8).  A block is one
    // "round" of an XCD's 32 CUs working on the same level (shared x / dout lines in that L2), a level's 64 slices
    // spread over two XCDs, and -- what matters most -- EVERY XCD starts with the long items (coarse hashed levels,
    // ~200 us each) and ends with the short ones (dense slab-range items, ~20 us), so the tail of the kernel is filled
    // with small work.  (Round 1 kept whole levels on one XCD: with 16 levels of unequal cost on 8 lists the heaviest
    // list (levels 6 + 12) alone took the kernel's whole duration and long items were still being STARTED after 300 us,
    // tools/trace_sliced.py.)
    uint32_t total_items = 0;
    for (uint32_t l = 0; l < g->n_levels; ++l) total_items += p.n_slices[l] * p.n_ranges[l];
    p.total_items = total_items;
    for (int i = 0; i < 8; ++i) p.items_per_xcd[i] = 0;
    const uint32_t sched_block = 1u << p.sched_shift;
    for (uint32_t blk = 0; blk * sched_block < total_items; ++blk) {
        const uint32_t left = total_items - blk * sched_block;
        p.items_per_xcd[blk & 7u] += left < sched_block ? left : sched_block;
    }
    for (uint32_t l = 0; l < EMER_MAX_LEVELS; ++l) p.xcd_of[l] = 0;  // (unused: kept for layout stability of the argument struct)
    // the global order: levels with the most expensive single items first
    for (uint32_t l = 0; l < EMER_MAX_LEVELS; ++l) p.order[l] = l;
    for (uint32_t a = 0; a + 1 < g->n_levels; ++a)
        for (uint32_t b = a + 1; b < g->n_levels; ++b) {
            const float ca = item_cost(g, p, p.order[a]), cb = item_cost(g, p, p.order[b]);
            if (cb > ca) { const uint32_t t = p.order[a]; p.order[a] = p.order[b]; p.order[b] = t; }
        }
    return p;
}

// bitmap row (group of 2^gsub slices) an entry belongs to
__device__ __forceinline__ uint32_t slice_of(const SlicePlan &p, uint32_t level, uint32_t idx) {
    return idx >> (p.shift[level] + p.gsub[level]);
}
__device__ __forceinline__ uint32_t bitmap_rows(const SlicePlan &p, uint32_t level) {
    return (p.n_slices[level] + (1u << p.gsub[level]) - 1u) >> p.gsub[level];
}


// 64x64 bit-matrix transpose across the 64 lanes of a wave (lane r holds row r; afterwards lane c holds
// column c): six butterfly stages, each swapping the off-diagonal blocks with lane ^ j -- all on VALU data-parallel
// primitives (DPP, v_permlane16/32_swap_b32 of gfx950), no ds_bpermute round trips through the LDS crossbar.
// value of lane ^ J for J in {1, 2, 4, 8, 16}, without touching the LDS crossbar: quad permutes (1, 2), row shifts
// selected by the lane's bit (4), a row rotation by 8, and the gfx950 row-pair swap v_permlane16_swap_b32 (16)
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v, int lane) {
    if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);        // quad_perm [2,3,0,1]
    if (J == 4) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, false);  // row_shl:4  <- lane + 4
        const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4  <- lane - 4
        return (lane & 4) ? dn : up;
    }
    if (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);       // row_ror:8
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // r[0]: odd rows <- lane - 16; r[1]: even rows <- lane + 16
    return (lane & 16) ? r[0] : r[1];
}
template <int J>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v, int lane) {
    return ((uint64_t)lane_xor_u32<J>((uint32_t)(v >> 32), lane) << 32) | lane_xor_u32<J>((uint32_t)v, lane);
}
template <int J>
__device__ __forceinline__ uint64_t transpose_stage(uint64_t x, uint64_t lm, int lane) {
    const uint64_t y = lane_xor_u64<J>(x, lane);
    return (lane & J) ? (((y & ~lm) >> J) | (x & ~lm)) : ((x & lm) | ((y & lm) << J));
}
__device__ __forceinline__ uint64_t wave_bit_transpose(uint64_t x, int lane) {
    {   // j = 32: whole dwords change halves of the wave
        const auto r = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)(x >> 32), false, false);
        x = ((uint64_t)r[1] << 32) | r[0];
    }
    x = transpose_stage<16>(x, 0x0000FFFF0000FFFFull, lane);
    x = transpose_stage<8>(x, 0x00FF00FF00FF00FFull, lane);
    x = transpose_stage<4>(x, 0x0F0F0F0F0F0F0F0Full, lane);
    x = transpose_stage<2>(x, 0x3333333333333333ull, lane);
    x = transpose_stage<1>(x, 0x5555555555555555ull, lane);
    return x;
}

// The owner-computes backward reads ONE bit per sample for each (level, slice).  A wave holding the 64-bit slice masks
// of 64 consecutive samples transposes them, so lane s holds the membership bits of slice s; the four waves of a
// workgroup (256 consecutive samples) stage their columns in LDS and wave 0 writes 32 contiguous bytes per bitmap row
// (one store instruction touches 64 lines instead of four doing so).  Layout: bitmaps[(level * 64 + s) * n_words + word].
// Must be called by ALL 256 threads of the workgroup (it synchronises).
template <int Q>
__device__ __forceinline__ void store_slice_bitmaps(uint64_t *__restrict__ bitmaps, const uint64_t (&mask)[Q], uint32_t level, uint32_t n_rows,
                                                    int64_t n0_block, int64_t n_words, int tid) {
    __shared__ uint64_t tile[Q][4][64];  // [row group][wave][row]: conflict-free writes and reads
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < Q; ++q) tile[q][wave][lane] = wave_bit_transpose(mask[q], lane);
    __syncthreads();
    // Q == 1: wave 0 writes the 64 rows; Q == 4: wave q writes rows 64 q .. 64 q + 63
    const int q = Q == 1 ? 0 : wave;
    const uint32_t row = (uint32_t)(q * 64 + lane);
    if ((Q > 1 || wave == 0) && row < n_rows) {
        const int64_t w0 = n0_block >> 6;  // first word of this workgroup (a multiple of 4)
        uint64_t *dst = bitmaps + ((int64_t)level * (64 * Q) + row) * n_words + w0;
        if ((n_words & 3) == 0 && w0 + 4 <= n_words) {
            reinterpret_cast<ulonglong2 *>(dst)[0] = make_ulonglong2(tile[q][0][lane], tile[q][1][lane]);
            reinterpret_cast<ulonglong2 *>(dst)[1] = make_ulonglong2(tile[q][2][lane], tile[q][3][lane]);
        } else {
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w0 + w < n_words) dst[w] = tile[q][w][lane];
        }
    }
}
// set bit `row` of a Q x 64-bit row mask held in registers
template <int Q>
__device__ __forceinline__ void set_row(uint64_t (&mask)[Q], uint32_t row) {
    const uint64_t b = 1ull << (row & 63u);
    if (Q == 1) { mask[0] |= b; return; }
#pragma unroll
    for (int q = 0; q < Q; ++q) mask[q] |= ((row >> 6) == (uint32_t)q) ? b : 0ull;
}

// ------------------------------------------------------------------------------------ forward
// JAC [r4]: rows n >= jac_row0 also store J[f][d] = d out[n][level][f] / d x[n][d] (jac[level][n - jac_row0][F][D]) -- the corner
// values are in registers here anyway, and the input gradient of the flow configs becomes a streaming contraction of J with dOut
// (hashgrid_bwd_input_jac_kernel) instead of a second pass of 2^D * L gathers per sample (hashgrid_bwd_input_kernel).  The encoding
// itself is computed exactly as without JAC (same loop, same order).
template <int D, int F, typename PT, int Q, bool JAC = false>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(const emer_grid_desc g, const float *__restrict__ x,
                                                           const PT *__restrict__ params, float *__restrict__ out,
                                                           int64_t sn, int64_t sl, int64_t N, uint32_t n_chunks, const LevelMap lmap,
                                                           const SlicePlan plan, uint64_t *__restrict__ masks,
                                                           float *__restrict__ jac = nullptr, int64_t jac_row0 = 0) {
    uint32_t level, chunk;
    if (!map_block(blockIdx.x, lmap, n_chunks, level, chunk)) return;
    const int64_t n = (int64_t)chunk * 256 + threadIdx.x;
    const bool valid = n < N;
    if (!valid && !masks) return;
    const LevelInfo li = level_info(g, level);
    const PT *__restrict__ table = params + (size_t)li.offset * F;

    uint64_t mask[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) mask[q] = 0ull;
    if (valid) {
        float xv[D], w[D];
        uint32_t gi[D];
        load_x<D>(x, n, xv);
        cell_of<D>(li, xv, gi, w);

        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.0f;
        const bool pow2 = (li.size & (li.size - 1u)) == 0u;
        bool paired = false;
        float vv[JAC ? (1 << D) : 1][F];   // JAC: every corner's features stay live for the differences along each axis
        // [r6] ... and for 16-byte fp32 entries (F = 4: the default static table and the xyzt tables) the pair is two 16-byte loads from ONE
        // 32-byte sector instead of two sectors of two lines: a quarter fewer lines per hashed level -- xyzt forward 516 -> 429 us, flow
        // table 472 -> 407 us, default static table 519 -> 503 us per million samples, dynamic step 6.18 -> 6.04 ms (same-session A/B,
        // profiles/r06_pair16.txt).  The Jacobian forward keeps the generic loop: paired it needs 130 registers, three waves per SIMD
        // instead of four, and loses what the pairs gain (448 vs 437 us; forced to 128 registers it spills: 575 us).
        if constexpr ((sizeof(PT) * F <= 8 || (sizeof(PT) == 4 && F == 4)) && !JAC) {
        if (li.hashed && pow2) {  // level-uniform
            paired = true;
            // Hashed power-of-two level: the x-neighbours of a (y, z[, t]) combination are idx0 = (x ^ h) & mask and
            // idx1 = ((x+1) ^ h) & mask.  For even x they are the two halves of ONE aligned entry pair {2k, 2k+1}:
            // a single double-width gather fetches both (a quarter fewer L1/TA lane requests on these levels).
            // Accumulation order is unchanged (corner-major, x fastest).
            const uint32_t primes[4] = {1u, 2654435761u, 805459861u, 3674653429u};
            const uint32_t maskv = li.size - 1u;
            const bool x_even = (gi[0] & 1u) == 0u;
            // inside the grid (gi[0] + 1 <= res) the two x-neighbours differ only in index bits below the slice bits when
            // the resolution is below the slice width: one membership bit covers both
            const bool x_pair_one_slice = li.res < (1u << plan.shift[level]) && !__ballot(gi[0] >= li.res);
#pragma unroll
            for (uint32_t m = 0; m < (1u << (D - 1)); ++m) {
                uint32_t h = 0;
                float t[D];
#pragma unroll
                for (int d = 1; d < D; ++d) {
                    const uint32_t bit = (m >> (d - 1)) & 1u;
                    h ^= (gi[d] + bit) * primes[d];
                    t[d] = bit ? w[d] : 1.0f - w[d];
                }
                const uint32_t idx0 = (gi[0] ^ h) & maskv, idx1 = ((gi[0] + 1u) ^ h) & maskv;
                float v0[F], v1[F];
                if (x_even) {
                    float e[2 * F];
                    load_feats<2 * F, PT>(table + (size_t)(idx0 & ~1u) * F, e);
#pragma unroll
                    for (int f = 0; f < F; ++f) { v0[f] = (idx0 & 1u) ? e[F + f] : e[f]; v1[f] = (idx0 & 1u) ? e[f] : e[F + f]; }
                } else {
                    load_feats<F, PT>(table + (size_t)idx0 * F, v0);
                    load_feats<F, PT>(table + (size_t)idx1 * F, v1);
                }
                float wa = 1.0f - w[0], wb = w[0];  // ((t0*t1)*t2)*t3, as the generic loop
#pragma unroll
                for (int d = 1; d < D; ++d) { wa *= t[d]; wb *= t[d]; }
#pragma unroll
                for (int f = 0; f < F; ++f) acc[f] += wa * v0[f];
#pragma unroll
                for (int f = 0; f < F; ++f) acc[f] += wb * v1[f];
                if (masks) {
                    set_row<Q>(mask, slice_of(plan, level, idx0));
                    if (!x_pair_one_slice) set_row<Q>(mask, slice_of(plan, level, idx1));  // (level-uniform)
                }
            }
        }
        }
        if (!paired) {
#pragma unroll
        for (uint32_t m = 0; m < (1u << D); ++m) {
            float wt = 1.0f;
            uint32_t c[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (m & (1u << d)) { wt *= w[d]; c[d] = gi[d] + 1u; }
                else { wt *= 1.0f - w[d]; c[d] = gi[d]; }
            }
            float v[F];
            const uint32_t idx = grid_index<D>(li, c);
            load_feats<F, PT>(table + (size_t)idx * F, v);
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] += wt * v[f];  // same order as the oracle (corner-major)
            if constexpr (JAC) {
#pragma unroll
                for (int f = 0; f < F; ++f) vv[m][f] = v[f];
            }
            if (masks) set_row<Q>(mask, slice_of(plan, level, idx));  // by-product for the owner-computes backward
        }
        }
        if constexpr (JAC) {
            // [r6] The encoding leaves FIRST.  hipcc used to schedule the Jacobian (and its four 16-byte stores) ahead of the accumulation
            // of the encoding; on gfx950 stores count in vmcnt like loads, so the waits that hand the gathered corners to the accumulation
            // (vmcnt 15 .. 0 in its model) then also waited for the Jacobian stores' acknowledgements -- a memory round trip in the
            // middle of every wave.  The barrier pins: accumulate, store the encoding, then the Jacobian (same-session A/B,
            // profiles/r06_xyzt_jac.txt: 448 -> 440 us at the 2048-ray shard, 1822 -> 1757 us at 8192 rays; results bitwise unchanged).
            {
                float *o = out + n * sn + (int64_t)level * sl;
                if (F == 2) { *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1 < F ? 1 : 0]); }
                else if (F == 4) { *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1 < F ? 1 : 0], acc[2 < F ? 2 : 0], acc[3 < F ? 3 : 0]); }
                else {
#pragma unroll
                    for (int f = 0; f < F; ++f) o[f] = acc[f];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (n >= jac_row0) {
                // J[f][gd] = scale * sum over the corners m with bit gd clear of prod_{d != gd} t_d(m) * (v[m | gd] - v[m])[f]:
                // the same differences, weights and corner order as hashgrid_bwd_input_kernel forms after projecting on dOut
                float J[F][D];
#pragma unroll
                for (int gd = 0; gd < D; ++gd) {
                    float a[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) a[f] = 0.0f;
#pragma unroll
                    for (uint32_t m = 0; m < (1u << D); ++m) {
                        if (m & (1u << gd)) continue;
                        float wt = li.scale;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            if (d == gd) continue;
                            wt *= (m & (1u << d)) ? w[d] : 1.0f - w[d];
                        }
#pragma unroll
                        for (int f = 0; f < F; ++f) a[f] += wt * (vv[m | (1u << gd)][f] - vv[m][f]);
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) J[f][gd] = a[f];
                }
                float *jp = jac + (((int64_t)level * (N - jac_row0)) + (n - jac_row0)) * (F * D);
                if constexpr ((F * D) % 4 == 0) {
#pragma unroll
                    for (int i = 0; i < F * D; i += 4)
                        *reinterpret_cast<float4 *>(jp + i) = make_float4(J[i / D][i % D], J[(i + 1) / D][(i + 1) % D], J[(i + 2) / D][(i + 2) % D], J[(i + 3) / D][(i + 3) % D]);
                } else {
#pragma unroll
                    for (int i = 0; i < F * D; ++i) jp[i] = J[i / D][i % D];
                }
            }
        }
        if constexpr (!JAC) {
        float *o = out + n * sn + (int64_t)level * sl;
        if (F == 2) { *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1 < F ? 1 : 0]); }
        else if (F == 4) { *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1 < F ? 1 : 0], acc[2 < F ? 2 : 0], acc[3 < F ? 3 : 0]); }
        else {
#pragma unroll
            for (int f = 0; f < F; ++f) o[f] = acc[f];
        }
        }
    }
    if (masks)  // the whole workgroup takes part (tail lanes carry an empty mask)
        store_slice_bitmaps<Q>(masks, mask, level, bitmap_rows(plan, level), (int64_t)chunk * 256, (N + 63) >> 6, (int)threadIdx.x);
}

// ------------------------------------------------------------------------ backward (params)
template <int D, int F, typename GT>
__global__ __launch_bounds__(256) void hashgrid_bwd_params_kernel(const emer_grid_desc g, const float *__restrict__ x,
                                                                  const float *__restrict__ dout, int64_t sn, int64_t sl,
                                                                  GT *__restrict__ grad, int64_t N, uint32_t n_chunks, const LevelMap lmap) {
    uint32_t level, chunk;
    if (!map_block(blockIdx.x, lmap, n_chunks, level, chunk)) return;
    const int64_t n = (int64_t)chunk * 256 + threadIdx.x;
    if (n >= N) return;
    const LevelInfo li = level_info(g, level);
    GT *__restrict__ table = grad + (size_t)li.offset * F;

    float go[F];
    const float *gp = dout + n * sn + (int64_t)level * sl;
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; ++f) { go[f] = gp[f]; any |= (go[f] != 0.0f); }
    if (!any) return;  // exact zeros add nothing (masked / fully occluded samples)

    float xv[D], w[D];
    uint32_t gi[D];
    load_x<D>(x, n, xv);
    cell_of<D>(li, xv, gi, w);
#pragma unroll
    for (uint32_t m = 0; m < (1u << D); ++m) {
        float wt = 1.0f;
        uint32_t c[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (m & (1u << d)) { wt *= w[d]; c[d] = gi[d] + 1u; }
            else { wt *= 1.0f - w[d]; c[d] = gi[d]; }
        }
        float v[F];
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = wt * go[f];
        atomic_add_feats<F>(table + (size_t)grid_index<D>(li, c) * F, v);
    }
}


// ------------------------------------------------------- backward (params), owner-computes
// Measured on MI355X (tools/atomic_probe.hip): L2 float atomics retire ~21 G cache-line requests/s no matter the table
// size, dtype (f32 = pk_f16 = f64) or scope, so the tcnn-style global scatter (2^D * L * F * N = 268 M requests at the
// metric shape) costs 13-20 ms.  The scatter is turned inside out instead (DESIGN.md section 4.1 has the full account):
//   * a level's table is cut into <= 64 contiguous slices that fit a CU's LDS as DOUBLE accumulators (T = 2^19, F = 2:
//     64 x 8192 entries x 16 B = 128 KiB).  gfx950's ds_add_f32 runs at 0.37 lane-ops/clk/CU, ds_add_f64 at 2.6
//     (tools/lds_probe.hip), so double is both faster and more accurate than the upstream fp32 atomics;
//   * the FORWARD kernel emits, per (level, slice), a bitmap with ONE bit per sample (bit-transposed across the wave,
//     4 words per row and workgroup through LDS): a slice owner reads 128 KiB of bitmap instead of touching every sample;
//   * a persistent workgroup (1024 threads, one per CU) OWNS one (level, slice[, sample range]) work item at a time.  Each
//     lane holds one 64-sample bitmap word; the hits of a wave's 64 words are compacted analytically (DPP scans, ranks
//     through a 4096-bit head vector, select of the k-th set bit) so that lane l of chunk c materialises
//     hit 64 c + l directly -- dense lanes, sample order, no queue, no barrier in the loop;
//   * the hits are reprocessed in GROUPS of two 64-hit chunks, software-pipelined over two register sets: the x / dout
//     gathers of group g + 1 (32-bit offsets from scalar bases) are in flight while group g is consumed -- recompute cell +
//     weights, then ds_add_f64 the corners that live in the slice.  Hashed power-of-two levels add one x-pair per hit (a
//     rare second pair goes through a small per-wave queue, drained between groups); dense coarse levels reduce runs of
//     equal cells with a segmented DPP scan first (one v_fmac_f32_dpp per value and step), so one lane per run touches
//     the LDS.  The stream is instantiated once per kind of level (dense / paired hashed / other hashed);
//   * at the end the slice is written with plain coalesced stores (every entry of a hashed level is owned by exactly one
//     workgroup: no global atomics, no memset of the 49 MB gradient); dense levels, whose sample stream is also cut in
//     ranges for balance, merge their non-zero entries into a pre-zeroed level with L2 atomics.
// Work items are consumed through per-XCD atomic cursors in ONE global order (longest items first, blocks of 32 dealt to
// the XCDs round-robin); a workgroup whose list is empty steals from the others.  Placement only affects speed.
// Determinism: the accumulation order inside a slice depends on wave scheduling, but sums are formed in double and rounded
// to fp32 once, so two runs agree to the last ulp or two (tests/test_a_metric_shape_gpu.py: <= 1e-6 relative).
#ifdef EMER_SLICED_TRACE
// Debug build only (tools/trace_sliced.py): per-work-item timeline of the owner-computes backward.
// trace[0] = item counter; record i at trace[8 + 4 i] = {level | slice << 8 | range << 24 | block << 40, start, end, hits}
__device__ unsigned long long *g_sliced_trace = nullptr;
#endif
constexpr int kSliceThreads = 1024;   // owner workgroup: 1024 lanes with a 128 KiB slice, one per CU (two 512-lane owners per CU: measured +9 %, DESIGN 4.1 [r4])
constexpr int kSliceWaves = kSliceThreads / 64;
#ifndef EMER_STRIDED_MAX_RES
#define EMER_STRIDED_MAX_RES 420
#endif
constexpr uint32_t kStridedHitsMaxRes = EMER_STRIDED_MAX_RES;          // one-feature hashed levels (no run reduction) up to this resolution spread a wave's hits over distant samples


// ---- DPP wave scans (gfx9 data-parallel primitives: a VALU operand modifier, no LDS round trip) -------------------
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes whose source lane does not exist, or whose row is
// masked off, keep `old`.  row_shr:n = 0x110 + n (inside a 16-lane row), row_bcast15 = 0x142 (lane 15 of each row to
// the next row), row_bcast31 = 0x143 (lane 31 to rows 2 and 3), wave_shr:1 = 0x138, wave_shl:1 = 0x130.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t src, uint32_t old = 0u) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, false));
}
// value of the previous / next lane of the wave (lane 0 / lane 63 get `edge`)
__device__ __forceinline__ uint32_t wave_prev_u32(uint32_t v, uint32_t edge) { return dpp_u32<0x138, 0xF>(v, edge); }
__device__ __forceinline__ uint32_t wave_next_u32(uint32_t v, uint32_t edge) { return dpp_u32<0x130, 0xF>(v, edge); }

// Segmented inclusive wave scan with shared segment heads: step masks first (one set per 64 lanes), then one
// multiply-add per value and step.  The scan operator on (head flag, value) pairs is
// (f1, v1) (+) (f2, v2) = (f1 | f2, f2 ? v2 : v1 + v2); keep[s] = 1.0 where the lane still ACCEPTS the partner's value
// at step s (no head seen so far between the partner and itself), else 0.0.
struct RunMasks {
    float keep[6];
    bool need[6];  // (wave-uniform) some lane still accepts a value at this step: runs longer than 2^s lanes exist
};
__device__ __forceinline__ RunMasks run_masks(bool head) {
    RunMasks m;
    uint32_t f = head ? 1u : 0u;
    m.keep[0] = f ? 0.0f : 1.0f; m.need[0] = __ballot(f == 0u) != 0ull; f |= dpp_u32<0x111, 0xF>(f);
    m.keep[1] = f ? 0.0f : 1.0f; m.need[1] = __ballot(f == 0u) != 0ull; f |= dpp_u32<0x112, 0xF>(f);
    m.keep[2] = f ? 0.0f : 1.0f; m.need[2] = __ballot(f == 0u) != 0ull; f |= dpp_u32<0x114, 0xF>(f);
    m.keep[3] = f ? 0.0f : 1.0f; m.need[3] = __ballot(f == 0u) != 0ull; f |= dpp_u32<0x118, 0xF>(f);
    m.keep[4] = f ? 0.0f : 1.0f; m.need[4] = __ballot(f == 0u) != 0ull; f |= dpp_u32<0x142, 0xA>(f);
    m.keep[5] = f ? 0.0f : 1.0f; m.need[5] = __ballot(f == 0u) != 0ull;
    return m;
}
// One scan step = ONE instruction per value: v_fmac_f32 with the DPP modifier on its first source, v += dpp(v) * keep.
// Lanes whose source lane does not exist, or whose row is masked off, are disabled by the modifier (bound_ctrl off) and
// keep v -- exactly "add nothing".  The compiler does not fold update_dpp into the multiply-add (it emitted a zero
// initialisation, a v_mov_b32_dpp and a v_fmac_f32 per step: 288 instead of 96 instructions for the 16 reductions of a
// chunk), so the step is written out.  A DPP source written by the previous VALU instruction needs two wait states, the
// assembler does not add them inside inline asm: every step starts with s_nop 1 and walks all values before the next.
#define EMER_DPP_STEP(CTRL, KEEP)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                                                      \
        if (i == 0) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " CTRL " bank_mask:0xf" : "+v"(v[i]) : "v"(KEEP)); \
        else asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL " bank_mask:0xf" : "+v"(v[i]) : "v"(KEEP));                  \
    }
template <int NV>
__device__ __forceinline__ void run_reduce_dpp(float (&v)[NV], const RunMasks &m) {
    // a step whose keep mask is zero on every lane adds nothing: with eight or more values per step (the pair sums of the
    // four-feature grids) it is skipped (wave-uniform test; short runs on the fine levels need one or two of the six steps:
    // xyzt grid 1424 -> 1388 us); with fewer values the test costs what it saves (main grid: +1 %)
    if (NV < 8 || m.need[0]) { EMER_DPP_STEP("row_shr:1 row_mask:0xf", m.keep[0]) }
    if (NV < 8 || m.need[1]) { EMER_DPP_STEP("row_shr:2 row_mask:0xf", m.keep[1]) }
    if (NV < 8 || m.need[2]) { EMER_DPP_STEP("row_shr:4 row_mask:0xf", m.keep[2]) }
    if (NV < 8 || m.need[3]) { EMER_DPP_STEP("row_shr:8 row_mask:0xf", m.keep[3]) }
    if (NV < 8 || m.need[4]) { EMER_DPP_STEP("row_bcast:15 row_mask:0xa", m.keep[4]) }
    if (NV < 8 || m.need[5]) { EMER_DPP_STEP("row_bcast:31 row_mask:0xc", m.keep[5]) }
}
#undef EMER_DPP_STEP

// Dense-level drain helper: the 64 queued samples of a wave are consecutive samples of a few rays, so
// they form RUNS that share one cell (and therefore all 2^D corner entries).  Values are reduced per run
// with a segmented wave scan; only the last lane of each run touches the LDS (one ds_add_f64 per run and
// corner instead of a same-address serialisation across the run).
template <int NV>
__device__ __forceinline__ void run_reduce(float (&v)[NV], int run_start, int lane) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const bool take = lane - off >= run_start;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float t = __shfl_up(v[i], off, kWave);
            if (take) v[i] += t;
        }
    }
}


// ---- analytic hit compaction (used by the owner-computes backward) ----------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_sum_u32(uint32_t v, int lane) {
    (void)lane;
    v += dpp_u32<0x111, 0xF>(v);
    v += dpp_u32<0x112, 0xF>(v);
    v += dpp_u32<0x114, 0xF>(v);
    v += dpp_u32<0x118, 0xF>(v);
    v += dpp_u32<0x142, 0xA>(v);
    v += dpp_u32<0x143, 0xC>(v);
    return v;
}
// position of the k-th (0-based) set bit of w; k < popcount(w)
__device__ __forceinline__ uint32_t select64(uint64_t w, uint32_t k) {
    uint32_t x = (uint32_t)w, base = 0, c = (uint32_t)__popc((uint32_t)w);
    if (k >= c) { k -= c; x = (uint32_t)(w >> 32); base = 32; }
    c = (uint32_t)__popc(x & 0xFFFFu); if (k >= c) { k -= c; x >>= 16; base += 16; }
    c = (uint32_t)__popc(x & 0xFFu);   if (k >= c) { k -= c; x >>= 8;  base += 8; }
    c = (uint32_t)__popc(x & 0xFu);    if (k >= c) { k -= c; x >>= 4;  base += 4; }
    c = (uint32_t)__popc(x & 0x3u);    if (k >= c) { k -= c; x >>= 2;  base += 2; }
    if (k >= (x & 1u)) base += 1;
    return base;
}
// the same with the last three levels replaced by one LDS byte look-up: lut[byte * 8 + k] = position of the k-th set bit
// of `byte` (2 KiB per workgroup, filled once): 21 VALU instead of 42 per 64 hits in the hottest loop of the backward
// 0: never, 1: always, 2: three-dimensional grids only.  History: round 2 measured the plain VALU select 1 % faster (the vector pipes
// were ~40 % busy then, the extra LDS round trip on the hit -> sample chain cost more than 21 instructions); since the run-reduced adds
// the kernel is vector-issue bound (SQ_ACTIVE_INST_VALU ~70 % of the SIMD cycles, profiles/r04a_grid_counters.json) and the look-up
// wins on the D = 3 grids -- [r5] same-session A/B: main grid 535 -> 516 us, default static 663 -> 655, proposal grids 252 -> 246 --
// while the D4 / F4 xyzt tables, whose LDS is the busier unit, lose 0.6 %.
#ifndef EMER_SELECT_LUT
#define EMER_SELECT_LUT 2
#endif
template <int D> constexpr bool use_select_lut() { return EMER_SELECT_LUT == 1 || (EMER_SELECT_LUT == 2 && D == 3); }
template <int D> constexpr uint32_t select_lut_bytes() { return use_select_lut<D>() ? 2048u : 0u; }
__device__ __forceinline__ uint32_t select64_lut(uint64_t w, uint32_t k, const uint8_t *lut) {
    uint32_t x = (uint32_t)w, base = 0, c = (uint32_t)__popc((uint32_t)w);
    if (k >= c) { k -= c; x = (uint32_t)(w >> 32); base = 32; }
    c = (uint32_t)__popc(x & 0xFFFFu); if (k >= c) { k -= c; x >>= 16; base += 16; }
    c = (uint32_t)__popc(x & 0xFFu);   if (k >= c) { k -= c; x >>= 8;  base += 8; }
    return base + lut[((x & 0xFFu) << 3) | (k & 7u)];
}

// ---- pair helpers of the owner-computes backward (hashed power-of-two levels) -------------------------------------
constexpr uint32_t kPairQueue = 72;        // ring capacity per wave (words): drain threshold - 1 + one chunk of 64
#ifndef EMER_QUEUE_DRAIN
#define EMER_QUEUE_DRAIN 8
#endif
constexpr uint32_t kPairQueueDrain = EMER_QUEUE_DRAIN;    // drain once this many second pairs wait (they are rare: ~1 per 1000 hits)
constexpr uint32_t kPairQueueShift = 24;   // entry = sample id << 8 | remaining pair mask: ids below 2^24, masks up to 8 bits (D <= 4)
// hash contributions of the non-x dimensions for both corner values: hd[d][b] = (gi[d] + b) * prime_d
template <int D>
__device__ __forceinline__ void hash_terms(const uint32_t (&gi)[D], uint32_t (&hd)[D][2]) {
    const uint32_t primes[4] = {1u, 2654435761u, 805459861u, 3674653429u};
    hd[0][0] = hd[0][1] = 0u;
#pragma unroll
    for (int d = 1; d < D; ++d) { hd[d][0] = gi[d] * primes[d]; hd[d][1] = hd[d][0] + primes[d]; }
}
// bit m set iff pair m (corner bits of dims 1..D-1) lives in the slice: the x index only touches bits below the slice
// bits, so the test is on the hash alone: ((h ^ slice_first) & slice_bits) == 0
template <int D>
__device__ __forceinline__ uint32_t pair_matches(const uint32_t (&hd)[D][2], uint32_t slice_want, uint32_t slice_bits) {
    uint32_t match = 0;
    const uint32_t y0 = hd[1][0] ^ slice_want, y1 = hd[1][1] ^ slice_want;
#pragma unroll
    for (uint32_t m = 0; m < (1u << (D - 1)); ++m) {
        uint32_t h = (m & 1u) ? y1 : y0;
#pragma unroll
        for (int d = 2; d < D; ++d) h ^= hd[d][(m >> (d - 1)) & 1u];
        if ((h & slice_bits) == 0u) match |= 1u << m;
    }
    return match;
}
// every lane adds its lowest matching pair (if any) and clears it from `match`
template <int D, int F>
__device__ __forceinline__ void add_pair(double *acc, const uint32_t (&gi)[D], const float (&w)[D], const uint32_t (&hd)[D][2],
                                         const float (&go)[F], uint32_t &match, uint32_t local_mask) {
    const bool has = match != 0u;
    const uint32_t m = has ? (uint32_t)__ffs((int)match) - 1u : 0u;
    match &= match - 1u;  // (0 stays 0)
    uint32_t h = 0;
    float wa = 1.0f - w[0], wb = w[0];  // same product order as the generic path: ((t0*t1)*t2)*t3
#pragma unroll
    for (int d = 1; d < D; ++d) {
        const bool bit = (m >> (d - 1)) & 1u;
        h ^= bit ? hd[d][1] : hd[d][0];
        const float t = bit ? w[d] : 1.0f - w[d];
        wa *= t; wb *= t;
    }
    const uint32_t l0 = (gi[0] ^ h) & local_mask, l1 = ((gi[0] + 1u) ^ h) & local_mask;  // offsets inside the slice
    if (F == 2) {
        // An entry is F doubles, so "feature f of a random entry" reaches only half of the 32 bank pairs; odd lanes
        // therefore add feature 1 in the first instruction and feature 0 in the second: each instruction's addresses cover
        // all bank pairs (-1.5 % on the main grid; nothing on the F = 4 grids, whose limit is the LDS atomic rate itself).
        const bool odd = (__lane_id() & 1u) != 0u;
        const float g0 = odd ? go[1 % F] : go[0], g1 = odd ? go[0] : go[1 % F];
        const uint32_t f0 = odd ? 1u : 0u, f1 = f0 ^ 1u;
        if (has) {
            atomicAdd(acc + (size_t)l0 * F + f0, (double)(wa * g0));  // ds_add_f64
            atomicAdd(acc + (size_t)l1 * F + f0, (double)(wb * g0));
            atomicAdd(acc + (size_t)l0 * F + f1, (double)(wa * g1));
            atomicAdd(acc + (size_t)l1 * F + f1, (double)(wb * g1));
        }
        return;
    }
    if (has) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            atomicAdd(acc + (size_t)l0 * F + f, (double)(wa * go[f]));  // ds_add_f64
            atomicAdd(acc + (size_t)l1 * F + f, (double)(wb * go[f]));
        }
    }
}
// Coarse hashed levels: consecutive samples of a ray share cells, so the hits of a slice come in RUNS of equal cells whose
// first matching pair -- and therefore the two LDS entries -- coincide.  Like the dense levels, the 2 F values of a run
// are summed with a segmented DPP scan and only the run's last lane touches the LDS: r-fold fewer ds_add_f64 and no
// same-address serialisation (round 1 / first half of round 2 spread such hits over distant lanes instead and still paid
// every add).  Lanes of a run have the same cell, hence the same match mask.
// Measured (same-session A/B, training distribution): main grid D3/F2 592 -> 532 us, xyzt D4/F4 1955 -> 1422 us, default static
// D3/F4 842 -> 661 us with EVERY paired hashed level on this path (a threshold at res 420 / 900 gives 552 / 545 and 1496 / 1437);
// the one-feature proposal grids lose (243 -> 251 at res 900, 280 on all levels: two adds per hit do not pay for the scan),
// so F = 1 keeps the plain path with strided hit order on its coarse levels.
#ifndef EMER_RUN_RES
#define EMER_RUN_RES 0x40000000   // hashed levels up to this resolution take the run-reduced path (0: off)
#endif
template <int D, int F>
__device__ __forceinline__ void add_pair_runs(double *acc, const uint32_t (&gi)[D], const float (&w)[D], const uint32_t (&hd)[D][2],
                                              const float (&go)[F], uint32_t &match, uint32_t local_mask, bool valid, int lane) {
    // run heads: a lane whose cell differs from its predecessor's (invalid lanes are runs of their own)
    uint32_t diff = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const uint32_t key = valid ? gi[d] : 0xFFFFFFFFu - (uint32_t)lane;
        diff |= key ^ wave_prev_u32(key, ~key);
    }
    const bool head = diff != 0u;  // (lane 0 compares with ~key: always a head)
    const RunMasks rm = run_masks(head);
    const bool next_head = wave_next_u32(head ? 1u : 0u, 1u) != 0u;
    const bool has = match != 0u;
    const uint32_t m = has ? (uint32_t)__ffs((int)match) - 1u : 0u;
    match &= match - 1u;
    uint32_t h = 0;
    float wa = 1.0f - w[0], wb = w[0];
#pragma unroll
    for (int d = 1; d < D; ++d) {
        const bool bit = (m >> (d - 1)) & 1u;
        h ^= bit ? hd[d][1] : hd[d][0];
        const float t = bit ? w[d] : 1.0f - w[d];
        wa *= t; wb *= t;
    }
    const uint32_t l0 = (gi[0] ^ h) & local_mask, l1 = ((gi[0] + 1u) ^ h) & local_mask;
    float v[2 * F];
#pragma unroll
    for (int f = 0; f < F; ++f) { v[f] = has ? wa * go[f] : 0.0f; v[F + f] = has ? wb * go[f] : 0.0f; }
    run_reduce_dpp<2 * F>(v, rm);
    if (has && (lane == 63 || next_head)) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            atomicAdd(acc + (size_t)l0 * F + f, (double)v[f]);  // ds_add_f64
            atomicAdd(acc + (size_t)l1 * F + f, (double)v[F + f]);
        }
    }
}
// Queued second pairs: lane l takes entry l, reloads the sample (cache hits: it was loaded a moment ago), and adds the
// queued pairs.
template <int D, int F>
__device__ __forceinline__ void drain_pair_queue(double *acc, const LevelInfo &li, const float *__restrict__ x, const float *__restrict__ dl,
                                                 int64_t sn, const uint32_t *Qw, uint32_t q_head, uint32_t count, uint32_t slice_want,
                                                 uint32_t slice_bits, uint32_t local_mask, int lane) {
    (void)slice_want; (void)slice_bits;
    const bool on = (uint32_t)lane < count;
    const uint32_t e = on ? Qw[(q_head + (uint32_t)lane) % kPairQueue] : 0u;
    uint32_t n = e >> 8;
    uint32_t match = e & 0xFFu;
    // hipcc 7.2 (gfx950) folded the former `id = e & 0xFFFFFF` INTO the x address as `e * 12` (mask dropped: memory
    // fault); the id is made opaque here so the address arithmetic cannot be rewritten across the unpacking
    asm volatile("" : "+v"(n));
    float xv[D], go[F], w[D];
    uint32_t gi[D], hd[D][2];
    load_x<D>(x, (int64_t)n, xv);
#pragma unroll
    for (int f = 0; f < F; ++f) go[f] = dl[(int64_t)n * sn + f];
    cell_of<D>(li, xv, gi, w);
    hash_terms<D>(gi, hd);
    add_pair<D, F>(acc, gi, w, hd, go, match, local_mask);
    while (__ballot(match != 0u)) add_pair<D, F>(acc, gi, w, hd, go, match, local_mask);
}

// chunks (of 64 hits) per register set of the software-pipelined drain: two sets are live, D + F + 1 registers per chunk
#ifndef EMER_PIPE_K
#define EMER_PIPE_K 2
#endif
template <int D, int F> constexpr int kPipeK() {
    constexpr int per = D + F + 1, k = 28 / per;  // register budget of one set
    return k > EMER_PIPE_K ? EMER_PIPE_K : (k < 1 ? 1 : k);
}
// one group of hits: sample ids, validity and the gathered x / dout values of up to K chunks of 64 hits
template <int D, int F, int K>
struct HitGroup {
    float xs[K][D], go[K][F];
    uint32_t ns[K];      // sample ids (the second-pair queue stores them)
    bool vld[K];
    uint32_t take;       // hits in the group (wave-uniform)
};
constexpr int kScanWords = 64 + 64 + 32 + 32;  // per-wave LDS scratch of the compaction, in u64: words, head bit-vector, offsets, head bases

template <int D, int F>
__global__ __launch_bounds__(kSliceThreads) void hashgrid_bwd_params_sliced_kernel(const emer_grid_desc g, const SlicePlan plan,
                                                                                   const float *__restrict__ x,
                                                                                   const float *__restrict__ dout, int64_t sn, int64_t sl,
                                                                                   const uint64_t *__restrict__ masks,
                                                                                   uint32_t *__restrict__ work_ctr,
                                                                                   float *__restrict__ grad, int64_t N,
                                                                                   uint32_t accumulate) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ uint32_t s_item;
    // Persistent workgroups with XCD-affine work lists.  Each XCD has a list of (level, slice, range) items -- whole
    // levels, so that a level's streamed inputs are fetched into ONE L2 -- consumed through an atomic cursor.  A
    // workgroup whose own list is exhausted steals from the other XCDs' lists, which absorbs whatever the static cost
    // model got wrong for the actual sample distribution.  Placement only affects speed, never results.
    const uint32_t my_xcd = blockIdx.x & 7u;
    // byte-select table behind everything else in the LDS: lut[b * 8 + k] = index of the k-th set bit of b (0 if none)
    uint8_t *sel_lut = nullptr;
    if constexpr (use_select_lut<D>()) {
    sel_lut = reinterpret_cast<uint8_t *>(reinterpret_cast<uint64_t *>(smem + (size_t)plan.max_local * F) + (size_t)kSliceWaves * kScanWords)
                       + (size_t)kSliceWaves * kPairQueue * sizeof(uint32_t);
    for (uint32_t e = threadIdx.x; e < 2048u; e += kSliceThreads) {
        uint32_t b = e >> 3, k = e & 7u, pos = 0;
        for (uint32_t i = 0; i < 8u; ++i)
            if ((b >> i) & 1u) { if (k == 0u) { pos = i; break; } --k; }
        sel_lut[e] = (uint8_t)pos;
    }
    __syncthreads();
    }
  for (;;) {
    if (threadIdx.x == 0) {
        uint32_t it = 0xFFFFFFFFu;
        for (uint32_t t = 0; t < 8u; ++t) {
            const uint32_t xx = (my_xcd + t) & 7u;
            if (plan.items_per_xcd[xx] == 0u) continue;
            const uint32_t jj = atomicAdd(work_ctr + xx, 1u);
            if (jj < plan.items_per_xcd[xx]) { it = (xx << 24) | jj; break; }
        }
        s_item = it;
    }
    __syncthreads();
    const uint32_t item = s_item;
    __syncthreads();  // s_item may be rewritten by thread 0 right after the item is finished
    if (item == 0xFFFFFFFFu) return;
    const uint32_t xcd = item >> 24;
    // local index on the XCD's list -> global item index (blocks of kSchedBlock dealt round-robin) -> (level, slice, range)
    uint32_t j = item & 0xFFFFFFu;
    j = ((((j >> plan.sched_shift) << 3) + xcd) << plan.sched_shift) + (j & ((1u << plan.sched_shift) - 1u));
    uint32_t level = 0, slice = 0, range = 0;
    for (uint32_t oi = 0; oi < g.n_levels; ++oi) {
        level = plan.order[oi];
        const uint32_t nb = plan.n_slices[level] * plan.n_ranges[level];
        if (j < nb) { slice = j % plan.n_slices[level]; range = j / plan.n_slices[level]; break; }
        j -= nb;
    }
    const LevelInfo li = level_info(g, level);
#ifdef EMER_SLICED_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    unsigned long long trace_hits = 0;
#endif
    const bool dense_rt = !li.hashed;
    // [r5] wide pairs: levels whose resolution reaches the slice width (the xyzt tables' res 4424 / 8192 against 4096-entry slices)
    // are pairable too -- the x-neighbours still share a slice unless x sits at the last position of a slice-wide block, a 1-in-4096
    // event handled like the out-of-range wrap (wave-uniform fallback to the per-corner path).  Before, those levels took the per-corner
    // path for EVERY hit: 680 us per work item against 450 us for the level below them (tools/trace_sliced.py).
    const bool pairable_rt = li.hashed && (li.size & (li.size - 1u)) == 0u && true;
    const bool run_reduced = pairable_rt && F >= 2 && li.res <= (uint32_t)EMER_RUN_RES;  // coarse hashed level: runs of equal cells are summed before the LDS
    const bool consecutive = dense_rt || run_reduced || li.res > kStridedHitsMaxRes;  // hit -> lane assignment, see the compaction below
    const uint32_t shift = plan.shift[level], n_ranges = plan.n_ranges[level];
    const uint32_t first = slice << shift;
    const uint32_t n_local = ((li.size - first) < (1u << shift)) ? (li.size - first) : (1u << shift);
    // sample range of this workgroup (whole stream for hashed levels), in 64-sample bitmap words
    const int64_t n_words = (N + 63) >> 6;
    const int64_t words_per_range = ceil_div_dev(n_words, (int64_t)n_ranges);
    const int64_t w_begin = (int64_t)range * words_per_range;
    const int64_t w_end = (w_begin + words_per_range < n_words) ? w_begin + words_per_range : n_words;

    double *acc = smem;                                                                // [max_local * F] (ds_add_f64)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // wave-private scratch of the hit compaction, behind the accumulators
    uint64_t *scratch = reinterpret_cast<uint64_t *>(smem + (size_t)plan.max_local * F) + (size_t)wave * kScanWords;
    uint64_t *Wl = scratch;
    unsigned long long *Hv = reinterpret_cast<unsigned long long *>(scratch + 64);  // (one type for plain and atomic accesses)
    uint32_t *El = reinterpret_cast<uint32_t *>(scratch + 128), *Hx = reinterpret_cast<uint32_t *>(scratch + 160);

    for (uint32_t i = threadIdx.x; i < n_local * F; i += kSliceThreads) acc[i] = 0.0;
    __syncthreads();
    // second-pair queue (pairable levels): wave-private ring of kPairQueue words behind the compaction scratch
    uint32_t *Qw = reinterpret_cast<uint32_t *>(reinterpret_cast<uint64_t *>(smem + (size_t)plan.max_local * F) + (size_t)kSliceWaves * kScanWords)
                   + (size_t)wave * kPairQueue;
    const bool use_queue = pairable_rt && N <= (1ll << kPairQueueShift) && (1u << (D - 1)) <= 8u;
    const uint32_t slice_bits = (li.size - 1u) & ~((1u << shift) - 1u), slice_want = first, local_mask = (1u << shift) - 1u;

    const float *__restrict__ dl = dout + (int64_t)level * sl;
    const uint64_t *__restrict__ bm = masks + ((int64_t)level * (64 * plan.mask_q) + (slice >> plan.gsub[level])) * n_words;  // 1 bit per sample
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    // Every lane holds one 64-sample word of the bitmap; a trip of the workgroup covers 1024 words and the next trip's
    // word is loaded while the current one is consumed.  The hits of a wave's 64 words are COMPACTED ANALYTICALLY --
    // no per-bit peeling, whose iteration count is the largest popcount in the wave (bursty bitmaps: a few words hold
    // most bits) at a few per cent lane utilisation, and no LDS queue:
    //   p_i = popcount(word_i); an exclusive wave scan gives every word its offset E_i in the wave's hit list;
    //   the non-empty words are packed to the front of an LDS array (ballot prefix), and a 4096-bit "head" vector gets
    //   bit E_i set for each of them.  Hit number j then belongs to packed word rank(j) = (#head bits at positions <= j)
    //   - 1, a popcount over one head word plus a scanned base, and is the (j - E)-th set bit of that word (select64).
    // So lane l of chunk c materialises hit 64 c + l directly: 64 hits per ~45 instructions at full lane utilisation,
    // already in sample order (which the dense levels' run reduction needs).
    // Word of a trip held by this lane: hashed = thread id; dense = interleaved over the waves so that short ranges
    // still occupy all 16 waves (lane t of wave w holds word t * 16 + w; words of a wave stay in increasing order).
    //
    // The whole hit stream of the item is instantiated once per KIND of level (0 dense, 1 hashed with x-pairs that share a
    // slice, 2 any other hashed level): one consume path per copy keeps the software-pipelined loop below small enough for
    // the register allocator to leave the in-flight gathers alone (with all three paths in one loop body it split their
    // live ranges with copies placed right behind the loads, i.e. it waited for them at once).
    auto run_item = [&](auto kind_c) __attribute__((always_inline)) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr bool dense = KIND == 0, pairable = KIND == 1;
    uint32_t q_head = 0, q_len = 0;
    const int64_t my_word = dense ? (int64_t)lane * kSliceWaves + wave : (int64_t)threadIdx.x;
    const int64_t wave_word0 = dense ? wave : wave * 64;
    const uint32_t lane_word_shift = dense ? (kSliceWaves == 16 ? 4u : 3u) : 0u;  // log2 of the word stride between neighbouring lanes (= log2 kSliceWaves)
    static_assert(kSliceWaves == 16 || kSliceWaves == 8, "lane_word_shift assumes 16 or 8 waves");
    // ---- the hit stream of this item, as GROUPS of up to KG chunks of 64 hits --------------------------------------
    // next_trip() compacts the next non-empty 1024-word trip into the wave's scratch (hv / hexcl / total / n_chunks);
    // issue(G) materialises the next KG chunks of the current trip -- sample ids, then the x / dout gathers, left IN
    // FLIGHT in G's registers -- and consume(G) does the arithmetic and the LDS adds.  The two are
    // software-pipelined over two register sets: the gathers of group g + 1 (and the compaction of its trip) are issued
    // before group g is consumed, so a wave overlaps its own gather latency instead of relying on the three other
    // waves of its SIMD (the slice fills the LDS: one 1024-thread workgroup per CU, four waves per SIMD).
    constexpr int KG = kPipeK<D, F>();
    int64_t wbase = w_begin;          // next trip to compact
    uint32_t trip_w0 = 0;             // first word (32-bit) of the trip held in the scratch
    uint32_t total = 0, n_chunks = 0, c0 = 0, hexcl = 0;
    uint64_t hv = 0;
    uint64_t pre = (w_begin + my_word < w_end) ? bm[w_begin + my_word] : 0ull;
    auto next_trip = [&]() __attribute__((always_inline)) -> bool {
        while (wbase < w_end) {
            const uint64_t wv = pre;
            trip_w0 = (uint32_t)wbase;
            wbase += kSliceThreads;
            pre = (wbase + my_word < w_end) ? bm[wbase + my_word] : 0ull;
            const uint32_t p = (uint32_t)__popcll(wv);
            const uint32_t incl = wave_inclusive_sum_u32(p, lane);
            total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);  // (an SGPR: everything derived from it -- chunk counts, group sizes -- stays scalar)
            if (total == 0u) continue;
#ifdef EMER_SLICED_TRACE
            if (lane == 0) trace_hits += total;
#endif
            const uint32_t excl = incl - p;
            const unsigned long long nzm = __ballot(p != 0u);
            Hv[lane] = 0ull;
            if (p != 0u) {
                const uint32_t rk = (uint32_t)__popcll(nzm & lt_mask);
                Wl[rk] = wv;
                El[rk] = excl | ((uint32_t)lane << 16);
                atomicOr(Hv + (excl >> 6), 1ull << (excl & 63u));  // ds_or_b64
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the packed words / head bits of all lanes are visible
            hv = Hv[lane];
            const uint32_t hp = (uint32_t)__popcll(hv);
            hexcl = wave_inclusive_sum_u32(hp, lane) - hp;  // head bits before this lane's head word
            if (!consecutive) { Hx[lane] = hexcl; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
            n_chunks = (total + 63u) >> 6;
            c0 = 0;
            return true;
        }
        return false;
    };
    auto issue = [&](HitGroup<D, F, KG> &G) __attribute__((always_inline)) -> bool {
        bool have = true;
        if (c0 >= n_chunks) {  // wave-uniform
            have = next_trip();
            if (!have) { total = 0u; n_chunks = 0u; c0 = 0u; }  // stream exhausted: the group below is empty (dummy gathers of sample 0)
        }
        // (the gathers below are issued unconditionally, in straight-line code: a group that is merged with another
        // definition of its registers makes the compiler copy the loaded values -- and wait for them -- right here)
        G.take = (total - 64u * c0) < 64u * KG ? (total - 64u * c0) : 64u * KG;
        const uint32_t last_hit = total ? total - 1u : 0u, last_chunk = n_chunks ? n_chunks - 1u : 0u;
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            uint32_t c = c0 + (uint32_t)k;
            const bool live = c < n_chunks;                                          // wave-uniform
            c = live ? c : last_chunk;
            // Lane l takes hit 64 c + l (sample order) wherever runs of equal cells are reduced before the LDS (dense levels,
            // paired hashed levels with F >= 2) and on FINE levels, where neighbouring hits share x / dout cache lines.  On the
            // COARSE levels of one-feature grids (plain adds) consecutive samples of a ray share cells and would serialise
            // on the same LDS address: there lane l takes hit l * n_chunks + c, so the lanes of one instruction work on hits
            // far apart (different rays).
            uint32_t n = 0u;               // chunks past the end of the trip fetch sample 0 (cached) and are never consumed
            bool in_range = false;
            if (live) {                     // wave-uniform: only the sample id is merged (one register), the gathers below are unconditional
            uint32_t j = consecutive ? 64u * c + (uint32_t)lane : __umul24((uint32_t)lane, n_chunks) + c;  // (n_chunks <= 64: full-rate multiply)
            in_range = j < total;
            j = in_range ? j : last_hit;                                             // out-of-range lanes repeat the last hit (masked below)
            uint64_t hr;
            uint32_t hx;
            if (consecutive) {
                const uint32_t hr_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hv, (int)c);
                const uint32_t hr_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hv >> 32), (int)c);
                hx = (uint32_t)__builtin_amdgcn_readlane((int)hexcl, (int)c);
                hr = ((uint64_t)hr_hi << 32) | hr_lo;
            } else {
                hr = Hv[j >> 6];
                hx = Hx[j >> 6];
            }
            const uint64_t upto = ((j & 63u) == 63u) ? ~0ull : ((2ull << (j & 63u)) - 1ull);
            const uint32_t rank = hx + (uint32_t)__popcll(hr & upto) - 1u;
            const uint64_t wq = Wl[rank];
            const uint32_t el = El[rank];
            uint32_t bit;
            if constexpr (use_select_lut<D>()) bit = select64_lut(wq, j - (el & 0xFFFFu), sel_lut);
            else bit = select64(wq, j - (el & 0xFFFFu));
            n = ((trip_w0 + (uint32_t)wave_word0 + ((el >> 16) << lane_word_shift)) << 6) + bit;  // (32-bit: n < 2^28)
            }
            G.vld[k] = in_range && live;
            G.ns[k] = n;
            // 32-bit byte offsets from the (uniform) bases: the host checked N * 16 < 2^32, so the gathers use the
            // scalar-base + 32-bit-offset addressing mode instead of 64-bit multiply-adds per lane (sn == F)
            // (n * 12 as shifts: v_mul_lo_u32 is a quarter-rate instruction)
            uint32_t xoff;
            if (D == 3) {
                // n * 12 as shift-adds: the compiler folds (n << 3) + (n << 2) back into v_mul_lo_u32, a quarter-rate instruction
                uint32_t n4 = n << 2;
                asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(xoff) : "v"(n), "v"(n4));
            } else {
                xoff = n * (uint32_t)(D * 4);
            }
            const float *xp = reinterpret_cast<const float *>(reinterpret_cast<const char *>(x) + xoff);
            const float *gp = reinterpret_cast<const float *>(reinterpret_cast<const char *>(dl) + (uint32_t)(n * (uint32_t)(F * 4)));
            load_x<D>(xp, 0, G.xs[k]);
            if (F == 2) { float2 t = *reinterpret_cast<const float2 *>(gp); G.go[k][0] = t.x; G.go[k][1 < F ? 1 : 0] = t.y; }
            else if (F == 4) { float4 t = *reinterpret_cast<const float4 *>(gp); G.go[k][0] = t.x; G.go[k][1 < F ? 1 : 0] = t.y; G.go[k][2 < F ? 2 : 0] = t.z; G.go[k][3 < F ? 3 : 0] = t.w; }
            else {
#pragma unroll
                for (int f = 0; f < F; ++f) G.go[k][f] = gp[f];
            }
        }
        c0 += (uint32_t)KG;
        return have;
    };
    auto consume = [&](HitGroup<D, F, KG> &G) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            if ((uint32_t)(k * 64) >= G.take) break;  // wave-uniform
            const bool valid = G.vld[k];
            float w[D];
            uint32_t gi[D];
            cell_of<D>(li, G.xs[k], gi, w);
            if (!valid) {
#pragma unroll
                for (int f = 0; f < F; ++f) G.go[k][f] = 0.0f;
            }
            if constexpr (dense) {
                // ---- run-segmented reduction: lanes with the same cell as their predecessor join its run
                uint32_t cell = 0, mul = 1;
#pragma unroll
                for (int d = 0; d < D; ++d) { cell += gi[d] * mul; mul *= li.res + 1u; }
                if (!valid) cell = 0xFFFFFFFFu;
                const uint32_t prev = wave_prev_u32(cell, ~cell);
                const bool head = cell != prev;  // (lane 0 compares with ~cell: always a head)
                const RunMasks rm = run_masks(head);
                const bool next_head = wave_next_u32(head ? 1u : 0u, 1u) != 0u;
                const bool tail = valid && (lane == 63 || next_head);
                // dense index of corner m = index of the cell + a level-uniform offset (same arithmetic mod 2^32 as
                // grid_index: the strides are scalars, so the per-corner multiplies collapse to one add)
                uint32_t stride_d[D], cell_idx = 0;
                {
                    uint32_t st = 1;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        stride_d[d] = (st <= li.size) ? st : 0u;
                        if (st <= li.size) st *= li.res;
                        cell_idx += gi[d] * stride_d[d];
                    }
                }
#pragma unroll
                for (uint32_t m = 0; m < (1u << D); ++m) {
                    float wt = 1.0f;
                    uint32_t off_m = 0;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if (m & (1u << d)) off_m += stride_d[d];
                        wt *= (m & (1u << d)) ? w[d] : 1.0f - w[d];
                    }
                    float v[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) v[f] = wt * G.go[k][f];
                    run_reduce_dpp<F>(v, rm);
                    uint32_t idx = cell_idx + off_m;
                    if ((li.size & (li.size - 1u)) == 0u) idx &= li.size - 1u;
                    else if (idx >= li.size) { idx -= li.size; if (idx >= li.size) idx %= li.size; }
                    if (tail && (idx >> shift) == slice) {
#pragma unroll
                        for (int f = 0; f < F; ++f) atomicAdd(acc + (size_t)(idx - first) * F + f, (double)v[f]);
                    }
                }
            } else if (pairable && !__ballot(valid && (gi[0] >= li.res || (gi[0] & local_mask) == local_mask))) {
                // hashed power-of-two level whose resolution is below the slice width: the two x-corners of a
                // (y, z[, t]) combination differ only in index bits BELOW the slice bits, so they always share a
                // slice -- for cells inside the grid (gi[0] + 1 <= res < slice width).  Inputs outside [0, 1] wrap
                // (tcnn semantics) and may put the two x-corners in different slices: a wave holding such a hit
                // takes the generic per-corner path below (wave-uniform test, never taken by EmerNeRF's own inputs).
                // A hit has ONE pair in this slice, now and then a second one (the 2^(D-1) pairs of a sample fall in
                // ~independent slices): every lane adds its first matching pair here on dense lanes; the rare further
                // pairs are queued (sample id + remaining pair mask) and drained a few at a time, again on dense
                // lanes, instead of running a second, 95 % masked, pair body after every chunk.
                uint32_t hd[D][2];
                hash_terms<D>(gi, hd);
                // (the x index reaches into the slice bits when the resolution exceeds the slice width: it joins the test)
                uint32_t match = valid ? pair_matches<D>(hd, slice_want ^ (gi[0] & slice_bits), slice_bits) : 0u;
                if (run_reduced) add_pair_runs<D, F>(acc, gi, w, hd, G.go[k], match, local_mask, valid, lane);  // (level-uniform)
                else add_pair<D, F>(acc, gi, w, hd, G.go[k], match, local_mask);
                if (use_queue) {
                    const bool more = match != 0u;
                    const unsigned long long mb = __ballot(more);
                    if (mb) {  // wave-uniform
                        const uint32_t n_more = (uint32_t)__popcll(mb);
                        if (q_len + n_more <= kPairQueue) {
                            if (more) Qw[(q_head + q_len + (uint32_t)__popcll(mb & lt_mask)) % kPairQueue] = (G.ns[k] << 8) | match;
                            q_len += n_more;
                        } else {
                            // ring full (the queue is drained between groups, see below): add the further pairs right here,
                            // on masked lanes -- no gathers inside the chunk loop, so the compiler's wait counts for the
                            // in-flight gathers of the following chunks stay exact
                            while (__ballot(match != 0u)) add_pair<D, F>(acc, gi, w, hd, G.go[k], match, local_mask);
                        }
                    }
                } else {
                    add_pair<D, F>(acc, gi, w, hd, G.go[k], match, local_mask);
                    while (__ballot(match != 0u)) add_pair<D, F>(acc, gi, w, hd, G.go[k], match, local_mask);
                }
            } else {
#pragma unroll
                for (uint32_t m = 0; m < (1u << D); ++m) {
                    uint32_t c[D];
                    float wt = 1.0f;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        c[d] = gi[d] + ((m >> d) & 1u);
                        wt *= (m & (1u << d)) ? w[d] : 1.0f - w[d];
                    }
                    const uint32_t idx = grid_index<D>(li, c);
                    if (valid && (idx >> shift) == slice) {
#pragma unroll
                        for (int f = 0; f < F; ++f) atomicAdd(acc + (size_t)(idx - first) * F + f, (double)(wt * G.go[k][f]));  // ds_add_f64
                    }
                }
            }
        }
        if constexpr (pairable) {
            // queued second pairs are drained BETWEEN groups (its gathers would otherwise sit inside the chunk loop and make
            // the wait counts of the pipelined gathers conservative: vmcnt(0) after every chunk)
            if (q_len >= kPairQueueDrain) {  // wave-uniform
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint32_t done = q_len < 64u ? q_len : 64u;
                drain_pair_queue<D, F>(acc, li, x, dl, sn, Qw, q_head, done, slice_want, slice_bits, local_mask, lane);
                q_head = (q_head + done) % kPairQueue; q_len -= done;
            }
        }
    };
    {
        HitGroup<D, F, KG> ga;
        HitGroup<D, F, KG> gb;
        bool more = issue(ga);
        while (more) {
            const bool more_b = issue(gb);   // gathers of the next group in flight ...
            consume(ga);                     // ... while this one is consumed
            if (!more_b) break;
            more = issue(ga);
            consume(gb);
        }
    }
    while (q_len) {  // wave-uniform: second pairs still queued
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t done = q_len < 64u ? q_len : 64u;
        drain_pair_queue<D, F>(acc, li, x, dl, sn, Qw, q_head, done, slice_want, slice_bits, local_mask, lane);
        q_head = (q_head + done) % kPairQueue; q_len -= done;
    }
    };  // run_item
    if (dense_rt) run_item(std::integral_constant<int, 0>{});
    else if (pairable_rt) run_item(std::integral_constant<int, 1>{});
    else run_item(std::integral_constant<int, 2>{});
    __syncthreads();
    // write the slice.  One range: every entry is owned by exactly this workgroup -> plain coalesced stores.
    // Several ranges (dense levels): the host zeroed the level; merge the non-zero entries with L2 atomics.
    float *__restrict__ out = grad + ((size_t)li.offset + first) * F;
    if (n_ranges == 1u) {
        // (accumulate: a further evaluation of the same table in this step ADDS to what the first one wrote -- every entry still has one
        // owner, so a plain read-modify-write)
        if (accumulate) { for (uint32_t i = threadIdx.x; i < n_local * F; i += kSliceThreads) out[i] += (float)acc[i]; }
        else { for (uint32_t i = threadIdx.x; i < n_local * F; i += kSliceThreads) out[i] = (float)acc[i]; }
    } else {
        for (uint32_t i = threadIdx.x; i < n_local * F; i += kSliceThreads) {
            const float v = (float)acc[i];
            if (v != 0.0f) __hip_atomic_fetch_add(out + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();  // the accumulators are re-zeroed by the next item
#ifdef EMER_SLICED_TRACE
    if (g_sliced_trace && threadIdx.x == 0) {
        const unsigned long long slot = atomicAdd(g_sliced_trace, 1ull);
        unsigned long long *rec = g_sliced_trace + 8 + 4 * slot;
        rec[0] = (unsigned long long)level | ((unsigned long long)slice << 8) | ((unsigned long long)range << 24) | ((unsigned long long)blockIdx.x << 40);
        rec[1] = trace_t0; rec[2] = wall_clock64(); rec[3] = trace_hits;  // (hits of wave 0 only: 1/16 of the item's)
    }
#endif
  }
}

struct ZeroRegions {
    float *p[EMER_MAX_LEVELS + 1];
    uint32_t n[EMER_MAX_LEVELS + 1];
    int32_t count;
};
__global__ __launch_bounds__(256) void zero_regions_kernel(const ZeroRegions z) {
    float *__restrict__ p = z.p[blockIdx.y];
    const uint32_t n = z.n[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) p[i] = 0.0f;
}

// Slice bitmaps for callers that did not get them from the forward pass.
template <int D, int Q>
__global__ __launch_bounds__(256) void hashgrid_slice_masks_kernel(const emer_grid_desc g, const SlicePlan plan,
                                                                   const float *__restrict__ x, uint64_t *__restrict__ masks,
                                                                   int64_t N, uint32_t n_chunks, const LevelMap lmap) {
    uint32_t level, chunk;
    if (!map_block(blockIdx.x, lmap, n_chunks, level, chunk)) return;
    const int64_t n = (int64_t)chunk * 256 + threadIdx.x;
    const LevelInfo li = level_info(g, level);
    uint64_t mask[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) mask[q] = 0ull;
    if (n < N) {
        float xv[D], w[D];
        uint32_t gi[D];
        load_x<D>(x, n, xv);
        cell_of<D>(li, xv, gi, w);
#pragma unroll
        for (uint32_t m = 0; m < (1u << D); ++m) {
            uint32_t c[D];
#pragma unroll
            for (int d = 0; d < D; ++d) c[d] = gi[d] + ((m >> d) & 1u);
            set_row<Q>(mask, slice_of(plan, level, grid_index<D>(li, c)));
        }
    }
    store_slice_bitmaps<Q>(masks, mask, level, bitmap_rows(plan, level), (int64_t)chunk * 256, (N + 63) >> 6, (int)threadIdx.x);
}

// ------------------------------------------------------------------------- backward (input)
// One thread per sample, all levels: deterministic, no atomics.  Only the flow configs reach this.
template <int D, int F, typename PT>
__global__ __launch_bounds__(256) void hashgrid_bwd_input_kernel(const emer_grid_desc g, const float *__restrict__ x,
                                                                 const PT *__restrict__ params,
                                                                 const float *__restrict__ dout, int64_t sn, int64_t sl,
                                                                 float *__restrict__ dx, int64_t N) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float xv[D], gx[D];
    load_x<D>(x, n, xv);
#pragma unroll
    for (int d = 0; d < D; ++d) gx[d] = 0.0f;
    for (uint32_t l = 0; l < g.n_levels; ++l) {
        const LevelInfo li = level_info(g, l);
        const PT *__restrict__ table = params + (size_t)li.offset * F;
        float w[D], go[F];
        uint32_t gi[D];
        cell_of<D>(li, xv, gi, w);
        const float *gp = dout + n * sn + (int64_t)l * sl;
#pragma unroll
        for (int f = 0; f < F; ++f) go[f] = gp[f];
        // gather all 2^D corners once, projected on dOut: s[m] = sum_f dOut_f * table[corner m][f]
        float s[1 << D];
#pragma unroll
        for (uint32_t m = 0; m < (1u << D); ++m) {
            uint32_t c[D];
#pragma unroll
            for (int d = 0; d < D; ++d) c[d] = gi[d] + ((m >> d) & 1u);
            float v[F];
            load_feats<F, PT>(table + (size_t)grid_index<D>(li, c) * F, v);
            float a = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) a += go[f] * v[f];
            s[m] = a;
        }
#pragma unroll
        for (int gd = 0; gd < D; ++gd) {
            float acc = 0.0f;
#pragma unroll
            for (uint32_t m = 0; m < (1u << D); ++m) {
                if (m & (1u << gd)) continue;  // enumerate corners with bit gd == 0
                float wt = li.scale;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (d == gd) continue;
                    wt *= (m & (1u << d)) ? w[d] : 1.0f - w[d];
                }
                acc += wt * (s[m | (1u << gd)] - s[m]);
            }
            gx[gd] += acc;
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) dx[n * D + d] = gx[d];
}

// The same gradient from the Jacobians the JAC forward stored: dx[row0 + j][d] = sum_l sum_f dOut[l][row0 + j][f] * J[l][j][f][d].
// Streaming (F D + F floats per sample and level, coalesced), no table gathers; levels summed in the gather kernel's order.
template <int D, int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_input_jac_kernel(const float *__restrict__ jac, const float *__restrict__ dout, int64_t sn,
                                                                     int64_t sl, float *__restrict__ dx, int64_t n_rows, uint32_t n_levels) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_rows) return;
    float gx[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gx[d] = 0.0f;
    for (uint32_t l = 0; l < n_levels; ++l) {
        const float *jp = jac + ((int64_t)l * n_rows + j) * (F * D);
        const float *gp = dout + j * sn + (int64_t)l * sl;
        float J[F * D], go[F];
        if constexpr ((F * D) % 4 == 0) {
#pragma unroll
            for (int i = 0; i < F * D; i += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(jp + i);
                J[i] = t.x; J[i + 1] = t.y; J[i + 2] = t.z; J[i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < F * D; ++i) J[i] = jp[i];
        }
#pragma unroll
        for (int f = 0; f < F; ++f) go[f] = gp[f];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float a = 0.0f;
#pragma unroll
            for (int f = 0; f < F; ++f) a += go[f] * J[f * D + d];
            gx[d] += a;
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) dx[j * D + d] = gx[d];
}

// ------------------------------------------------------------------------------- dispatch
template <typename Fn>
static int dispatch_df(uint32_t D, uint32_t F, Fn &&fn) {
#define EMER_CASE(d, f) if (D == d && F == f) return fn(std::integral_constant<int, d>{}, std::integral_constant<int, f>{});
    EMER_CASE(2, 1) EMER_CASE(2, 2) EMER_CASE(2, 4) EMER_CASE(2, 8)
    EMER_CASE(3, 1) EMER_CASE(3, 2) EMER_CASE(3, 4) EMER_CASE(3, 8)
    EMER_CASE(4, 1) EMER_CASE(4, 2) EMER_CASE(4, 4) EMER_CASE(4, 8)
#undef EMER_CASE
    set_error("hashgrid: unsupported (n_dims=%u, n_features=%u); need D in 2..4, F in {1,2,4,8}", D, F);
    return EMER_E_INVALID;
}

static int check_desc(const emer_grid_desc *g) {
    EMER_REQUIRE(g != nullptr, "hashgrid: null descriptor");
    EMER_REQUIRE(g->n_levels >= 1 && g->n_levels <= EMER_MAX_LEVELS, "hashgrid: n_levels=%u out of range", g->n_levels);
    EMER_REQUIRE(g->n_entries > 0, "hashgrid: descriptor not initialised (call emer_grid_desc_init)");
    return EMER_OK;
}

}  // namespace emer

using namespace emer;

extern "C" int emer_hashgrid_fwd(const emer_grid_desc *g, const float *x, const void *params, int param_dtype,
                                 float *out, int64_t sn, int64_t sl, uint64_t *slice_masks, int64_t n, void *stream) {
    if (int rc = check_desc(g)) return rc;
    EMER_REQUIRE(n >= 0, "hashgrid_fwd: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x && params && out, "hashgrid_fwd: null pointer");
    EMER_REQUIRE(param_dtype == EMER_F32 || param_dtype == EMER_F16, "hashgrid_fwd: bad param_dtype %d", param_dtype);
    const uint32_t n_chunks = (uint32_t)ceil_div(n, 256);
    uint32_t blocks = 0;
    const LevelMap lmap = make_level_map(g, n_chunks, &blocks);
    const SlicePlan plan = make_slice_plan(g);
    EMER_REQUIRE(!slice_masks || plan.ok, "hashgrid_fwd: slice bitmaps requested but a level needs more than 256 x 64 LDS slices");
    const ProfileEvents ev = take_profile_events();  // (null unless emer_profile_next armed them)
    return dispatch_df(g->n_dims, g->n_features, [&](auto d, auto f) {
        constexpr int D = decltype(d)::value, F = decltype(f)::value;
        const bool wide = slice_masks && plan.mask_q == 4;  // 256 bitmap rows per level
        if (param_dtype == EMER_F32) {
            if (wide)
                EMER_LAUNCH_PROFILED(ev, (hashgrid_fwd_kernel<D, F, float, 4>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, x,
                                     (const float *)params, out, sn, sl, n, n_chunks, lmap, plan, slice_masks, (float *)nullptr, (int64_t)0);
            else
                EMER_LAUNCH_PROFILED(ev, (hashgrid_fwd_kernel<D, F, float, 1>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, x,
                                     (const float *)params, out, sn, sl, n, n_chunks, lmap, plan, slice_masks, (float *)nullptr, (int64_t)0);
        } else {
            if (wide)
                EMER_LAUNCH_PROFILED(ev, (hashgrid_fwd_kernel<D, F, __half, 4>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, x,
                                     (const __half *)params, out, sn, sl, n, n_chunks, lmap, plan, slice_masks, (float *)nullptr, (int64_t)0);
            else
                EMER_LAUNCH_PROFILED(ev, (hashgrid_fwd_kernel<D, F, __half, 1>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, x,
                                     (const __half *)params, out, sn, sl, n, n_chunks, lmap, plan, slice_masks, (float *)nullptr, (int64_t)0);
        }
        return check_launch("hashgrid_fwd");
    });
}

// emer_hashgrid_fwd (fp32 tables) that also stores, for the rows jac_row0 .. n - 1, the Jacobian of the encoding with respect to the
// position: jac [n_levels][n - jac_row0][n_features][n_dims].  emer_hashgrid_bwd_input_jac contracts it with dOut.
extern "C" int emer_hashgrid_fwd_jac(const emer_grid_desc *g, const float *x, const float *params, float *out, int64_t sn, int64_t sl,
                                     uint64_t *slice_masks, float *jac, int64_t jac_row0, int64_t n, void *stream) {
    if (int rc = check_desc(g)) return rc;
    EMER_REQUIRE(n >= 0 && jac_row0 >= 0 && jac_row0 <= n, "hashgrid_fwd_jac: bad row range (n=%lld, jac_row0=%lld)", (long long)n, (long long)jac_row0);
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x && params && out && (jac || jac_row0 == n), "hashgrid_fwd_jac: null pointer");
    EMER_REQUIRE(((uintptr_t)jac % 16) == 0, "hashgrid_fwd_jac: jac must be 16-byte aligned");
    const uint32_t n_chunks = (uint32_t)ceil_div(n, 256);
    uint32_t blocks = 0;
    const LevelMap lmap = make_level_map(g, n_chunks, &blocks);
    const SlicePlan plan = make_slice_plan(g);
    EMER_REQUIRE(!slice_masks || plan.ok, "hashgrid_fwd_jac: slice bitmaps requested but a level needs more than 256 x 64 LDS slices");
    const ProfileEvents ev = take_profile_events();
    return dispatch_df(g->n_dims, g->n_features, [&](auto d, auto f) {
        constexpr int D = decltype(d)::value, F = decltype(f)::value;
        if (slice_masks && plan.mask_q == 4)
            EMER_LAUNCH_PROFILED(ev, (hashgrid_fwd_kernel<D, F, float, 4, true>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, x, params, out,
                                 sn, sl, n, n_chunks, lmap, plan, slice_masks, jac, jac_row0);
        else
            EMER_LAUNCH_PROFILED(ev, (hashgrid_fwd_kernel<D, F, float, 1, true>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, x, params, out,
                                 sn, sl, n, n_chunks, lmap, plan, slice_masks, jac, jac_row0);
        return check_launch("hashgrid_fwd_jac");
    });
}

// dx [n_rows][n_dims] from jac [n_levels][n_rows][n_features][n_dims] and dout (row j of dout at dout + j * sn + level * sl: pass the
// pointer of the first of the n_rows rows)
extern "C" int emer_hashgrid_bwd_input_jac(const emer_grid_desc *g, const float *jac, const float *dout, int64_t sn, int64_t sl, float *dx,
                                           int64_t n_rows, void *stream) {
    if (int rc = check_desc(g)) return rc;
    EMER_REQUIRE(n_rows >= 0, "hashgrid_bwd_input_jac: negative n_rows");
    if (n_rows == 0) return EMER_OK;
    EMER_REQUIRE(jac && dout && dx && ((uintptr_t)jac % 16) == 0, "hashgrid_bwd_input_jac: null or misaligned pointer");
    const uint32_t blocks = (uint32_t)ceil_div(n_rows, 256);
    return dispatch_df(g->n_dims, g->n_features, [&](auto d, auto f) {
        constexpr int D = decltype(d)::value, F = decltype(f)::value;
        hipLaunchKernelGGL((hashgrid_bwd_input_jac_kernel<D, F>), dim3(blocks), dim3(256), 0, as_stream(stream), jac, dout, sn, sl, dx, n_rows,
                           g->n_levels);
        return check_launch("hashgrid_bwd_input_jac");
    });
}

extern "C" int emer_hashgrid_bwd_params(const emer_grid_desc *g, const float *x, const float *dout, int64_t sn,
                                        int64_t sl, void *grad, int grad_dtype, int64_t n, void *stream) {
    if (int rc = check_desc(g)) return rc;
    EMER_REQUIRE(n >= 0, "hashgrid_bwd_params: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x && dout && grad, "hashgrid_bwd_params: null pointer");
    EMER_REQUIRE(grad_dtype == EMER_F32 || grad_dtype == EMER_F16, "hashgrid_bwd_params: bad grad_dtype %d", grad_dtype);
    EMER_REQUIRE(!(grad_dtype == EMER_F16 && (g->n_features & 1u)), "hashgrid_bwd_params: fp16 gradients need an even n_features");
    const uint32_t n_chunks = (uint32_t)ceil_div(n, 256);
    uint32_t blocks = 0;
    const LevelMap lmap = make_level_map(g, n_chunks, &blocks);
    return dispatch_df(g->n_dims, g->n_features, [&](auto d, auto f) {
        constexpr int D = decltype(d)::value, F = decltype(f)::value;
        if (grad_dtype == EMER_F32) {
            hipLaunchKernelGGL((hashgrid_bwd_params_kernel<D, F, float>), dim3(blocks), dim3(256), 0, as_stream(stream), *g,
                               x, dout, sn, sl, (float *)grad, n, n_chunks, lmap);
        } else {
            if constexpr (F % 2 == 0)
                hipLaunchKernelGGL((hashgrid_bwd_params_kernel<D, F, __half>), dim3(blocks), dim3(256), 0, as_stream(stream),
                                   *g, x, dout, sn, sl, (__half *)grad, n, n_chunks, lmap);
        }
        return check_launch("hashgrid_bwd_params");
    });
}


// 1 if the owner-computes backward supports this grid (every level fits <= 64 LDS slices), else 0.
extern "C" int emer_hashgrid_sliced_supported(const emer_grid_desc *g) {
    if (check_desc(g)) return 0;
    return make_slice_plan(g).ok ? 1 : 0;
}

// Bitmap rows per level (64 or 256): the slice bitmaps of emer_hashgrid_fwd / emer_hashgrid_slice_masks hold
// n_levels * rows * ceil(n / 64) words (+ EMER_SLICE_MASK_SCRATCH).  0 when the grid is not supported.
extern "C" int emer_hashgrid_mask_rows(const emer_grid_desc *g) {
    if (check_desc(g)) return 0;
    const SlicePlan plan = make_slice_plan(g);
    return plan.ok ? (int)(64u * plan.mask_q) : 0;
}

// Slice bitmaps [L][rows][ceil(N/64)] (u64) for emer_hashgrid_bwd_params_sliced when the forward did not emit them.
extern "C" int emer_hashgrid_slice_masks(const emer_grid_desc *g, const float *x, uint64_t *slice_masks, int64_t n, void *stream) {
    if (int rc = check_desc(g)) return rc;
    EMER_REQUIRE(n >= 0, "hashgrid_slice_masks: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x && slice_masks, "hashgrid_slice_masks: null pointer");
    const SlicePlan plan = make_slice_plan(g);
    EMER_REQUIRE(plan.ok, "hashgrid_slice_masks: a level needs more than 256 x 64 LDS slices");
    const uint32_t n_chunks = (uint32_t)ceil_div(n, 256);
    uint32_t blocks = 0;
    const LevelMap lmap = make_level_map(g, n_chunks, &blocks);
    return dispatch_df(g->n_dims, g->n_features, [&](auto d, auto) {
        constexpr int D = decltype(d)::value;
        if (plan.mask_q == 4)
            hipLaunchKernelGGL((hashgrid_slice_masks_kernel<D, 4>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, plan, x, slice_masks, n,
                               n_chunks, lmap);
        else
            hipLaunchKernelGGL((hashgrid_slice_masks_kernel<D, 1>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, plan, x, slice_masks, n,
                               n_chunks, lmap);
        return check_launch("hashgrid_slice_masks");
    });
}

// Owner-computes variant: OVERWRITES grad (f32) -- every entry of every level is written exactly
// once, so the caller does not zero the buffer.  The slice bitmaps come from emer_hashgrid_fwd (or
// emer_hashgrid_slice_masks) for the SAME x.
static int hashgrid_bwd_params_sliced_range(const emer_grid_desc *g, const float *x, const float *dout, int64_t sn, int64_t sl,
                                            uint64_t *slice_masks, float *grad, int64_t n, uint32_t level_begin, uint32_t level_end, void *stream,
                                            bool accumulate = false);

extern "C" int emer_hashgrid_bwd_params_sliced(const emer_grid_desc *g, const float *x, const float *dout, int64_t sn,
                                               int64_t sl, uint64_t *slice_masks, float *grad, int64_t n, void *stream) {
    if (int rc = check_desc(g)) return rc;
    return hashgrid_bwd_params_sliced_range(g, x, dout, sn, sl, slice_masks, grad, n, 0u, g->n_levels, stream);
}

// [r5] grad += the table gradient of this evaluation (a further evaluation of the same encoder in one step: warped positions, chunked
// training) -- the write-out adds instead of storing, so the caller needs no second 40 MB buffer and no add launch.
extern "C" int emer_hashgrid_bwd_params_sliced_add(const emer_grid_desc *g, const float *x, const float *dout, int64_t sn,
                                                   int64_t sl, uint64_t *slice_masks, float *grad, int64_t n, void *stream) {
    if (int rc = check_desc(g)) return rc;
    return hashgrid_bwd_params_sliced_range(g, x, dout, sn, sl, slice_masks, grad, n, 0u, g->n_levels, stream, true);
}

// The same for the levels [level_begin, level_end) only: the entries of the other levels are neither read nor written.  Two calls
// that partition the levels give the one-call result; a data-parallel trainer launches the collective of the first call's levels
// (a contiguous range of the table) while the second call computes (emernerf_amd/trainer.py).
extern "C" int emer_hashgrid_bwd_params_sliced_levels(const emer_grid_desc *g, const float *x, const float *dout, int64_t sn, int64_t sl,
                                                      uint64_t *slice_masks, float *grad, int64_t n, int32_t level_begin, int32_t level_end,
                                                      void *stream) {
    if (int rc = check_desc(g)) return rc;
    EMER_REQUIRE(level_begin >= 0 && level_begin <= level_end && level_end <= (int32_t)g->n_levels, "hashgrid_bwd_params_sliced_levels: bad level range [%d, %d)",
                 level_begin, level_end);
    if (level_begin == level_end) return EMER_OK;
    return hashgrid_bwd_params_sliced_range(g, x, dout, sn, sl, slice_masks, grad, n, (uint32_t)level_begin, (uint32_t)level_end, stream);
}

// The work-item plan of the owner-computes backward, for tests and diagnostics (host arithmetic only: no GPU needed): per level the number
// of LDS slices and of sample ranges (1 = every entry written once with plain stores; > 1 = merged with atomics: dense levels, and the
// half- or quarter-size tail items of grids without enough dense filler, EMER_TAIL_SPLIT).  Returns the total number of work items, or a
// negative error code (EMER_E_INVALID when the grid needs the global-atomic backward).  Arrays of at least n_levels entries; NULL = skip.
extern "C" int emer_hashgrid_sliced_plan(const emer_grid_desc *g, uint32_t *n_slices, uint32_t *n_ranges) {
    if (int rc = check_desc(g)) return rc;
    const SlicePlan plan = make_slice_plan(g);
    EMER_REQUIRE(plan.ok, "hashgrid_sliced_plan: a level needs more than 256 x 64 LDS slices");
    for (uint32_t l = 0; l < g->n_levels; ++l) {
        if (n_slices) n_slices[l] = plan.n_slices[l];
        if (n_ranges) n_ranges[l] = plan.n_ranges[l];
    }
    return (int)plan.total_items;
}

// Where to cut for emer_hashgrid_bwd_params_sliced_levels: the first level k of the fine range [k, L) that a caller launches
// first.  The persistent owners take work items in rounds (one item per CU at a time, three rounds for the cfg-2 table), so a
// cut costs nothing only where it falls between rounds: [k, L) is the largest set of finest levels whose items fill at most ONE
// round of the resident owners (cfg 2: levels 12..15, 4 x 64 slices = 256 items; measured 157 + 411 us against 536 us for the
// single launch, while a cut by bytes -- k = 10 -- costs 676 us: profiles/r04_table_split.txt).  0: no useful cut (the whole
// grid is one round, or the finest level alone is more than one).
extern "C" int emer_hashgrid_sliced_split_level(const emer_grid_desc *g) {
    if (check_desc(g) != EMER_OK) return 0;
    const SlicePlan plan = make_slice_plan(g);
    if (!plan.ok) return 0;
    const uint32_t owners = 256;
    if (plan.total_items <= owners) return 0;
    uint32_t items = 0, k = g->n_levels;
    while (k > 1u) {
        const uint32_t add = plan.n_slices[k - 1u] * plan.n_ranges[k - 1u];
        if (items + add > owners) break;
        items += add;
        --k;
    }
    return k < g->n_levels ? (int)k : 0;
}

static int hashgrid_bwd_params_sliced_range(const emer_grid_desc *g, const float *x, const float *dout, int64_t sn, int64_t sl,
                                            uint64_t *slice_masks, float *grad, int64_t n, uint32_t level_begin, uint32_t level_end, void *stream,
                                            bool accumulate) {
    EMER_REQUIRE(n >= 0 && n < (1ll << 28), "hashgrid_bwd_params_sliced: n out of range (byte offsets of the gathers are 32-bit: n < 2^28)");
    EMER_REQUIRE(sn == (int64_t)g->n_features, "hashgrid_bwd_params_sliced: dout must be level-major with packed features (stride_n == n_features)");
    EMER_REQUIRE(x && dout && grad && slice_masks, "hashgrid_bwd_params_sliced: null pointer");
    const uint32_t F = g->n_features;
    SlicePlan plan = make_slice_plan(g);
    EMER_REQUIRE(plan.ok, "hashgrid_bwd_params_sliced: a level needs more than 256 x 64 LDS slices; use emer_hashgrid_bwd_params");
    if (level_begin != 0u || level_end != g->n_levels) {
        // a level outside the range contributes no work item (the bitmap layout -- mask_q, gsub, shift -- stays the whole grid's: the
        // forward wrote the bitmaps for all levels); the item lists are dealt again over what is left
        uint32_t left_items = 0;
        for (uint32_t l = 0; l < g->n_levels; ++l) {
            if (l < level_begin || l >= level_end) plan.n_slices[l] = 0;
            left_items += plan.n_slices[l] * plan.n_ranges[l];
        }
        plan.total_items = left_items;
        for (int i = 0; i < 8; ++i) plan.items_per_xcd[i] = 0;
        const uint32_t sched_block = 1u << plan.sched_shift;
        for (uint32_t blk = 0; blk * sched_block < left_items; ++blk) {
            const uint32_t left = left_items - blk * sched_block;
            plan.items_per_xcd[blk & 7u] += left < sched_block ? left : sched_block;
        }
    }
    uint32_t total_items = 0;
    for (int i = 0; i < 8; ++i) total_items += plan.items_per_xcd[i];
    if (total_items == 0u) return EMER_OK;
    // Zero, in ONE launch, the levels that are merged with atomics and the work cursors (the 16 scratch words behind
    // the bitmaps).  A kernel rather than hipMemsetAsync nodes: one launch instead of up to six, and hipGraph replays of
    // memset nodes proved unreliable on ROCm 7.2 (gradients drifted after a few replays).
    ZeroRegions zr;
    zr.count = 0;
    uint32_t *work_ctr = reinterpret_cast<uint32_t *>(slice_masks + (size_t)g->n_levels * (64 * plan.mask_q) * (size_t)ceil_div(n, 64));
    zr.p[zr.count] = reinterpret_cast<float *>(work_ctr); zr.n[zr.count] = 8u; ++zr.count;
    for (uint32_t l = level_begin; l < level_end; ++l) {
        if (plan.n_ranges[l] > 1u && !accumulate) {   // (accumulate: the merging atomics add onto what is there)
            zr.p[zr.count] = grad + (size_t)g->offset[l] * F; zr.n[zr.count] = g->size[l] * F; ++zr.count;
        }
    }
    hipLaunchKernelGGL(zero_regions_kernel, dim3(64, (uint32_t)zr.count), dim3(256), 0, as_stream(stream), zr);
    if (int rc = check_launch("hashgrid_bwd_params_sliced(zero)")) return rc;
    // persistent grid: one workgroup per CU (the LDS slice fills a CU), block b lands on XCD b % 8
    uint32_t n_blocks = 256;
    if (total_items < n_blocks) n_blocks = (total_items + 7u) / 8u * 8u;
    const size_t lds_base = (size_t)plan.max_local * F * sizeof(double) + (size_t)kSliceWaves * (kScanWords * sizeof(uint64_t) + kPairQueue * sizeof(uint32_t));
    return dispatch_df(g->n_dims, g->n_features, [&](auto d, auto f) {
        constexpr int D = decltype(d)::value, FF = decltype(f)::value;
        auto kern = hashgrid_bwd_params_sliced_kernel<D, FF>;
        const size_t lds = lds_base + select_lut_bytes<D>();
        if (int rc = reserve_lds(reinterpret_cast<const void *>(kern), lds, "hashgrid_bwd_params_sliced")) return rc;
        const ProfileEvents ev = take_profile_events();
        EMER_LAUNCH_PROFILED(ev, kern, dim3(n_blocks), dim3(kSliceThreads), lds, as_stream(stream), *g, plan, x, dout, sn, sl,
                           slice_masks, work_ctr, grad, n, accumulate ? 1u : 0u);
        return check_launch("hashgrid_bwd_params_sliced");
    });
}

#ifdef EMER_SLICED_TRACE
extern "C" int emer_debug_sliced_trace(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sliced_trace), &buf, sizeof(buf)) == hipSuccess ? EMER_OK : EMER_E_LAUNCH;
}
#endif

extern "C" int emer_hashgrid_bwd_input(const emer_grid_desc *g, const float *x, const void *params, int param_dtype,
                                       const float *dout, int64_t sn, int64_t sl, float *dx, int64_t n, void *stream) {
    if (int rc = check_desc(g)) return rc;
    EMER_REQUIRE(n >= 0, "hashgrid_bwd_input: negative n");
    if (n == 0) return EMER_OK;
    EMER_REQUIRE(x && params && dout && dx, "hashgrid_bwd_input: null pointer");
    EMER_REQUIRE(param_dtype == EMER_F32 || param_dtype == EMER_F16, "hashgrid_bwd_input: bad param_dtype %d", param_dtype);
    const uint32_t blocks = (uint32_t)ceil_div(n, 256);
    return dispatch_df(g->n_dims, g->n_features, [&](auto d, auto f) {
        constexpr int D = decltype(d)::value, F = decltype(f)::value;
        if (param_dtype == EMER_F32)
            hipLaunchKernelGGL((hashgrid_bwd_input_kernel<D, F, float>), dim3(blocks), dim3(256), 0, as_stream(stream), *g, x,
                               (const float *)params, dout, sn, sl, dx, n);
        else
            hipLaunchKernelGGL((hashgrid_bwd_input_kernel<D, F, __half>), dim3(blocks), dim3(256), 0, as_stream(stream), *g,
                               x, (const __half *)params, dout, sn, sl, dx, n);
        return check_launch("hashgrid_bwd_input");
    });
}
