// sfx_host.hip -- host runtime of libsfx: the C ABI of include/sfx.h over the gfx950 kernels.
//
// One world = one HIP device, one stream.  Every buffer is allocated at upload time; the
// tick itself (sfx_step) only enqueues kernels and memsets on the world's stream and never
// allocates, synchronises or copies to the host.  There is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "sfx_kernels.h"

using namespace sfx;

// worlds alive per device in this process, and whether other processes use the device too (see sfx_tick_substep)
#include <atomic>
constexpr int kMaxDevices = 64;
static std::atomic<int> g_live_worlds[kMaxDevices];
static const bool g_shared_device = [] { const char* e = getenv("SFX_SHARED_DEVICE"); return e && e[0] == '1'; }();

constexpr uint32_t kTargetBatches = 224;  // batches per phase the grouping aims at (< 256 CUs)

// ---- static device scratch of sfx_query_points_static (the entry point that allocates nothing) ----------------
// Holds one query index (DESIGN.md "Contract / spatial queries") for a world of at most kStaticMaxBodies bodies on
// a grid of at most kStaticMaxCells buckets.  Every GPU the library runs on holds its own copy of the array, at its
// own address, until the process ends; `scratch` below is only the host-side stand-in of the symbol, the device
// address is looked up once per device (scratch_address) and everything on the host goes through that pointer.
constexpr uint32_t kStaticMaxBodies = 131072u;
constexpr uint32_t kStaticMaxCells = 1u << 18;
constexpr size_t kStaticZeroBytes = ((sizeof(sfx::Counters) + ((size_t)kStaticMaxCells + 1 + sfx::kScanItems) * 4 + 255) / 256) * 256;
constexpr size_t kStaticFillBytes = ((((size_t)kStaticMaxCells + 1 + sfx::kScanItems) * 4 + 255) / 256) * 256;
constexpr size_t kStaticScratchBytes = kStaticZeroBytes + kStaticFillBytes + 256 /* bounds */ +
                                       (size_t)kStaticMaxBodies * (4 * 4 + 2 * 16 + 2 * 8);   // = 10.5 MB
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static unsigned char* g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each device, nullptr = not looked up yet
static std::mutex g_scratch_mutex;


namespace {

thread_local std::string g_create_error;

struct Event2 { hipEvent_t a, b; int cls; uint64_t items; };

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)) == hipSuccess ? 0 : -1;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
};

}  // namespace

struct sfx_world {
    sfx_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    bool own_stream = true, group_fixed = false;
    int device = 0;
    // resolved
    float cell = 1.0f, inv_cell = 1.0f;
    uint32_t gxl = 2, gyl = 2, ncell = 16, ncell1 = 0, max_pairs = 0;   // ncell1: cells of the grid's second level (0: one level)
    // scene
    // n counts ENTRIES: the nb bodies (entry k < nb = body k with its own collider), then one entry per further collider
    // (include/sfx.h sfx_collider.body); every per-entry device array has n elements, the host vectors below nb
    uint32_t n = 0, nb = 0, m = 0, n_mats = 0, n_poly = 0;
    bool have_bodies = false, have_colliders = false, have_mats = false;
    std::vector<float> h_inv_mass, h_inv_inertia, h_pos_x, h_pos_y;
    std::vector<uint32_t> h_cbody;     // body of entry nb + k
    DevBuf<uint32_t> cbody;
    std::vector<sfx_collider> h_col;
    std::vector<float> h_poly;
    DevBuf<float4> pose, prev, vel, dpos, shape, lpose, poly, mats, aabb, sorted_aabb;   // dpos: sfx_device.h BodyArrays
    DevBuf<float2> mass;
    uint32_t vel_epoch = 0;        // derive-on-load: stamp of the current substep's velocity solve (sfx_step only)
    uint32_t vel_prev_epoch = 0;   // stamp of the previous substep of this tick when it left bodies underived, else 0
    bool dpos_clean = false;       // every displacement record is zero (the one-launch tick leaves them so and relies on it)
    uint32_t resident_pos_i = 0, resident_vel_i = 0;   // ... of k_tile_pos / k_tile_vel<256, integration on load>
    uint32_t resident_tick = 0, resident_tick_solve = 0;    // workgroups of k_tile_tick<256, with / without narrowphase> the device holds at once
    bool in_step = false;          // sfx_step is driving the substeps (it sees the whole tick)
    DevBuf<uint4> filt, sorted_filt;   // filter words by entry id / in slot order (written by k_finalize_slots)
    DevBuf<uint32_t> slot_of;          // slot of every entry; only worlds in which a dynamic body has further colliders
    bool dyn_compounds = false;        // ... (rebuild_filt): k_pairs then looks up where a further collider's body sits
    // broadphase.  `zero` is one allocation cleared by a single memset per tick:
    // [Counters][cell_start: ncell + 1 + pad][tile_cnt: ntiles + 1 + pad][tile_fill: ntiles][tile_flag: ntiles][tile_batch: ntiles]
    DevBuf<uint32_t> zero;
    Counters* ctr = nullptr;
    uint32_t *cell_start = nullptr, *tile_cnt = nullptr, *tile_fill = nullptr, *tile_flag = nullptr, *tile_batch = nullptr;
    size_t zero_bytes = 0;         // one half of `zero`: the tick's counters, cell counts and tile tables
    uint32_t zero_half = 0;        // the half this tick uses; the other one is cleared while it runs (broadphase)
    DevBuf<uint32_t> bucket, cell_fill, tmp_id, sorted_id, large_tmp, slot_cnt, slot_off, scan_sums, active;
    DevBuf<BatchDesc> batches;
    DevBuf<uint4> torder;
    DevBuf<uint32_t> done, bbody, bruns;
    DevBuf<float2> bmass;   // masses in bbody order
    uint32_t epoch = 0;            // one per fused multi-phase launch; never reset, so T.done needs no clearing
    // workgroups of the solve kernels the device holds at once, per workgroup size variant (512, 256 threads)
    uint32_t resident_pos_v[3] = {0, 0, 0}, resident_vel_v[3] = {0, 0, 0};
    bool counted = false;          // this world is in g_live_worlds[device]
    int tile_slot = 0;             // ticket counter of the next tile-solve launch (alternates)
    uint32_t ctr_skipped = 0;      // ticks since the last asynchronous read-back of the counters
    uint32_t ctr_age = 0;          // ticks begun since the pending read-back was enqueued
    bool ctr_fast = false;         // the scene is growing (adopt_counters): counters read back every tick, host at most kFastLag ticks ahead
    uint32_t prev_pairs = 0, prev_batches = 0;   // of the counters adopted before
    int tile_variant_fixed = -1;   // SFX_TILE_NT=512|256|128 pins the variant (tuning); -1: the default (256)
    int tile_variant_pos = -1, tile_variant_vel = -1;  // SFX_TILE_NT_POS / SFX_TILE_NT_VEL: per kernel (tuning)
    bool fuse_phases = true;
    DevBuf<int2> cellxy, sorted_cell;
    DevBuf<uint4> rawp, tpair;
    DevBuf<uint2> fpairs, fsorted, hit_tmp;
    uint32_t large_cap = 0, ntiles = 4, fb_cap = 0, tlx = 0, tly = 0;
    // colouring of global edge lists (joints at upload, the fallback list every tick)
    DevBuf<unsigned long long> used, claim;
    DevBuf<uint32_t> left_list;
    DevBuf<uint8_t> pcolor, prop;
    DevBuf<Counters> jctr;
    DevBuf<uint32_t> sticky;       // overflow bits of every tick since the last report (device side, see k_zero)
    uint32_t ovf_host = 0;         // the same, as far as the host has seen them (polled counters)
    // slab halos
    DevBuf<uint32_t> halo_send_idx, halo_recv_idx;
    DevBuf<float4> halo_send, halo_recv;
    uint32_t n_halo_send = 0, n_halo_recv = 0;
    DevBuf<float> halo_x0;         // x of every body when the halo lists were set
    float halo_guard = 0.0f;       // > 0: k_aabb_count flags a body further than this from halo_x0 (kOvfHalo)
    // exchange / compute overlap (sfx_halo_overlap; sfx_kernels.h k_near_closure): the velocity sweep as two launches, the
    // pack and the transfer on a stream of their own behind the first
    bool halo_overlap = false;
    hipStream_t halo_stream = nullptr;
    hipEvent_t ev_pre = nullptr, ev_near = nullptr, ev_xchg = nullptr;
    bool near_open = false;        // the halo stream holds this substep's near launch: a pack goes right behind it
    DevBuf<uint32_t> near_list;
    std::vector<uint8_t> h_halo;   // per body: on a halo list (kFiltHalo of its entries)
    // spatial query index (allocated at the first query; DESIGN.md "Contract / spatial queries")
    DevBuf<uint32_t> q_zero, q_bucket, q_cell_fill, q_tmp_id, q_sorted_id, q_large_tmp, q_bounds, q_nhits;
    DevBuf<unsigned char> q_in, q_out;   // query transfer buffers, grown on demand and kept
    DevBuf<float4> q_aabb, q_sorted_aabb;
    DevBuf<int2> q_cellxy, q_sorted_cell;
    size_t q_zero_bytes = 0;
    uint32_t q_n = 0, q_ncell = 0;
    bool q_valid = false;
    // tick in pieces
    StepParams sp{};
    bool tick_open = false;
    uint32_t substeps_done = 0;
    hipStream_t own_stream_handle = nullptr;
    DevBuf<unsigned long long> dbg;  // -DSFX_TILE_TIMING builds: stage timing of k_tile_pos
    // contact records
    DevBuf<float4> c_rec;   // kRecPlanes planes of one 16-byte word per slot (sfx_device.h ContactArrays)
    DevBuf<float2> c_lam;
    // joints (colour-sorted)
    DevBuf<uint2> j_bodies;
    DevBuf<float4> j_anchors;
    DevBuf<float4> j_params;
    DevBuf<uint32_t> j_count_dev;
    std::vector<sfx_distance_constraint> h_dist;  // as uploaded: the joint list is h_dist, then h_angle
    std::vector<sfx_angle_constraint> h_angle;
    bool joint_damped = false;
    std::vector<uint32_t> joint_perm;
    std::vector<uint8_t> joint_color_sorted;
    std::vector<uint32_t> joint_color_start;  // n_joint_colors + 1
    // joints by component (k_joint_groups): groups of whole components, their joints colour by colour; 0 groups = a launch per colour
    uint32_t joint_groups = 0;
    DevBuf<uint32_t> jg_cstart, jg_index;
    // asynchronous pose read-back (sfx_readback_begin / _wait): two slots, each a device snapshot + a pinned host copy
    hipStream_t rb_stream = nullptr;
    DevBuf<float4> rb_dev[2];
    float4* rb_host[2] = {nullptr, nullptr};
    hipEvent_t rb_snap[2] = {nullptr, nullptr}, rb_done[2] = {nullptr, nullptr};
    uint32_t rb_count[2] = {0, 0};
    uint32_t rb_cap = 0;          // bodies the slots hold
    uint64_t rb_head = 0, rb_tail = 0;
    // async counter read-back (stale by one tick, never waited for inside sfx_step)
    Counters* h_ctr = nullptr;       // pinned
    hipEvent_t ev_ctr = nullptr;
    bool ctr_pending = false, ctr_valid = false;
    Counters stale{};
    uint32_t launches = 0;
    uint32_t color_grid = 1024, tile_grid[4] = {1024, 1024, 1024, 1024}, tile_grid_max = 8192;
    uint32_t group = 4;  // tiles a colouring workgroup cuts into batches (from the pair density of an earlier tick)
    uint32_t group_cap = kMaxGroup;  // halved whenever a batch ran out of LDS slots for colliders outside its cells (kOvfExt)
    bool ticked = false;
    // profiling
    bool profiling = false;
    uint32_t prof_mask = 0;        // sfx_set_profiling(w, 2 * mask): events around the launches of these classes only, collected on demand
    std::vector<Event2> ev_used;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    sfx_kernel_time ktime[SFX_K_CLASSES]{};
    hipEvent_t t_begin = nullptr, t_end = nullptr;
};

namespace {

int fail(sfx_world* w, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (w) w->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(w, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(w, SFX_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

#define ALLOC(w, buf, count)                                                             \
    do {                                                                                 \
        if ((buf).alloc(count) != 0)                                                     \
            return fail(w, SFX_ERR_NOMEM, "hipMalloc of %zu bytes failed for " #buf,     \
                        (size_t)(count) * sizeof(*(buf).p));                             \
    } while (0)

// device address of the static scratch array on `device`, which must be the current device (looked up once per device)
int scratch_address(int device, unsigned char** out, hipError_t* err) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) { *err = hipErrorInvalidDevice; return -1; }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        unsigned char* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) { *err = e != hipSuccess ? e : hipErrorInvalidSymbol; return -1; }
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

void resolve_defaults(sfx_config& c) {
    if (c.substeps == 0) c.substeps = 1;
    if (c.color_rounds == 0) c.color_rounds = 12;
    if (c.color_rounds > 60) c.color_rounds = 60;
    if (c.tile_cells != 0u && c.tile_cells != SFX_TILE_WHOLE_GRID)   // (0: automatic, WHOLE_GRID: one tile -- resolve_grid)
        c.tile_cells = c.tile_cells >= 8 ? 8u : c.tile_cells >= 4 ? 4u : 2u;   // at least two cells: DESIGN.md "Contract / tiles"
    if (c.speculative_margin == 0.0f) c.speculative_margin = 0.010f;
    if (c.speculative_margin < 0.0f) c.speculative_margin = 0.0f;
    if (c.restitution_threshold == 0.0f) c.restitution_threshold = 2.0f;
    if (c.vel_pad_max == 0.0f) c.vel_pad_max = 0.03f;
    if (!(c.max_depenetration_velocity > 0.0f)) c.max_depenetration_velocity = INFINITY;
    if (c.vel_pad_max < 0.0f) c.vel_pad_max = 0.0f;
}

inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

uint32_t ceil_log2(uint32_t v) {
    uint32_t l = 0;
    while ((1u << l) < v && l < 31) ++l;
    return l;
}

float shape_diameter(const sfx_world* w, const sfx_collider& c) {
    switch (c.type) {
    case SFX_SHAPE_CIRCLE: return 2.0f * c.p[2];
    case SFX_SHAPE_RECT: return 2.0f * (sqrtf(c.p[0] * c.p[0] + c.p[1] * c.p[1]) + c.p[2]);
    case SFX_SHAPE_CAPSULE: return 2.0f * (c.p[0] + c.p[2]);
    case SFX_SHAPE_POLY: {
        float mx = 0.0f;
        for (uint32_t k = 0; k < c.poly_count; ++k) {
            float x = w->h_poly[2 * (c.poly_offset + k)], y = w->h_poly[2 * (c.poly_offset + k) + 1];
            float d = sqrtf(x * x + y * y);
            if (d > mx) mx = d;
        }
        return 2.0f * (mx + c.p[2]);
    }
    default: return 0.0f;
    }
}

// the body an entry belongs to; is that body dynamic (an entry beyond the bodies has no mass of its own)
uint32_t owner(const sfx_world* w, uint32_t e) { return e < w->nb ? e : w->h_cbody[e - w->nb]; }
bool is_dynamic(const sfx_world* w, uint32_t e) { const uint32_t i = owner(w, e); return w->h_inv_mass[i] > 0.0f || w->h_inv_inertia[i] > 0.0f; }

// DESIGN.md "Contract / grid": identical rule in oracle/sfx_oracle.c
void resolve_grid(sfx_world* w) {
    float cell = w->cfg.cell_size;
    if (!(cell > 0.0f)) {
        float dmax = 0.0f;
        for (uint32_t i = 0; i < w->n; ++i) {
            if (!is_dynamic(w, i) || w->h_col[i].type == SFX_SHAPE_NONE) continue;
            float d = shape_diameter(w, w->h_col[i]);
            if (d > dmax) dmax = d;
        }
        cell = dmax > 0.0f ? 1.25f * (dmax + 2.0f * (w->cfg.aabb_margin + w->cfg.vel_pad_max)) : 1.0f;
    }
    w->cell = cell;
    w->inv_cell = 1.0f / cell;
    uint32_t gx = w->cfg.grid_log2[0], gy = w->cfg.grid_log2[1];
    if (gx == 0 || gy == 0) {
        uint32_t side = (uint32_t)std::ceil(std::sqrt(2.0 * (double)(w->n ? w->n : 1)));
        uint32_t l = ceil_log2(side);
        if (l > 12) l = 12;
        gx = gy = l;
        // the 2^(2l) buckets take the shape of the scene (DESIGN.md "Contract / grid"): extent of the dynamic colliders'
        // centres at upload, in cells, rounded up to powers of two; then at least 2^(2l) buckets (the shorter side
        // grows) and at most four times as many (the longer side shrinks).  Same rule, same operations as the oracle's.
        float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        for (uint32_t i = 0; i < w->n; ++i) {
            if (!is_dynamic(w, i) || w->h_col[i].type == SFX_SHAPE_NONE) continue;
            const float x = w->h_pos_x[owner(w, i)], y = w->h_pos_y[owner(w, i)];
            if (x < xmin) xmin = x;
            if (x > xmax) xmax = x;
            if (y < ymin) ymin = y;
            if (y > ymax) ymax = y;
        }
        if (xmax >= xmin && ymax >= ymin) {
            float ex = floorf((xmax - xmin) * w->inv_cell) + 3.0f, ey = floorf((ymax - ymin) * w->inv_cell) + 3.0f;
            if (!(ex < 8192.0f)) ex = 8192.0f;
            if (!(ey < 8192.0f)) ey = 8192.0f;
            uint32_t bx = ceil_log2((uint32_t)ex), by = ceil_log2((uint32_t)ey);
            const uint32_t tot = 2u * l;
            bx = std::min(std::max(bx, 2u), 12u);
            by = std::min(std::max(by, 2u), 12u);
            while (bx + by < tot) { if (by <= bx && by < 12) ++by; else if (bx < 12) ++bx; else break; }
            while (bx + by > tot + 2u) { if (bx >= by && bx > 2) --bx; else if (by > 2) --by; else break; }
            gx = bx; gy = by;
        }
    }
    // at least 4 tiles per side, so that the 2x2 tile blocks of one phase never meet on the torus
    w->tlx = w->cfg.tile_cells == 8 ? 3u : w->cfg.tile_cells == 4 ? 2u : 1u;
    gx = std::min(std::max(gx, w->tlx + 2u), 12u);
    gy = std::min(std::max(gy, 3u), 12u);
    // tile height: as wide by default; a power of two; the grid height at most (= strips)
    w->tly = w->tlx;
    if (w->cfg.tile_cells_y) {
        w->tly = 1;
        while (w->tly < 31 && (2u << w->tly) <= w->cfg.tile_cells_y) w->tly++;
    } else {
        // automatic: colliders per tile column (centre x, wrapped like the cells) at upload time
        const uint32_t ncol = 1u << (gx - w->tlx);
        std::vector<uint32_t> cnt(ncol, 0u);
        for (uint32_t i = 0; i < w->n; ++i) {
            if (w->h_col[i].type == SFX_SHAPE_NONE) continue;
            float f = floorf(w->h_pos_x[owner(w, i)] * w->inv_cell);
            if (!(f > -1073741824.0f)) f = -1073741824.0f;
            if (f > 1073741824.0f) f = 1073741824.0f;
            const int32_t cx = (int32_t)f;
            cnt[((uint32_t)cx >> w->tlx) & (ncol - 1u)]++;
        }
        uint32_t worst = 0;
        for (uint32_t c = 0; c < ncol; ++c) worst = std::max(worst, cnt[c] + cnt[(c + 1u) & (ncol - 1u)]);
        if (worst <= 7u * SFX_TILE_MAX_BODIES / 8u) w->tly = gy;
    }
    w->tly = std::min(w->tly, gy);
    if (w->tly < gy && gy < w->tly + 2u) w->tly = gy;   // fewer than four tile rows: one row (strips)
    // one tile for the whole grid (DESIGN.md "Contract / tiles"): asked for, or automatic for small worlds -- every pair is
    // then an interior pair of the one unit: one colouring, no seam phase.  Same rule as the oracle's.
    // (automatic only while the colliders that may be too large for a base cell -- the ones a batch keeps in its kMaxExt
    // outside-body slots -- all fit one batch: a level of many platforms stays tiled)
    static_assert(SFX_ONE_TILE_LARGE <= kMaxExt, "one batch holds every large collider of a one-tile world");
    uint32_t nshape = 0, nbig = 0;
    for (uint32_t i = 0; i < w->n; ++i) {
        if (w->h_col[i].type == SFX_SHAPE_NONE) continue;
        nshape++;
        if (!(shape_diameter(w, w->h_col[i]) + 2.0f * (w->cfg.aabb_margin + w->cfg.vel_pad_max) <= 0.9f * cell)) nbig++;
    }
    if (w->cfg.tile_cells == SFX_TILE_WHOLE_GRID ||
        (w->cfg.tile_cells == 0u && w->cfg.tile_cells_y == 0u && nshape <= SFX_ONE_TILE_COLLIDERS && nbig <= SFX_ONE_TILE_LARGE)) {
        w->tlx = gx; w->tly = gy;
    }
    w->gxl = gx; w->gyl = gy;
    w->ncell = (1u << gx) * (1u << gy);
    // second level: cells eight times as wide on the same torus (gx >= tile width + 2 >= 3 and gy >= 3 above)
    w->ncell1 = (w->cfg.flags & SFX_CFG_ONE_LEVEL) ? 0u : (1u << (gx - 3u)) * (1u << (gy - 3u));
    w->ntiles = w->ncell >> (w->tlx + w->tly);
}

GridParams grid_params(const sfx_world* w) {
    GridParams g;
    g.inv_cell = w->inv_cell;
    g.small_lim = 0.9f * w->cell;
    g.gxl = w->gxl; g.gyl = w->gyl; g.ncell = w->ncell;
    g.ncell_all = w->ncell + w->ncell1;
    g.gxl1 = w->gxl - 3u; g.gyl1 = w->gyl - 3u;
    g.inv_cell1 = w->inv_cell * 0.125f;
    g.medium_lim = w->ncell1 ? 0.9f * (8.0f * w->cell) : 0.0f;
    g.tlx = w->tlx; g.tly = w->tly; g.ntiles = w->ntiles;
    g.pstride = w->ntiles;   // every phase (class of pairs) may have a batch per tile
    return g;
}

// ---- launch bookkeeping -------------------------------------------------------------
struct Prof {
    sfx_world* w;
    int cls;
    bool on;
    size_t idx = 0;
    Prof(sfx_world* w_, int cls_, uint64_t items) : w(w_), cls(cls_), on(w_->profiling || ((w_->prof_mask >> cls_) & 1u)) {
        w->launches++;
        if (!on) return;
        if (w->ev_pool.empty()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            w->ev_pool.emplace_back(a, b);
        }
        auto pr = w->ev_pool.back();
        w->ev_pool.pop_back();
        w->ev_used.push_back({pr.first, pr.second, cls, items});
        idx = w->ev_used.size() - 1;
        (void)hipEventRecord(pr.first, w->stream);
    }
    ~Prof() {
        if (on) (void)hipEventRecord(w->ev_used[idx].b, w->stream);
    }
};

int collect_profile(sfx_world* w) {
    if (w->ev_used.empty()) return SFX_OK;
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (auto& e : w->ev_used) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            w->ktime[e.cls].total_ms += ms;
            w->ktime[e.cls].launches += 1;
            w->ktime[e.cls].items += e.items;
        }
        w->ev_pool.emplace_back(e.a, e.b);
    }
    w->ev_used.clear();
    return SFX_OK;
}

BodyArrays body_arrays(sfx_world* w) {
    BodyArrays B;
    B.pose = w->pose.p; B.prev = w->prev.p; B.vel = w->vel.p; B.dpos = w->dpos.p; B.mass = w->mass.p;
    B.shape = w->shape.p; B.lpose = w->lpose.p; B.filt = w->filt.p; B.poly = w->poly.p; B.mats = w->mats.p;
    B.cbody = w->cbody.p; B.nb = w->nb;
    return B;
}

ContactArrays contact_arrays(sfx_world* w) {
    ContactArrays C;
    C.rec = w->c_rec.p; C.lam = w->c_lam.p; C.nslots = w->max_pairs + w->fb_cap;
    return C;
}

// The tick's zero-initialised block (counters, cell counts, tile tables) exists twice: a tick works in one half while its
// first kernel clears the other for the next tick, so that no tick starts with a clearing launch of its own.
void point_zero_block(sfx_world* w) {
    const size_t pad = kScanItems;
    uint32_t* base = w->zero.p + (size_t)w->zero_half * (w->zero_bytes / 4);
    w->ctr = reinterpret_cast<Counters*>(base);
    w->cell_start = base + sizeof(Counters) / 4;
    w->tile_cnt = w->cell_start + ((size_t)w->ncell + w->ncell1 + 1 + pad);
    w->tile_fill = w->tile_cnt + ((size_t)w->ntiles + 1 + pad);
    w->tile_flag = w->tile_fill + w->ntiles;
    w->tile_batch = w->tile_flag + w->ntiles;   // [4 phases][ntiles]
}

// exclusive scan of data[0..count) in place (count <= 8192 * 4096); copy receives the result too
int scan_u32(sfx_world* w, uint32_t* data, uint32_t count, uint32_t* copy) {
    const uint32_t nb = cdiv(count, kScanTile);
    if (nb > 1024u * kSumItems) return fail(w, SFX_ERR_UNSUPPORTED, "scan of %u items exceeds 32M", count);
    if (nb <= 1) {
        Prof p(w, SFX_K_SCAN, count);
        hipLaunchKernelGGL(k_scan_local, dim3(1), dim3(kBlock), 0, w->stream, data, count, w->scan_sums.p, copy, 1);
    } else {
        { Prof p(w, SFX_K_SCAN, count);
          hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(kBlock), 0, w->stream, data, count, w->scan_sums.p, (uint32_t*)nullptr, 0); }
        const int raw = nb <= (uint32_t)kBlock;  // few tiles: k_scan_add sums the totals before its tile itself
        if (!raw) { Prof p(w, SFX_K_SCAN, nb);
          hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, w->stream, w->scan_sums.p, nb); }
        { Prof p(w, SFX_K_SCAN, count);
          hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kBlock), 0, w->stream, data, count, w->scan_sums.p, copy, raw); }
    }
    return SFX_OK;
}

uint32_t edge_grid(uint32_t cap) { return std::min(std::max(cdiv(cap, kBlock), 1u), 2048u); }

TileArrays tile_arrays(sfx_world* w) {
    TileArrays T;
    T.cell_start = w->cell_start; T.sorted_id = w->sorted_id.p; T.tile_start = w->tile_cnt;
    T.active = w->active.p; T.batches = w->batches.p; T.tpair = w->tpair.p; T.torder = w->torder.p; T.bbody = w->bbody.p; T.bmass = w->bmass.p; T.bruns = w->bruns.p; T.tile_batch = w->tile_batch; T.done = w->done.p; T.fsorted = w->fsorted.p;
    T.max_pairs = w->max_pairs; T.fb_cap = w->fb_cap;
    T.near_list = w->near_list.p; T.near_cap = (uint32_t)w->near_list.n;
    T.dbg = w->dbg.p;
    return T;
}

// colour one global edge list with a single workgroup (k_color_global)
void color_global(sfx_world* w, const uint2* edges, const uint32_t* n_edges, uint32_t cap, int key_by_index,
                  uint2* sorted, uint32_t* start, uint32_t* n_colors_out, Counters* ctr, int tick_totals) {
    GlobalColor A;
    A.edges = edges; A.n_edges = n_edges; A.cap = cap; A.key_by_index = key_by_index;
    A.ends_are_colliders = key_by_index ? 0 : 1;   // joints name bodies, the fallback list colliders
    A.filt = w->filt.p; A.used = w->used.p; A.claim = w->claim.p;
    A.color = w->pcolor.p; A.prop = w->prop.p; A.left_list = w->left_list.p;
    A.sorted = sorted; A.start = start; A.n_colors_out = n_colors_out;
    A.rounds = w->cfg.color_rounds; A.ctr = ctr; A.tick_totals = tick_totals;
    A.tile_start = w->tile_cnt; A.ntiles = w->ntiles; A.max_pairs = w->max_pairs;
    Prof p(w, SFX_K_COLOR, cap);
    hipLaunchKernelGGL(k_color_global, dim3(1), dim3(kGlobalColorBlock), 0, w->stream, A);
}

#ifndef SFX_COLOR_Z   // workgroups that colour the batches of one group of tiles side by side (build knob for probes)
#define SFX_COLOR_Z(group) ((group) > 1u ? 2u : 1u)
#endif

// the velocity sweep runs as a near and a far launch (k_near_closure)
bool split_sweep(const sfx_world* w) {
    return w->halo_overlap && w->fuse_phases && w->near_list.n >= 4u * (size_t)w->ntiles && (w->n_halo_send || w->n_halo_recv);
}

int broadphase(sfx_world* w, float dt) {
    const uint32_t n = w->n;
    const GridParams g = grid_params(w);
    // this tick takes the half that the previous tick cleared; the previous tick's half is cleared by k_aabb_count on the way
    // (by k_zero when the block is large against the body count: a 4096 x 4096 grid with a few hundred bodies)
    uint4* stale_half = reinterpret_cast<uint4*>(w->zero.p + (size_t)w->zero_half * (w->zero_bytes / 4));
    w->zero_half ^= 1u;
    point_zero_block(w);
    const uint32_t gb = cdiv(std::max(n, 1u), kBlock);
    uint32_t n16 = (uint32_t)(w->zero_bytes / 16);
    if (n16 > 16u * gb * kBlock) {
      Prof p(w, SFX_K_AABB_COUNT, 0);
      hipLaunchKernelGGL(k_zero, dim3(cdiv(n16, kBlock)), dim3(kBlock), 0, w->stream, stale_half, n16, w->sticky.p);
      n16 = 0u;
    }
    { Prof p(w, SFX_K_AABB_COUNT, n);
      hipLaunchKernelGGL(k_aabb_count, dim3(gb), dim3(kBlock), 0, w->stream, n, w->pose.p, w->vel.p, w->shape.p,
                         w->lpose.p, w->poly.p, w->cfg.aabb_margin, w->cfg.vel_pad_max, dt, g, w->aabb.p, w->bucket.p, w->cellxy.p, w->cell_start,
                         w->large_tmp.p, w->large_cap, w->ctr, stale_half, n16, w->sticky.p,
                         (w->halo_guard > 0.0f && w->halo_x0.n == n) ? w->halo_x0.p : (const float*)nullptr, w->halo_guard); }
    int rc = scan_u32(w, w->cell_start, w->ncell + w->ncell1 + 1, w->cell_fill.p);
    if (rc) return rc;
    { Prof p(w, SFX_K_SCATTER, n);
      hipLaunchKernelGGL(k_scatter, dim3(gb), dim3(kBlock), 0, w->stream, n, w->bucket.p, w->cell_fill.p, w->tmp_id.p); }
    uint32_t* const slot_of = (w->dyn_compounds && w->slot_of.n >= n) ? w->slot_of.p : nullptr;
    if (slot_of) HIPCHK(w, hipMemsetAsync(slot_of, 0xFF, sizeof(uint32_t) * n, w->stream));   // kNoSlot: entries without a shape get none
    { Prof p(w, SFX_K_SCATTER, n);
      hipLaunchKernelGGL(k_finalize_slots, dim3(gb), dim3(kBlock), 0, w->stream, n, g, w->cell_start, w->tmp_id.p,
                         w->bucket.p, w->aabb.p, w->cellxy.p, w->large_tmp.p, w->large_cap, w->sorted_id.p,
                         w->sorted_aabb.p, w->sorted_cell.p, w->ctr, w->filt.p, w->sorted_filt.p, slot_of); }
    PairArrays P;
    P.slot_of = slot_of;
    P.slot_cnt = w->slot_cnt.p; P.slot_off = w->slot_off.p; P.tile_cnt = w->tile_cnt; P.tile_fill = w->tile_fill; P.tile_flag = w->tile_flag;
    P.rawp = w->rawp.p; P.fpairs = w->fpairs.p; P.hit_tmp = w->hit_tmp.p; P.max_pairs = w->max_pairs; P.fb_cap = w->fb_cap;
    // one 8-lane group per slot: 32 slots per block
    const uint32_t gp = std::min(std::max(cdiv(n + 1, kBlock / kGroup), 1u), 8192u);
    { Prof p(w, SFX_K_PAIR_COUNT, n);
      if (slot_of) hipLaunchKernelGGL((k_pairs<false, true>), dim3(gp), dim3(kBlock), 0, w->stream, n, g, w->cell_start, w->sorted_id.p,
                                      w->sorted_aabb.p, w->sorted_cell.p, w->sorted_filt.p, P, w->ctr, w->large_cap);
      else hipLaunchKernelGGL((k_pairs<false, false>), dim3(gp), dim3(kBlock), 0, w->stream, n, g, w->cell_start, w->sorted_id.p,
                              w->sorted_aabb.p, w->sorted_cell.p, w->sorted_filt.p, P, w->ctr, w->large_cap); }
    { Prof p(w, SFX_K_TILE_LIST, w->ntiles);
      if (n / std::max(w->ntiles, 1u) >= 32u)
          hipLaunchKernelGGL(k_tile_count<64>, dim3(cdiv(w->ntiles, kBlock / 64)), dim3(kBlock), 0, w->stream, g, w->cell_start,
                             w->slot_cnt.p, w->tile_cnt, w->tile_fill, w->slot_off.p);
      else
          hipLaunchKernelGGL(k_tile_count<8>, dim3(cdiv(w->ntiles, kBlock / 8)), dim3(kBlock), 0, w->stream, g, w->cell_start,
                             w->slot_cnt.p, w->tile_cnt, w->tile_fill, w->slot_off.p); }
    if (w->ntiles + 1u <= (uint32_t)kScanTile) {
      Prof p(w, SFX_K_TILE_LIST, w->ntiles);
      hipLaunchKernelGGL(k_tile_list_scan, dim3(1), dim3(kBlock), 0, w->stream, g, w->cell_start,
                         w->tile_cnt, w->tile_flag, w->active.p, w->ctr);
    } else {
      { Prof p(w, SFX_K_TILE_LIST, w->ntiles);
        hipLaunchKernelGGL(k_tile_list, dim3(cdiv(w->ntiles, kBlock)), dim3(kBlock), 0, w->stream, g, w->cell_start,
                           w->tile_cnt, w->tile_flag, w->active.p, w->ctr); }
      rc = scan_u32(w, w->tile_cnt, w->ntiles + 1, nullptr);
      if (rc) return rc;
    }
    { Prof p(w, SFX_K_PAIR_FILL, n);
      if (slot_of) hipLaunchKernelGGL((k_pairs<true, true>), dim3(gp), dim3(kBlock), 0, w->stream, n, g, w->cell_start, w->sorted_id.p,
                                      w->sorted_aabb.p, w->sorted_cell.p, w->sorted_filt.p, P, w->ctr, w->large_cap);
      else hipLaunchKernelGGL((k_pairs<true, false>), dim3(gp), dim3(kBlock), 0, w->stream, n, g, w->cell_start, w->sorted_id.p,
                              w->sorted_aabb.p, w->sorted_cell.p, w->sorted_filt.p, P, w->ctr, w->large_cap); }
    ReachParams RP;
    RP.pose = w->pose.p; RP.vel = w->vel.p; RP.lpose = w->lpose.p; RP.poly = w->poly.p;
    RP.margin = w->cfg.aabb_margin; RP.vel_pad_max = w->cfg.vel_pad_max; RP.dt = dt; RP.sm = w->cfg.speculative_margin;
    { Prof p(w, SFX_K_COLOR, w->max_pairs);
      const uint32_t est = w->ctr_valid ? w->stale.n_tile_pairs + w->stale.n_tile_pairs / 4u + 16384u : w->max_pairs;
      hipLaunchKernelGGL(k_reach, dim3(cdiv(std::min(est, w->max_pairs), kBlock)), dim3(kBlock), 0, w->stream, w->tile_cnt,
                         w->ntiles, w->max_pairs, w->rawp.p, w->shape.p, RP); }
    { Prof p(w, SFX_K_COLOR, w->max_pairs);
      hipLaunchKernelGGL(k_batch_color<kBatchP>, dim3(w->color_grid, (w->tly == w->gyl) ? 2u : 4u, SFX_COLOR_Z(w->group)), dim3(kColorThreads), 0, w->stream, g, w->group,
                         w->cell_start, w->sorted_id.p, w->sorted_filt.p, w->tile_cnt, w->active.p, w->rawp.p, w->tpair.p,
                         w->batches.p, w->shape.p, w->torder.p, w->tile_batch, w->bbody.p, (uint32_t)w->bbody.n, w->mass.p, w->bmass.p, w->bruns.p, (uint32_t)(w->bruns.n / 64), w->max_pairs, w->cfg.color_rounds, w->ctr, w->dbg.p); }
    color_global(w, w->fpairs.p, &w->ctr->n_fb, w->fb_cap, 0, w->fsorted.p, w->ctr->fb_start, &w->ctr->n_fb_colors, w->ctr, 1);
    if (split_sweep(w)) {   // slab world with the exchange overlapped: which batches the halo waits for
      Prof p(w, SFX_K_COLOR, w->ntiles);
      hipLaunchKernelGGL(k_near_closure, dim3(1), dim3(1024), 0, w->stream, g, (w->tly == w->gyl) ? 2u : 4u, w->batches.p, w->active.p,
                         w->tile_batch, w->near_list.p, (uint32_t)w->near_list.n, w->ctr);
    }
    HIPCHK(w, hipGetLastError());
    return SFX_OK;
}

// poll the async counter copy of an earlier tick (never blocks)
void adopt_counters(sfx_world* w);
void poll_counters(sfx_world* w) {
    if (w->ctr_pending && hipEventQuery(w->ev_ctr) == hipSuccess) {
        w->stale = *w->h_ctr;
        w->ovf_host |= w->stale.overflow;
        w->ctr_pending = false;
        adopt_counters(w);
    }
}

// new counters in w->stale: the launch sizes of the next ticks follow them
void adopt_counters(sfx_world* w) {
    {
        w->ctr_valid = true;
        // tiles per batch: about kTargetBatches batches per phase (one workgroup each, all resident at
        // once), but no batch beyond ~2/3 of the pair and body capacities on average
        const uint32_t ntl = w->stale.n_active[0];
        if (!w->group_fixed && ntl) {
            const uint32_t per_tile = std::max(w->stale.n_tile_pairs / ntl, 1u);
            const uint32_t want = cdiv(ntl + 1u, kTargetBatches);   // every phase walks all the tiles
            const uint32_t fit = std::max((2u * kBatchP / 3u) / per_tile, 1u);
            w->group = std::min(std::min(std::max(want, 1u), fit), (uint32_t)kMaxGroup);
        }
        // a batch's pairs touched more than kMaxExt colliders outside its cells (many mid-size or large colliders over one
        // batch of tiles): that tick dropped those pairs and says so; from now on batches hold half as many tiles
        if ((w->stale.overflow & kOvfExt) && w->group_cap > 1u) w->group_cap = std::max(1u, std::min(w->group, w->group_cap) / 2u);
        if (!w->group_fixed) w->group = std::min(w->group, w->group_cap);
        // The grids follow these counters with an eighth of headroom -- and never fall below a floor that costs nothing
        // (workgroups without a batch or a group leave at once): a scene that grows between two read-backs then runs on
        // a small grid, not on the handful of workgroups of the scene it was (the chains config: 100 k links fall onto
        // an empty floor and pile up within sixty ticks).
        uint32_t groups = 0, batches = 0;
        for (int ph = 0; ph < 4; ++ph) {
            const uint32_t b = w->stale.n_batches[ph];
            batches += b;
            w->tile_grid[ph] = std::min(std::max(b + b / 8u + 4u, 64u), w->tile_grid_max);
        }
        groups = cdiv(w->stale.n_listed[0], w->group);
        w->color_grid = std::min(std::max(groups + groups / 8u + 4u, 128u), 8192u);
        // growing: more than an eighth more pairs or batches than the counters before said -- until it stops, the counters
        // are read back after every tick and the host stays close behind them (sfx_tick_begin, substeps_done)
        w->ctr_fast = w->stale.n_pairs > w->prev_pairs + w->prev_pairs / 8u + 1024u || batches > w->prev_batches + w->prev_batches / 8u + 16u;
        w->prev_pairs = w->stale.n_pairs; w->prev_batches = batches;
    }
}

int ready(sfx_world* w) {
    if (!w) return SFX_ERR_INVALID;
    if (!w->have_bodies || !w->have_colliders || !w->have_mats)
        return fail(w, SFX_ERR_INVALID, "bodies, colliders and materials must be uploaded before this call");
    return SFX_OK;
}

// every per-entry device array at n_entries elements (contents undefined afterwards)
int alloc_entries(sfx_world* w, uint32_t n_entries) {
    const uint32_t n = n_entries;
    w->n = n;
    w->n_halo_send = w->n_halo_recv = 0;   // the halo lists and the guard's snapshot index the old entries
    w->halo_guard = 0.0f;
    w->h_halo.clear();
    const size_t c = std::max(n, 1u);
    ALLOC(w, w->pose, c); ALLOC(w, w->prev, c); ALLOC(w, w->vel, c); ALLOC(w, w->dpos, c); ALLOC(w, w->mass, c);
    HIPCHK(w, hipMemset(w->dpos.p, 0, sizeof(float4) * c));   // entries that are never integrated (further colliders) keep a zero displacement
    ALLOC(w, w->filt, c); ALLOC(w, w->shape, c); ALLOC(w, w->lpose, c); ALLOC(w, w->aabb, c); ALLOC(w, w->sorted_aabb, c); ALLOC(w, w->sorted_filt, c);
    ALLOC(w, w->bucket, c); ALLOC(w, w->tmp_id, c); ALLOC(w, w->sorted_id, c); ALLOC(w, w->large_tmp, c);
    ALLOC(w, w->slot_cnt, 3 * ((size_t)n + 1)); ALLOC(w, w->slot_off, 2 * ((size_t)n + 1)); ALLOC(w, w->hit_tmp, kHitTmp * ((size_t)n + 1)); ALLOC(w, w->cellxy, c); ALLOC(w, w->sorted_cell, c);
    ALLOC(w, w->used, c); ALLOC(w, w->claim, c * kClaim);
    ALLOC(w, w->cbody, std::max<size_t>(n - w->nb, 1));
    w->large_cap = n;
    return SFX_OK;
}

// the bodies' state (nb elements each) to the device; the entries beyond the bodies mirror their bodies, without mass
int write_state(sfx_world* w, std::vector<float4>& hp, std::vector<float4>& hv, std::vector<float2>& hm) {
    const uint32_t n = w->n;
    hp.resize(std::max(n, 1u)); hv.resize(std::max(n, 1u)); hm.resize(std::max(n, 1u));
    for (uint32_t e = w->nb; e < n; ++e) {
        const uint32_t b = w->h_cbody[e - w->nb];
        hp[e] = hp[b]; hv[e] = hv[b]; hm[e] = make_float2(0.0f, 0.0f);
    }
    HIPCHK(w, hipMemcpyAsync(w->pose.p, hp.data(), sizeof(float4) * n, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipMemcpyAsync(w->prev.p, hp.data(), sizeof(float4) * n, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipMemcpyAsync(w->vel.p, hv.data(), sizeof(float4) * n, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipMemcpyAsync(w->mass.p, hm.data(), sizeof(float2) * n, hipMemcpyHostToDevice, w->stream));
    if (n > w->nb) HIPCHK(w, hipMemcpyAsync(w->cbody.p, w->h_cbody.data(), sizeof(uint32_t) * (n - w->nb), hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return SFX_OK;
}

// further colliders take the pose and velocity of their bodies (k_mirror)
void mirror_entries(sfx_world* w) {
    if (w->n <= w->nb) return;
    const uint32_t nsec = w->n - w->nb;
    Prof p(w, SFX_K_INTEGRATE, nsec);
    hipLaunchKernelGGL(k_mirror, dim3(cdiv(nsec, kBlock)), dim3(kBlock), 0, w->stream, nsec, w->nb, w->cbody.p, w->pose.p, w->vel.p);
}

int sync_counters(sfx_world* w, Counters* out, const Counters* dev) {
    HIPCHK(w, hipMemcpyAsync(out, dev, sizeof(Counters), hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return SFX_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

uint32_t sfx_abi_version(void) { return SFX_ABI_VERSION; }

#ifndef SFX_BUILD_ID
#define SFX_BUILD_ID "unknown"
#endif
// (the id sits behind a marker so that starframe_amd/build.py can read it from the FILE: loading the library to ask it
// would pin the old copy in a process that then rebuilds it)
static const char g_build_marker[] = "sfx-build-id:" SFX_BUILD_ID;
const char* sfx_build_id(void) { return g_build_marker + 13; }

void sfx_default_config(sfx_config* cfg) {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(sfx_config);
    cfg->substeps = 4;
    cfg->gravity[0] = 0.0f; cfg->gravity[1] = -9.81f;
    cfg->aabb_margin = 0.01f;
    resolve_defaults(*cfg);
}

int sfx_create(const sfx_config* cfg, sfx_world** out) {
    if (!cfg || !out) return fail(nullptr, SFX_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(sfx_config))
        return fail(nullptr, SFX_ERR_INVALID, "sfx_config.struct_size %u != %zu", cfg->struct_size, sizeof(sfx_config));
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SFX_ERR_DEVICE, "no HIP device available (%s); libsfx has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, SFX_ERR_INVALID, "device %d out of range (%d devices)", cfg->device, ndev);
    sfx_world* w = new (std::nothrow) sfx_world();
    if (!w) return fail(nullptr, SFX_ERR_NOMEM, "out of host memory");
    w->cfg = *cfg;
    resolve_defaults(w->cfg);
    w->device = cfg->device;
    if (w->device < kMaxDevices) { g_live_worlds[w->device].fetch_add(1); w->counted = true; }
    if ((e = hipSetDevice(w->device)) != hipSuccess) {   // before the occupancy queries below: they are taken on the current device
        fail(nullptr, SFX_ERR_DEVICE, "hipSetDevice(%d) failed: %s", w->device, hipGetErrorString(e));
        sfx_destroy(w);
        return SFX_ERR_DEVICE;
    }
    // SFX_NULL_STREAM=1 runs the world on the device's default stream (profilers that cannot
    // follow a second queue)
    {   // how many workgroups of the solve kernels the device holds at once: a fused multi-phase launch is
        // that large (more would only queue up; batches are handed out by ticket to running workgroups, so
        // the launch makes progress whatever the grid size)
        hipDeviceProp_t prop;
        bool ok = hipGetDeviceProperties(&prop, w->device) == hipSuccess;
        auto occ = [&](auto kernel, int nt, uint32_t& out) {
            int b = 0;
            if (ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, kernel, nt, 0) == hipSuccess)
                out = (uint32_t)std::max(b, 0) * (uint32_t)prop.multiProcessorCount;
            else ok = false;
        };
        occ(k_tile_pos<512>, 512, w->resident_pos_v[0]); occ(k_tile_vel<512>, 512, w->resident_vel_v[0]);
        occ(k_tile_pos<256>, 256, w->resident_pos_v[1]); occ(k_tile_vel<256>, 256, w->resident_vel_v[1]);
        occ(k_tile_pos<128>, 128, w->resident_pos_v[2]); occ(k_tile_vel<128>, 128, w->resident_vel_v[2]);
        occ(k_tile_tick<256, true>, 256, w->resident_tick); occ(k_tile_tick<256, false>, 256, w->resident_tick_solve);
        occ((k_tile_pos<256, true>), 256, w->resident_pos_i); occ((k_tile_vel<256, true>), 256, w->resident_vel_i);
        auto variant = [](int v) { return v == 512 ? 0 : v == 256 ? 1 : v == 128 ? 2 : -1; };
        if (const char* nt = getenv("SFX_TILE_NT")) {
            const int v = atoi(nt);
            w->tile_variant_fixed = variant(v);
        }
        if (const char* nt = getenv("SFX_TILE_NT_POS")) { const int v = atoi(nt); w->tile_variant_pos = variant(v); }
        if (const char* nt = getenv("SFX_TILE_NT_VEL")) { const int v = atoi(nt); w->tile_variant_vel = variant(v); }
        const char* nf = getenv("SFX_NO_FUSE");   // tuning knob: one launch per phase
        w->fuse_phases = !(nf && nf[0] == '1') && ok && w->resident_pos_v[0] >= 32 && w->resident_vel_v[0] >= 32;
    }
    if (const char* gr = getenv("SFX_GROUP")) {  // tuning knob: tiles per colouring group
        w->group = std::min(std::max(1, atoi(gr)), kMaxGroup);
        w->group_fixed = true;
    }
    if (const char* gm = getenv("SFX_TILE_GRID_MAX")) {  // tuning knob: cap on the workgroups of a tile-solve launch
        w->tile_grid_max = std::max(1, atoi(gm));
        for (auto& tg : w->tile_grid) tg = std::min(tg, w->tile_grid_max);
    }
    const char* ns = getenv("SFX_NULL_STREAM");
    w->own_stream = !(ns && ns[0] == '1');
    if ((e = hipSetDevice(w->device)) != hipSuccess ||
        (w->own_stream && (e = hipStreamCreate(&w->own_stream_handle)) != hipSuccess) ||
        (e = hipEventCreate(&w->ev_ctr)) != hipSuccess || (e = hipEventCreate(&w->t_begin)) != hipSuccess ||
        (e = hipEventCreate(&w->t_end)) != hipSuccess ||
        (e = hipHostMalloc(reinterpret_cast<void**>(&w->h_ctr), sizeof(Counters))) != hipSuccess) {
        fail(nullptr, SFX_ERR_DEVICE, "device set-up failed: %s", hipGetErrorString(e));
        sfx_destroy(w);
        return SFX_ERR_DEVICE;
    }
    w->stream = w->own_stream_handle;
    {   // the static scratch of sfx_query_points_static: its address on this device, with the device current
        unsigned char* sp = nullptr;
        hipError_t se = hipSuccess;
        if (w->device < kMaxDevices && scratch_address(w->device, &sp, &se) != 0) {
            fail(nullptr, SFX_ERR_DEVICE, "hipGetSymbolAddress(scratch) failed on device %d: %s", w->device, hipGetErrorString(se));
            sfx_destroy(w);
            return SFX_ERR_DEVICE;
        }
    }
    if (w->jctr.alloc(1) || w->scan_sums.alloc(1024 * kSumItems) || w->left_list.alloc(kLeftCap) ||
        w->j_count_dev.alloc(1) || w->sticky.alloc(4) || hipMemset(w->sticky.p, 0, 16) != hipSuccess) {
        fail(nullptr, SFX_ERR_NOMEM, "hipMalloc failed");
        sfx_destroy(w);
        return SFX_ERR_NOMEM;
    }
    *out = w;
    return SFX_OK;
}

void sfx_destroy(sfx_world* w) {
    if (!w) return;
    if (w->counted) g_live_worlds[w->device].fetch_sub(1);
    (void)hipSetDevice(w->device);
    (void)hipStreamSynchronize(w->stream);
    for (auto& e : w->ev_used) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : w->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    w->pose.release(); w->prev.release(); w->vel.release(); w->dpos.release(); w->shape.release(); w->lpose.release(); w->poly.release();
    w->mats.release(); w->aabb.release(); w->sorted_aabb.release(); w->mass.release(); w->filt.release(); w->sorted_filt.release(); w->slot_of.release();
    w->bucket.release(); w->zero.release(); w->cell_fill.release(); w->tmp_id.release();
    w->active.release(); w->batches.release(); w->torder.release(); w->done.release(); w->bbody.release(); w->bmass.release(); w->bruns.release(); w->rawp.release(); w->tpair.release();
    w->fpairs.release(); w->fsorted.release(); w->hit_tmp.release();
    w->sorted_id.release(); w->large_tmp.release(); w->slot_cnt.release(); w->slot_off.release(); w->scan_sums.release();
    w->cellxy.release(); w->sorted_cell.release();
    w->used.release(); w->claim.release(); w->left_list.release();
    w->pcolor.release(); w->prop.release(); w->jctr.release(); w->sticky.release(); w->dbg.release();
    w->halo_send_idx.release(); w->halo_recv_idx.release(); w->halo_send.release(); w->halo_recv.release(); w->halo_x0.release();
    w->q_zero.release(); w->q_bucket.release(); w->q_cell_fill.release(); w->q_tmp_id.release(); w->q_sorted_id.release();
    w->q_large_tmp.release(); w->q_bounds.release(); w->q_nhits.release(); w->q_aabb.release(); w->q_sorted_aabb.release();
    w->q_cellxy.release(); w->q_sorted_cell.release(); w->q_in.release(); w->q_out.release();
    w->c_rec.release(); w->c_lam.release();
    w->j_bodies.release(); w->j_anchors.release(); w->j_params.release(); w->j_count_dev.release(); w->jg_cstart.release(); w->jg_index.release();
    if (w->rb_stream) (void)hipStreamSynchronize(w->rb_stream);
    for (int k = 0; k < 2; ++k) {
        w->rb_dev[k].release();
        if (w->rb_host[k]) (void)hipHostFree(w->rb_host[k]);
        if (w->rb_snap[k]) (void)hipEventDestroy(w->rb_snap[k]);
        if (w->rb_done[k]) (void)hipEventDestroy(w->rb_done[k]);
    }
    if (w->rb_stream) (void)hipStreamDestroy(w->rb_stream);
    if (w->h_ctr) (void)hipHostFree(w->h_ctr);
    if (w->ev_ctr) (void)hipEventDestroy(w->ev_ctr);
    if (w->t_begin) (void)hipEventDestroy(w->t_begin);
    if (w->t_end) (void)hipEventDestroy(w->t_end);
    if (w->halo_stream) { (void)hipStreamSynchronize(w->halo_stream); (void)hipStreamDestroy(w->halo_stream); }
    if (w->ev_pre) (void)hipEventDestroy(w->ev_pre);
    if (w->ev_near) (void)hipEventDestroy(w->ev_near);
    if (w->ev_xchg) (void)hipEventDestroy(w->ev_xchg);
    w->near_list.release();
    if (w->own_stream_handle) (void)hipStreamDestroy(w->own_stream_handle);
    delete w;
}

const char* sfx_last_error(sfx_world* w) { return w ? w->err.c_str() : g_create_error.c_str(); }

static int rebuild_filt(sfx_world* w) {
    if (!w->have_bodies) return SFX_OK;
    std::vector<uint4> f(std::max(w->n, 1u));
    w->dyn_compounds = false;
    for (uint32_t i = 0; i < w->n; ++i) {
        const bool dyn = is_dynamic(w, i);
        const uint32_t ob = owner(w, i);
        const bool halo = ob < w->h_halo.size() && w->h_halo[ob];
        uint4 v = make_uint4(0, 0, 0, (dyn ? kFiltDynamic : 0u) | ((dyn && i >= w->nb) ? kFiltFurther : 0u) | (halo ? kFiltHalo : 0u) | (ob << 8));
        if (w->have_colliders) { v.x = w->h_col[i].category; v.y = w->h_col[i].mask; v.z = w->h_col[i].group; }
        f[i] = v;
        if (dyn && i >= w->nb) w->dyn_compounds = true;
    }
    if (w->dyn_compounds && w->slot_of.n < w->n) ALLOC(w, w->slot_of, w->n);
    HIPCHK(w, hipMemcpyAsync(w->filt.p, f.data(), sizeof(uint4) * w->n, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return SFX_OK;
}

int sfx_upload_bodies(sfx_world* w, uint32_t n, const float* pos, const float* rot, const float* vel,
                      const float* angvel, const float* inv_mass, const float* inv_inertia) {
    if (!w) return SFX_ERR_INVALID;
    if (n && (!pos || !rot || !vel || !angvel || !inv_mass || !inv_inertia)) return fail(w, SFX_ERR_INVALID, "null array");
    if (n > (1u << 24)) return fail(w, SFX_ERR_UNSUPPORTED, "more than 16M bodies per world");
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    const bool resize = !w->have_bodies || n != w->nb;
    if (resize) {   // another body count: colliders (and the further colliders' entries), joints and halo lists go
        w->nb = n;
        w->h_cbody.clear();
        w->have_colliders = false;
        w->m = 0;
        w->h_dist.clear(); w->h_angle.clear(); w->joint_damped = false;
        w->joint_perm.clear(); w->joint_color_sorted.clear(); w->joint_color_start.assign(1, 0);
        int rc = alloc_entries(w, n);
        if (rc) return rc;
    }
    std::vector<float4> hp(std::max(n, 1u)), hv(std::max(n, 1u));
    std::vector<float2> hm(std::max(n, 1u));
    w->h_inv_mass.assign(inv_mass, inv_mass + n);
    w->h_inv_inertia.assign(inv_inertia, inv_inertia + n);
    w->h_pos_x.resize(n); w->h_pos_y.resize(n);
    for (uint32_t i = 0; i < n; ++i) { w->h_pos_x[i] = pos[2 * i]; w->h_pos_y[i] = pos[2 * i + 1]; }
    for (uint32_t i = 0; i < n; ++i) {
        hp[i] = make_float4(pos[2 * i], pos[2 * i + 1], rot[2 * i], rot[2 * i + 1]);
        hv[i] = make_float4(vel[2 * i], vel[2 * i + 1], angvel[i], 0.0f);
        hm[i] = make_float2(inv_mass[i], inv_inertia[i]);
    }
    int rc = write_state(w, hp, hv, hm);
    if (rc) return rc;
    w->have_bodies = true;
    w->ticked = false;
    w->q_valid = false;
    return rebuild_filt(w);
}

int sfx_upload_materials(sfx_world* w, uint32_t n, const sfx_material* materials) {
    if (!w || (n && !materials)) return fail(w, SFX_ERR_INVALID, "null argument");
    if (n == 0 || n > (1u << 24)) return fail(w, SFX_ERR_INVALID, "material count %u out of range", n);
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    std::vector<float4> hm(n);
    for (uint32_t i = 0; i < n; ++i)
        hm[i] = make_float4(materials[i].restitution, materials[i].static_friction, materials[i].dynamic_friction, 0.0f);
    ALLOC(w, w->mats, n);
    HIPCHK(w, hipMemcpy(w->mats.p, hm.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    w->n_mats = n;
    w->have_mats = true;
    return SFX_OK;
}

int sfx_upload_colliders(sfx_world* w, uint32_t n, const sfx_collider* col, const float* poly_verts,
                         uint32_t n_poly_verts) {
    if (!w || (n && !col)) return fail(w, SFX_ERR_INVALID, "null argument");
    if (!w->have_bodies) return fail(w, SFX_ERR_INVALID, "upload bodies before colliders");
    if (n < w->nb) return fail(w, SFX_ERR_INVALID, "collider count %u < body count %u (collider k < bodies is body k's own: SFX_SHAPE_NONE for none)", n, w->nb);
    if (n > (1u << 24)) return fail(w, SFX_ERR_UNSUPPORTED, "more than 16M bodies + further colliders per world");
    for (uint32_t k = w->nb; k < n; ++k)
        if (col[k].body >= w->nb) return fail(w, SFX_ERR_INVALID, "collider %u: body %u out of range", k, col[k].body);
    if (n_poly_verts && !poly_verts) return fail(w, SFX_ERR_INVALID, "null vertex pool");
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    {   // entries: the bodies, then the further colliders; the bodies' state is carried over when their number changes
        std::vector<float4> hp(std::max(w->nb, 1u)), hv(std::max(w->nb, 1u));
        std::vector<float2> hm(std::max(w->nb, 1u));
        HIPCHK(w, hipMemcpy(hp.data(), w->pose.p, sizeof(float4) * w->nb, hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(hv.data(), w->vel.p, sizeof(float4) * w->nb, hipMemcpyDeviceToHost));
        HIPCHK(w, hipMemcpy(hm.data(), w->mass.p, sizeof(float2) * w->nb, hipMemcpyDeviceToHost));
        w->h_cbody.resize(n - w->nb);
        for (uint32_t k = w->nb; k < n; ++k) w->h_cbody[k - w->nb] = col[k].body;
        if (n != w->n) { int rc = alloc_entries(w, n); if (rc) return rc; }
        int rc = write_state(w, hp, hv, hm);
        if (rc) return rc;
    }
    w->h_col.assign(col, col + n);
    w->h_poly.assign(poly_verts, poly_verts + 2 * (size_t)n_poly_verts);
    w->n_poly = n_poly_verts;
    std::vector<float4> hpoly(std::max(n_poly_verts, 1u), make_float4(0, 0, 0, 0));
    std::vector<float4> hs(std::max(n, 1u)), hl(std::max(n, 1u));
    for (uint32_t i = 0; i < n; ++i) {
        const sfx_collider& c = col[i];
        if (c.type > SFX_SHAPE_POLY) return fail(w, SFX_ERR_INVALID, "collider %u: unknown shape type %u", i, c.type);
        if (c.type != SFX_SHAPE_NONE && w->have_mats && c.material >= w->n_mats)
            return fail(w, SFX_ERR_INVALID, "collider %u: material %u out of range", i, c.material);
        if (c.material >= (1u << 23)) return fail(w, SFX_ERR_INVALID, "collider %u: material index too large", i);
        if (!std::isfinite(c.local_pos[0]) || !std::isfinite(c.local_pos[1]) || !std::isfinite(c.local_angle))
            return fail(w, SFX_ERR_INVALID, "collider %u: local pose is not finite", i);
        // the shape's local pose (DESIGN.md "Contract / collider pose"); the only trigonometry of the contract, once, on the host
        const bool local = c.local_pos[0] != 0.0f || c.local_pos[1] != 0.0f || c.local_angle != 0.0f;
        hl[i] = make_float4(c.local_pos[0], c.local_pos[1], c.local_angle == 0.0f ? 1.0f : cosf(c.local_angle),
                            c.local_angle == 0.0f ? 0.0f : sinf(c.local_angle));
        const uint32_t bits = c.type | (c.material << 8) | (local ? 0x80000000u : 0u);
        float4 s = make_float4(0, c.p[0], c.p[1], c.p[2]);
        memcpy(&s.x, &bits, 4);
        if (c.type == SFX_SHAPE_POLY) {
            if (c.poly_count < 3 || c.poly_count > SFX_MAX_POLY_VERTS || (uint64_t)c.poly_offset + c.poly_count > n_poly_verts)
                return fail(w, SFX_ERR_INVALID, "collider %u: bad polygon range", i);
            memcpy(&s.y, &c.poly_offset, 4);
            memcpy(&s.z, &c.poly_count, 4);
            for (uint32_t k = 0; k < c.poly_count; ++k) {
                const uint32_t k0 = c.poly_offset + k, k1 = c.poly_offset + (k + 1) % c.poly_count;
                const float ex = poly_verts[2 * k1] - poly_verts[2 * k0], ey = poly_verts[2 * k1 + 1] - poly_verts[2 * k0 + 1];
                const float len = sqrtf(ex * ex + ey * ey);
                hpoly[k0] = make_float4(poly_verts[2 * k0], poly_verts[2 * k0 + 1], ey / len, -ex / len);
            }
        }
        hs[i] = s;
    }
    ALLOC(w, w->poly, hpoly.size());
    HIPCHK(w, hipMemcpy(w->poly.p, hpoly.data(), sizeof(float4) * hpoly.size(), hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->shape.p, hs.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->lpose.p, hl.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    w->have_colliders = true;
    resolve_grid(w);
    w->max_pairs = w->cfg.max_pairs ? w->cfg.max_pairs : 12u * n;
    if (w->max_pairs < 1024u) w->max_pairs = 1024u;
    // fallback list: pairs with a large dynamic collider and pairs of tiles over capacity
    w->fb_cap = std::max(65536u, w->max_pairs / 8u);
    const size_t pad = kScanItems;
    const size_t words = sizeof(Counters) / 4 + ((size_t)w->ncell + w->ncell1 + 1 + pad) + ((size_t)w->ntiles + 1 + pad) +
                         6 * (size_t)w->ntiles;   // tile_fill, tile_flag, tile_batch per phase
    const size_t words16 = (words + 3) & ~(size_t)3;  // cleared by k_zero in 16-byte units
    ALLOC(w, w->zero, 2 * words16);
    w->zero_bytes = words16 * 4;
    w->zero_half = 0;
    point_zero_block(w);
    ALLOC(w, w->cell_fill, (size_t)w->ncell + w->ncell1 + 1 + pad);
    const size_t max_batches = 4 * (size_t)w->ntiles;   // per phase at most one per tile
    ALLOC(w, w->active, 2 * (size_t)w->ntiles); ALLOC(w, w->batches, max_batches);
    ALLOC(w, w->rawp, w->max_pairs); ALLOC(w, w->tpair, w->max_pairs); ALLOC(w, w->torder, w->max_pairs);
    ALLOC(w, w->done, max_batches);
    ALLOC(w, w->bruns, 64 * std::min<size_t>(max_batches, w->max_pairs));  // a batch holds at least one tile and one pair
    // a collider lies in at most four tile blocks; plus the outside (large) colliders of each batch: at most kMaxExt per
    // batch, one kept pair each at least
    ALLOC(w, w->bbody, 4 * (size_t)n + 16 + std::min<size_t>((size_t)kMaxExt * w->ntiles, w->max_pairs));
    ALLOC(w, w->bmass, w->bbody.n);
    HIPCHK(w, hipMemset(w->done.p, 0, sizeof(uint32_t) * max_batches));
    ALLOC(w, w->fpairs, w->fb_cap); ALLOC(w, w->fsorted, w->fb_cap);
    const size_t mc = std::max<size_t>(w->fb_cap, w->m);
    ALLOC(w, w->pcolor, mc); ALLOC(w, w->prop, mc);
    const size_t mp = (size_t)w->max_pairs + w->fb_cap;  // record slots: tile pairs, then the fallback list
    ALLOC(w, w->c_rec, kRecPlanes * mp); ALLOC(w, w->c_lam, mp);
    HIPCHK(w, hipMemset(w->c_rec.p, 0, sizeof(float4) * kRecPlanes * mp));
    HIPCHK(w, hipMemset(w->zero.p, 0, 2 * w->zero_bytes));
    HIPCHK(w, hipMemset(w->sticky.p, 0, sizeof(uint32_t)));   // every table is new: earlier overflows no longer apply
    w->ovf_host = 0;
    w->ctr_valid = false; w->ctr_pending = false;
    w->ctr_fast = false; w->prev_pairs = w->prev_batches = 0;
    w->ticked = false;
    w->q_valid = false;
    return rebuild_filt(w);
}

namespace {

// Joints by component (sfx_kernels.h k_joint_groups): the connected components of the joint graph, packed whole into groups
// of at most kJointGroupCap joints; per group its joints colour by colour.  hb: the colour-sorted joints' bodies.  A world
// with a larger component (a net, a long rope) keeps a launch per colour (joint_groups = 0).  SFX_JOINT_GROUPS=0: never.
int group_joints(sfx_world* w, const std::vector<uint2>& hb, uint32_t nc) {
    static const bool off = [] { const char* e = getenv("SFX_JOINT_GROUPS"); return e && e[0] == '0'; }();
    const uint32_t m = (uint32_t)hb.size();
    w->joint_groups = 0;
    if (off || m == 0 || nc == 0) return SFX_OK;
    std::vector<uint32_t> parent(w->nb);
    for (uint32_t i = 0; i < w->nb; ++i) parent[i] = i;
    auto find = [&](uint32_t x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    for (uint32_t d = 0; d < m; ++d) {
        const uint32_t a = hb[d].x, b = hb[d].y;
        if (a == SFX_WORLD_BODY || b == SFX_WORLD_BODY) continue;
        const uint32_t ra = find(a), rb = find(b);
        if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
    }
    // component of a joint = the root of one of its bodies; components in the order of their first joint
    std::vector<uint32_t> comp_of(m), comp_id(w->nb, 0xFFFFFFFFu), comp_size;
    for (uint32_t d = 0; d < m; ++d) {
        const uint32_t body = hb[d].x != SFX_WORLD_BODY ? hb[d].x : hb[d].y;
        if (body == SFX_WORLD_BODY) return SFX_OK;   // (a joint between the world and the world: nothing to group by)
        const uint32_t r = find(body);
        if (comp_id[r] == 0xFFFFFFFFu) { comp_id[r] = (uint32_t)comp_size.size(); comp_size.push_back(0); }
        comp_of[d] = comp_id[r];
        if (++comp_size[comp_of[d]] > kJointGroupCap) return SFX_OK;
    }
    std::vector<uint32_t> group_of_comp(comp_size.size());
    uint32_t ng = 0, fill = 0;
    for (size_t c = 0; c < comp_size.size(); ++c) {
        if (c == 0 || fill + comp_size[c] > kJointGroupCap) { ++ng; fill = 0; }
        fill += comp_size[c];
        group_of_comp[c] = ng - 1;
    }
    // per group and colour: counts, then offsets, then the joints (d ascending inside a colour)
    std::vector<uint32_t> cstart((size_t)ng * (nc + 1), 0), index(m);
    for (uint32_t d = 0; d < m; ++d) cstart[(size_t)group_of_comp[comp_of[d]] * (nc + 1) + w->joint_color_sorted[d] + 1]++;
    uint32_t run = 0;
    for (uint32_t gq = 0; gq < ng; ++gq) {
        uint32_t* cs = &cstart[(size_t)gq * (nc + 1)];
        cs[0] = run;
        for (uint32_t c = 0; c < nc; ++c) { const uint32_t cnt = cs[c + 1]; cs[c + 1] = cs[c] + cnt; }
        run = cs[nc];
    }
    std::vector<uint32_t> cur(cstart);
    for (uint32_t d = 0; d < m; ++d) index[cur[(size_t)group_of_comp[comp_of[d]] * (nc + 1) + w->joint_color_sorted[d]]++] = d;
    ALLOC(w, w->jg_cstart, cstart.size()); ALLOC(w, w->jg_index, m);
    HIPCHK(w, hipMemcpy(w->jg_cstart.p, cstart.data(), sizeof(uint32_t) * cstart.size(), hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->jg_index.p, index.data(), sizeof(uint32_t) * m, hipMemcpyHostToDevice));
    w->joint_groups = ng;
    return SFX_OK;
}

// colours the joint list (distance constraints, then angle constraints) and uploads it colour-sorted
int rebuild_joints(sfx_world* w) {
    const uint32_t md = (uint32_t)w->h_dist.size(), m = md + (uint32_t)w->h_angle.size();
    w->m = m;
    w->joint_perm.assign(m, 0);
    w->joint_color_sorted.assign(m, 0);
    w->joint_color_start.assign(1, 0);
    w->joint_damped = false;
    w->joint_groups = 0;
    if (m == 0) return SFX_OK;
    // colour the joint graph once, on the device, with the same kernels as the contact graph
    std::vector<uint2> ab(m);
    for (uint32_t k = 0; k < md; ++k) ab[k] = make_uint2(w->h_dist[k].body_a, w->h_dist[k].body_b);
    for (uint32_t k = md; k < m; ++k) ab[k] = make_uint2(w->h_angle[k - md].body_a, w->h_angle[k - md].body_b);
    DevBuf<uint2> d_ab;
    DevBuf<uint8_t> color_save, prop_save;
    ALLOC(w, d_ab, m);
    // colour/prop scratch must hold m entries even before colliders are uploaded
    const bool own_scratch = w->pcolor.n < m;
    if (own_scratch) { ALLOC(w, color_save, m); ALLOC(w, prop_save, m); std::swap(color_save.p, w->pcolor.p); std::swap(prop_save.p, w->prop.p); }
    HIPCHK(w, hipMemcpy(d_ab.p, ab.data(), sizeof(uint2) * m, hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->j_count_dev.p, &m, sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(w, hipMemsetAsync(w->jctr.p, 0, sizeof(Counters), w->stream));
    int rc = SFX_OK;
    color_global(w, d_ab.p, w->j_count_dev.p, m, 1, nullptr, nullptr, nullptr, w->jctr.p, 0);
    std::vector<uint8_t> colors(m);
    Counters jc{};
    {
        hipError_t e = hipMemcpyAsync(colors.data(), w->pcolor.p, m, hipMemcpyDeviceToHost, w->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&jc, w->jctr.p, sizeof(Counters), hipMemcpyDeviceToHost, w->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
        if (e != hipSuccess) rc = fail(w, SFX_ERR_DEVICE, "joint colouring failed: %s", hipGetErrorString(e));
    }
    if (own_scratch) { std::swap(color_save.p, w->pcolor.p); std::swap(prop_save.p, w->prop.p); }
    d_ab.release(); color_save.release(); prop_save.release();
    if (rc) return rc;
    if (jc.overflow) return fail(w, SFX_ERR_CAPACITY, "joint graph needs more than %u colours or %u serial edges", SFX_MAX_COLORS, kLeftCap);
    const uint32_t nc = jc.n_colors;
    w->joint_color_start.assign(nc + 1, 0);
    for (uint32_t k = 0; k < m; ++k) w->joint_color_start[colors[k] + 1]++;
    for (uint32_t q = 0; q < nc; ++q) w->joint_color_start[q + 1] += w->joint_color_start[q];
    std::vector<uint32_t> fill(w->joint_color_start.begin(), w->joint_color_start.end() - 1);
    for (uint32_t k = 0; k < m; ++k) {
        if (colors[k] >= nc) continue;  // cannot happen unless the colouring overflowed (rejected above)
        const uint32_t dst = fill[colors[k]]++;
        w->joint_perm[dst] = k;
        w->joint_color_sorted[dst] = colors[k];
    }
    std::vector<uint2> hb(m);
    std::vector<float4> ha(m);
    std::vector<float4> hp(m);
    for (uint32_t d = 0; d < m; ++d) {
        const uint32_t k = w->joint_perm[d];
        hb[d] = ab[k];
        if (k < md) {
            const sfx_distance_constraint& j = w->h_dist[k];
            ha[d] = make_float4(j.anchor_a[0], j.anchor_a[1], j.anchor_b[0], j.anchor_b[1]);
            hp[d] = make_float4(j.rest_length, j.compliance, 0.0f, j.damping);
            memcpy(&hp[d].z, &j.limit, 4);
            if (j.damping > 0.0f) w->joint_damped = true;
        } else {
            const sfx_angle_constraint& j = w->h_angle[k - md];
            const uint32_t kind = SFX_JOINT_ANGLE;
            ha[d] = make_float4(j.min_angle, j.max_angle, 0.0f, 0.0f);
            hp[d] = make_float4(0.0f, j.compliance, 0.0f, j.damping);
            memcpy(&hp[d].z, &kind, 4);
            if (j.damping > 0.0f) w->joint_damped = true;
        }
    }
    ALLOC(w, w->j_bodies, m); ALLOC(w, w->j_anchors, m); ALLOC(w, w->j_params, m);
    HIPCHK(w, hipMemcpy(w->j_bodies.p, hb.data(), sizeof(uint2) * m, hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->j_anchors.p, ha.data(), sizeof(float4) * m, hipMemcpyHostToDevice));
    HIPCHK(w, hipMemcpy(w->j_params.p, hp.data(), sizeof(float4) * m, hipMemcpyHostToDevice));
    return group_joints(w, hb, nc);
}

int constraints_ready(sfx_world* w, uint32_t m, const void* c) {
    if (!w || (m && !c)) return fail(w, SFX_ERR_INVALID, "null argument");
    if (!w->have_bodies) return fail(w, SFX_ERR_INVALID, "upload bodies before constraints");
    if (w->tick_open) return fail(w, SFX_ERR_INVALID, "constraints cannot change inside a tick");
    return SFX_OK;
}

bool body_ok(const sfx_world* w, uint32_t b) { return b == SFX_WORLD_BODY || b < w->nb; }

}  // namespace

int sfx_upload_constraints(sfx_world* w, uint32_t m, const sfx_distance_constraint* c) {
    int rc = constraints_ready(w, m, c);
    if (rc) return rc;
    for (uint32_t k = 0; k < m; ++k) {
        if (!body_ok(w, c[k].body_a) || !body_ok(w, c[k].body_b)) return fail(w, SFX_ERR_INVALID, "constraint %u: body index out of range", k);
        if (c[k].limit > SFX_LIMIT_STRUT) return fail(w, SFX_ERR_INVALID, "constraint %u: unknown limit %u", k, c[k].limit);
        if (!(c[k].damping >= 0.0f)) return fail(w, SFX_ERR_INVALID, "constraint %u: negative damping", k);
        if (c[k].damping > 0.0f && c[k].limit != SFX_LIMIT_NONE)
            return fail(w, SFX_ERR_INVALID, "constraint %u: damping needs a two-sided constraint (SFX_LIMIT_NONE)", k);
    }
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->h_dist.assign(c, c + m);
    return rebuild_joints(w);
}

int sfx_upload_angle_constraints(sfx_world* w, uint32_t m, const sfx_angle_constraint* c) {
    int rc = constraints_ready(w, m, c);
    if (rc) return rc;
    const float pi = 3.14159265358979323846f;
    for (uint32_t k = 0; k < m; ++k) {
        if (!body_ok(w, c[k].body_a) || !body_ok(w, c[k].body_b)) return fail(w, SFX_ERR_INVALID, "angle constraint %u: body index out of range", k);
        if (!(c[k].min_angle > -pi && c[k].min_angle <= c[k].max_angle && c[k].max_angle < pi))
            return fail(w, SFX_ERR_INVALID, "angle constraint %u: needs -pi < min_angle <= max_angle < pi", k);
        if (!(c[k].damping >= 0.0f) || !(c[k].compliance >= 0.0f)) return fail(w, SFX_ERR_INVALID, "angle constraint %u: negative damping or compliance", k);
    }
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->h_angle.assign(c, c + m);
    return rebuild_joints(w);
}

int sfx_debug_broadphase(sfx_world* w, float frame_dt) {
    int rc = ready(w);
    if (rc) return rc;
    HIPCHK(w, hipSetDevice(w->device));
    w->launches = 0;
    rc = broadphase(w, frame_dt);
    if (rc) return rc;
    w->ticked = true;
    return SFX_OK;
}

int sfx_tick_begin(sfx_world* w, float frame_dt) {
    int rc = ready(w);
    if (rc) return rc;
    if (!(frame_dt > 0.0f)) return fail(w, SFX_ERR_INVALID, "frame_dt must be positive");
    HIPCHK(w, hipSetDevice(w->device));
    // The launch sizes follow the counters of an earlier tick.  A caller that enqueues ticks without ever synchronising
    // could run arbitrarily far ahead of them -- a scene whose pair count grows a thousandfold meanwhile would run its tile
    // sweeps on the few workgroups of the empty scene (measured on the chains config: 4.4 ms per tick instead of 0.5) --
    // so the host waits for a read-back that is kMaxLag ticks old (the device still has that many ticks queued: no bubble).
    static const uint32_t max_lag = [] { const char* e = getenv("SFX_CTR_MAX_LAG"); return e ? (uint32_t)std::max(0, atoi(e)) : 16u; }();
    constexpr uint32_t kFastLag = 3;   // while the scene grows (three ticks are still queued on the device: no bubble)
    const uint32_t lag = w->ctr_fast ? std::min(max_lag, kFastLag) : max_lag;
    if (w->ctr_pending && max_lag && ++w->ctr_age > lag) HIPCHK(w, hipEventSynchronize(w->ev_ctr));
    poll_counters(w);
    if (w->ovf_host)
        return fail(w, SFX_ERR_CAPACITY, "capacity exceeded in an earlier tick (bits %u, listed by sfx_sync): its results are incomplete; sfx_sync reports "
                    "and clears the condition (bit 1: raise sfx_config.max_pairs; bit 128: the halo guard, cut the slabs again)", w->ovf_host);
    w->launches = 0;
    rc = broadphase(w, frame_dt);
    if (rc) return rc;
    StepParams& sp = w->sp;
    sp.h = frame_dt / (float)w->cfg.substeps;
    sp.inv_h = 1.0f / sp.h;
    sp.inv_h2 = sp.inv_h * sp.inv_h;
    sp.gx = w->cfg.gravity[0]; sp.gy = w->cfg.gravity[1];
    const float gl = sqrtf(sp.gx * sp.gx + sp.gy * sp.gy);
    sp.vthr = w->cfg.restitution_threshold * gl * sp.h;
    sp.flags = w->cfg.flags;
    sp.max_corr = w->cfg.max_depenetration_velocity * sp.h;
    sp.damp_lin = 1.0f / (1.0f + sp.h * w->cfg.linear_damping);
    sp.damp_ang = 1.0f / (1.0f + sp.h * w->cfg.angular_damping);
    w->tick_open = true;
    w->substeps_done = 0;
    w->ticked = true;
    return SFX_OK;
}

static int substeps_done(sfx_world* w, uint32_t count, bool mirror);

// Small worlds (sfx_kernels.h k_small_substeps): whole substeps in one workgroup, the bodies in LDS.  Measured on MI355X
// (profiles/r02/small_world_paths.log): it saves launches and hand-offs but gives up the device -- one workgroup steps through
// the colours of ALL batches of a phase where the launch-per-stage path runs every batch on a CU of its own -- so it wins
// where there is one batch anyway: the worlds that are ONE tile (automatic up to SFX_ONE_TILE_COLLIDERS colliders: x1.02 to
// x1.39 on every scene tried) and loses on tiled worlds from a few hundred bodies on (1000 circles x0.70, soup of 2000 x0.37).
// Taken by itself exactly for one-tile worlds; SFX_CFG_ONE_WORKGROUP takes it for any world of at most kSmallBodies bodies,
// SFX_CFG_MULTI_LAUNCH (or SFX_SMALL_WORLDS=0 in the environment) never.  Not under profiling (the launch-per-stage path has
// the per-stage events).  Both paths give the same bits.
static const bool g_small_off = [] { const char* e = getenv("SFX_SMALL_WORLDS"); return e && e[0] == '0'; }();

static bool small_world(const sfx_world* w) {
    if (g_small_off || (w->cfg.flags & SFX_CFG_MULTI_LAUNCH) || w->profiling || w->prof_mask || w->nb > kSmallBodies ||
        w->joint_color_start.size() > (size_t)SFX_MAX_COLORS + 1)
        return false;
    if (w->cfg.flags & SFX_CFG_ONE_WORKGROUP) return true;
    return w->tlx == w->gxl && w->tly == w->gyl;
}

static int small_substeps(sfx_world* w, uint32_t nsub) {
    const StepParams sp = w->sp;
    w->q_valid = false;
    w->dpos_clean = false;
    JointArrays J;
    J.bodies = w->j_bodies.p; J.anchors = w->j_anchors.p; J.params = w->j_params.p;
    SmallJoints SJ;
    SJ.ncolors = (uint32_t)w->joint_color_start.size() - 1;
    SJ.damped = w->joint_damped ? 1u : 0u;
    for (uint32_t c = 0; c <= SFX_MAX_COLORS; ++c) SJ.start[c] = w->joint_color_start[std::min<size_t>(c, SJ.ncolors)];
    const uint32_t nph = (w->tly == w->gyl) ? 2u : 4u;
    // (test hook: SFX_SMALL_TABLED=0 makes the kernel take its batches one after the other, the path of worlds with
    // more batches than its LDS tables hold)
    const char* te = getenv("SFX_SMALL_TABLED");
    const uint32_t max_tabled = (te && te[0] == '0') ? 0u : kSmallBatches;
    ++w->launches;
    hipLaunchKernelGGL(k_small_substeps, dim3(1), dim3(kSmallThreads), 0, w->stream, nsub, nph, w->n, grid_params(w), tile_arrays(w),
                       w->ctr, body_arrays(w), contact_arrays(w), J, SJ, sp, w->cfg.speculative_margin, max_tabled);
    HIPCHK(w, hipGetLastError());
    return substeps_done(w, nsub, false);
}

static bool tick_fused(const sfx_world* w, bool one_substep);
static int tick_substeps(sfx_world* w, bool one_substep);

int sfx_tick_substep(sfx_world* w) {
    if (!w) return SFX_ERR_INVALID;
    if (!w->tick_open) return fail(w, SFX_ERR_INVALID, "sfx_tick_substep without sfx_tick_begin");
    HIPCHK(w, hipSetDevice(w->device));
    if (small_world(w)) return small_substeps(w, 1u);
    if (!w->in_step && tick_fused(w, true)) return tick_substeps(w, true);   // integration on load, substep by substep (slab drivers)

    const StepParams sp = w->sp;
    w->q_valid = false;
    w->dpos_clean = false;   // (k_integrate zeroes the displacements at the start of a substep, nobody after its sweep)
    const BodyArrays B = body_arrays(w);
    const ContactArrays C = contact_arrays(w);
    JointArrays J;
    J.bodies = w->j_bodies.p; J.anchors = w->j_anchors.p; J.params = w->j_params.p;
    // grids follow the batch counts of an earlier tick (stale counters, read back
    // asynchronously); the kernels loop over the batch tables, so any grid size is correct
    const TileArrays T = tile_arrays(w);
    const GridParams g = grid_params(w);
    const uint32_t gb = cdiv(std::max(w->nb, 1u), kBlock);   // integration and derive walk the bodies
    const uint32_t njc = (uint32_t)w->joint_color_start.size() - 1;
    { Prof p(w, SFX_K_INTEGRATE, w->nb);
      hipLaunchKernelGGL(k_integrate, dim3(gb), dim3(kBlock), 0, w->stream, w->nb, w->pose.p, w->prev.p, w->vel.p, w->dpos.p, w->mass.p, sp, w->vel_prev_epoch); }
    // Derive-on-load (sfx_kernels.h): inside sfx_step, with no joint damping between the derive and the velocity solve and
    // no halos read between substeps, the velocity solve derives a body's velocity where it first loads it and the bodies
    // it does not touch are derived by the next integration (or by one k_derive after the last substep): three launches of
    // k_derive less per tick.  SFX_FUSE_DERIVE=0 keeps the separate pass.
    static const bool fuse_derive_off = [] { const char* e = getenv("SFX_FUSE_DERIVE"); return e && e[0] == '0'; }();
    const bool fuse_derive = w->in_step && !fuse_derive_off && !w->joint_damped && w->n_halo_send == 0 && w->n_halo_recv == 0;
    const bool last_substep = w->substeps_done + 1u >= w->cfg.substeps;
    if (fuse_derive && ++w->vel_epoch == 0u) w->vel_epoch = 1u;   // (0 means "off")
    const uint32_t vstamp = fuse_derive ? w->vel_epoch : 0u;
    if (w->joint_groups) {   // every colour in one launch, component by component (k_joint_groups)
        Prof p(w, SFX_K_SOLVE_JOINTS, w->m);
        hipLaunchKernelGGL(k_joint_groups<false>, dim3(w->joint_groups), dim3(kJointGroupThreads), 0, w->stream, njc, w->jg_cstart.p, w->jg_index.p, B, J, sp.inv_h2);
    }
    for (uint32_t jc = 0; !w->joint_groups && jc < njc; ++jc) {
        const uint32_t start = w->joint_color_start[jc], cnt = w->joint_color_start[jc + 1] - start;
        if (!cnt) continue;
        Prof p(w, SFX_K_SOLVE_JOINTS, cnt);
        hipLaunchKernelGGL(k_solve_joints, dim3(cdiv(cnt, kBlock)), dim3(kBlock), 0, w->stream, start, cnt, B, J, sp.inv_h2);
    }
    // full-height strips have one tile row: only the phases 0 and 1 (tile x parity) exist
    const uint32_t nph = (w->tly == w->gyl) ? 2u : 4u;
    uint32_t total_grid = 0;
    for (uint32_t ph = 0; ph < nph; ++ph) total_grid += w->tile_grid[ph];
    // Workgroup size of the tile solves.  A batch is a dependent chain (one short step per colour) that keeps one or
    // two waves busy, so what counts is that every batch of the sweep is on the device at once (a later phase's batch
    // fetches its records, body ids and masses while it waits for its neighbours) and that the barriers are cheap:
    // 256 threads (three workgroups per CU by LDS).  With batches handed out by workgroup index and a 7/8 residency
    // margin, 512 threads used to win on the 100 k pile; with tickets 256 wins on every BASELINE config (measured:
    // pyramid 0.605 vs 0.631 ms, 10 k circles 0.385 vs 0.401, 100 k pile 0.775 vs 0.795, 500 k slab 2.13 vs 2.56).
    const int tv = w->tile_variant_fixed >= 0 ? w->tile_variant_fixed : 1;
    const int tvp = w->tile_variant_pos >= 0 ? w->tile_variant_pos : tv, tvv = w->tile_variant_vel >= 0 ? w->tile_variant_vel : tv;
    const uint32_t res_pos = w->resident_pos_v[tvp], res_vel = w->resident_vel_v[tvv];
    // Fused launches: workgroup b may start with batch b (no ticket round trip) while this world is the only one
    // on its device -- the grid is then resident as a whole.  With other worlds on the device (logical slabs, several
    // ranks per GPU: SFX_SHARED_DEVICE=1 says so for other processes) every batch is handed out by ticket.
    static const int race_selftest = [] { const char* e = getenv("SFX_RACE_SELFTEST"); return (e && e[0] == '1') ? kTileRaceSelfTest : 0; }();   // libsfx_races.so only
    const int first = ((w->counted && g_live_worlds[w->device].load() == 1 && !g_shared_device) ? kTileStaticFirst : 0) | race_selftest;
    auto launch_pos = [&](uint32_t grid, uint32_t ph0, uint32_t np, uint32_t ep, int fb) {
        fb |= (w->tile_slot ^= 1) ? kTileSlot : 0;
        const uint32_t nt = tvp == 0 ? 512u : tvp == 1 ? 256u : 128u;
        if (tvp == 2) hipLaunchKernelGGL(k_tile_pos<128>, dim3(grid), dim3(nt), 0, w->stream, ph0, np, ep, fb, g, T, w->ctr, B, C, sp.max_corr, w->cfg.speculative_margin, sp.vthr, sp, 0u);
        else if (tvp == 0) hipLaunchKernelGGL(k_tile_pos<512>, dim3(grid), dim3(nt), 0, w->stream, ph0, np, ep, fb, g, T, w->ctr, B, C, sp.max_corr, w->cfg.speculative_margin, sp.vthr, sp, 0u);
        else hipLaunchKernelGGL(k_tile_pos<256>, dim3(grid), dim3(nt), 0, w->stream, ph0, np, ep, fb, g, T, w->ctr, B, C, sp.max_corr, w->cfg.speculative_margin, sp.vthr, sp, 0u);
    };
    auto launch_vel = [&](uint32_t grid, uint32_t ph0, uint32_t np, uint32_t ep, int fb, hipStream_t st = nullptr) {
        if (!(fb & kTileOnlyNear)) fb |= (w->tile_slot ^= 1) ? kTileSlot : 0;   // (the near launch takes no tickets)
        if (!st) st = w->stream;
        const uint32_t nt = tvv == 0 ? 512u : tvv == 1 ? 256u : 128u;
        if (tvv == 2) hipLaunchKernelGGL(k_tile_vel<128>, dim3(grid), dim3(nt), 0, st, ph0, np, ep, fb, g, T, w->ctr, B, C, sp, vstamp);
        else if (tvv == 0) hipLaunchKernelGGL(k_tile_vel<512>, dim3(grid), dim3(nt), 0, st, ph0, np, ep, fb, g, T, w->ctr, B, C, sp, vstamp);
        else hipLaunchKernelGGL(k_tile_vel<256>, dim3(grid), dim3(nt), 0, st, ph0, np, ep, fb, g, T, w->ctr, B, C, sp, vstamp);
    };
    mirror_entries(w);   // the narrowphase reads colliders (entries), the solves write bodies
    // The narrowphase inside the position sweep (sfx_kernels.h narrow_batch): every batch computes its own pairs' records
    // before it solves them -- one launch less per substep, and the batches of the later phases do that work while they
    // wait for their neighbours.  Same bits.  SFX_FUSE_NARROW=0 keeps the k_narrow launch (2: later phases only).
    static const int fuse_narrow_mode = [] { const char* e = getenv("SFX_FUSE_NARROW"); return e ? atoi(e) : 1; }();
    const int fuse_narrow = !fuse_narrow_mode ? 0 : fuse_narrow_mode == 2 ? kTileNarrow : (kTileNarrow | kTileNarrowAll);
    if (!(fuse_narrow & kTileNarrowAll)) {   // manifolds of all solver pairs at the poses integration and joints left (one pair per lane)
        const uint32_t est = w->ctr_valid ? w->stale.n_pairs : w->max_pairs;
        Prof p(w, SFX_K_NARROW, w->ctr_valid ? w->stale.n_kept : 0);
        static const uint32_t ndiv = [] { const char* e = getenv("SFX_NARROW_DIV"); return e ? (uint32_t)std::max(1, atoi(e)) : 1u; }();
        hipLaunchKernelGGL(k_narrow, dim3(std::min(std::max(cdiv(est, kBlock * ndiv), 1u), 16384u)), dim3(kBlock), 0, w->stream, T, w->ctr,
                           B, C, w->cfg.speculative_margin, sp.vthr, fuse_narrow ? 1u : 0u);
    }
    if (w->fuse_phases) {
        // all phases in one launch, batches wait for the neighbouring batches of earlier phases
        Prof p(w, SFX_K_SOLVE_POS, w->ctr_valid ? w->stale.n_kept : 0);
        launch_pos(std::min(total_grid, res_pos), 0u, nph, ++w->epoch, kTileFallback | first | fuse_narrow);
    } else {
        for (uint32_t ph = 0; ph < nph; ++ph) {
            Prof p(w, SFX_K_SOLVE_POS, w->ctr_valid ? w->stale.n_kept / nph : 0);
            launch_pos(w->tile_grid[ph], ph, 1u, 0u, (ph + 1 == nph ? kTileFallback : 0) | kTileStaticFirst | fuse_narrow);
        }
    }
    if (!fuse_derive) {
      Prof p(w, SFX_K_DERIVE_VEL, w->nb);
      hipLaunchKernelGGL(k_derive, dim3(gb), dim3(kBlock), 0, w->stream, w->nb, w->pose.p, w->prev.p, w->vel.p, w->dpos.p, w->mass.p, sp, 0u);
    }
    if (w->joint_damped && w->joint_groups) {
        Prof p(w, SFX_K_SOLVE_JOINTS, w->m);
        hipLaunchKernelGGL(k_joint_groups<true>, dim3(w->joint_groups), dim3(kJointGroupThreads), 0, w->stream, njc, w->jg_cstart.p, w->jg_index.p, B, J, sp.h);
    }
    for (uint32_t jc = 0; w->joint_damped && !w->joint_groups && jc < njc; ++jc) {
        const uint32_t start = w->joint_color_start[jc], cnt = w->joint_color_start[jc + 1] - start;
        if (!cnt) continue;
        Prof p(w, SFX_K_SOLVE_JOINTS, cnt);
        hipLaunchKernelGGL(k_damp_joints, dim3(cdiv(cnt, kBlock)), dim3(kBlock), 0, w->stream, start, cnt, B, J, sp.h);
    }
    // The split is taken when the last counters read back say the tick can be split (no fallback list: k_near_closure) --
    // a tick that turns out otherwise runs its whole sweep in the near launch, on that launch's few workgroups: slow,
    // right, and over when the next counters arrive.
    if (w->fuse_phases && split_sweep(w) && w->ctr_valid && w->stale.n_near != kNoSplit && res_vel >= 64u) {
        // the batches the halo waits for on the halo stream -- the pack and the transfer follow them there -- and all the
        // others beside them on the world's stream, under the same epoch (k_near_closure).  Both launches fit on the
        // device together: the far batches that wait for a near batch always find it running or done.
        const uint32_t ep = ++w->epoch;
        const uint32_t grid_near = std::min(std::max(w->stale.n_near + w->stale.n_near / 4u + 4u, 8u), res_vel / 4u);
        const uint32_t grid_far = std::min(total_grid, res_vel - grid_near);
        HIPCHK(w, hipEventRecord(w->ev_pre, w->stream));
        HIPCHK(w, hipStreamWaitEvent(w->halo_stream, w->ev_pre, 0));
        launch_vel(grid_near, 0u, nph, ep, kTileFallback | kTileOnlyNear | race_selftest, w->halo_stream);
        HIPCHK(w, hipEventRecord(w->ev_near, w->halo_stream));
        { Prof p(w, SFX_K_SOLVE_VEL, w->ctr_valid ? w->stale.n_kept : 0);
          launch_vel(grid_far, 0u, nph, ep, kTileFallback | kTileOnlyFar | first); }
        HIPCHK(w, hipStreamWaitEvent(w->stream, w->ev_near, 0));   // whatever follows on the world's stream follows both launches
        w->near_open = true;
    } else if (w->fuse_phases) {
        Prof p(w, SFX_K_SOLVE_VEL, w->ctr_valid ? w->stale.n_kept : 0);
        launch_vel(std::min(total_grid, res_vel), 0u, nph, ++w->epoch, kTileFallback | first);
    } else {
        for (uint32_t ph = 0; ph < nph; ++ph) {
            Prof p(w, SFX_K_SOLVE_VEL, w->ctr_valid ? w->stale.n_kept / nph : 0);
            launch_vel(w->tile_grid[ph], ph, 1u, 0u, (ph + 1 == nph ? kTileFallback : 0) | kTileStaticFirst);
        }
    }
    if (fuse_derive && last_substep) {   // the bodies no pair touched in the last substep
      Prof p(w, SFX_K_DERIVE_VEL, w->nb);
      hipLaunchKernelGGL(k_derive, dim3(gb), dim3(kBlock), 0, w->stream, w->nb, w->pose.p, w->prev.p, w->vel.p, w->dpos.p, w->mass.p, sp, vstamp);
    }
    w->vel_prev_epoch = (fuse_derive && !last_substep) ? vstamp : 0u;
    HIPCHK(w, hipGetLastError());
    return substeps_done(w, 1u, true);
}

// after `count` substeps have been enqueued: at the end of the tick, the mirror entries (unless the kernel that ran the
// substeps has written them) and the asynchronous read-back of the tick's counters
static int substeps_done(sfx_world* w, uint32_t count, bool mirror) {
    if ((w->substeps_done += count) >= w->cfg.substeps) {
        if (mirror) mirror_entries(w);   // between ticks the entries are where their bodies are (queries, AABBs, debug hooks)
        // end of the tick: start the asynchronous read-back of its counters
        w->tick_open = false;
        static const uint32_t every = [] { const char* e = getenv("SFX_CTR_EVERY"); return e ? (uint32_t)std::max(1, atoi(e)) : 8u; }();
        // the counters only size the next ticks' grids (every kernel loops over what it is given), so they are read
        // back every few ticks: the copy costs the stream about 6 us
        if (!w->ctr_pending && (!w->ctr_valid || ++w->ctr_skipped >= (w->ctr_fast ? 1u : every))) {
            w->ctr_skipped = 0;
            w->ctr_age = 0;
            HIPCHK(w, hipMemcpyAsync(w->h_ctr, w->ctr, sizeof(Counters), hipMemcpyDeviceToHost, w->stream));
            HIPCHK(w, hipEventRecord(w->ev_ctr, w->stream));
            w->ctr_pending = true;
        }
        if (w->profiling) return collect_profile(w);
    }
    return SFX_OK;
}

// Integration on load (DESIGN.md "Integration on load"; sfx_kernels.h k_tile_pos / k_tile_vel<256, true>, k_tick_free): sfx_step,
// which sees the whole tick, runs the substeps of worlds whose substeps are contact sweeps and nothing else -- no joints, no
// further colliders (their mirrors are refreshed between the stages), no halos (exchanged between substeps) -- WITHOUT
// k_integrate: a body's pose and velocity at the start of a substep are a function of its record at the end of the substep
// before, computed by the narrowphase where it loads a body and by the velocity sweep where it derives one; the bodies no
// batch holds are stepped through all substeps by one k_tick_free at the end of the tick.  Same bits as the path with
// k_integrate (tests/test_gpu_fused_tick.py).  SFX_FUSE_TICK=0 keeps that path; 1 and 2 are the two forms of "the substeps of
// a tick as one launch" (k_tile_tick: sweeps as ticketed items that wait for the batches around them instead of launch
// boundaries), bit-identical too, measured slower and kept as knobs (DESIGN.md "The substeps of a tick as one launch").
static const int g_fuse_tick_mode = [] { const char* e = getenv("SFX_FUSE_TICK"); return e ? atoi(e) : 3; }();   // 0: off; 3 (default): the staged launches with the integration on load;
                                                                                    // measured and not kept as defaults: 1: one launch per tick, 2: per substep a narrowphase launch + one launch of both sweeps
static const bool g_fuse_tick_off = g_fuse_tick_mode == 0;

// one_substep: asked by sfx_tick_substep, which runs ONE substep and must leave every body final for its caller (a slab driver
// packs its halo next): the default form only, halos allowed (a ghost is a dynamic body whose record the unpack refreshes),
// k_tick_free after every substep; not with the exchange overlapped (the near launch's pack would overtake k_tick_free).
static bool tick_fused(const sfx_world* w, bool one_substep) {
    static const bool other_knobs = [] {   // the launch-per-stage variants these knobs ask for
        const char* a = getenv("SFX_FUSE_DERIVE"); const char* b = getenv("SFX_FUSE_NARROW");
        return (a && a[0] == '0') || (b && atoi(b) != 1);
    }();
    if (g_fuse_tick_off || other_knobs || !w->fuse_phases || w->resident_tick < 32u || w->resident_pos_i < 32u || w->resident_vel_i < 32u) return false;
    if (w->joint_color_start.size() > 1 || w->joint_damped || w->n != w->nb) return false;
    if (one_substep ? (g_fuse_tick_mode != 3 || split_sweep(w)) : (w->n_halo_send || w->n_halo_recv)) return false;
    if (w->tile_variant_fixed >= 0 && w->tile_variant_fixed != 1) return false;
    if ((w->tile_variant_pos >= 0 && w->tile_variant_pos != 1) || (w->tile_variant_vel >= 0 && w->tile_variant_vel != 1)) return false;
    if (((uint64_t)w->max_pairs + w->fb_cap) * kRecPlanes * 16ull >= 0xFFFFFFF0ull) return false;   // record planes addressed by 32-bit byte offsets
    return true;
}

static int tick_substeps(sfx_world* w, bool one_substep) {
    const StepParams sp = w->sp;
    const uint32_t nsub = one_substep ? 1u : w->cfg.substeps;   // substeps of this call (one_substep: k_tick_free writes every body after it, so it is "substep 0" again)
    w->q_valid = false;
    const BodyArrays B = body_arrays(w);
    const ContactArrays C = contact_arrays(w);
    const TileArrays T = tile_arrays(w);
    const GridParams g = grid_params(w);
    if (!w->dpos_clean) {   // the launch-per-stage path (or the upload) left displacements behind
        HIPCHK(w, hipMemsetAsync(w->dpos.p, 0, sizeof(float4) * (size_t)w->n, w->stream));
        w->dpos_clean = true;
    }
    const uint32_t nph = (w->tly == w->gyl) ? 2u : 4u;
    uint32_t total_grid = 0;
    for (uint32_t ph = 0; ph < nph; ++ph) total_grid += w->tile_grid[ph];
    if (w->vel_epoch > 0xFFFF0000u) w->vel_epoch = 0u;
    const uint32_t vstamp0 = w->vel_epoch + 1u;   // substep s stamps vstamp0 + s (never 0)
    w->vel_epoch += nsub;
    static const int race_selftest = [] { const char* e = getenv("SFX_RACE_SELFTEST"); return (e && e[0] == '1') ? kTileRaceSelfTest : 0; }();
    const int fl0 = ((w->counted && g_live_worlds[w->device].load() == 1 && !g_shared_device) ? kTileStaticFirst : 0) | race_selftest;
    if (g_fuse_tick_mode == 3) {
        // Integration on load, a launch per sweep (DESIGN.md "Integration on load"): the staged launches without k_integrate --
        // the narrowphase at the head of the position sweep integrates the bodies it loads, the velocity sweep the ones it derives
        const uint32_t grid_p = std::max(std::min(total_grid, w->resident_pos_i), 1u), grid_v = std::max(std::min(total_grid, w->resident_vel_i), 1u);
        for (uint32_t s = 0; s < nsub; ++s) {
            { Prof p(w, SFX_K_SOLVE_POS, w->ctr_valid ? w->stale.n_kept : 0);
              const int fl = fl0 | kTileFallback | kTileNarrow | kTileNarrowAll | ((w->tile_slot ^= 1) ? kTileSlot : 0);
              hipLaunchKernelGGL((k_tile_pos<256, true>), dim3(grid_p), dim3(256), 0, w->stream, 0u, nph, ++w->epoch, fl, g, T, w->ctr, B, C, sp.max_corr,
                                 w->cfg.speculative_margin, sp.vthr, sp, s); }
            { Prof p(w, SFX_K_SOLVE_VEL, w->ctr_valid ? w->stale.n_kept : 0);
              const int fl = fl0 | kTileFallback | ((w->tile_slot ^= 1) ? kTileSlot : 0);
              hipLaunchKernelGGL((k_tile_vel<256, true>), dim3(grid_v), dim3(256), 0, w->stream, 0u, nph, ++w->epoch, fl, g, T, w->ctr, B, C, sp, vstamp0 + s); }
        }
    } else if (g_fuse_tick_mode == 2) {
        // one launch per substep, the narrowphase (with the integration on load) in a launch of its own in front of it
        const uint32_t grid = std::max(std::min(2u * total_grid, w->resident_tick_solve), 1u);
        const uint32_t est = w->ctr_valid ? w->stale.n_pairs : w->max_pairs;
        for (uint32_t s = 0; s < nsub; ++s) {
            { Prof p(w, SFX_K_NARROW, w->ctr_valid ? w->stale.n_kept : 0);
              hipLaunchKernelGGL(k_narrow_tick, dim3(std::min(std::max(cdiv(est, kBlock), 1u), 16384u)), dim3(kBlock), 0, w->stream, T, w->ctr, B, C,
                                 w->cfg.speculative_margin, sp, s); }
            const uint32_t epoch0 = w->epoch;
            w->epoch += 2u;
            const int fl = fl0 | ((w->tile_slot ^= 1) ? kTileSlot : 0);
            Prof p(w, SFX_K_TICK, w->ctr_valid ? (uint64_t)w->stale.n_kept : 0);
            hipLaunchKernelGGL((k_tile_tick<256, false>), dim3(grid), dim3(256), 0, w->stream, 1u, s, nph, epoch0, vstamp0 + s, fl, g, T, w->ctr, B, C, sp,
                               w->cfg.speculative_margin);
        }
    } else {
        const uint32_t epoch0 = w->epoch;   // sweep u publishes epoch0 + u + 1
        w->epoch += 2u * nsub;
        const int fl = fl0 | ((w->tile_slot ^= 1) ? kTileSlot : 0);
        // two sweeps can be under way at once (a batch of the next sweep starts when the batches around it are done)
        const uint32_t grid = std::max(std::min(2u * total_grid, w->resident_tick), 1u);
        Prof p(w, SFX_K_TICK, w->ctr_valid ? (uint64_t)w->stale.n_kept * nsub : 0);
        hipLaunchKernelGGL((k_tile_tick<256, true>), dim3(grid), dim3(256), 0, w->stream, nsub, 0u, nph, epoch0, vstamp0, fl, g, T, w->ctr, B, C, sp,
                           w->cfg.speculative_margin);
    }
    { Prof p(w, SFX_K_INTEGRATE, w->nb);
      hipLaunchKernelGGL(k_tick_free, dim3(cdiv(std::max(w->nb, 1u), kBlock)), dim3(kBlock), 0, w->stream, w->nb, w->pose.p, w->prev.p,
                         w->vel.p, w->mass.p, sp, nsub, vstamp0 + nsub - 1u); }
    w->vel_prev_epoch = 0u;
    HIPCHK(w, hipGetLastError());
    return substeps_done(w, nsub, true);
}

int sfx_step(sfx_world* w, float frame_dt) {
    int rc = sfx_tick_begin(w, frame_dt);
    if (rc == SFX_OK && small_world(w)) return small_substeps(w, w->cfg.substeps);   // the whole tick in one launch
    if (rc == SFX_OK && tick_fused(w, false)) return tick_substeps(w, false);        // integration on load (or the substeps in one launch)
    w->in_step = true;   // the substeps below are the whole tick: velocities may stay underived between them
    for (uint32_t s = 0; rc == SFX_OK && s < w->cfg.substeps; ++s) rc = sfx_tick_substep(w);
    w->in_step = false;
    if (rc != SFX_OK) w->vel_prev_epoch = 0u;
    return rc;
}

int sfx_set_stream(sfx_world* w, void* hip_stream) {
    if (!w) return SFX_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : w->own_stream_handle;
    return SFX_OK;
}

int sfx_get_stream(sfx_world* w, void** hip_stream) {
    if (!w || !hip_stream) return SFX_ERR_INVALID;
    *hip_stream = reinterpret_cast<void*>(w->stream);
    return SFX_OK;
}

int sfx_halo_counts(sfx_world* w, uint32_t* n_send, uint32_t* n_recv) {
    if (!w) return SFX_ERR_INVALID;
    if (n_send) *n_send = w->n_halo_send;
    if (n_recv) *n_recv = w->n_halo_recv;
    return SFX_OK;
}

int sfx_halo_set(sfx_world* w, const uint32_t* send_idx, uint32_t n_send, const uint32_t* recv_idx, uint32_t n_recv) {
    if (!w || (n_send && !send_idx) || (n_recv && !recv_idx)) return fail(w, SFX_ERR_INVALID, "null argument");
    if (!w->have_bodies) return fail(w, SFX_ERR_INVALID, "upload bodies before the halo lists");
    for (uint32_t k = 0; k < n_send; ++k)
        if (send_idx[k] >= w->nb) return fail(w, SFX_ERR_INVALID, "halo send index %u out of range", send_idx[k]);
    for (uint32_t k = 0; k < n_recv; ++k)
        if (recv_idx[k] >= w->nb) return fail(w, SFX_ERR_INVALID, "halo receive index %u out of range", recv_idx[k]);
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    ALLOC(w, w->halo_send_idx, n_send); ALLOC(w, w->halo_recv_idx, n_recv);
    ALLOC(w, w->halo_send, 2 * (size_t)n_send); ALLOC(w, w->halo_recv, 2 * (size_t)n_recv);
    if (n_send) HIPCHK(w, hipMemcpy(w->halo_send_idx.p, send_idx, sizeof(uint32_t) * n_send, hipMemcpyHostToDevice));
    if (n_recv) HIPCHK(w, hipMemcpy(w->halo_recv_idx.p, recv_idx, sizeof(uint32_t) * n_recv, hipMemcpyHostToDevice));
    w->n_halo_send = n_send; w->n_halo_recv = n_recv;
    if (w->halo_overlap && w->near_list.n < 4u * (size_t)w->ntiles) ALLOC(w, w->near_list, 4u * (size_t)std::max(w->ntiles, 1u));
    {   // the halo bodies' mark in the filter words (kFiltHalo): the colouring finds the batches that touch them
        w->h_halo.assign(w->nb, 0);
        for (uint32_t k = 0; k < n_send; ++k) w->h_halo[send_idx[k]] = 1;
        for (uint32_t k = 0; k < n_recv; ++k) w->h_halo[recv_idx[k]] = 1;
        const int rc = rebuild_filt(w);
        if (rc) return rc;
    }
    // the guard's reference: where every body is now, when the lists are cut
    ALLOC(w, w->halo_x0, std::max(w->n, 1u));
    hipLaunchKernelGGL(k_halo_snapshot, dim3(cdiv(std::max(w->n, 1u), kBlock)), dim3(kBlock), 0, w->stream, w->n, w->pose.p, w->halo_x0.p);
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return SFX_OK;
}

int sfx_halo_guard(sfx_world* w, float max_move) {
    if (!w) return SFX_ERR_INVALID;
    if (!(max_move >= 0.0f)) return fail(w, SFX_ERR_INVALID, "sfx_halo_guard: max_move must be >= 0");
    if (max_move > 0.0f && w->halo_x0.n != std::max(w->n, 1u)) return fail(w, SFX_ERR_INVALID, "sfx_halo_guard: call sfx_halo_set first");
    w->halo_guard = max_move;
    return SFX_OK;
}

int sfx_halo_pack(sfx_world* w) {
    if (!w) return SFX_ERR_INVALID;
    if (!w->n_halo_send) return SFX_OK;
    HIPCHK(w, hipSetDevice(w->device));
    hipStream_t st = w->stream;
    if (w->halo_overlap) {   // on the halo stream, behind the near launch of this substep (or behind whatever the world's stream holds)
        st = w->halo_stream;
        if (!w->near_open) {   // no split launch before it: behind whatever the world's stream holds
            HIPCHK(w, hipEventRecord(w->ev_pre, w->stream));
            HIPCHK(w, hipStreamWaitEvent(st, w->ev_pre, 0));
        }
        w->near_open = false;
    }
    Prof p(w, SFX_K_HALO, w->n_halo_send);
    hipLaunchKernelGGL(k_halo_pack, dim3(cdiv(w->n_halo_send, kBlock)), dim3(kBlock), 0, st, w->n_halo_send,
                       w->halo_send_idx.p, w->pose.p, w->vel.p, w->halo_send.p);
    return SFX_OK;
}

// the stream the halo's pack and transfer are enqueued on
static hipStream_t xfer_stream(const sfx_world* w) { return w->halo_overlap ? w->halo_stream : w->stream; }

int sfx_halo_stream(sfx_world* w, void** hip_stream) {
    if (!w || !hip_stream) return SFX_ERR_INVALID;
    *hip_stream = reinterpret_cast<void*>(xfer_stream(w));
    return SFX_OK;
}

int sfx_halo_overlap(sfx_world* w, int on) {
    if (!w) return SFX_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    if (w->halo_stream) HIPCHK(w, hipStreamSynchronize(w->halo_stream));
    if (w->tick_open) return fail(w, SFX_ERR_INVALID, "sfx_halo_overlap inside a tick");
    if (on && !w->halo_stream) {
        HIPCHK(w, hipStreamCreateWithFlags(&w->halo_stream, hipStreamNonBlocking));
        HIPCHK(w, hipEventCreateWithFlags(&w->ev_pre, hipEventDisableTiming));
        HIPCHK(w, hipEventCreateWithFlags(&w->ev_near, hipEventDisableTiming));
        HIPCHK(w, hipEventCreateWithFlags(&w->ev_xchg, hipEventDisableTiming));
    }
    if (on && w->near_list.n < 4u * (size_t)w->ntiles) ALLOC(w, w->near_list, 4u * (size_t)std::max(w->ntiles, 1u));
    w->halo_overlap = on != 0;
    w->near_open = false;
    return SFX_OK;
}

int sfx_halo_unpack(sfx_world* w) {
    if (!w) return SFX_ERR_INVALID;
    if (!w->n_halo_recv) return SFX_OK;
    HIPCHK(w, hipSetDevice(w->device));
    w->q_valid = false;
    if (w->halo_overlap) {   // the world's stream goes on when the transfer on the halo stream has arrived
        HIPCHK(w, hipEventRecord(w->ev_xchg, w->halo_stream));
        HIPCHK(w, hipStreamWaitEvent(w->stream, w->ev_xchg, 0));
    }
    Prof p(w, SFX_K_HALO, w->n_halo_recv);
    hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv(w->n_halo_recv, kBlock)), dim3(kBlock), 0, w->stream, w->n_halo_recv,
                       w->halo_recv_idx.p, w->halo_recv.p, w->pose.p, w->vel.p);
    return SFX_OK;
}

int sfx_halo_buffers(sfx_world* w, float** send_dev, float** recv_dev) {
    if (!w) return SFX_ERR_INVALID;
    if (send_dev) *send_dev = reinterpret_cast<float*>(w->halo_send.p);
    if (recv_dev) *recv_dev = reinterpret_cast<float*>(w->halo_recv.p);
    return SFX_OK;
}

int sfx_halo_download(sfx_world* w, float* send_host) {
    if (!w || (w->n_halo_send && !send_host)) return fail(w, SFX_ERR_INVALID, "null argument");
    int rc = sfx_halo_pack(w);
    if (rc) return rc;
    if (w->n_halo_send)
        HIPCHK(w, hipMemcpyAsync(send_host, w->halo_send.p, sizeof(float4) * 2 * w->n_halo_send, hipMemcpyDeviceToHost, xfer_stream(w)));
    HIPCHK(w, hipStreamSynchronize(xfer_stream(w)));
    return SFX_OK;
}

int sfx_halo_upload(sfx_world* w, const float* recv_host) {
    if (!w || (w->n_halo_recv && !recv_host)) return fail(w, SFX_ERR_INVALID, "null argument");
    if (!w->n_halo_recv) return SFX_OK;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipMemcpyAsync(w->halo_recv.p, recv_host, sizeof(float4) * 2 * w->n_halo_recv, hipMemcpyHostToDevice, xfer_stream(w)));
    HIPCHK(w, hipStreamSynchronize(xfer_stream(w)));  // recv_host is borrowed for this call only
    return sfx_halo_unpack(w);
}

int sfx_halo_copy(sfx_world* dst, uint32_t dst_first, sfx_world* src, uint32_t src_first, uint32_t count) {
    if (!dst || !src) return SFX_ERR_INVALID;
    if (dst->device != src->device) return fail(dst, SFX_ERR_UNSUPPORTED, "sfx_halo_copy needs both worlds on one device");
    if ((uint64_t)dst_first + count > dst->n_halo_recv || (uint64_t)src_first + count > src->n_halo_send)
        return fail(dst, SFX_ERR_INVALID, "halo copy range out of bounds");
    if (!count) return SFX_OK;
    HIPCHK(dst, hipSetDevice(dst->device));
    // ordered after the source's pack and before the destination's unpack: on the streams the halos travel on
    hipStream_t ss = xfer_stream(src), ds = xfer_stream(dst);
    if (ss != ds) {
        hipEvent_t ev;
        HIPCHK(dst, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(dst, hipEventRecord(ev, ss));
        HIPCHK(dst, hipStreamWaitEvent(ds, ev, 0));
        HIPCHK(dst, hipEventDestroy(ev));
    }
    HIPCHK(dst, hipMemcpyAsync(dst->halo_recv.p + 2 * (size_t)dst_first, src->halo_send.p + 2 * (size_t)src_first,
                               sizeof(float4) * 2 * count, hipMemcpyDeviceToDevice, ds));
    if (ss != ds) {  // the source must not pack again before this copy has read its buffer
        hipEvent_t ev;
        HIPCHK(dst, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIPCHK(dst, hipEventRecord(ev, ds));
        HIPCHK(dst, hipStreamWaitEvent(ss, ev, 0));
        HIPCHK(dst, hipEventDestroy(ev));
    }
    return SFX_OK;
}

// ---- spatial queries ------------------------------------------------------------------
namespace {

// The buffers of a query index: the library's own (allocated at the first query) or carved out of the static device
// scratch array of sfx_query_points_static.
struct QueryBuffers {
    uint32_t* zero;      // [Counters][cell_start: ncell + 1 + pad], cleared by k_zero
    size_t zero_bytes;
    uint32_t *cell_fill, *bucket, *tmp_id, *sorted_id, *large_tmp, *bounds;
    float4 *aabb, *sorted_aabb;
    int2 *cellxy, *sorted_cell;
};

size_t query_zero_words(const sfx_world* w) {
    const size_t words = sizeof(Counters) / 4 + ((size_t)w->ncell + 1 + kScanItems);
    return (words + 3) & ~(size_t)3;  // cleared in 16-byte units
}

// builds the query index at the current poses into Q: the broadphase's counting sort on tight AABBs
// (margin, pad and dt = 0, row-major buckets), then the bounds of the small boxes.  Enqueues only.
int build_query_index(sfx_world* w, const QueryBuffers& Q) {
    const uint32_t n = w->n;
    Counters* qctr = reinterpret_cast<Counters*>(Q.zero);
    uint32_t* qcs = Q.zero + sizeof(Counters) / 4;
    GridParams g = grid_params(w);
    g.tlx = 0; g.tly = 0; g.ntiles = w->ncell; g.pstride = 0;  // plain row-major buckets
    g.ncell_all = w->ncell; g.medium_lim = 0.0f;                 // the query index has one level
    const uint32_t gb = cdiv(std::max(n, 1u), kBlock);
    { Prof p(w, SFX_K_QUERY, 0);
      const uint32_t n16 = (uint32_t)(Q.zero_bytes / 16);
      hipLaunchKernelGGL(k_zero, dim3(cdiv(n16, kBlock)), dim3(kBlock), 0, w->stream, reinterpret_cast<uint4*>(Q.zero), n16, (uint32_t*)nullptr); }
    HIPCHK(w, hipMemsetAsync(Q.bounds, 0, 4 * sizeof(uint32_t), w->stream));
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_aabb_count, dim3(gb), dim3(kBlock), 0, w->stream, n, w->pose.p, w->vel.p, w->shape.p, w->lpose.p, w->poly.p,
                         0.0f, 0.0f, 0.0f, g, Q.aabb, Q.bucket, Q.cellxy, qcs, Q.large_tmp, w->large_cap, qctr,
                         (uint4*)nullptr, 0u, (uint32_t*)nullptr, (const float*)nullptr, 0.0f); }
    int rc = scan_u32(w, qcs, w->ncell + 1, Q.cell_fill);
    if (rc) return rc;
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_scatter, dim3(gb), dim3(kBlock), 0, w->stream, n, Q.bucket, Q.cell_fill, Q.tmp_id); }
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_finalize_slots, dim3(gb), dim3(kBlock), 0, w->stream, n, g, qcs, Q.tmp_id, Q.bucket,
                         Q.aabb, Q.cellxy, Q.large_tmp, w->large_cap, Q.sorted_id, Q.sorted_aabb,
                         Q.sorted_cell, qctr, (const uint4*)nullptr, (uint4*)nullptr, (uint32_t*)nullptr); }
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_query_bounds, dim3(std::min(gb, 1024u)), dim3(kBlock), 0, w->stream, qcs, w->ncell,
                         Q.sorted_aabb, Q.bounds); }
    HIPCHK(w, hipGetLastError());
    return SFX_OK;
}

QueryBuffers own_query_buffers(sfx_world* w) {
    QueryBuffers Q{};
    Q.zero = w->q_zero.p; Q.zero_bytes = w->q_zero_bytes;
    Q.cell_fill = w->q_cell_fill.p; Q.bucket = w->q_bucket.p; Q.tmp_id = w->q_tmp_id.p; Q.sorted_id = w->q_sorted_id.p;
    Q.large_tmp = w->q_large_tmp.p; Q.bounds = w->q_bounds.p;
    Q.aabb = w->q_aabb.p; Q.sorted_aabb = w->q_sorted_aabb.p; Q.cellxy = w->q_cellxy.p; Q.sorted_cell = w->q_sorted_cell.p;
    return Q;
}

// (re)builds the library-owned query index when the poses changed since the last query
int ensure_query_index(sfx_world* w) {
    int rc = ready(w);
    if (rc) return rc;
    if (w->tick_open) return fail(w, SFX_ERR_INVALID, "spatial queries are not allowed inside an open tick");
    HIPCHK(w, hipSetDevice(w->device));
    const uint32_t n = w->n;
    const size_t pad = kScanItems;
    if (w->q_n != n || w->q_ncell != w->ncell || !w->q_zero.p) {
        const size_t c = std::max(n, 1u);
        const size_t words16 = query_zero_words(w);
        ALLOC(w, w->q_zero, words16);
        w->q_zero_bytes = words16 * 4;
        ALLOC(w, w->q_cell_fill, (size_t)w->ncell + 1 + pad);
        ALLOC(w, w->q_bucket, c); ALLOC(w, w->q_tmp_id, c); ALLOC(w, w->q_sorted_id, c); ALLOC(w, w->q_large_tmp, c);
        ALLOC(w, w->q_aabb, c); ALLOC(w, w->q_sorted_aabb, c); ALLOC(w, w->q_cellxy, c); ALLOC(w, w->q_sorted_cell, c);
        ALLOC(w, w->q_bounds, 4); ALLOC(w, w->q_nhits, 1);
        w->q_n = n; w->q_ncell = w->ncell;
        w->q_valid = false;
    }
    if (w->q_valid) return SFX_OK;
    rc = build_query_index(w, own_query_buffers(w));
    if (rc) return rc;
    w->q_valid = true;
    return SFX_OK;
}

// transfer buffers of the queries: grown when a call needs more, never shrunk
int query_buffers(sfx_world* w, size_t in_bytes, size_t out_bytes) {
    if (w->q_in.n < in_bytes) ALLOC(w, w->q_in, in_bytes + in_bytes / 2);
    if (w->q_out.n < out_bytes) ALLOC(w, w->q_out, out_bytes + out_bytes / 2);
    return SFX_OK;
}

QueryIndex query_index_of(sfx_world* w, const QueryBuffers& B) {
    QueryIndex Q;
    Q.cell_start = B.zero + sizeof(Counters) / 4;
    Q.sorted_id = B.sorted_id; Q.sorted_aabb = B.sorted_aabb; Q.sorted_cell = B.sorted_cell;
    Q.ctr = reinterpret_cast<const Counters*>(B.zero);
    Q.bounds = B.bounds;
    Q.gxl = w->gxl; Q.gyl = w->gyl; Q.ncell = w->ncell; Q.large_cap = w->large_cap;
    Q.inv_cell = w->inv_cell; Q.cell = w->cell;
    return Q;
}

QueryIndex query_index(sfx_world* w) { return query_index_of(w, own_query_buffers(w)); }

}  // namespace

int sfx_query_points(sfx_world* w, uint32_t n, const float* points, uint32_t mask, uint32_t* body, uint32_t* count) {
    if (!w || (n && (!points || !body))) return fail(w, SFX_ERR_INVALID, "null argument");
    int rc = ensure_query_index(w);
    if (rc || n == 0) return rc;
    rc = query_buffers(w, sizeof(float2) * n, 2 * sizeof(uint32_t) * (size_t)n);
    if (rc) return rc;
    float2* d_pts = reinterpret_cast<float2*>(w->q_in.p);
    uint32_t* d_out = reinterpret_cast<uint32_t*>(w->q_out.p);
    HIPCHK(w, hipMemcpyAsync(d_pts, points, sizeof(float2) * n, hipMemcpyHostToDevice, w->stream));
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_query_points, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, w->stream, n, d_pts, mask, query_index(w),
                         body_arrays(w), d_out, d_out + n); }
    hipError_t e = hipMemcpyAsync(body, d_out, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, w->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(count, d_out + n, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, w->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
    if (e != hipSuccess) return fail(w, SFX_ERR_DEVICE, "point query failed: %s", hipGetErrorString(e));
    return w->profiling ? collect_profile(w) : SFX_OK;
}

int sfx_query_aabbs(sfx_world* w, uint32_t n, const float* boxes, uint32_t mask, uint64_t* hits, size_t cap, size_t* n_hits) {
    if (!w || (n && !boxes) || !n_hits || (cap && !hits)) return fail(w, SFX_ERR_INVALID, "null argument");
    *n_hits = 0;
    int rc = ensure_query_index(w);
    if (rc || n == 0) return rc;
    const uint32_t dcap = (uint32_t)std::min<size_t>(cap, 0xFFFFFFF0u);
    rc = query_buffers(w, sizeof(float4) * n, sizeof(unsigned long long) * std::max<size_t>(dcap, 1));
    if (rc) return rc;
    float4* d_box = reinterpret_cast<float4*>(w->q_in.p);
    unsigned long long* d_hits = reinterpret_cast<unsigned long long*>(w->q_out.p);
    HIPCHK(w, hipMemcpyAsync(d_box, boxes, sizeof(float4) * n, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipMemsetAsync(w->q_nhits.p, 0, sizeof(uint32_t), w->stream));
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_query_aabbs, dim3(std::min(cdiv(n, kBlock / 64), 4096u)), dim3(kBlock), 0, w->stream, n, d_box, mask,
                         query_index(w), body_arrays(w), d_hits, dcap, w->q_nhits.p, 0); }
    uint32_t total = 0;
    hipError_t e = hipMemcpyAsync(&total, w->q_nhits.p, sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
    const size_t got = std::min<size_t>(total, dcap);
    if (e == hipSuccess && got) e = hipMemcpy(hits, d_hits, sizeof(uint64_t) * got, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(w, SFX_ERR_DEVICE, "box query failed: %s", hipGetErrorString(e));
    size_t kept = got;
    if (w->n > w->nb && total <= dcap) {   // the kernel names colliders: their bodies, each once per query
        for (size_t h = 0; h < got; ++h) hits[h] = (hits[h] & 0xFFFFFFFF00000000ull) | owner(w, (uint32_t)hits[h]);
        std::sort(hits, hits + got);
        kept = (size_t)(std::unique(hits, hits + got) - hits);
    } else {
        std::sort(hits, hits + got);
    }
    *n_hits = total > dcap ? total : kept;
    if (w->profiling) { rc = collect_profile(w); if (rc) return rc; }
    if (total > dcap) return fail(w, SFX_ERR_CAPACITY, "box query found %u hits, room for %zu", total, cap);
    return SFX_OK;
}

int sfx_query_rays(sfx_world* w, uint32_t n, const float* rays, const float* max_t, uint32_t mask, sfx_ray_hit* hits) {
    if (!w || (n && (!rays || !max_t || !hits))) return fail(w, SFX_ERR_INVALID, "null argument");
    int rc = ensure_query_index(w);
    if (rc || n == 0) return rc;
    rc = query_buffers(w, (sizeof(float4) + sizeof(float)) * (size_t)n, sizeof(sfx_ray_hit) * (size_t)n);
    if (rc) return rc;
    float4* d_rays = reinterpret_cast<float4*>(w->q_in.p);
    float* d_mt = reinterpret_cast<float*>(d_rays + n);
    sfx_ray_hit* d_out = reinterpret_cast<sfx_ray_hit*>(w->q_out.p);
    HIPCHK(w, hipMemcpyAsync(d_rays, rays, sizeof(float4) * n, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipMemcpyAsync(d_mt, max_t, sizeof(float) * n, hipMemcpyHostToDevice, w->stream));
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_query_rays, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, w->stream, n, d_rays, d_mt, mask, query_index(w),
                         body_arrays(w), d_out); }
    hipError_t e = hipMemcpyAsync(hits, d_out, sizeof(sfx_ray_hit) * n, hipMemcpyDeviceToHost, w->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
    if (e != hipSuccess) return fail(w, SFX_ERR_DEVICE, "ray query failed: %s", hipGetErrorString(e));
    return w->profiling ? collect_profile(w) : SFX_OK;
}

// Box and ray queries for callers that keep their data on the device: device pointers in and out,
// everything enqueued on the world's stream, no copy and no synchronisation.  The query index is the library's own (built
// by the first query after the poses changed; its buffers are allocated by the first query of a world, like the host
// variants'), so only sfx_query_points_static is the entry point that allocates nothing.
int sfx_query_aabbs_device(sfx_world* w, uint32_t n, const float* boxes_dev, uint32_t mask, uint64_t* hits_dev, uint32_t cap,
                           uint32_t* n_hits_dev) {
    if (!w || (n && !boxes_dev) || !n_hits_dev || (cap && !hits_dev)) return fail(w, SFX_ERR_INVALID, "null argument");
    int rc = ensure_query_index(w);
    if (rc) return rc;
    HIPCHK(w, hipMemsetAsync(n_hits_dev, 0, sizeof(uint32_t), w->stream));
    if (n == 0) return SFX_OK;
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_query_aabbs, dim3(std::min(cdiv(n, kBlock / 64), 4096u)), dim3(kBlock), 0, w->stream, n,
                         reinterpret_cast<const float4*>(boxes_dev), mask, query_index(w), body_arrays(w),
                         reinterpret_cast<unsigned long long*>(hits_dev), cap, n_hits_dev, 1); }
    HIPCHK(w, hipGetLastError());
    return SFX_OK;
}

int sfx_query_rays_device(sfx_world* w, uint32_t n, const float* rays_dev, const float* max_t_dev, uint32_t mask,
                          sfx_ray_hit* hits_dev) {
    if (!w || (n && (!rays_dev || !max_t_dev || !hits_dev))) return fail(w, SFX_ERR_INVALID, "null argument");
    int rc = ensure_query_index(w);
    if (rc || n == 0) return rc;
    { Prof p(w, SFX_K_QUERY, n);
      hipLaunchKernelGGL(k_query_rays, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, w->stream, n, reinterpret_cast<const float4*>(rays_dev),
                         max_t_dev, mask, query_index(w), body_arrays(w), hits_dev); }
    HIPCHK(w, hipGetLastError());
    return SFX_OK;
}

// The point query without any allocation: same kernels as sfx_query_points, the query index in the static device
// scratch array instead of library-allocated buffers, inputs and outputs in DEVICE memory, everything enqueued on the
// world's stream (no copy, no synchronisation).  The index is rebuilt by every call (the scratch is shared by all
// worlds of the device and keeps the last call's contents: the counters, the cell histogram and the bounds are
// cleared first).  Calls that overlap on one GPU (two streams, two threads, two processes do NOT share it: the
// array is per process) race on the scratch.
int sfx_query_points_static(sfx_world* w, uint32_t n, const float* points_dev, uint32_t mask, uint32_t* body_dev,
                            uint32_t* count_dev) {
    if (!w || (n && (!points_dev || !body_dev))) return fail(w, SFX_ERR_INVALID, "null argument");
    int rc = ready(w);
    if (rc) return rc;
    if (w->tick_open) return fail(w, SFX_ERR_INVALID, "spatial queries are not allowed inside an open tick");
    if (w->n > kStaticMaxBodies || w->ncell > kStaticMaxCells)
        return fail(w, SFX_ERR_UNSUPPORTED, "sfx_query_points_static: %u bodies / %u grid cells exceed the static scratch (%u bodies, %u cells); use sfx_query_points",
                    w->n, w->ncell, kStaticMaxBodies, kStaticMaxCells);
    if (w->profiling || w->prof_mask) return fail(w, SFX_ERR_INVALID, "sfx_query_points_static: switch profiling off (event pools allocate)");
    if (w->device >= kMaxDevices || !g_scratch_ptr[w->device]) return fail(w, SFX_ERR_DEVICE, "static scratch address unknown for device %d", w->device);
    HIPCHK(w, hipSetDevice(w->device));
    if (n == 0) return SFX_OK;
    // carve the index out of the scratch through its DEVICE address (never through the symbol's host-side name)
    unsigned char* ptr = g_scratch_ptr[w->device];
    QueryBuffers Q{};
    const size_t nb = kStaticMaxBodies;
    Q.zero = reinterpret_cast<uint32_t*>(ptr); Q.zero_bytes = query_zero_words(w) * 4; ptr += kStaticZeroBytes;
    Q.cell_fill = reinterpret_cast<uint32_t*>(ptr); ptr += kStaticFillBytes;
    Q.bounds = reinterpret_cast<uint32_t*>(ptr); ptr += 256;
    Q.aabb = reinterpret_cast<float4*>(ptr); ptr += 16 * nb;
    Q.sorted_aabb = reinterpret_cast<float4*>(ptr); ptr += 16 * nb;
    Q.cellxy = reinterpret_cast<int2*>(ptr); ptr += 8 * nb;
    Q.sorted_cell = reinterpret_cast<int2*>(ptr); ptr += 8 * nb;
    Q.bucket = reinterpret_cast<uint32_t*>(ptr); ptr += 4 * nb;
    Q.tmp_id = reinterpret_cast<uint32_t*>(ptr); ptr += 4 * nb;
    Q.sorted_id = reinterpret_cast<uint32_t*>(ptr); ptr += 4 * nb;
    Q.large_tmp = reinterpret_cast<uint32_t*>(ptr); ptr += 4 * nb;
    static_assert(kStaticZeroBytes + kStaticFillBytes + 256 + (size_t)kStaticMaxBodies * 64 == kStaticScratchBytes, "scratch layout");
    if (Q.zero_bytes > kStaticZeroBytes) return fail(w, SFX_ERR_UNSUPPORTED, "sfx_query_points_static: grid too large for the scratch");
    rc = build_query_index(w, Q);
    if (rc) return rc;
    hipLaunchKernelGGL(k_query_points, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, w->stream, n, reinterpret_cast<const float2*>(points_dev),
                       mask, query_index_of(w, Q), body_arrays(w), body_dev, count_dev);
    HIPCHK(w, hipGetLastError());
    return SFX_OK;
}

int sfx_sync(sfx_world* w) {
    if (!w) return SFX_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    poll_counters(w);
    // Overflow bits live in three places: the tick's own counter block (the last tick's), the device's sticky word
    // (every earlier tick's: folded in when its block is cleared, two ticks later) and what the asynchronous counter
    // copies showed the host.  A sync point reports all of them, once, and clears them.
    uint32_t bits = w->ovf_host, dev_sticky = 0;
    HIPCHK(w, hipMemcpy(&dev_sticky, w->sticky.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    bits |= dev_sticky;
    if (w->ticked && w->have_colliders && !w->tick_open) {
        int rc = sync_counters(w, &w->stale, w->ctr);
        if (rc) return rc;
        w->ctr_skipped = 0;
        adopt_counters(w);
        bits |= w->stale.overflow;
    }
    if (bits) {
        w->ovf_host = 0;
        HIPCHK(w, hipMemset(w->sticky.p, 0, sizeof(uint32_t)));
        if (w->ticked && w->have_colliders && !w->tick_open) { HIPCHK(w, hipMemset(&w->ctr->overflow, 0, sizeof(uint32_t))); w->stale.overflow = 0; }
        return fail(w, SFX_ERR_CAPACITY, "capacity exceeded in a tick since the last sfx_sync (bits %u: 1 pairs, 2 colours, 4 serial colouring, 8 large list, 16 fallback list, 32 a fused solve launch gave up waiting, 64 more than 64 large colliders over one batch of tiles, 128 a body moved beyond the halo guard: cut the slabs again, 256 two pairs of one colour wrote the same body [libsfx_races.so only])",
                    bits);
    }
    return SFX_OK;
}

int sfx_download_poses(sfx_world* w, float* pos, float* rot) {
    if (!w || !w->have_bodies) return fail(w, SFX_ERR_INVALID, "no bodies");
    HIPCHK(w, hipSetDevice(w->device));
    std::vector<float4> hp(std::max(w->nb, 1u));
    HIPCHK(w, hipMemcpyAsync(hp.data(), w->pose.p, sizeof(float4) * w->nb, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (uint32_t i = 0; i < w->nb; ++i) {
        if (pos) { pos[2 * i] = hp[i].x; pos[2 * i + 1] = hp[i].y; }
        if (rot) { rot[2 * i] = hp[i].z; rot[2 * i + 1] = hp[i].w; }
    }
    return SFX_OK;
}

// ---- pose read-back for rendering: the copy to the host overlaps the next tick
int sfx_readback_begin(sfx_world* w) {
    if (!w || !w->have_bodies) return fail(w, SFX_ERR_INVALID, "no bodies");
    if (w->tick_open) return fail(w, SFX_ERR_INVALID, "sfx_readback_begin inside a tick");
    if (w->rb_head - w->rb_tail >= 2) return fail(w, SFX_ERR_INVALID, "two read-backs are already in flight: call sfx_readback_wait first");
    HIPCHK(w, hipSetDevice(w->device));
    if (!w->rb_stream) {
        HIPCHK(w, hipStreamCreateWithFlags(&w->rb_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(w, hipEventCreateWithFlags(&w->rb_snap[k], hipEventDisableTiming));
            HIPCHK(w, hipEventCreateWithFlags(&w->rb_done[k], hipEventDisableTiming));
        }
    }
    if (w->rb_cap < w->nb) {   // first use, or more bodies than before: nothing is in flight in a slot that is free
        if (w->rb_head != w->rb_tail) return fail(w, SFX_ERR_INVALID, "the body count grew while a read-back is in flight");
        for (int k = 0; k < 2; ++k) {
            if (w->rb_host[k]) { HIPCHK(w, hipHostFree(w->rb_host[k])); w->rb_host[k] = nullptr; }
            ALLOC(w, w->rb_dev[k], w->nb);
            HIPCHK(w, hipHostMalloc((void**)&w->rb_host[k], sizeof(float4) * w->nb, hipHostMallocDefault));
        }
        w->rb_cap = w->nb;
    }
    const int k = (int)(w->rb_head & 1u);
    const size_t bytes = sizeof(float4) * w->nb;
    // snapshot in stream order (the next tick may overwrite the poses), then the transfer on its own stream
    HIPCHK(w, hipMemcpyAsync(w->rb_dev[k].p, w->pose.p, bytes, hipMemcpyDeviceToDevice, w->stream));
    HIPCHK(w, hipEventRecord(w->rb_snap[k], w->stream));
    HIPCHK(w, hipStreamWaitEvent(w->rb_stream, w->rb_snap[k], 0));
    HIPCHK(w, hipMemcpyAsync(w->rb_host[k], w->rb_dev[k].p, bytes, hipMemcpyDeviceToHost, w->rb_stream));
    HIPCHK(w, hipEventRecord(w->rb_done[k], w->rb_stream));
    w->rb_count[k] = w->nb;
    w->rb_head++;
    return SFX_OK;
}

int sfx_readback_wait(sfx_world* w, const float** poses, uint32_t* n) {
    if (!w || !poses) return fail(w, SFX_ERR_INVALID, "null argument");
    if (w->rb_head == w->rb_tail) return fail(w, SFX_ERR_INVALID, "sfx_readback_wait without sfx_readback_begin");
    HIPCHK(w, hipSetDevice(w->device));
    const int k = (int)(w->rb_tail & 1u);
    HIPCHK(w, hipEventSynchronize(w->rb_done[k]));
    *poses = reinterpret_cast<const float*>(w->rb_host[k]);
    if (n) *n = w->rb_count[k];
    w->rb_tail++;
    return SFX_OK;
}

int sfx_download_velocities(sfx_world* w, float* vel, float* angvel) {
    if (!w || !w->have_bodies) return fail(w, SFX_ERR_INVALID, "no bodies");
    HIPCHK(w, hipSetDevice(w->device));
    std::vector<float4> hv(std::max(w->nb, 1u));
    HIPCHK(w, hipMemcpyAsync(hv.data(), w->vel.p, sizeof(float4) * w->nb, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (uint32_t i = 0; i < w->nb; ++i) {
        if (vel) { vel[2 * i] = hv[i].x; vel[2 * i + 1] = hv[i].y; }
        if (angvel) angvel[i] = hv[i].z;
    }
    return SFX_OK;
}

int sfx_debug_aabbs(sfx_world* w, float* out) {
    int rc = ready(w);
    if (rc) return rc;
    if (!out || !w->ticked) return fail(w, SFX_ERR_INVALID, "no broadphase has run");
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipMemcpyAsync(out, w->aabb.p, sizeof(float4) * w->n, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return SFX_OK;
}

// all candidate pairs of the last broadphase: tile-owned pairs (with owner tile and colour),
// then the fallback list
struct HostPair { uint32_t a, b, tile; uint8_t color; };

static int fetch_pairs(sfx_world* w, std::vector<HostPair>& out, Counters& c) {
    int rc = ready(w);
    if (rc) return rc;
    if (!w->ticked) return fail(w, SFX_ERR_INVALID, "no broadphase has run");
    HIPCHK(w, hipSetDevice(w->device));
    rc = sync_counters(w, &c, w->ctr);
    if (rc) return rc;
    out.clear();
    const uint32_t ntp = c.n_tile_pairs, nfb = c.n_fb;
    std::vector<uint4> tp(ntp);
    if (ntp) HIPCHK(w, hipMemcpy(tp.data(), w->tpair.p, sizeof(uint4) * ntp, hipMemcpyDeviceToHost));
    out.reserve((size_t)ntp + nfb);
    for (uint32_t k = 0; k < ntp; ++k) {   // unit = class << 28 | owner tile (include/sfx.h sfx_debug_pair_colors)
        const uint32_t u = tp[k].w >> 8;
        out.push_back({tp[k].x, tp[k].y, ((u >> 22) << 28) | (u & 0x3FFFFFu), (uint8_t)(tp[k].w & 0xFFu)});
    }
    if (nfb) {
        // the raw list holds every fallback pair; colours come from the colour-major copy
        std::vector<uint2> fp(nfb), fs(c.fb_start[SFX_MAX_COLORS]);
        HIPCHK(w, hipMemcpy(fp.data(), w->fpairs.p, sizeof(uint2) * nfb, hipMemcpyDeviceToHost));
        if (!fs.empty()) HIPCHK(w, hipMemcpy(fs.data(), w->fsorted.p, sizeof(uint2) * fs.size(), hipMemcpyDeviceToHost));
        std::vector<std::pair<uint64_t, uint8_t>> col;
        col.reserve(fs.size());
        for (uint32_t q = 0; q < SFX_MAX_COLORS; ++q)
            for (uint32_t k = c.fb_start[q]; k < c.fb_start[q + 1]; ++k)
                col.emplace_back(((uint64_t)fs[k].x << 32) | fs[k].y, (uint8_t)q);
        std::sort(col.begin(), col.end());
        for (uint32_t k = 0; k < nfb; ++k) {
            const uint64_t key = ((uint64_t)fp[k].x << 32) | fp[k].y;
            auto it = std::lower_bound(col.begin(), col.end(), std::make_pair(key, (uint8_t)0));
            const uint8_t cc = (it != col.end() && it->first == key) ? it->second : (uint8_t)SFX_NO_COLOR;
            out.push_back({fp[k].x, fp[k].y, SFX_FALLBACK_TILE, cc});
        }
    }
    return SFX_OK;
}

int sfx_debug_pairs(sfx_world* w, uint64_t* keys, size_t cap, size_t* n) {
    std::vector<HostPair> p;
    Counters c;
    int rc = fetch_pairs(w, p, c);
    if (rc) return rc;
    std::vector<uint64_t> k(p.size());
    for (size_t i = 0; i < p.size(); ++i) k[i] = ((uint64_t)p[i].a << 32) | p[i].b;
    std::sort(k.begin(), k.end());
    if (n) *n = k.size();
    if (keys) memcpy(keys, k.data(), sizeof(uint64_t) * std::min(cap, k.size()));
    return SFX_OK;
}

int sfx_debug_pair_colors(sfx_world* w, uint64_t* keys, uint8_t* colors, uint32_t* tiles, size_t cap, size_t* n) {
    std::vector<HostPair> p;
    Counters c;
    int rc = fetch_pairs(w, p, c);
    if (rc) return rc;
    if (n) *n = p.size();
    for (size_t k = 0; k < p.size() && k < cap; ++k) {
        if (keys) keys[k] = ((uint64_t)p[k].a << 32) | p[k].b;
        if (colors) colors[k] = p[k].color;
        if (tiles) tiles[k] = p[k].tile;
    }
    return SFX_OK;
}

int sfx_debug_contacts(sfx_world* w, sfx_contact* out, size_t cap, size_t* n) {
    std::vector<HostPair> hp;
    Counters c;
    int rc = fetch_pairs(w, hp, c);
    if (rc) return rc;
    std::vector<uint2> p(hp.size());
    for (size_t i = 0; i < hp.size(); ++i) p[i] = make_uint2(hp[i].a, hp[i].b);
    std::sort(p.begin(), p.end(), [](const uint2& a, const uint2& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    if (n) *n = p.size();
    const uint32_t np = (uint32_t)std::min(cap, p.size());
    if (!np || !out) return SFX_OK;
    DevBuf<uint2> dp;
    DevBuf<sfx_contact> dc;
    ALLOC(w, dp, np); ALLOC(w, dc, np);
    hipError_t e = hipMemcpy(dp.p, p.data(), sizeof(uint2) * np, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_contacts_debug, dim3(cdiv(np, kBlock)), dim3(kBlock), 0, w->stream, np, dp.p, body_arrays(w),
                           w->cfg.speculative_margin, dc.p);
        e = hipStreamSynchronize(w->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dc.p, sizeof(sfx_contact) * np, hipMemcpyDeviceToHost);
    dp.release(); dc.release();
    if (e != hipSuccess) return fail(w, SFX_ERR_DEVICE, "contact debug pass failed: %s", hipGetErrorString(e));
    return SFX_OK;
}

int sfx_debug_joint_order(sfx_world* w, uint32_t* perm, uint8_t* colors, size_t cap, size_t* n) {
    if (!w) return SFX_ERR_INVALID;
    if (n) *n = w->m;
    for (size_t k = 0; k < w->m && k < cap; ++k) {
        if (perm) perm[k] = w->joint_perm[k];
        if (colors) colors[k] = w->joint_color_sorted[k];
    }
    return SFX_OK;
}

int sfx_debug_near_batches(sfx_world* w, uint32_t* n_near, uint32_t* n_batches) {
    if (!w || !n_near || !n_batches) return SFX_ERR_INVALID;
    *n_near = 0; *n_batches = 0;
    if (!w->ticked || !w->ctr) return SFX_OK;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    Counters c;
    HIPCHK(w, hipMemcpy(&c, w->ctr, sizeof(Counters), hipMemcpyDeviceToHost));
    *n_near = c.n_near;
    for (int ph = 0; ph < 4; ++ph) *n_batches += c.n_batches[ph];
    return SFX_OK;
}

int sfx_stats(sfx_world* w, sfx_frame_stats* out) {
    if (!w || !out) return SFX_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    out->n_bodies = w->nb;
    out->n_joint_colors = (uint32_t)w->joint_color_start.size() - 1;
    out->launches = w->launches;
    if (!w->have_bodies || !w->have_colliders || !w->ticked) return SFX_OK;
    HIPCHK(w, hipSetDevice(w->device));
    const uint32_t saved = w->launches;
    // the stats live in their own counter block so the tick's counters stay untouched
    HIPCHK(w, hipMemsetAsync(w->jctr.p, 0, sizeof(Counters), w->stream));
    const uint32_t grid = std::min(std::max(cdiv(std::max(w->n, 1u), kBlock), 1u), 1024u);
    hipLaunchKernelGGL(k_stats_bodies, dim3(grid), dim3(kBlock), 0, w->stream, w->n, w->vel.p, w->mass.p, w->jctr.p);
    if (w->have_mats)
        hipLaunchKernelGGL(k_stats_pairs, dim3(edge_grid(w->max_pairs)), dim3(kBlock), 0, w->stream, w->ctr, w->jctr.p,
                           tile_arrays(w), body_arrays(w), contact_arrays(w), w->cfg.speculative_margin);
    Counters c{}, s{};
    int rc = sync_counters(w, &s, w->jctr.p);
    if (rc) return rc;
    rc = sync_counters(w, &c, w->ctr);
    if (rc) return rc;
    w->launches = saved;
    out->n_small = c.n_small; out->n_large = c.n_large + c.n_medium;   /* everything not in a base cell */ out->n_pairs = c.n_pairs; out->n_colors = c.n_colors;
    uint32_t dev_sticky = 0;
    HIPCHK(w, hipMemcpy(&dev_sticky, w->sticky.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    out->n_leftover = c.n_leftover; out->overflow = c.overflow | dev_sticky | w->ovf_host;   // this tick's and every unreported earlier one's
    out->n_tiles = c.n_active[0];
    out->n_fallback = c.n_fb; out->max_tile_pairs = c.max_tile_pairs; out->max_tile_bodies = c.max_tile_bodies;
    out->n_solver_pairs = c.n_kept + std::min(c.n_fb, w->fb_cap);
    out->n_contacts = s.n_contacts;
    out->n_manifolds = s.n_manifolds;
    out->n_manifolds2 = s.n_manifolds2;
    memcpy(&out->max_depth, &s.max_depth_bits, 4);
    out->kinetic_energy = s.ke;
    for (int k = 0; k < SFX_MAX_COLORS; ++k) out->color_count[k] = c.color_count[k] + s.color_count[k];  // fallback list + tiles
    return SFX_OK;
}

int sfx_set_profiling(sfx_world* w, int on) {
    if (!w) return SFX_ERR_INVALID;
    if (!on && (w->profiling || w->prof_mask)) collect_profile(w);
    w->profiling = on == 1;                           // every launch, collected (with a synchronisation) at the end of each tick
    w->prof_mask = on > 1 ? (uint32_t)on >> 1 : 0u;   // chosen classes only, collected by sfx_kernel_times / when switched off
    return SFX_OK;
}

int sfx_kernel_times(sfx_world* w, sfx_kernel_time* out, int reset) {
    if (!w || !out) return SFX_ERR_INVALID;
    int rc = collect_profile(w);
    if (rc) return rc;
    memcpy(out, w->ktime, sizeof(w->ktime));
    if (reset) memset(w->ktime, 0, sizeof(w->ktime));
    return SFX_OK;
}

#if defined(SFX_TILE_TIMING) || defined(SFX_STEP_PROBE)
// development hook: stage timing in 10 ns units, 16 slots.  k_tile_pos: [load, narrowphase, colour loop, whole batch,
// batches, colours, wait, write-back]; k_tile_vel: [8 wait, 9 load, 10 colour loop, 11 write-back, 12 whole batch, 13 batches]
int sfx_debug_tile_timing(sfx_world* w, unsigned long long* out, int reset) {
    if (!w) return SFX_ERR_INVALID;
    if (!w->dbg.p) { if (w->dbg.alloc(128)) return SFX_ERR_NOMEM; (void)hipMemset(w->dbg.p, 0, 1024); }
    (void)hipStreamSynchronize(w->stream);
    if (out) (void)hipMemcpy(out, w->dbg.p, 1024, hipMemcpyDeviceToHost);
    if (reset) (void)hipMemset(w->dbg.p, 0, 1024);
    return SFX_OK;
}
#endif

const char* sfx_kernel_class_name(int cls) {
    static const char* names[SFX_K_CLASSES] = {
        "aabb_count", "scan", "scatter", "pair_count", "pair_fill", "tile_list", "color", "integrate",
        "solve_joints", "solve_pos", "derive_vel", "solve_vel", "stats", "halo", "query", "narrow", "tick"};
    return (cls >= 0 && cls < SFX_K_CLASSES) ? names[cls] : "?";
}

int sfx_timer_begin(sfx_world* w) {
    if (!w) return SFX_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipEventRecord(w->t_begin, w->stream));
    return SFX_OK;
}

int sfx_timer_end(sfx_world* w, double* ms) {
    if (!w || !ms) return SFX_ERR_INVALID;
    HIPCHK(w, hipSetDevice(w->device));
    HIPCHK(w, hipEventRecord(w->t_end, w->stream));
    HIPCHK(w, hipEventSynchronize(w->t_end));
    float f = 0.0f;
    HIPCHK(w, hipEventElapsedTime(&f, w->t_begin, w->t_end));
    *ms = f;
    return SFX_OK;
}

}  // extern "C"
