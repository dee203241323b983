// sfx_kernels.h -- the HIP kernels of the physics tick (gfx950, wave64).
//
// K1  k_integrate          substep integration over the body records (streaming)
// K2  k_aabb_count         world AABB + home cell + per-cell histogram (atomics)
// K3b k_scan_*             exclusive prefix scan, block scan in LDS with wave shuffles
// K3c k_scatter / k_finalize_slots   counting sort into tile-major cell order, ascending id per cell
// K3d k_pairs<FILL>        candidate pairs, wave ballot compaction, binned by owner tile
// K5  k_tile_list / k_tile_color / k_color_global   solver tiles + edge colouring (LDS / one workgroup)
// K6  k_tile_pos           per tile and phase: bodies of the 2x2 tile block in LDS, narrowphase of
//                          the tile's pairs, XPBD manifold block solves colour by colour
// K7  k_derive             velocities from pose deltas
// K8  k_tile_vel           restitution + friction per manifold, same tiling
//
// All of it is HBM/latency-bound gather-scatter work: no MFMA anywhere.
#pragma once

#include "sfx_device.h"

namespace sfx {

constexpr int kBlock = 256;
constexpr uint32_t kMaxP = SFX_TILE_MAX_PAIRS;      // pairs a tile may own
constexpr uint32_t kMaxB = SFX_TILE_MAX_BODIES;     // grid-sorted colliders in a 2x2 tile block
constexpr int kClaim = 4;                           // claim slots per body in the colouring
#ifndef SFX_TILE_THREADS
#define SFX_TILE_THREADS 512
#endif
#ifndef SFX_TILE_MIN_WAVES
#define SFX_TILE_MIN_WAVES 1
#endif
constexpr int kTileThreads = SFX_TILE_THREADS;      // workgroup of the tile solves = pairs on the register path
constexpr uint32_t kBatchP = kMaxP;                 // pairs of a batch of tiles
constexpr int kMaxGroup = 16;                       // tiles per batch at most
constexpr uint32_t kExt = 0xFFFFu;                  // local index of a body outside the block (large static)
constexpr uint32_t kMaxExt = 64;                    // distinct outside bodies a batch can hold in LDS behind its block bodies
constexpr uint32_t kMaxBX = kMaxB + kMaxExt;        // LDS body slots of a tile solve
constexpr int kScanItems = 16;                      // items per thread in the block scan
constexpr int kScanTile = kBlock * kScanItems;      // 4096 items per block
constexpr uint32_t kLeftCap = 4096;                 // leftover edges the serial finish takes
constexpr uint32_t kOvfPairs = 1u, kOvfColors = 2u, kOvfLeftover = 4u, kOvfLarge = 8u, kOvfFallback = 16u, kOvfStall = 32u, kOvfExt = 64u, kOvfHalo = 128u, kOvfRace = 256u;

// stage timing of the tile kernels and the colouring (development builds with -DSFX_TILE_TIMING; `T.dbg` or `dbg` in scope)
#ifdef SFX_TILE_TIMING
#define SFX_TT(var) const unsigned long long var = wall_clock64()
#define SFX_TT_ADD(slot, a, b) if (threadIdx.x == 0 && T.dbg) atomicAdd(&T.dbg[slot], (b) - (a))
#define SFX_TT_MAX(slot, a, b) if (threadIdx.x == 0 && T.dbg) atomicMax(&T.dbg[slot], (b) - (a))
#else
#define SFX_TT(var)
#define SFX_TT_ADD(slot, a, b)
#define SFX_TT_MAX(slot, a, b)
#endif

#ifdef SFX_STEP_PROBE   // stage stamps of k_batch_color (see SFX_STAMP below: diagnostic builds only)
#define SFX_CSTAMP(t) unsigned long long t; do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SFX_CSTAMP(t)
#endif

// device-resident counters of one tick (one allocation, zeroed at the start of every tick)
struct Counters {
    uint32_t n_small;       // written by k_finalize_slots
    uint32_t n_large;
    uint32_t n_pairs;       // tile pairs + fallback pairs (clamped to the capacities)
    uint32_t overflow;      // kOvf* bits
    uint32_t n_colors;      // most colours of any tile / the fallback list
    uint32_t n_leftover;
    uint32_t n_contacts;    // stats kernel
    uint32_t max_depth_bits;
    float ke;
    uint32_t n_manifolds, n_manifolds2;
    uint32_t n_fb;          // pairs appended to the fallback list (may exceed its capacity)
    uint32_t n_fb_colors;
    uint32_t n_tile_pairs;  // pairs owned by tiles (clamped to max_pairs)
    uint32_t max_tile_pairs, max_tile_bodies;
    uint32_t n_active[4];   // [0]: tiles that own pairs (statistic), rest unused
    uint32_t ticket;        // last-workgroup detection of the phase-3 kernels
    uint32_t n_over_tiles;  // tiles that handed their pairs to the fallback list
    uint32_t body_cursor;   // entries written to the batches' body lists
    uint32_t tp_cursor;     // pairs written to the batch-ordered pair list
    uint32_t n_batches[4];  // batches per phase
    uint32_t n_listed[4];   // tiles on the per-phase lists
    uint32_t color_count[SFX_MAX_COLORS];
    uint32_t fb_start[SFX_MAX_COLORS + 1];  // colour-major offsets into the sorted fallback list
    uint32_t n_kept;        // tile pairs within reach (the solves' pairs; the fallback list is not tested)
    uint32_t run_cursor;    // batches that have written their colour run table
    uint32_t item_ticket[2];  // next batch of a tile-solve launch, handed out in item order; launches alternate between
                              // the two counters and every launch clears the one its successor will use
    uint32_t n_medium;      // colliders sorted into the grid's second level (written by k_finalize_slots)
    uint32_t n_near;        // batches on the near list (k_near_closure; slab worlds with the exchange overlapped)
    uint32_t fb_done;       // k_tile_tick: epoch0 + u + 1 once the fallback list of sweep u has been solved
};
static_assert(sizeof(Counters) % 16 == 0, "the per-tick block is cleared in 16-byte units");
static_assert(offsetof(Counters, overflow) == 12, "k_zero / k_aabb_count fold word 3 of the block into the sticky overflow word");

struct GridParams {
    float inv_cell;
    float small_lim;   // 0.9 * cell
    uint32_t gxl, gyl; // log2 grid dims (cells)
    uint32_t ncell;
    uint32_t tlx, tly; // log2 tile width / height (cells); tly == gyl: full-height strips
    uint32_t ntiles;
    uint32_t pstride;  // entries per phase in the tile lists and batch tables
    // second level (DESIGN.md "Contract / grid"): cells eight times as wide on the same torus for the MEDIUM colliders
    // (too large for a base cell, extent <= medium_lim); their buckets follow the base cells' in cell_start, so the
    // sorted order is small slots, medium slots, then the flat list.  medium_lim == 0: one level (ncell_all == ncell).
    uint32_t ncell_all;
    uint32_t gxl1, gyl1;
    float inv_cell1, medium_lim;
};

__device__ __forceinline__ uint32_t bucket1_of(int32_t cx, int32_t cy, const GridParams& g) {
    return g.ncell + ((((uint32_t)cy & ((1u << g.gyl1) - 1u)) << g.gxl1) | ((uint32_t)cx & ((1u << g.gxl1) - 1u)));
}

// wrapped tile index of true tile coordinates
__device__ __forceinline__ uint32_t tile_of(int32_t tx, int32_t ty, const GridParams& g) {
    const uint32_t lx = g.gxl - g.tlx, ly = g.gyl - g.tly;
    return (((uint32_t)ty & ((1u << ly) - 1u)) << lx) | ((uint32_t)tx & ((1u << lx) - 1u));
}

// bucket of a true cell: tile-major, so that a tile's cells -- and with them its colliders in
// the sorted order -- are contiguous
__device__ __forceinline__ uint32_t bucket_of(int32_t cx, int32_t cy, const GridParams& g) {
    const uint32_t mx = (1u << g.tlx) - 1u, my = (1u << g.tly) - 1u;
    return (tile_of(cx >> g.tlx, cy >> g.tly, g) << (g.tlx + g.tly)) | (((uint32_t)cy & my) << g.tlx) | ((uint32_t)cx & mx);
}

// clears the per-tick block (counters, cell and tile histograms) with 16-byte stores
// `sticky` (may be null): the overflow bits of the block being cleared are kept in that word, so that a capacity
// overflow of ANY tick stays visible until a synchronising call has reported it (sfx_sync)
__global__ __launch_bounds__(kBlock) void k_zero(uint4* __restrict__ p, uint32_t n16, uint32_t* __restrict__ sticky) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i == 0u && sticky && n16) { const uint32_t o = p[0].w; if (o) atomicOr(sticky, o); }
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ---------------------------------------------------------------- K1 / K7 streaming
// one body's integration step (DESIGN.md "Order"): returns the pose at the start of the substep in `q`
__device__ __forceinline__ void integrate_body(float4& p, float4& q, float4& v, const float2& w, const StepParams& sp) {
    q = p;
#ifndef SFX_PROBE_NOSOLVE   // (the timing probe keeps the scene where it is)
    if (w.x > 0.0f) {
        v.x = v.x + sp.gx * sp.h;
        v.y = v.y + sp.gy * sp.h;
    }
#endif
    if (w.x > 0.0f || w.y > 0.0f) {
        v.x = v.x * sp.damp_lin;
        v.y = v.y * sp.damp_lin;
        v.z = v.z * sp.damp_ang;
    }
    p.x = p.x + v.x * sp.h;
    p.y = p.y + v.y * sp.h;
    rotate_by(p.z, p.w, v.z * sp.h);
}

// Derive-on-load (DESIGN.md "Kernels / velocities derived where they are first read"): inside sfx_step the velocity solve
// derives a body's velocity when it first loads it and stamps the velocity record's unused fourth word with the substep's
// epoch; a body no pair touched is derived by the NEXT substep's integration (`derive_first`: the epoch whose stamp says
// "done"), or by one k_derive at the end of the tick.  0 = off (every body was derived by k_derive).
__device__ __forceinline__ void derive_body(const float4& p, const float4& q, float4& v, const StepParams& sp);

__global__ __launch_bounds__(kBlock) void k_integrate(uint32_t n, float4* __restrict__ pose, float4* __restrict__ prev,
                                                      float4* __restrict__ vel, float4* __restrict__ dpos,
                                                      const float2* __restrict__ mass, StepParams sp, uint32_t derive_first) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = pose[i], q;
    float4 v = vel[i];
    const float2 w = mass[i];
    // (a body whose velocity nobody derived in the last substep was in no pair's batch: its displacement is zero)
    if (derive_first && (w.x > 0.0f || w.y > 0.0f) && __float_as_uint(v.w) != derive_first) derive_body(p, prev[i], v, sp);
    dpos[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // the position sweep of this substep starts from the poses integrated here
    integrate_body(p, q, v, w, sp);
    prev[i] = q;
    if (w.x > 0.0f || w.y > 0.0f) vel[i] = v;
    pose[i] = p;
}

// velocity of one dynamic body from what the substep did to its pose (p now, q at the start of the substep)
__device__ __forceinline__ void derive_body(const float4& p, const float4& q, float4& v, const StepParams& sp) {
    v.x = (p.x - q.x) * sp.inv_h;
    v.y = (p.y - q.y) * sp.inv_h;
    float dc = p.z * q.z + p.w * q.w;
    float ds = p.w * q.z - p.z * q.w;
    if (dc > 1e-6f) {
        // tan(dtheta) has a pole at a quarter turn: what the substep's corrections add to the angular velocity is bounded
        // (DESIGN.md "Order"); a free spin, whose omega the substep does not change, is never touched
        float wn = (ds / dc) * sp.inv_h;
        const float dw = wn - v.z, lim = SFX_MAX_TAN_CORRECTION * sp.inv_h;
        if (dw > lim) wn = v.z + lim;
        if (dw < -lim) wn = v.z - lim;
        v.z = wn;
    }
}

// `done` != 0: bodies whose velocity record carries that stamp were derived by the velocity solve already
// (and had their displacement applied there); the others' displacement is applied here, before their velocity is derived
__global__ __launch_bounds__(kBlock) void k_derive(uint32_t n, float4* __restrict__ pose,
                                                    const float4* __restrict__ prev, float4* __restrict__ vel,
                                                    const float4* __restrict__ dpos,
                                                    const float2* __restrict__ mass, StepParams sp, uint32_t done) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float2 w = mass[i];
    if (!(w.x > 0.0f || w.y > 0.0f)) return;
    float4 v = vel[i];
    if (done && __float_as_uint(v.w) == done) return;
    float4 p = pose[i];
    const float4 d = dpos[i];
    if (d.x != 0.0f || d.y != 0.0f || d.z != 0.0f) { apply_delta(p, d); pose[i] = p; }
    derive_body(p, prev[i], v, sp);
    vel[i] = v;
}

// Further colliders (entries nb .. n) take the pose and velocity of their bodies: before every narrowphase and at the
// end of a tick (AABBs, queries and debug read entries; only the solves go through body_of)
__global__ __launch_bounds__(kBlock) void k_mirror(uint32_t nsec, uint32_t nb, const uint32_t* __restrict__ cbody,
                                                    float4* __restrict__ pose, float4* __restrict__ vel) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= nsec) return;
    const uint32_t b = cbody[k];
    pose[nb + k] = pose[b];
    vel[nb + k] = vel[b];
}

// ---------------------------------------------------------------- K2 AABB + histogram
__global__ __launch_bounds__(kBlock) void k_aabb_count(uint32_t n, const float4* __restrict__ pose,
                                                        const float4* __restrict__ vel, const float4* __restrict__ shape,
                                                        const float4* __restrict__ lpose,
                                                        const float4* __restrict__ poly, float margin, float vel_pad_max,
                                                        float dt, GridParams g, float4* __restrict__ aabb,
                                                        uint32_t* __restrict__ bucket, int2* __restrict__ cellxy,
                                                        uint32_t* __restrict__ cell_count, uint32_t* __restrict__ large_tmp,
                                                        uint32_t large_cap, Counters* __restrict__ ctr,
                                                        uint4* __restrict__ stale_half, uint32_t n16,
                                                        uint32_t* __restrict__ sticky, const float* __restrict__ guard_x0,
                                                        float guard_limit) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    // the zero block of the previous tick (the next tick's): cleared here, nothing in this tick reads it; its overflow
    // bits move to the sticky word first (thread 0 clears the unit that holds them itself, after reading it)
    if (i == 0u && n16 && sticky) { const uint32_t o = stale_half[0].w; if (o) atomicOr(sticky, o); }
    for (uint32_t k = i; k < n16; k += gridDim.x * kBlock) stale_half[k] = make_uint4(0u, 0u, 0u, 0u);
    if (i >= n) return;
    const float4 sh = shape[i];
    const float4 ps = pose[i];
    // slab worlds: a body that has moved further on x than the halo lists allow for since they were cut (sfx_halo_guard)
    if (guard_x0 && fabsf(ps.x - guard_x0[i]) > guard_limit) atomicOr(&ctr->overflow, kOvfHalo);
    float padx, pady;
    motion_pads(vel[i], sh, lpose, i, margin, vel_pad_max, dt, padx, pady);
    const float4 A = compute_aabb(shape_pose(ps, sh, lpose, i), sh, poly, padx, pady);
    aabb[i] = A;
    uint32_t bk = SFX_NO_BUCKET;
    if (shape_type(sh) != SFX_SHAPE_NONE) {
        if ((A.z - A.x) <= g.small_lim && (A.w - A.y) <= g.small_lim) {
            const int32_t cx = cell_coord(A.x, g.inv_cell), cy = cell_coord(A.y, g.inv_cell);
            bk = bucket_of(cx, cy, g);
            cellxy[i] = make_int2(cx, cy);
            atomicAdd(&cell_count[bk], 1u);
        } else if ((A.z - A.x) <= g.medium_lim && (A.w - A.y) <= g.medium_lim) {   // (never with one level: medium_lim == 0)
            const int32_t cx = cell_coord(A.x, g.inv_cell1), cy = cell_coord(A.y, g.inv_cell1);
            bk = bucket1_of(cx, cy, g);
            cellxy[i] = make_int2(cx, cy);
            atomicAdd(&cell_count[bk], 1u);
        } else {
            const uint32_t l = atomicAdd(&ctr->n_large, 1u);
            if (l < large_cap) large_tmp[l] = i;
            else atomicOr(&ctr->overflow, kOvfLarge);
        }
    }
    bucket[i] = bk;
}

// ---------------------------------------------------------------- K3b prefix scan
// Exclusive scan of u32.  Each block scans a 4096-item tile: per-thread serial scan of 16
// items (4 x 16-byte loads), wave scan with shuffles, 4 wave totals through LDS.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = sum
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds /* NT/64 + 1 */, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w2 = 0; w2 < NT / 64; ++w2) { uint32_t t = lds[w2]; lds[w2] = run; run += t; }
        lds[NT / 64] = run;
    }
    __syncthreads();
    uint32_t res = inc - v + lds[wave];
    *total = lds[NT / 64];
    __syncthreads();
    return res;
}

// data[0..count) scanned in place per tile; tile totals to sums[block].  If single != 0 the
// grid is one block and the result is final (copy, if given, receives it too).
__device__ __forceinline__ void scan_tile(uint32_t blk, uint32_t* data, uint32_t count, uint32_t* sums, uint32_t* copy,
                                          int single) {
    __shared__ uint32_t lds[kBlock / 64 + 1];
    const uint32_t base = blk * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    if (base + kScanItems <= count) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            uint4 t = *reinterpret_cast<const uint4*>(data + base + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) v[k] = (base + k < count) ? data[base + k] : 0u;
    }
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { uint32_t t = v[k]; v[k] = sum; sum += t; }
    uint32_t total;
    const uint32_t pre = block_exclusive_scan<kBlock>(sum, lds, &total);
    if (threadIdx.x == 0 && !single) sums[blk] = total;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] += pre;
    if (base + kScanItems <= count) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            uint4 t = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            *reinterpret_cast<uint4*>(data + base + 4 * q) = t;
            if (single && copy) *reinterpret_cast<uint4*>(copy + base + 4 * q) = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k)
            if (base + k < count) { data[base + k] = v[k]; if (single && copy) copy[base + k] = v[k]; }
    }
}

__global__ __launch_bounds__(kBlock) void k_scan_local(uint32_t* __restrict__ data, uint32_t count,
                                                        uint32_t* __restrict__ sums, uint32_t* __restrict__ copy,
                                                        int single) {
    scan_tile(blockIdx.x, data, count, sums, copy, single);
}

// exclusive scan of up to 8192 tile totals by one block of 1024 threads
constexpr int kSumItems = 8;
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t nsums) {
    __shared__ uint32_t lds[1024 / 64 + 1];
    uint32_t v[kSumItems];
    const uint32_t base = threadIdx.x * kSumItems;
#pragma unroll
    for (int k = 0; k < kSumItems; ++k) v[k] = (base + k < nsums) ? sums[base + k] : 0u;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kSumItems; ++k) { uint32_t t = v[k]; v[k] = sum; sum += t; }
    uint32_t total;
    const uint32_t pre = block_exclusive_scan<1024>(sum, lds, &total);
#pragma unroll
    for (int k = 0; k < kSumItems; ++k)
        if (base + k < nsums) sums[base + k] = v[k] + pre;
}

// raw != 0 (at most kBlock tiles): sums still holds the tile totals of k_scan_local and every block adds up
// the totals of the tiles before its own, which saves the k_scan_sums launch in between
__global__ __launch_bounds__(kBlock) void k_scan_add(uint32_t* __restrict__ data, uint32_t count,
                                                      const uint32_t* __restrict__ sums, uint32_t* __restrict__ copy,
                                                      int raw) {
    uint32_t add;
    if (raw) {
        __shared__ uint32_t part[kBlock / 64];
        uint32_t s = threadIdx.x < blockIdx.x ? sums[threadIdx.x] : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += (uint32_t)__shfl_xor((int)s, off, 64);
        if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = s;
        __syncthreads();
        add = 0u;
#pragma unroll
        for (int q = 0; q < kBlock / 64; ++q) add += part[q];
    } else {
        add = sums[blockIdx.x];
    }
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    if (base + kScanItems <= count) {
#pragma unroll
        for (int q = 0; q < kScanItems / 4; ++q) {
            uint4 t = *reinterpret_cast<const uint4*>(data + base + 4 * q);
            t.x += add; t.y += add; t.z += add; t.w += add;
            *reinterpret_cast<uint4*>(data + base + 4 * q) = t;
            if (copy) *reinterpret_cast<uint4*>(copy + base + 4 * q) = t;
        }
    } else {
        for (int k = 0; k < kScanItems; ++k)
            if (base + k < count) { uint32_t t = data[base + k] + add; data[base + k] = t; if (copy) copy[base + k] = t; }
    }
}

// ---------------------------------------------------------------- K3c counting sort
__global__ __launch_bounds__(kBlock) void k_scatter(uint32_t n, const uint32_t* __restrict__ bucket,
                                                     uint32_t* __restrict__ cell_fill, uint32_t* __restrict__ tmp_id) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t bk = bucket[i];
    if (bk == SFX_NO_BUCKET) return;
    tmp_id[atomicAdd(&cell_fill[bk], 1u)] = i;
}

// The atomic scatter leaves a cell's entries in arbitrary order; this pass ranks every entry
// inside its cell by body id and writes the final cell-ordered records.  Large colliders get
// the slots after the small ones, ascending id.  The slot order is what makes the pair list,
// and with it the colouring, reproducible.
__global__ __launch_bounds__(kBlock) void k_finalize_slots(uint32_t n, GridParams g, const uint32_t* __restrict__ cell_start,
                                                            const uint32_t* __restrict__ tmp_id,
                                                            const uint32_t* __restrict__ bucket,
                                                            const float4* __restrict__ aabb, const int2* __restrict__ cellxy,
                                                            const uint32_t* __restrict__ large_tmp, uint32_t large_cap,
                                                            uint32_t* __restrict__ sorted_id, float4* __restrict__ sorted_aabb,
                                                            int2* __restrict__ sorted_cell, Counters* __restrict__ ctr,
                                                            const uint4* __restrict__ filt, uint4* __restrict__ sorted_filt,
                                                            uint32_t* __restrict__ slot_of) {
    // (sorted_filt, may be null: the filter words in slot order too, so that the pair passes read a candidate's box,
    // cell, id and filter side by side instead of its filter through its id.  slot_of, may be null: the slot of every
    // entry -- worlds with dynamic compound bodies: k_pairs looks up where a further collider's body sits)
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t ns = cell_start[g.ncell_all];   // grid-sorted slots: the small colliders, then the medium ones
    const uint32_t nl = min(ctr->n_large, large_cap);
    if (p == 0) { ctr->n_small = cell_start[g.ncell]; ctr->n_medium = ns - cell_start[g.ncell]; }
    if (p < ns) {
        const uint32_t i = tmp_id[p];
        const uint32_t bk = bucket[i];
        const uint32_t s = cell_start[bk], e = cell_start[bk + 1];
        uint32_t rank = 0;
        for (uint32_t q = s; q < e; ++q) rank += (tmp_id[q] < i) ? 1u : 0u;
        const uint32_t dst = s + rank;
        sorted_id[dst] = i;
        sorted_aabb[dst] = aabb[i];
        sorted_cell[dst] = cellxy[i];
        if (sorted_filt) sorted_filt[dst] = filt[i];
        if (slot_of) slot_of[i] = dst;
    } else if (p - ns < nl) {
        const uint32_t i = large_tmp[p - ns];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < nl; ++q) rank += (large_tmp[q] < i) ? 1u : 0u;
        const uint32_t dst = ns + rank;
        sorted_id[dst] = i;
        sorted_aabb[dst] = aabb[i];
        sorted_cell[dst] = make_int2(0, 0);
        if (sorted_filt) sorted_filt[dst] = filt[i];
        if (slot_of) slot_of[i] = dst;
    }
}

// ---------------------------------------------------------------- K3d candidate pairs
// Eight lanes share one slot (eight slots per 64-lane wave): the candidates of a slot -- the
// entries of its own cell behind it, the four forward neighbour cells, then every large
// collider -- are spread over the eight lanes, tested, and the survivors compacted with
// wave-wide __ballot + popcount, so a slot's pairs land contiguously.
//
// Every pair goes to one of three streams (DESIGN.md "Contract / tiles"):
//   0  owned by the slot's own tile T      (both small, or small + large non-dynamic)
//   1  owned by the tile left of T         (the neighbour cell (-1, +1) lies in that tile)
//   2  the fallback list                   (a large dynamic collider is involved, or a further collider of a dynamic
//                                           body that does not sit beside its body's first collider: tile_end)
// The count pass adds the per-slot stream sizes to the tiles' pair counts; after
// k_tile_list + a scan of those counts the fill pass reserves room per slot and stream with
// one atomic each and writes (id a, id b, slot a, slot b).  Pairs of tiles flagged as over
// capacity are re-routed to the fallback list.
#ifndef SFX_PAIR_GROUP
#define SFX_PAIR_GROUP 8
#endif
constexpr int kGroup = SFX_PAIR_GROUP;    // lanes per slot (a power of two, 2 .. 16)
constexpr int kGroupLog = kGroup == 16 ? 4 : kGroup == 8 ? 3 : kGroup == 4 ? 2 : 1;
constexpr int kSlotsPerWave = 64 / kGroup;

struct PairArrays {
    uint32_t* slot_cnt;      // [3 * (n + 1)] per slot: stream sizes (count pass out, fill pass in)
    uint32_t* tile_cnt;      // count pass: pairs per tile (atomics); fill pass: scanned = tile start
    uint32_t* tile_fill;     // per tile: pairs of the tile's own slots (stream 0); the stream-1 pairs of its right neighbour's
                             // slots follow them in the tile's range (written by k_tile_count)
    const uint32_t* slot_off; // [2 * (n + 1)] per slot: pairs of its tile's earlier slots, stream 0 / stream 1 (k_tile_count)
    const uint32_t* tile_flag;  // != 0: tile hands its pairs to the fallback list
    uint4* rawp;             // tile-ordered pairs (id a, id b, slot a, slot b)
    uint2* fpairs;           // fallback list (id a, id b)
    uint2* hit_tmp;          // [kHitTmp * (n + 1)] per slot: the count pass's hits (id, slot) of stream 0 from the front
                             // and of stream 1 from the back, so that the fill pass copies instead of testing again
    const uint32_t* slot_of; // [n] slot of every entry (kNoSlot: none), or null: no dynamic body of this world has a further collider
    uint32_t max_pairs, fb_cap;
};
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr uint32_t kClassShift = 30u, kSlotMask = (1u << kClassShift) - 1u;  // slots < 2^24: the class of a tile pair sits above
constexpr uint32_t kHitTmp = 16;  // a slot with more tile pairs (or any fallback pair) is tested again by the fill pass

// The eight slot groups of a wave usually belong to one or two tiles (slots are sorted
// tile-major): the groups that name the same counter index add up, the first of them does one
// atomic for all, and every group gets its share.  All 64 lanes must call this.  `idx`, `c` are
// the group's (uniform inside the group); returns the group's base (FILL) or nothing useful.
template <bool WANT_BASE>
__device__ __forceinline__ uint32_t wave_group_add(uint32_t* __restrict__ arr, uint32_t idx, uint32_t c, int lane) {
    const int grp = lane >> kGroupLog;
    uint32_t tot = 0, pre = 0;
    int leader = grp;
#pragma unroll
    for (int gq = 0; gq < kSlotsPerWave; ++gq) {
        const uint32_t t = __shfl(idx, gq << kGroupLog, 64), cc = __shfl(c, gq << kGroupLog, 64);
        if (t == idx) {
            tot += cc;
            if (gq < grp) pre += cc;
            if (gq < leader) leader = gq;
        }
    }
    uint32_t rb = 0;
    if ((lane & (kGroup - 1)) == 0 && leader == grp && tot) {
        if (WANT_BASE) rb = atomicAdd(&arr[idx], tot);
        else atomicAdd(&arr[idx], tot);
    }
    if (!WANT_BASE) return 0u;
    rb = __shfl(rb, leader << kGroupLog, 64);
    return rb + pre;
}

// One end of a candidate pair that is a further collider of a DYNAMIC body (kFiltFurther).  The tile solves keep one
// state per body, at the slot of its first collider (entry = body id), so the pair may stand in a tile's list only when
// the further collider is small and lies in the home tile of that first collider, which must be small too -- and where
// the pair crosses a tile border in x (ddx) / y (ddy), in its cell column / row: the body then takes part in exactly
// the seam units its first collider alone could take part in, and the units of one class still share no body.  The
// slot word of the pair then names the BODY's slot (k_batch_color gives it the body's local index; the pair's ids
// stay the colliders': k_narrow and the reach test measure the shapes).  oracle/sfx_oracle.c tile_end_ok.
struct TileEnd { uint32_t slot; int2 cell; bool ok; };

__device__ __forceinline__ TileEnd tile_end(uint32_t slot, int2 c, uint4 f, uint32_t ns, const GridParams& g,
                                            const uint32_t* __restrict__ slot_of, const int2* __restrict__ sorted_cell) {
    TileEnd e;
    e.slot = slot; e.cell = c; e.ok = true;
    if (!(f.w & kFiltFurther)) return e;
    e.ok = false;
    if (!slot_of || slot >= ns) return e;
    const uint32_t ps = slot_of[f.w >> 8];
    if (ps >= ns) return e;   // (kNoSlot: the body's first collider has no shape)
    const int2 cp = sorted_cell[ps];
    e.ok = (cp.x >> g.tlx) == (c.x >> g.tlx) && (cp.y >> g.tly) == (c.y >> g.tly);
    e.slot = ps; e.cell = cp;
    return e;
}

__device__ __forceinline__ bool tile_end_ok(const TileEnd& e, int2 c, bool ddx, bool ddy) {
    return e.ok && (!ddx || e.cell.x == c.x) && (!ddy || e.cell.y == c.y);
}

// CMP: the world has dynamic bodies with further colliders (P.slot_of is there); without them no filter word carries
// kFiltFurther and the candidate loop is compiled without the lookups (they cost the count pass 4.6 us of 29 on the pile)
template <bool FILL, bool CMP>
__global__ __launch_bounds__(kBlock) void k_pairs(uint32_t n, GridParams g, const uint32_t* __restrict__ cell_start,
                                                   const uint32_t* __restrict__ sorted_id,
                                                   const float4* __restrict__ sorted_aabb,
                                                   const int2* __restrict__ sorted_cell, const uint4* __restrict__ sorted_filt,
                                                   PairArrays P, Counters* __restrict__ ctr, uint32_t large_cap) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & (kGroup - 1);          // lane inside the group
    const int gshift = lane & ~(kGroup - 1);     // first lane of the group
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kBlock) >> 6;
    const uint32_t ns = cell_start[g.ncell];       // small slots [0, ns)
    const uint32_t nsm = cell_start[g.ncell_all];  // medium slots [ns, nsm): sorted by level-1 cell; then the flat list
    const uint32_t nl = min(ctr->n_large, large_cap);
    const uint32_t nact = nsm + nl;
    const uint64_t below = (1ull << gl) - 1ull;  // group-local lanes before this one
    for (uint32_t p0 = wave * kSlotsPerWave; p0 < nact; p0 += nwaves * kSlotsPerWave) {
        const uint32_t p = p0 + (uint32_t)(lane / kGroup);
        const bool live = p < nact;
        uint32_t i = 0;
        float4 Ai = make_float4(0, 0, 0, 0);
        uint4 fi = make_uint4(0, 0, 0, 0);
        int2 c = make_int2(0, 0);
        if (live) { i = sorted_id[p]; Ai = sorted_aabb[p]; fi = sorted_filt[p]; c = sorted_cell[p]; }
        const int32_t txp = c.x >> g.tlx, typ = c.y >> g.tly;
        const uint32_t T0 = tile_of(txp, typ, g), T1 = tile_of(txp - 1, typ, g);
        TileEnd Ei;   // (p itself unless a further collider of a dynamic body)
        Ei.slot = p; Ei.cell = c; Ei.ok = true;
        if constexpr (CMP) Ei = tile_end(p, c, fi, ns, g, P.slot_of, sorted_cell);
        // stream bases (fill pass): lane 0 of the group reserves, the group reads them by shuffle
        uint32_t base[3] = {0u, 0u, 0u};
        bool to_fb[3] = {false, false, true};
        if (FILL) {
            // stream sizes from the count pass; pairs of tiles over capacity go to the fallback list
            uint32_t c0 = 0, c1 = 0, c2 = 0;
            if (live) { c0 = P.slot_cnt[3 * p]; c1 = P.slot_cnt[3 * p + 1]; c2 = P.slot_cnt[3 * p + 2]; }
            const bool f0 = c0 && P.tile_flag[T0], f1 = c1 && P.tile_flag[T1];
            const uint32_t fb_total = c2 + (f0 ? c0 : 0u) + (f1 ? c1 : 0u);
            // groups that are not live or have nothing for a stream get an index of their own
            const uint32_t none = 0xFFFFFF00u + (uint32_t)(lane >> kGroupLog);
            // places inside the tiles' ranges come from the prefix sums of k_tile_count: no cursor, no atomic, and the
            // pairs of a tile are in slot order whatever the waves' timing
            const uint32_t r0 = (live && c0 && !f0) ? P.slot_off[2 * p] : 0u;
            const uint32_t r1 = (live && c1 && !f1) ? P.tile_fill[T1] + P.slot_off[2 * p + 1] : 0u;
            const uint32_t r2 = wave_group_add<true>(&ctr->n_fb, fb_total ? 0u : none, fb_total, lane);
            base[2] = r2;
            base[0] = f0 ? r2 + c2 : ((live && c0) ? P.tile_cnt[T0] + r0 : 0u);
            base[1] = f1 ? r2 + c2 + (f0 ? c0 : 0u) : ((live && c1) ? P.tile_cnt[T1] + r1 : 0u);
            to_fb[0] = f0; to_fb[1] = f1;
        }
        uint32_t cnt[3] = {0u, 0u, 0u};
        // fill pass: a slot whose hits the count pass kept is copied, not tested again
        bool copy = false;
        uint32_t cc0 = 0, cc1 = 0;
        if (FILL && live) {
            cc0 = P.slot_cnt[3 * p]; cc1 = P.slot_cnt[3 * p + 1];
            copy = P.slot_cnt[3 * p + 2] == 0u && cc0 + cc1 <= kHitTmp;
        }
        const bool test = live && !copy;
        // Six candidate ranges per slot (own cell behind the slot, four forward neighbour cells, the large colliders),
        // laid end to end: the group's eight lanes stride over the joined list, so a slot with a handful of
        // candidates in all (the usual case: under one collider per bucket) takes one or two trips instead of one
        // per range.  The joined order is range-major, entry-minor -- the order the ranges were walked in one by
        // one -- so the compacted hits land exactly where they did.
        uint32_t qs_[6], off_[7];
        int32_t ncx_[6], ncy_[6];
        off_[0] = 0u;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            uint32_t qs = 0, qe = 0;
            int32_t ncx = 0, ncy = 0;
            if (test && p < nsm) {   // a grid-sorted slot: the same forward scheme at its own level (a medium slot: level 1)
                const bool lvl1 = p >= ns;
                if (r == 0) {
                    ncx = c.x; ncy = c.y;
                    qs = p + 1; qe = cell_start[(lvl1 ? bucket1_of(ncx, ncy, g) : bucket_of(ncx, ncy, g)) + 1];
                } else if (r < 5) {
                    const int dx = (r == 1) ? 1 : (r == 2) ? -1 : (r == 3) ? 0 : 1;
                    const int dy = (r == 1) ? 0 : 1;
                    ncx = c.x + dx; ncy = c.y + dy;
                    const uint32_t bk = lvl1 ? bucket1_of(ncx, ncy, g) : bucket_of(ncx, ncy, g);
                    qs = cell_start[bk]; qe = cell_start[bk + 1];
                } else {
                    qs = nsm; qe = nact;
                }
            } else if (test && r == 5) {
                qs = p + 1; qe = nact;  // large slot: later larges only
            }
            qs_[r] = qs; ncx_[r] = ncx; ncy_[r] = ncy;
            off_[r + 1] = off_[r] + (qe > qs ? qe - qs : 0u);
        }
        const bool small_slot = p < ns;
        const uint32_t L = off_[6];
        // a tested candidate joins its stream: count pass -> hit_tmp + stream sizes, fill pass -> the pair lists
        auto emit = [&](bool hit, uint32_t j, uint32_t q, int st, uint32_t cls) {
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const bool mine = hit && st == s3;
                const uint64_t bal = (__ballot(mine) >> gshift) & ((1ull << kGroup) - 1ull);
                if (!FILL && mine && s3 < 2) {
                    const uint32_t idx = cnt[s3] + (uint32_t)__popcll(bal & below);
                    if (idx < kHitTmp) P.hit_tmp[(size_t)p * kHitTmp + (s3 == 0 ? idx : kHitTmp - 1u - idx)] = make_uint2(j, q | (cls << kClassShift));
                }
                if (FILL && mine) {
                    const uint32_t dst = base[s3] + cnt[s3] + (uint32_t)__popcll(bal & below);
                    const bool swap = j < i;
                    if (to_fb[s3]) {
                        if (dst < P.fb_cap) P.fpairs[dst] = make_uint2(swap ? j : i, swap ? i : j);
                    } else if (dst < P.max_pairs) {
                        P.rawp[dst] = make_uint4(swap ? j : i, swap ? i : j, swap ? q : Ei.slot, (swap ? Ei.slot : q) | (cls << kClassShift));
                    }
                }
                cnt[s3] += (uint32_t)__popcll(bal);
            }
        };
        for (uint32_t x0 = 0; __any(x0 < L); x0 += kGroup) {
            const uint32_t x = x0 + (uint32_t)gl;
            int r = 0;
            uint32_t q = qs_[0] + x;
            int32_t ncx = ncx_[0], ncy = ncy_[0];
#pragma unroll
            for (int k = 1; k < 6; ++k)
                if (x >= off_[k]) { r = k; q = qs_[k] + (x - off_[k]); ncx = ncx_[k]; ncy = ncy_[k]; }
            bool hit = false;
            uint32_t j = 0;
            // stream: a neighbour cell in the tile to the left -> that tile's stream; a slot that is not small (all its
            // pairs) and a small slot's pair with a large dynamic collider -> the fallback list
            int st = (!small_slot) ? 2 : (r >= 1 && r < 5 && (ncx >> g.tlx) < txp) ? 1 : 0;
            // class of a tile pair (DESIGN.md "Contract / tiles"): 0 interior, 1 x seam, 2 y seam, 3 corner -- where the two
            // home tiles lie; it rides in the two top bits of the pair's second slot word
            uint32_t cls = 0u;
            bool ddx = false, ddy = false;
            if (small_slot && r >= 1 && r < 5) {
                // (wrapped tile coordinates: with one tile column / row every home tile is the same one)
                ddx = ((uint32_t)((ncx >> g.tlx) ^ txp) & ((1u << (g.gxl - g.tlx)) - 1u)) != 0u;
                ddy = ((uint32_t)((ncy >> g.tly) ^ typ) & ((1u << (g.gyl - g.tly)) - 1u)) != 0u;
                cls = ddx ? (ddy ? 3u : 1u) : (ddy ? 2u : 0u);
            }
            uint32_t qw = q;   // the candidate's slot word: its body's slot when it is a further collider of a dynamic body
            if (x < L) {
                // the candidate's cell, box, id and filter words: four loads side by side (the slot order makes them
                // sequential), nothing waits for another
                const int2 cq = sorted_cell[q];
                const float4 Aq = sorted_aabb[q];
                const uint32_t jq = sorted_id[q];
                const uint4 fj = sorted_filt[q];
                const bool ok = r >= 5 || (cq.x == ncx && cq.y == ncy);
                if (ok && aabb_overlap(Ai, Aq)) {
                    j = jq;
                    hit = pair_filter(fi, fj);
                    if (r == 5 && small_slot && (fj.w & kFiltDynamic)) st = 2;  // small against a large dynamic collider
                    if constexpr (CMP) {
                        if (hit && ((fi.w | fj.w) & kFiltFurther)) {   // a further collider of a dynamic body: in the tiles when it sits beside its body
                            const TileEnd Ej = tile_end(q, cq, fj, ns, g, P.slot_of, sorted_cell);
                            if (!tile_end_ok(Ei, c, ddx, ddy) || !tile_end_ok(Ej, cq, ddx, ddy)) st = 2;
                            qw = Ej.slot;
                        }
                    }
                }
            }
            emit(hit, j, qw, st, cls);
        }
        // Second level: a small slot against the medium colliders homed in the level-1 cells [c1(lo) - 1, c1(hi)] on both
        // axes (an overlapping medium collider's home cell lies there: its box is under 1.9 such cells wide).  Up to nine
        // cells, joined end to end like the ranges above; nothing of this runs in a scene without medium colliders.
        if (nsm > ns) {
            uint32_t qs1[9], off1[10];
            int32_t X0 = 0, Y0 = 0;
            uint32_t nx1 = 0, ny1 = 0;
            if (test && small_slot) {
                X0 = cell_coord(Ai.x, g.inv_cell1) - 1; Y0 = cell_coord(Ai.y, g.inv_cell1) - 1;
                nx1 = (uint32_t)(cell_coord(Ai.z, g.inv_cell1) - X0) + 1u; ny1 = (uint32_t)(cell_coord(Ai.w, g.inv_cell1) - Y0) + 1u;
                nx1 = min(nx1, 3u); ny1 = min(ny1, 3u);   // (a small box spans at most two level-1 cells per axis)
            }
            off1[0] = 0u;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                uint32_t qs = 0, qe = 0;
                if ((uint32_t)k < nx1 * ny1) {
                    const uint32_t bk = bucket1_of(X0 + (int32_t)((uint32_t)k % nx1), Y0 + (int32_t)((uint32_t)k / nx1), g);
                    qs = cell_start[bk]; qe = cell_start[bk + 1];
                }
                qs1[k] = qs;
                off1[k + 1] = off1[k] + (qe - qs);
            }
            const uint32_t L1 = off1[9];
            for (uint32_t x0 = 0; __any(x0 < L1); x0 += kGroup) {
                const uint32_t x = x0 + (uint32_t)gl;
                uint32_t k1 = 0, q = qs1[0] + x;
#pragma unroll
                for (int k = 1; k < 9; ++k)
                    if (x >= off1[k]) { k1 = (uint32_t)k; q = qs1[k] + (x - off1[k]); }
                bool hit = false;
                uint32_t j = 0;
                int st = 0;   // small + medium: the small collider's tile, unless the medium one is dynamic
                uint32_t qw = q;
                if (x < L1) {
                    const int2 cq = sorted_cell[q];
                    const float4 Aq = sorted_aabb[q];
                    const uint32_t jq = sorted_id[q];
                    const uint4 fj = sorted_filt[q];
                    if (cq.x == X0 + (int32_t)(k1 % nx1) && cq.y == Y0 + (int32_t)(k1 / nx1) && aabb_overlap(Ai, Aq)) {
                        j = jq;
                        hit = pair_filter(fi, fj);
                        if (fj.w & kFiltDynamic) st = 2;
                        if constexpr (CMP) {
                            if (hit && ((fi.w | fj.w) & kFiltFurther)) {
                                const TileEnd Ej = tile_end(q, cq, fj, ns, g, P.slot_of, sorted_cell);   // (a medium slot: never ok)
                                if (!tile_end_ok(Ei, c, false, false) || !tile_end_ok(Ej, cq, false, false)) st = 2;
                                qw = Ej.slot;
                            }
                        }
                    }
                }
                emit(hit, j, qw, st, 0u);
            }
        }
        if (FILL && copy) {
#pragma unroll
            for (int s3 = 0; s3 < 2; ++s3) {
                const uint32_t c = s3 == 0 ? cc0 : cc1;
                for (uint32_t h = (uint32_t)gl; h < c; h += kGroup) {
                    const uint2 e = P.hit_tmp[(size_t)p * kHitTmp + (s3 == 0 ? h : kHitTmp - 1u - h)];
                    const uint32_t j = e.x, q = e.y & kSlotMask, cls = e.y >> kClassShift, dst = base[s3] + h;
                    const bool swap = j < i;
                    if (to_fb[s3]) {
                        if (dst < P.fb_cap) P.fpairs[dst] = make_uint2(swap ? j : i, swap ? i : j);
                    } else if (dst < P.max_pairs) {
                        P.rawp[dst] = make_uint4(swap ? j : i, swap ? i : j, swap ? q : Ei.slot, (swap ? Ei.slot : q) | (cls << kClassShift));
                    }
                }
            }
        }
        if (!FILL) {
            if (live && gl == 0) { P.slot_cnt[3 * p] = cnt[0]; P.slot_cnt[3 * p + 1] = cnt[1]; P.slot_cnt[3 * p + 2] = cnt[2]; }
            // (the tiles' pair counts are summed from slot_cnt by k_tile_count: two atomics per wave on the tiles'
            // counters -- some fifty on each -- cost this pass 25 of its 47 us)
        }
    }
}

// ---------------------------------------------------------------- K5 solver tiles
// slot ranges of the four tiles of the block anchored at wrapped tile t, in the order
// (0,0) (1,0) (0,1) (1,1); off[k] = local index of the first body of tile k, off[4] = bodies
struct BlockRanges { uint32_t ts[4], te[4], off[5]; };

__device__ __forceinline__ void block_ranges(uint32_t t, const GridParams& g, const uint32_t* __restrict__ cell_start,
                                             BlockRanges& R) {
    const uint32_t lx = g.gxl - g.tlx;
    const int32_t tx = (int32_t)(t & ((1u << lx) - 1u)), ty = (int32_t)(t >> lx);
    const bool one_row = g.tly == g.gyl;  // strips: the block is two tiles, the upper two would repeat them
    const bool one_col = g.tlx == g.gxl;  // one tile column (the whole grid as one tile): the right-hand ones likewise
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t tk = tile_of(tx + (k & 1), ty + (k >> 1), g);
        R.ts[k] = cell_start[tk << (g.tlx + g.tly)];
        R.te[k] = ((one_row && k >= 2) || (one_col && (k & 1))) ? R.ts[k] : cell_start[(tk + 1u) << (g.tlx + g.tly)];
        R.off[k] = run;
        run += R.te[k] - R.ts[k];
    }
    R.off[4] = run;
}

__device__ __forceinline__ uint32_t block_local(const BlockRanges& R, uint32_t slot) {
    uint32_t l = kExt;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (slot >= R.ts[k] && slot < R.te[k]) l = R.off[k] + (slot - R.ts[k]);
    return l;
}

__device__ __forceinline__ uint32_t block_slot(const BlockRanges& R, uint32_t l) {
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (l >= R.off[k] && l < R.off[k + 1]) s = R.ts[k] + (l - R.off[k]);
    return s;
}

// Pairs per tile after the count pass, one wave per tile: the stream-0 counts of the tile's own slots (slots are
// sorted tile-major, a tile's colliders are one contiguous range) plus the stream-1 counts of the slots of the tile
// to its right, whose pairs with the neighbour cell (-1, +1) this tile owns.
template <int LPT>  // lanes per tile: 64 for tiles of hundreds of colliders (strips), 8 for the small tiles of wide grids
__global__ __launch_bounds__(kBlock) void k_tile_count(GridParams g, const uint32_t* __restrict__ cell_start,
                                                        const uint32_t* __restrict__ slot_cnt, uint32_t* __restrict__ tile_cnt,
                                                        uint32_t* __restrict__ tile_own, uint32_t* __restrict__ slot_off) {
    const uint32_t t = (blockIdx.x * kBlock + threadIdx.x) / (uint32_t)LPT;
    const uint32_t lane = threadIdx.x & (uint32_t)(LPT - 1);
    const bool live = t < g.ntiles;
    const uint32_t sh = g.tlx + g.tly, lx = g.gxl - g.tlx;
    const int32_t tx = (int32_t)(t & ((1u << lx) - 1u)), ty = (int32_t)(t >> lx);
    const uint32_t tr = tile_of(tx + 1, ty, g);
    uint32_t s0 = 0, e0 = 0, s1 = 0, e1 = 0;
    if (live) {
        s0 = cell_start[t << sh]; e0 = cell_start[(t + 1u) << sh];
        s1 = cell_start[tr << sh]; e1 = cell_start[(tr + 1u) << sh];
    }
    // the tile's own slots: exclusive prefix sums of both streams (where the fill pass puts each slot's pairs)
    uint32_t run0 = 0, run1 = 0;
    for (uint32_t base = s0; __any(base < e0); base += (uint32_t)LPT) {  // the groups of a wave keep in step for the shuffles
        const uint32_t p = base + lane;
        const bool in = p < e0;
        const uint32_t c0 = in ? slot_cnt[3u * p] : 0u, c1 = in ? slot_cnt[3u * p + 1u] : 0u;
        uint32_t i0 = c0, i1 = c1;
#pragma unroll
        for (int off = 1; off < LPT; off <<= 1) {
            const uint32_t u0 = (uint32_t)__shfl_up((int)i0, off, LPT), u1 = (uint32_t)__shfl_up((int)i1, off, LPT);
            if ((int)lane >= off) { i0 += u0; i1 += u1; }
        }
        if (in) { slot_off[2u * p] = run0 + i0 - c0; slot_off[2u * p + 1u] = run1 + i1 - c1; }
        run0 += (uint32_t)__shfl((int)i0, LPT - 1, LPT);
        run1 += (uint32_t)__shfl((int)i1, LPT - 1, LPT);
    }
    // the stream-1 pairs of the right neighbour's slots are this tile's as well
    uint32_t sum = 0;
    for (uint32_t p = s1 + lane; p < e1; p += (uint32_t)LPT) sum += slot_cnt[3u * p + 1u];
#pragma unroll
    for (int off = LPT / 2; off > 0; off >>= 1) sum += (uint32_t)__shfl_xor((int)sum, off, 64);
    if (live && lane == 0) { tile_cnt[t] = run0 + sum; tile_own[t] = run0; }
}

// One thread per tile, after the count pass: tiles over capacity are flagged (their pair count
// is zeroed so that the scan skips them), the others listed per phase.
__device__ __forceinline__ void tile_list_step(uint32_t t, const GridParams& g, const uint32_t* __restrict__ cell_start,
                                               uint32_t* tile_cnt, uint32_t* __restrict__ tile_flag,
                                               uint32_t* __restrict__ active, Counters* __restrict__ ctr) {
    const uint32_t cnt = t < g.ntiles ? tile_cnt[t] : 0u;
    // 0..3: listed under that phase, 4: over capacity, 5: nothing to do
    uint32_t kind = 5u, nb = 0u;
    if (cnt) {
        BlockRanges R;
        block_ranges(t, g, cell_start, R);
        nb = R.off[4];
        const uint32_t lx = g.gxl - g.tlx;
        (void)lx;
        kind = (cnt > kMaxP || nb > kMaxB) ? 4u : 0u;   // one list: every class (phase) of a sweep walks the same tiles
    }
    if (!__any(kind != 5u)) return;
    // one atomic per wave and list: the lanes of a kind take consecutive entries
    const int lane = threadIdx.x & 63;
    const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) {
        const uint64_t bal = __ballot(kind == k);
        if (!bal) continue;
        uint32_t base = 0;
        const int leader = __builtin_ctzll(bal);
        if (lane == leader) {
            const uint32_t c = (uint32_t)__popcll(bal);
            base = atomicAdd(k < 4 ? &ctr->n_listed[k] : &ctr->n_over_tiles, c);
        }
        base = __shfl(base, leader, 64);
        if (kind == k) {
            const uint32_t idx = base + (uint32_t)__popcll(bal & below);
            if (k < 4) active[k * g.pstride + idx] = t;
            else { tile_flag[t] = 1u; tile_cnt[t] = 0u; }
        }
    }
    // statistics: wave maxima, one atomic each
    uint32_t mp = (kind < 4u) ? cnt : 0u, mb = (kind < 4u) ? nb : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mp = max(mp, (uint32_t)__shfl_xor((int)mp, off, 64));
        mb = max(mb, (uint32_t)__shfl_xor((int)mb, off, 64));
    }
    if (lane == 0 && mp) { atomicMax(&ctr->max_tile_pairs, mp); atomicMax(&ctr->max_tile_bodies, mb); }
}

__global__ __launch_bounds__(kBlock) void k_tile_list(GridParams g, const uint32_t* __restrict__ cell_start,
                                                       uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ tile_flag,
                                                       uint32_t* __restrict__ active, Counters* __restrict__ ctr) {
    tile_list_step(blockIdx.x * kBlock + threadIdx.x, g, cell_start, tile_cnt, tile_flag, active, ctr);
}

// Grids of fewer than kScanTile tiles: one workgroup lists the tiles and then scans their pair counts (tile_cnt[0..ntiles]
// becomes the tiles' pair offsets), one launch instead of two.
__global__ __launch_bounds__(kBlock) void k_tile_list_scan(GridParams g, const uint32_t* __restrict__ cell_start,
                                                            uint32_t* tile_cnt, uint32_t* __restrict__ tile_flag,
                                                            uint32_t* __restrict__ active, Counters* __restrict__ ctr) {
    for (uint32_t t0 = 0; t0 < g.ntiles; t0 += kBlock)
        tile_list_step(t0 + threadIdx.x, g, cell_start, tile_cnt, tile_flag, active, ctr);
    __syncthreads();  // the zeroed counts of tiles over capacity are visible to the scan
    scan_tile(0u, tile_cnt, g.ntiles + 1u, nullptr, nullptr, 1);
}

// pair range of tile t after the scan, clamped to the pair capacity
__device__ __forceinline__ void tile_pair_range(uint32_t t, const uint32_t* __restrict__ tile_start, uint32_t max_pairs,
                                                uint32_t& start, uint32_t& np) {
    const uint32_t s = min(tile_start[t], max_pairs), e = min(tile_start[t + 1], max_pairs);
    start = s; np = e - s;
}

// A batch: up to kMaxGroup tiles of one phase solved together by one workgroup (their blocks
// are disjoint, so they do not interact; each tile is still coloured on its own, the batch only
// runs colour c of all its tiles in the same step).  Batching is scheduling only: results do
// not depend on which tiles share a batch.
constexpr uint32_t kBatchNear = 0x80000000u;   // bit 31 of BatchDesc.count: a NEAR batch (see k_near_closure)
constexpr uint32_t kNoSplit = 0xFFFFFFFFu;     // Counters.n_near of a tick whose velocity sweep is not split
struct BatchDesc {
    uint32_t first;    // index of the first tile on the phase's list
    uint32_t count;    // tiles | kBatchNear
    uint32_t bstart;   // first pair in tpair / first record slot
    uint32_t np;       // pairs within reach: the ones the solves take (the others follow them in tpair)
    uint32_t ncol;     // most colours of any of its tiles
    uint32_t nb;       // bodies of all its blocks
    uint32_t boff;     // first entry of its body list (body ids in local-index order)
    uint32_t roff;     // its colour run table: bruns[64 * roff + c] = first | end << 16 of colour c in its pair list
};

// body ranges of a batch in LDS: range r = 4 * (tile in batch) + (block tile 0..3)
struct BatchRanges {
    uint32_t rs[4 * kMaxGroup];       // first slot
    uint32_t off[4 * kMaxGroup + 1];  // local index of the range's first body
};

// all threads of the workgroup; tiles = the batch's tile ids (global memory)
__device__ __forceinline__ void batch_ranges(const uint32_t* __restrict__ tiles, uint32_t count, const GridParams& g,
                                             const uint32_t* __restrict__ cell_start, BatchRanges& S) {
    __syncthreads();
    if (threadIdx.x < 4u * count) {
        const uint32_t t = tiles[threadIdx.x >> 2], k = threadIdx.x & 3u;
        const uint32_t lx = g.gxl - g.tlx;
        const int32_t tx = (int32_t)(t & ((1u << lx) - 1u)), ty = (int32_t)(t >> lx);
        const uint32_t tk = tile_of(tx + (int32_t)(k & 1u), ty + (int32_t)(k >> 1), g);
        const uint32_t a = cell_start[tk << (g.tlx + g.tly)];
        const uint32_t b = ((g.tly == g.gyl && k >= 2u) || (g.tlx == g.gxl && (k & 1u))) ? a : cell_start[(tk + 1u) << (g.tlx + g.tly)];
        S.rs[threadIdx.x] = a;
        S.off[threadIdx.x + 1] = b - a;  // lengths first, prefix below
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        S.off[0] = 0;
        for (uint32_t r = 0; r < 4u * count; ++r) { const uint32_t len = S.off[r + 1]; S.off[r + 1] = run + len; run += len; }
    }
    __syncthreads();
}

// slot of local body l (binary search over the range offsets)
__device__ __forceinline__ uint32_t batch_slot(const BatchRanges& S, uint32_t nr, uint32_t l) {
    uint32_t lo = 0, hi = nr;  // largest r with off[r] <= l
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (S.off[mid] <= l) lo = mid; else hi = mid;
    }
    return S.rs[lo] + (l - S.off[lo]);
}

// local index of a slot known to belong to tile j of the batch (kExt when outside its block)
__device__ __forceinline__ uint32_t batch_local(const BatchRanges& S, uint32_t j, uint32_t slot) {
    uint32_t l = kExt;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t r = 4u * j + k, len = S.off[r + 1] - S.off[r];
        if (slot >= S.rs[r] && slot - S.rs[r] < len) l = S.off[r] + (slot - S.rs[r]);
    }
    return l;
}

// frame-start state and pads for the reach test of k_batch_color
struct ReachParams {
    const float4* pose;
    const float4* vel;
    const float4* lpose;
    const float4* poly;
    float margin, vel_pad_max, dt, sm;
};

// Reach test of every tile-owned pair (DESIGN.md "Contract / solver pairs"), one pair per lane:
// a pair that cannot touch in this tick gets bit 31 of its first slot word set.
constexpr uint32_t kOutOfReach = 0x80000000u;
// ... and its shape-pair class (circle-circle 2, circle-core 1, core-core 0, anything with a POLY 3: the narrowphase order
// of k_batch_color) in bits 28..29 of the same word: k_reach has the shapes in hand, the colouring would gather them again
constexpr uint32_t kShapeClassShift = 28u, kSlotWordMask = (1u << 24) - 1u;
constexpr uint32_t kNoPair = 0xFFFFFFFFu;  // torder entry of a pair that is not solved this tick

__global__ __launch_bounds__(kBlock) void k_reach(const uint32_t* __restrict__ tile_start, uint32_t ntiles, uint32_t max_pairs,
                                                   uint4* __restrict__ rawp, const float4* __restrict__ shape, ReachParams RP) {
    const uint32_t total = min(tile_start[ntiles], max_pairs);
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < total; k += gridDim.x * kBlock) {
        const uint4 rp = rawp[k];
        const float4 sa = shape[rp.x], sb = shape[rp.y];
        float pax, pay, pbx, pby;
        motion_pads(RP.vel[rp.x], sa, RP.lpose, rp.x, RP.margin, RP.vel_pad_max, RP.dt, pax, pay);
        motion_pads(RP.vel[rp.y], sb, RP.lpose, rp.y, RP.margin, RP.vel_pad_max, RP.dt, pbx, pby);
        const bool keep = pair_in_reach(shape_pose(RP.pose[rp.x], sa, RP.lpose, rp.x), sa, sqrtf(pax * pax + pay * pay),
                                        shape_pose(RP.pose[rp.y], sb, RP.lpose, rp.y), sb, sqrtf(pbx * pbx + pby * pby), RP.poly, RP.sm);
        const uint32_t ta = __float_as_uint(sa.x) & 0xFFu, tb = __float_as_uint(sb.x) & 0xFFu;
        const uint32_t ncirc = (ta == SFX_SHAPE_CIRCLE ? 1u : 0u) + (tb == SFX_SHAPE_CIRCLE ? 1u : 0u);
        const uint32_t scl = (ta == SFX_SHAPE_POLY || tb == SFX_SHAPE_POLY) ? 3u : 2u - ncirc;
        rawp[k].z = rp.z | (keep ? 0u : kOutOfReach) | (scl << kShapeClassShift);
    }
}

// Forms the batches and edge-colours every tile's pair graph, all state in LDS (DESIGN.md
// "Contract / colouring").  Pairs that failed the reach test are not coloured; they are listed
// behind the batch's pairs with SFX_PRUNED_COLOR.  A workgroup takes `group` consecutive tiles of a phase's list and
// cuts them greedily into batches of at most kBatchP pairs and kMaxB bodies.  Per batch it writes the pairs colour-major as
// (id a, id b, local a | local b << 16, colour | tile << 8) and a BatchDesc on the phase's table.
#ifndef SFX_COLOR_THREADS
#define SFX_COLOR_THREADS 512
#endif
constexpr int kColorThreads = SFX_COLOR_THREADS;
template <uint32_t MP>
__global__ __launch_bounds__(kColorThreads, 4) void k_batch_color(GridParams g, uint32_t group, const uint32_t* __restrict__ cell_start,
                                                         const uint32_t* __restrict__ sorted_id,
                                                         const uint4* __restrict__ filt /* in slot order (sorted_filt) */,
                                                         const uint32_t* __restrict__ tile_start,
                                                         const uint32_t* __restrict__ list, const uint4* __restrict__ rawp,
                                                         uint4* __restrict__ tpair, BatchDesc* __restrict__ batches,
                                                         const float4* __restrict__ shape, uint4* __restrict__ torder,
                                                         uint32_t* __restrict__ tile_batch,
                                                         uint32_t* __restrict__ bbody, uint32_t bbody_cap,
                                                         const float2* __restrict__ mass, float2* __restrict__ bmass,
                                                         uint32_t* __restrict__ bruns, uint32_t bruns_cap,
                                                         uint32_t max_pairs, uint32_t rounds, Counters* __restrict__ ctr,
                                                         unsigned long long* __restrict__ dbg) {
    struct { unsigned long long* dbg; } T = {dbg};  // for the stage-timing macros
    (void)T;
    constexpr uint32_t MB = kMaxB;
    constexpr uint32_t kBlock = kColorThreads;  // shadows the file-level constant
    __shared__ unsigned long long s_used[MB];
    __shared__ unsigned long long s_claim[MB * kClaim];
    __shared__ uint8_t s_dyn[MB];
    __shared__ uint32_t s_a[MP], s_b[MP];
    __shared__ uint16_t s_la[MP], s_lb[MP];
    // the leftover lists of the serial finish live in the claim table, which is dead once the rounds are
    // over: 8 KB less LDS, so that two of these workgroups fit on a CU
    static_assert(2 * MP * sizeof(uint16_t) <= MB * kClaim * sizeof(unsigned long long), "leftover lists alias s_claim");
    uint16_t* const s_left = reinterpret_cast<uint16_t*>(s_claim);
    uint16_t* const s_srt = s_left + MP;
    __shared__ uint8_t s_col[MP], s_scl[MP], s_tj[MP];   // colour, shape-pair class (from k_reach), tile of the batch
    __shared__ uint32_t s_hist[SFX_MAX_COLORS], s_base[SFX_MAX_COLORS];
    __shared__ uint32_t s_nleft, s_ovf, s_bstart, s_nbat, s_npruned, s_ptail, s_ptail_first;
    __shared__ uint32_t s_cls[4];
    __shared__ uint32_t s_bidx, s_boff;
    __shared__ uint32_t s_ext[kMaxExt], s_next;  // large colliders outside the block that the batch's pairs touch
    __shared__ BatchRanges S;
    __shared__ uint32_t s_tile[kMaxGroup], s_tnb[kMaxGroup], s_tnp[kMaxGroup], s_tstart[kMaxGroup];
    __shared__ uint32_t s_bfirst[kMaxGroup + 1], s_poff[kMaxGroup + 1];
    __shared__ uint8_t s_mark[MB];        // block bodies that a kept pair of this class touches: the batch's LDS bodies
    __shared__ uint16_t s_remap[MB];      // block-local index -> index in the batch's (compact) body list
    __shared__ uint32_t s_scan[kColorThreads / 64 + 1];
    __shared__ uint32_t s_np, s_nbc, s_near;
    const uint32_t tid = threadIdx.x;
    const uint32_t cls = blockIdx.y;      // the class of pairs (= phase of the sweep) this workgroup forms batches of
    const uint32_t nlist = ctr->n_listed[0];
    const uint32_t* lst = list;
    group = min(max(group, 1u), (uint32_t)kMaxGroup);
    const uint32_t ngroups = (nlist + group - 1u) / group;
    for (uint32_t gi = blockIdx.x; gi < ngroups; gi += gridDim.x) {
        const uint32_t e0 = gi * group, cnt = min(group, nlist - e0);
        __syncthreads();
        SFX_CSTAMP(cq0);
        if (tid < cnt) {
            const uint32_t t = lst[e0 + tid];
            BlockRanges R;
            block_ranges(t, g, cell_start, R);
            uint32_t start, np;
            tile_pair_range(t, tile_start, max_pairs, start, np);
            s_tile[tid] = t; s_tnb[tid] = min(R.off[4], kMaxB); s_tnp[tid] = min(np, MP); s_tstart[tid] = start;
        }
        __syncthreads();
        if (tid == 0) {  // greedy cut into batches
            uint32_t nbat = 0, nb = 0, np = 0;
            s_bfirst[0] = 0;
            for (uint32_t j = 0; j < cnt; ++j) {
                if (j > s_bfirst[nbat] && (nb + s_tnb[j] > kMaxB || np + s_tnp[j] > kBatchP)) {
                    s_bfirst[++nbat] = j; nb = 0; np = 0;
                }
                nb += s_tnb[j]; np += s_tnp[j];
            }
            s_bfirst[nbat + 1] = cnt;
            s_nbat = nbat + 1;
        }
        __syncthreads();
        const uint32_t nbat = s_nbat;
        SFX_CSTAMP(cq1);
        for (uint32_t bi = blockIdx.z; bi < nbat; bi += gridDim.z) {  // the batches of a group are coloured side by side
            const uint32_t f = s_bfirst[bi], c = s_bfirst[bi + 1] - f;
            SFX_TT(tc0);
            batch_ranges(lst + e0 + f, c, g, cell_start, S);
            if (tid == 0) {
                uint32_t run = 0;
                for (uint32_t j = 0; j < c; ++j) { s_poff[j] = run; run += s_tnp[f + j]; }
                s_poff[c] = run;
                s_nleft = 0u; s_ovf = 0u; s_npruned = 0u; s_np = 0u; s_near = 0u;
            }
            if (tid < kMaxExt) s_ext[tid] = 0xFFFFFFFFu;
            if (tid < SFX_MAX_COLORS) s_hist[tid] = 0u;
            if (tid < 4) s_cls[tid] = 0u;
            __syncthreads();
            SFX_CSTAMP(cq2);
            const uint32_t nb = min(S.off[4u * c], MB), np_all = min(s_poff[c], MP);
            for (uint32_t l = tid; l < nb; l += kBlock) {
                s_dyn[l] = (uint8_t)(filt[batch_slot(S, 4u * c, l)].w & (kFiltDynamic | kFiltHalo));   // (filt: the filter words in slot order)
                s_mark[l] = 0u;
                s_used[l] = 0ull;
#pragma unroll
                for (int k = 0; k < kClaim; ++k) s_claim[l * kClaim + k] = ~0ull;
            }
            SFX_CSTAMP(cq3);
            __syncthreads();  // the marks are cleared before the pairs set them
            int open_edges = 0;
            for (uint32_t e0 = tid; e0 < np_all; e0 += kBlock) {  // the tiles' pairs of this class, in no particular order
                uint32_t j = 0;
                while (j + 1 < c && s_poff[j + 1] <= e0) ++j;
                const uint4 rp = rawp[s_tstart[f + j] + (e0 - s_poff[j])];
                if ((rp.w >> kClassShift) != cls) continue;
                const uint32_t e = atomicAdd(&s_np, 1u);
                s_a[e] = rp.x; s_b[e] = rp.y;
                s_la[e] = (uint16_t)batch_local(S, j, rp.z & kSlotWordMask);
                s_scl[e] = (uint8_t)((rp.z >> kShapeClassShift) & 3u);
                s_lb[e] = (uint16_t)batch_local(S, j, rp.w & kSlotMask);
                s_tj[e] = (uint8_t)j;
                const bool keep = !(rp.z & kOutOfReach);
                s_col[e] = keep ? SFX_NO_COLOR : SFX_PRUNED_COLOR;
                if (!keep) atomicAdd(&s_npruned, 1u);
                open_edges |= keep ? 1 : 0;
                if (keep) {  // (plain stores of the same value: no lane minds who wins)
                    if (s_la[e] != kExt && s_la[e] < MB) s_mark[s_la[e]] = 1u;
                    if (s_lb[e] != kExt && s_lb[e] < MB) s_mark[s_lb[e]] = 1u;
                }
                if (keep && (s_la[e] == kExt || s_lb[e] == kExt)) {  // an end outside the block: give it an LDS slot of the solves
                    const uint32_t id = s_la[e] == kExt ? rp.x : rp.y;
                    uint32_t k = 0;
                    for (; k < kMaxExt; ++k) {
                        const uint32_t old = atomicCAS(&s_ext[k], 0xFFFFFFFFu, id);
                        if (old == 0xFFFFFFFFu || old == id) break;
                    }
                    if (k == kMaxExt) atomicOr(&s_ovf, 2u);  // more outside bodies than slots: the pair is dropped, the tick reports it
                }
            }
            int any_open = __syncthreads_or(open_edges);
            SFX_CSTAMP(cq4);
            const uint32_t np = s_np;
            if (np == 0u) continue;  // no pair of this class in these tiles: no batch (uniform: s_np is read after the barrier)
            {   // the batch's body list holds the touched block bodies only: the units of one class may share a tile's bodies
                // in their blocks, never in their pairs -- what a batch does not touch it must not write back either
                uint32_t m[(MB + kColorThreads - 1) / kColorThreads], cnt = 0u;
#pragma unroll
                for (uint32_t q = 0; q < (MB + kColorThreads - 1) / kColorThreads; ++q) {
                    const uint32_t l = tid * ((MB + kColorThreads - 1) / kColorThreads) + q;
                    m[q] = (l < nb && s_mark[l]) ? 1u : 0u;
                    cnt += m[q];
                }
                uint32_t total;
                uint32_t pre = block_exclusive_scan<kColorThreads>(cnt, s_scan, &total);
#pragma unroll
                for (uint32_t q = 0; q < (MB + kColorThreads - 1) / kColorThreads; ++q) {
                    const uint32_t l = tid * ((MB + kColorThreads - 1) / kColorThreads) + q;
                    if (l < nb) s_remap[l] = (uint16_t)pre;
                    pre += m[q];
                }
                if (tid == 0) s_nbc = total;
            }
            __syncthreads();
            const uint32_t nbc = s_nbc;
            SFX_CSTAMP(cq5);
            if (tid < 64u) {  // the batch's body list: touched block bodies, then the outside bodies (filled from slot 0 on)
                static_assert(kMaxExt == 64, "one lane per slot");
                const unsigned long long have = __ballot(s_ext[tid] != 0xFFFFFFFFu);
                const uint32_t next = (uint32_t)__popcll(have);
                uint32_t boff = 0u;
                if (tid == 0) boff = atomicAdd(&ctr->body_cursor, nbc + next);
                boff = (uint32_t)__shfl((int)boff, 0, 64);
                if (tid == 0) { s_boff = boff; s_next = next; }
                if (tid < next && boff + nbc + tid < bbody_cap) { bbody[boff + nbc + tid] = s_ext[tid]; bmass[boff + nbc + tid] = mass[s_ext[tid]]; }
            }
            __syncthreads();
            SFX_CSTAMP(cq6);
            for (uint32_t l = tid; l < nb; l += kBlock)  // the tile solves load their bodies from this list
                if (s_mark[l] && s_boff + s_remap[l] < bbody_cap) {
                    const uint32_t id = sorted_id[batch_slot(S, 4u * c, l)];
                    bbody[s_boff + s_remap[l]] = id;
                    bmass[s_boff + s_remap[l]] = mass[id];   // the masses in list order: the tile solves fetch them beside the ids, not after them
                    if (s_dyn[l] & kFiltHalo) s_near = 1u;   // the batch touches a halo body (sent or received): see k_near_closure
                }
            SFX_TT(tc1);
            SFX_CSTAMP(cq7);
#if defined(SFX_PROBE_COLOR_STAGE) && SFX_PROBE_COLOR_STAGE == 1   // timing probe: a batch ends after its set-up (results are wrong)
            continue;
#endif
#define TDYN(l) ((l) != kExt && (l) < MB && (s_dyn[(l)] & kFiltDynamic))
            // A thread owns the same (at most MP / threads) edges in every round: their ends, keys, dynamic flags, colour and
            // proposal stay in registers, so a round is one LDS read (the ends' used colours / the claims) and one LDS
            // atomic per edge and phase instead of a chain of dependent reads.
            constexpr uint32_t EPT = (MP + kBlock - 1) / kBlock;
            uint32_t e_l[EPT], e_a[EPT], e_b[EPT];
            uint8_t e_col[EPT], e_prop[EPT], e_dyn[EPT];
#pragma unroll
            for (uint32_t k = 0; k < EPT; ++k) {
                const uint32_t e = tid + k * kBlock;
                e_l[k] = 0u; e_a[k] = 0u; e_b[k] = 0u; e_dyn[k] = 0u; e_prop[k] = SFX_NO_COLOR;
                e_col[k] = 0u;   // (anything but SFX_NO_COLOR: an edge that does not exist is never open)
                if (e < np) {
                    const uint32_t la = s_la[e], lb = s_lb[e];
                    e_l[k] = la | (lb << 16); e_a[k] = s_a[e]; e_b[k] = s_b[e]; e_col[k] = s_col[e];
                    e_dyn[k] = (uint8_t)((TDYN(la) ? 1u : 0u) | (TDYN(lb) ? 2u : 0u));
                }
            }
            for (uint32_t r = 0; r < rounds && any_open; ++r) {  // everything coloured: later rounds would do nothing
                SFX_TT_ADD(98, 0ull, 1ull);
#pragma unroll
                for (uint32_t k = 0; k < EPT; ++k) {  // propose
                    if (e_col[k] != SFX_NO_COLOR) continue;
                    const uint32_t la = e_l[k] & 0xFFFFu, lb = e_l[k] >> 16;
                    const bool da = e_dyn[k] & 1u, db = e_dyn[k] & 2u;
                    const unsigned long long freem = ~((da ? s_used[la] : 0ull) | (db ? s_used[lb] : 0ull));
                    if (freem == 0ull) { e_prop[k] = SFX_NO_COLOR; atomicOr(&s_ovf, 1u); continue; }
                    const uint32_t cc = (uint32_t)__builtin_ctzll(freem);  // first fit
                    const uint32_t h = edge_hash(e_a[k], e_b[k], r);
                    e_prop[k] = (uint8_t)cc;
                    const unsigned long long p = edge_prio(e_a[k], e_b[k], r, h);
                    if (da) atomicMin(&s_claim[la * kClaim + (cc & (kClaim - 1u))], p);
                    if (db) atomicMin(&s_claim[lb * kClaim + (cc & (kClaim - 1u))], p);
                }
                __syncthreads();
                open_edges = 0;
#pragma unroll
                for (uint32_t k = 0; k < EPT; ++k) {  // resolve
                    if (e_col[k] != SFX_NO_COLOR) continue;
                    const uint32_t cc = e_prop[k];
                    if (cc == SFX_NO_COLOR) { open_edges = 1; continue; }
                    const uint32_t la = e_l[k] & 0xFFFFu, lb = e_l[k] >> 16;
                    const bool da = e_dyn[k] & 1u, db = e_dyn[k] & 2u;
                    const uint32_t h = edge_hash(e_a[k], e_b[k], r);
                    const unsigned long long p = edge_prio(e_a[k], e_b[k], r, h);
                    const bool win = (!da || s_claim[la * kClaim + (cc & (kClaim - 1u))] == p) &&
                                     (!db || s_claim[lb * kClaim + (cc & (kClaim - 1u))] == p);
                    if (win) {
                        e_col[k] = (uint8_t)cc;
                        s_col[tid + k * kBlock] = (uint8_t)cc;
                        if (da) atomicOr(&s_used[la], 1ull << cc);
                        if (db) atomicOr(&s_used[lb], 1ull << cc);
                    } else {
                        open_edges = 1;
                    }
                }
                any_open = __syncthreads_or(open_edges);
            }
            SFX_TT(tc2);
            SFX_CSTAMP(cq8);
#if defined(SFX_PROBE_COLOR_STAGE) && SFX_PROBE_COLOR_STAGE == 2   // ... after the colouring rounds
            continue;
#endif
            // serial first-fit finish over the leftovers, ascending key
            for (uint32_t e = tid; e < np; e += kBlock)
                if (s_col[e] == SFX_NO_COLOR) s_left[atomicAdd(&s_nleft, 1u)] = (uint16_t)e;
            __syncthreads();
            const uint32_t nleft = s_nleft;
            if (nleft) {
                for (uint32_t q = tid; q < nleft; q += kBlock) {
                    const uint32_t e = s_left[q];
                    const unsigned long long key = ((unsigned long long)s_a[e] << 24) | s_b[e];
                    uint32_t rank = 0;
                    for (uint32_t q2 = 0; q2 < nleft; ++q2) {
                        const uint32_t e2 = s_left[q2];
                        rank += ((((unsigned long long)s_a[e2] << 24) | s_b[e2]) < key) ? 1u : 0u;
                    }
                    s_srt[rank] = (uint16_t)e;
                }
                __syncthreads();
                if (tid == 0) {
                    for (uint32_t q = 0; q < nleft; ++q) {
                        const uint32_t e = s_srt[q], la = s_la[e], lb = s_lb[e];
                        const bool da = TDYN(la), db = TDYN(lb);
                        const unsigned long long freem = ~((da ? s_used[la] : 0ull) | (db ? s_used[lb] : 0ull));
                        if (freem == 0ull) { s_ovf |= 1u; continue; }  // stays uncoloured: dropped from the solve
                        const uint32_t cc = (uint32_t)__builtin_ctzll(freem);
                        s_col[e] = (uint8_t)cc;
                        if (da) s_used[la] |= 1ull << cc;
                        if (db) s_used[lb] |= 1ull << cc;
                    }
                    atomicAdd(&ctr->n_leftover, nleft);
                }
                __syncthreads();
            }
#undef TDYN
            SFX_TT(tc3);
#if defined(SFX_PROBE_COLOR_STAGE) && SFX_PROBE_COLOR_STAGE == 3   // ... after the serial finish
            continue;
#endif
            // the batch's places in the colour-run table, the pair list and the phase's batch table: the three returning
            // atomics leave together and come back while the histogram below is made (one after the other, each behind its
            // own wait, they were three round trips on the batch's chain; reserved at the START of a batch they made every
            // workgroup queue for its body-list offset: + 9 us, measured)
            uint32_t r_roff = 0u, r_bstart = 0u, r_bidx = 0u;
            if (tid == 0) {
                r_roff = atomicAdd(&ctr->run_cursor, 1u);
                r_bstart = atomicAdd(&ctr->tp_cursor, np);
                r_bidx = atomicAdd(&ctr->n_batches[cls], 1u);
            }
            // colour-major order inside the batch (uncoloured pairs last)
            for (uint32_t e = tid; e < np; e += kBlock) {
                const uint32_t cc = s_col[e];
                if (cc < SFX_MAX_COLORS) atomicAdd(&s_hist[cc], 1u);
            }
            __syncthreads();
            if (tid < 64u) {  // the first wave: exclusive scan of the colour histogram with shuffles
                static_assert(SFX_MAX_COLORS == 64, "one lane per colour");
                const uint32_t hc = s_hist[tid];
                const uint32_t inc = wave_inclusive_scan(hc);
                s_base[tid] = inc - hc;
                const uint64_t usedc = __ballot(hc != 0u);
                const uint32_t nc = usedc ? 64u - (uint32_t)__builtin_clzll(usedc) : 0u;
                const uint32_t run = (uint32_t)__shfl((int)inc, 63, 64);
                const uint32_t roff = (uint32_t)__shfl((int)r_roff, 0, 64);
                if (roff < bruns_cap) bruns[64u * roff + tid] = (inc - hc) | (inc << 16);  // what the tile solves loop over
              if (tid == 0) {
                s_nleft = run;  // reused: cursor of the uncoloured tail
                const uint32_t kept = np - s_npruned;
                s_ptail = kept;  // cursor of the pruned pairs behind them
                s_ptail_first = kept;
                atomicAdd(&ctr->n_kept, kept);
                const uint32_t bstart = r_bstart;
                s_bstart = bstart;
                const uint32_t ph = cls;
                BatchDesc d;
                d.first = e0 + f;
                d.count = c | (s_near ? kBatchNear : 0u); d.bstart = bstart; d.np = kept; d.ncol = nc; d.nb = nbc + s_next; d.boff = s_boff; d.roff = roff;
                const uint32_t bidx = r_bidx;
                batches[ph * g.pstride + bidx] = d;
                s_bidx = bidx;
                atomicMax(&ctr->n_colors, nc);
                if (s_ovf & 1u) atomicOr(&ctr->overflow, kOvfColors);
                if (s_ovf & 2u) atomicOr(&ctr->overflow, kOvfExt);
              }
            }
            // (the per-colour pair counts of sfx_frame_stats are tallied by k_stats_pairs when somebody asks: adding every
            // batch's histogram to one global table cost ~9000 device-scope atomics on two cache lines per tick)
            __syncthreads();
            const uint32_t bstart = s_bstart;
            if (tid < c) tile_batch[cls * g.ntiles + s_tile[f + tid]] = s_bidx + 1u;
            // narrowphase order: the batch's pairs grouped by shape-pair class (circle-circle, circle-
            // polygon core, core-core, anything with a POLY), so that the lanes of a wave take the
            // same branch of the narrowphase.  (The class came with the pair: k_reach wrote it.)
            for (uint32_t e = tid; e < np; e += kBlock) {
                if (s_col[e] == SFX_PRUNED_COLOR) continue;
                atomicAdd(&s_cls[s_scl[e]], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t run = 0;
                for (int k = 0; k < 4; ++k) { const uint32_t c4 = s_cls[k]; s_cls[k] = run; run += c4; }
            }
            __syncthreads();
            for (uint32_t e = tid; e < np; e += kBlock) {
                const uint32_t cc = s_col[e];
                const uint32_t dst = (cc < SFX_MAX_COLORS) ? atomicAdd(&s_base[cc], 1u)
                                     : (cc == SFX_PRUNED_COLOR) ? atomicAdd(&s_ptail, 1u) : atomicAdd(&s_nleft, 1u);
                uint32_t la = s_la[e], lb = s_lb[e];
                if (cc == SFX_PRUNED_COLOR) { la = 0u; lb = 0u; }  // never solved: its ends may not be on the body list
                else {
                    if (la != kExt) la = la < MB ? s_remap[la] : kExt;
                    if (lb != kExt) lb = lb < MB ? s_remap[lb] : kExt;
                }
                if ((la == kExt || lb == kExt) && cc != SFX_PRUNED_COLOR) {  // the outside end's LDS slot (kExt stays when there was none left)
                    const bool a_out = s_la[e] == kExt;
                    const uint32_t id = a_out ? s_a[e] : s_b[e];
                    uint32_t k = 0;
                    while (k < s_next && s_ext[k] != id) ++k;
                    const uint32_t lx = k < s_next ? nbc + k : kExt;
                    if (a_out) la = lx; else lb = lx;
                }
                tpair[bstart + dst] = make_uint4(s_a[e], s_b[e], la | (lb << 16), cc | (((cls << 22) | s_tile[f + s_tj[e]]) << 8));
                if (cc != SFX_PRUNED_COLOR) torder[bstart + atomicAdd(&s_cls[s_scl[e]], 1u)] = make_uint4(s_a[e], s_b[e], bstart + dst, cls);
            }
            for (uint32_t e = s_ptail_first + tid; e < np; e += kBlock) torder[bstart + e] = make_uint4(0u, 0u, kNoPair, 0u);  // pruned pairs: no narrowphase
            __syncthreads();
            SFX_TT(tc4);
            SFX_CSTAMP(cq9);
#ifdef SFX_STEP_PROBE
            if (blockIdx.x == 5u && blockIdx.y == 0u && blockIdx.z == 0u && tid == 0u && dbg) {
                const unsigned long long st_[10] = {cq0, cq1, cq2, cq3, cq4, cq5, cq6, cq7, cq8, cq9};
                for (int k_ = 0; k_ < 9; ++k_) atomicAdd(&dbg[80 + k_], st_[k_ + 1] - st_[k_]);
                atomicAdd(&dbg[89], 1ull);
            }
#endif
            SFX_TT_ADD(96, tc0, tc1); SFX_TT_ADD(97, tc1, tc2); SFX_TT_ADD(99, tc2, tc3); SFX_TT_ADD(100, tc3, tc4);
            SFX_TT_ADD(101, 0ull, 1ull); SFX_TT_ADD(102, tc0, tc4); SFX_TT_MAX(103, tc0, tc4);
        }
    }
}

// Edge colouring of one global edge list by ONE workgroup (joints at upload, the fallback list
// every tick): same rounds as k_tile_color, with the tables in global memory.  Optionally writes
// the edges colour-major (`sorted`) with the colour offsets (`start`, SFX_MAX_COLORS + 1).
constexpr int kGlobalColorBlock = 1024;

struct GlobalColor {
    const uint2* edges;            // (a, b), SFX_WORLD_BODY allowed
    const uint32_t* n_edges;       // device-side count
    uint32_t cap;                  // list capacity (count is clamped to it)
    int key_by_index;              // joints: keyed by edge index, pairs: by (a, b)
    const uint4* filt;             // .w = kFiltDynamic | ... | body << 8
    int ends_are_colliders;        // the fallback list: an edge's ends are colliders, the conflicts are between their bodies
    unsigned long long* used;      // [n_bodies]
    unsigned long long* claim;     // [n_bodies][kClaim]
    uint8_t* color;                // [cap] out
    uint8_t* prop;                 // [cap] scratch
    uint32_t* left_list;           // [kLeftCap] scratch
    uint2* sorted;                 // optional [cap]
    uint32_t* start;               // optional [SFX_MAX_COLORS + 1]
    uint32_t* n_colors_out;        // optional
    uint32_t rounds;
    Counters* ctr;                 // overflow / n_leftover / colour stats
    int tick_totals;               // 1: also publish n_pairs, n_tile_pairs and the overflow bits of the tick
    const uint32_t* tile_start;    // tick_totals: scanned tile pair counts
    uint32_t ntiles, max_pairs;
};

__global__ __launch_bounds__(kGlobalColorBlock) void k_color_global(GlobalColor A) {
    __shared__ uint32_t s_hist[SFX_MAX_COLORS], s_base[SFX_MAX_COLORS];
    __shared__ uint32_t s_nleft;
    __shared__ uint32_t s_raw[kLeftCap];
    __shared__ uint32_t s_srt[kLeftCap];
    const uint32_t tid = threadIdx.x;
    uint32_t ne = *A.n_edges;
    if (ne > A.cap) {
        if (tid == 0) atomicOr(&A.ctr->overflow, kOvfFallback);
        ne = A.cap;
    }
    if (A.tick_totals && tid == 0) {
        uint32_t total = A.tile_start[A.ntiles];
        if (total > A.max_pairs) { atomicOr(&A.ctr->overflow, kOvfPairs); total = A.max_pairs; }
        A.ctr->n_tile_pairs = total;
        A.ctr->n_pairs = total + ne;
        A.ctr->n_fb = ne;
        A.ctr->n_active[0] = A.ctr->n_listed[0] + A.ctr->n_listed[1] + A.ctr->n_listed[2] + A.ctr->n_listed[3];
    }
    if (tid < SFX_MAX_COLORS) s_hist[tid] = 0u;
    if (tid == 0) s_nleft = 0u;
    __syncthreads();
    if (ne == 0) {
        if (tid <= SFX_MAX_COLORS && A.start) A.start[tid] = 0u;
        if (tid == 0 && A.n_colors_out) *A.n_colors_out = 0u;
        return;
    }
#define GDYN(x) ((x) != SFX_WORLD_BODY && (A.filt[(x)].w & kFiltDynamic))
    // the ends of an edge as bodies: what the colours must keep apart (the keys stay the list's own ids)
    auto ends = [&](uint32_t e) {
        uint2 ab = A.edges[e];
        if (A.ends_are_colliders) { ab.x = A.filt[ab.x].w >> 8; ab.y = A.filt[ab.y].w >> 8; }
        return ab;
    };
    for (uint32_t e = tid; e < ne; e += kGlobalColorBlock) {
        const uint2 ab = ends(e);
        A.color[e] = SFX_NO_COLOR;
        if (GDYN(ab.x)) { A.used[ab.x] = 0ull; for (int k = 0; k < kClaim; ++k) A.claim[(size_t)ab.x * kClaim + k] = ~0ull; }
        if (GDYN(ab.y)) { A.used[ab.y] = 0ull; for (int k = 0; k < kClaim; ++k) A.claim[(size_t)ab.y * kClaim + k] = ~0ull; }
    }
    __syncthreads();
    for (uint32_t r = 0; r < A.rounds; ++r) {
        for (uint32_t e = tid; e < ne; e += kGlobalColorBlock) {  // propose
            if (A.color[e] != SFX_NO_COLOR) continue;
            const uint2 ab = ends(e), kk = A.edges[e];
            const bool da = GDYN(ab.x), db = GDYN(ab.y);
            const unsigned long long freem = ~((da ? A.used[ab.x] : 0ull) | (db ? A.used[ab.y] : 0ull));
            if (freem == 0ull) { A.prop[e] = SFX_NO_COLOR; atomicOr(&A.ctr->overflow, kOvfColors); continue; }
            const uint32_t ku = A.key_by_index ? (e >> 24) : kk.x, kv = A.key_by_index ? (e & 0xFFFFFFu) : kk.y;
            const uint32_t c = (uint32_t)__builtin_ctzll(freem);  // first fit
            const uint32_t h = edge_hash(ku, kv, r);
            A.prop[e] = (uint8_t)c;
            const unsigned long long p = edge_prio(ku, kv, r, h);
            if (da) atomicMin(&A.claim[(size_t)ab.x * kClaim + (c & (kClaim - 1u))], p);
            if (db) atomicMin(&A.claim[(size_t)ab.y * kClaim + (c & (kClaim - 1u))], p);
        }
        __syncthreads();
        for (uint32_t e = tid; e < ne; e += kGlobalColorBlock) {  // resolve
            if (A.color[e] != SFX_NO_COLOR) continue;
            const uint32_t c = A.prop[e];
            if (c == SFX_NO_COLOR) continue;
            const uint2 ab = ends(e), kk = A.edges[e];
            const bool da = GDYN(ab.x), db = GDYN(ab.y);
            const uint32_t ku = A.key_by_index ? (e >> 24) : kk.x, kv = A.key_by_index ? (e & 0xFFFFFFu) : kk.y;
            const uint32_t h = edge_hash(ku, kv, r);
            const unsigned long long p = edge_prio(ku, kv, r, h);
            const bool win = (!da || A.claim[(size_t)ab.x * kClaim + (c & (kClaim - 1u))] == p) &&
                             (!db || A.claim[(size_t)ab.y * kClaim + (c & (kClaim - 1u))] == p);
            if (win) {
                A.color[e] = (uint8_t)c;
                if (da) atomicOr(&A.used[ab.x], 1ull << c);
                if (db) atomicOr(&A.used[ab.y], 1ull << c);
            }
        }
        __syncthreads();
    }
    // serial first-fit finish, ascending key
    for (uint32_t e = tid; e < ne; e += kGlobalColorBlock) {
        if (A.color[e] != SFX_NO_COLOR) continue;
        const uint32_t l = atomicAdd(&s_nleft, 1u);
        if (l < kLeftCap) s_raw[l] = e;
    }
    __syncthreads();
    uint32_t nleft = s_nleft;
    if (nleft) {
        if (nleft > kLeftCap) {
            if (tid == 0) atomicOr(&A.ctr->overflow, kOvfLeftover);
            nleft = kLeftCap;
        }
        for (uint32_t q = tid; q < nleft; q += kGlobalColorBlock) {
            const uint32_t e = s_raw[q];
            const uint2 ab = A.edges[e];
            const unsigned long long key = A.key_by_index ? (unsigned long long)e : (((unsigned long long)ab.x << 24) | ab.y);
            uint32_t rank = 0;
            for (uint32_t q2 = 0; q2 < nleft; ++q2) {
                const uint32_t e2 = s_raw[q2];
                const uint2 ab2 = A.edges[e2];
                const unsigned long long key2 = A.key_by_index ? (unsigned long long)e2 : (((unsigned long long)ab2.x << 24) | ab2.y);
                rank += (key2 < key) ? 1u : 0u;
            }
            s_srt[rank] = e;
        }
        __syncthreads();
        if (tid == 0) {
            for (uint32_t q = 0; q < nleft; ++q) {
                const uint32_t e = s_srt[q];
                const uint2 ab = ends(e);
                const bool da = GDYN(ab.x), db = GDYN(ab.y);
                const unsigned long long freem = ~((da ? A.used[ab.x] : 0ull) | (db ? A.used[ab.y] : 0ull));
                if (freem == 0ull) { atomicOr(&A.ctr->overflow, kOvfColors); continue; }
                const uint32_t c = (uint32_t)__builtin_ctzll(freem);
                A.color[e] = (uint8_t)c;
                if (da) A.used[ab.x] |= 1ull << c;
                if (db) A.used[ab.y] |= 1ull << c;
            }
            atomicAdd(&A.ctr->n_leftover, nleft);
        }
        __syncthreads();
    }
#undef GDYN
    for (uint32_t e = tid; e < ne; e += kGlobalColorBlock) {
        const uint32_t c = A.color[e];
        if (c < SFX_MAX_COLORS) atomicAdd(&s_hist[c], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0, nc = 0;
        for (uint32_t c = 0; c < SFX_MAX_COLORS; ++c) {
            s_base[c] = run;
            if (A.start) A.start[c] = run;
            if (s_hist[c]) nc = c + 1;
            run += s_hist[c];
        }
        if (A.start) A.start[SFX_MAX_COLORS] = run;
        if (A.n_colors_out) *A.n_colors_out = nc;
        atomicMax(&A.ctr->n_colors, nc);
    }
    if (tid < SFX_MAX_COLORS && s_hist[tid]) atomicAdd(&A.ctr->color_count[tid], s_hist[tid]);
    __syncthreads();
    if (A.sorted) {
        for (uint32_t e = tid; e < ne; e += kGlobalColorBlock) {
            const uint32_t c = A.color[e];
            if (c < SFX_MAX_COLORS) A.sorted[atomicAdd(&s_base[c], 1u)] = A.edges[e];
        }
    }
}

// ---------------------------------------------------------------- K6 / K8 tile solves
struct TileArrays {
    const uint32_t* cell_start;
    const uint32_t* sorted_id;
    const uint32_t* tile_start;   // scanned pair counts
    const uint32_t* active;       // per phase tile lists
    const BatchDesc* batches;     // per phase batch tables
    const uint4* tpair;           // colour-major per batch
    const uint4* torder;          // narrowphase order, per batch grouped by shape-pair class: (body a, body b, position in tpair
                                  // = record slot, -); slot kNoPair: skip.  The ids ride along so that k_narrow has one dependent load less
    const uint32_t* bbody;        // per batch (BatchDesc.boff): body ids in local-index order
    const float2* bmass;          // ... and their masses, in the same order (written once per frame by k_batch_color)
    const uint32_t* bruns;        // per batch (BatchDesc.roff): 64 colour runs, first | end << 16
    const uint32_t* tile_batch;   // [phase][tile]: 1 + index of the tile's batch on that phase's table (0: none this tick)
    uint32_t* done;               // per batch slot: epoch of the last fused launch that finished it
    const uint32_t* near_list;    // (phase << 24 | batch) of the NEAR batches in phase order (k_near_closure), ctr->n_near of them
    uint32_t near_cap;
    const uint2* fsorted;         // fallback list, colour-major
    uint32_t max_pairs;           // record slot of fallback pair k = max_pairs + k
    uint32_t fb_cap;
    unsigned long long* dbg;      // stage timing (builds with -DSFX_TILE_TIMING only), else unused
};


__device__ __forceinline__ uint32_t rec_count(const float4& a) { return __float_as_uint(a.z) & 0xFFu; }
__device__ __forceinline__ float4* rec_plane(const ContactArrays& C, int pl) { return C.rec + (size_t)pl * C.nslots; }

__device__ __forceinline__ void store_record(const ContactArrays& C, uint32_t slot, const PairRec& r) {
    rec_plane(C, kPlA)[slot] = make_float4(r.nx, r.ny, __uint_as_float((uint32_t)r.count | (r.block ? 256u : 0u)), r.iK[0]);
    if (r.count == 0) return;
    rec_plane(C, kPlB)[slot] = make_float4(r.ca[0], r.cb[0], r.d0[0], r.tgt[0]);
    rec_plane(C, kPlC)[slot] = make_float4(r.cta, r.ctb, r.mu_s, r.mu_d);
    rec_plane(C, kPlD)[slot] = make_float4(r.At[0], r.invS, r.Att, 0.0f);
    if (r.count == 2) {
        rec_plane(C, kPlE)[slot] = make_float4(r.ca[1], r.cb[1], r.d0[1], r.tgt[1]);
        rec_plane(C, kPlF)[slot] = make_float4(r.i00, r.i01, r.i11, r.K01);
        rec_plane(C, kPlG)[slot] = make_float4(r.iK[1], r.At[1], r.invS0, r.invS1);
    }
}

__device__ __forceinline__ void load_record(const ContactArrays& C, uint32_t slot, PairRec& r) {
    const float4 h0 = rec_plane(C, kPlA)[slot];
    const uint32_t bits = __float_as_uint(h0.z);
    r.count = (int)(bits & 0xFFu);
    r.block = (bits & 256u) != 0u;
    r.nx = h0.x; r.ny = h0.y; r.iK[0] = h0.w;
    r.mu_s = r.mu_d = 0.0f;
    r.ca[0] = r.ca[1] = r.cb[0] = r.cb[1] = r.cta = r.ctb = 0.0f;
    r.d0[0] = r.d0[1] = r.tgt[0] = r.tgt[1] = 0.0f;
    r.iK[1] = r.i00 = r.i01 = r.i11 = r.K01 = 0.0f;
    r.Att = r.At[0] = r.At[1] = r.invS = r.invS0 = r.invS1 = 0.0f;
    r.l0 = r.l1 = 0.0f;
    if (r.count == 0) return;
    const float2 lam = C.lam[slot];
    r.l0 = lam.x; r.l1 = lam.y;
    const float4 hb = rec_plane(C, kPlB)[slot], hc = rec_plane(C, kPlC)[slot], hd = rec_plane(C, kPlD)[slot];
    r.ca[0] = hb.x; r.cb[0] = hb.y; r.d0[0] = hb.z; r.tgt[0] = hb.w;
    r.cta = hc.x; r.ctb = hc.y; r.mu_s = hc.z; r.mu_d = hc.w;
    r.At[0] = hd.x; r.invS = hd.y; r.Att = hd.z;
    if (r.count == 2) {
        const float4 he = rec_plane(C, kPlE)[slot], hf = rec_plane(C, kPlF)[slot], hg = rec_plane(C, kPlG)[slot];
        r.ca[1] = he.x; r.cb[1] = he.y; r.d0[1] = he.z; r.tgt[1] = he.w;
        r.i00 = hf.x; r.i01 = hf.y; r.i11 = hf.z; r.K01 = hf.w;
        r.iK[1] = hg.x; r.At[1] = hg.y; r.invS0 = hg.z; r.invS1 = hg.w;
    }
}

// Narrowphase of one substep (DESIGN.md "Contract / sweep order"): the manifolds of every solver pair
// -- the tile pairs in the batches' class-grouped order, then the fallback list -- at the poses the
// integration (and the joints) left, one pair per lane, records to the contact arrays.  Every pair
// is independent, so this is the one stage of the contact sweep that fills the whole device.
#ifndef SFX_NARROW_WAVES
#define SFX_NARROW_WAVES 1   // waves per SIMD the register budget of k_narrow must allow (build knob)
#endif
__global__ __launch_bounds__(kBlock, SFX_NARROW_WAVES) void k_narrow(TileArrays T, const Counters* __restrict__ ctr, BodyArrays B,
                                                    ContactArrays C, float sm, float vthr, uint32_t first_phase_only) {
    const uint32_t ntp = min(ctr->tp_cursor, T.max_pairs);
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap) ? ctr->fb_start[SFX_MAX_COLORS] : 0u;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < ntp + nfb; j += gridDim.x * kBlock) {
        uint32_t a, b, slot;
        if (j < ntp) {
            const uint4 o = T.torder[j];
            slot = o.z;
            if (slot == kNoPair) continue;
            if (first_phase_only && o.w != 0u) continue;   // the batches of the later phases compute their own (kTileNarrow)
            a = o.x; b = o.y;
        } else {
            const uint2 ab = T.fsorted[j - ntp];
            a = ab.x; b = ab.y; slot = T.max_pairs + (j - ntp);
        }
        PairRec r;
        const float4 pa = B.pose[a], pb = B.pose[b], sa = B.shape[a], sb = B.shape[b];
        // the masses are the BODIES' (a further collider has none of its own)
        const float2 wa = B.mass[body_of(B, a)], wb = B.mass[body_of(B, b)];
        prepare_pair(pa, sa, B.vel[a], wa, pb, sb, B.vel[b], wb, shape_pose(pa, sa, B.lpose, a), shape_pose(pb, sb, B.lpose, b), B.poly, B.mats, sm, vthr, r);
        store_record(C, slot, r);
    }
}

// The narrowphase inside the position sweep (kTileNarrow): the workgroup that will solve a batch first computes the
// records of the batch's pairs, in the batch's class-grouped order, exactly as k_narrow's lanes would (same function, same
// inputs: the position sweep moves displacements, not poses or velocities, so a batch of a later phase sees what k_narrow
// would have seen -- and does this work while the batches it waits for are still solving).  The records go through the
// contact arrays: the velocity sweep reads them too.  Every storing wave drains its stores; the caller's next barrier
// makes them visible to the workgroup's other waves (same CU, write-through L1).
__device__ __forceinline__ void narrow_batch(const TileArrays& T, const BodyArrays& B, const ContactArrays& C, uint32_t start,
                                             uint32_t np, float sm, float vthr, uint32_t nt) {
    // (measured, profiles/r03/ab_fused_narrow.log: order entries fetched a round ahead: +-0; the bodies' records too: +6 us
    // per launch -- the registers they take cost the compute more than the round trip they hide)
    for (uint32_t k = threadIdx.x; k < np; k += nt) {
        const uint4 o = T.torder[start + k];
        const bool skip = o.z == kNoPair;
        const uint32_t a = skip ? 0u : o.x, b = skip ? 0u : o.y;   // (no branch between the order entry and the gathers)
        const float4 pa = B.pose[a], pb = B.pose[b], sa = B.shape[a], sb = B.shape[b], va = B.vel[a], vb = B.vel[b];
        const float2 wa = B.mass[body_of(B, a)], wb = B.mass[body_of(B, b)];   // the masses are the BODIES' (a further collider has none of its own)
        if (skip) continue;
        PairRec r;
        prepare_pair(pa, sa, va, wa, pb, sb, vb, wb, shape_pose(pa, sa, B.lpose, a), shape_pose(pb, sb, B.lpose, b), B.poly, B.mats, sm, vthr, r);
        store_record(C, o.z, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void narrow_fallback(const TileArrays& T, const Counters* __restrict__ ctr, const BodyArrays& B,
                                                const ContactArrays& C, float sm, float vthr) {
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap) ? ctr->fb_start[SFX_MAX_COLORS] : 0u;
    for (uint32_t k = threadIdx.x; k < nfb; k += blockDim.x) {
        const uint2 ab = T.fsorted[k];
        const uint32_t a = ab.x, b = ab.y;
        PairRec r;
        const float4 pa = B.pose[a], pb = B.pose[b], sa = B.shape[a], sb = B.shape[b];
        const float2 wa = B.mass[body_of(B, a)], wb = B.mass[body_of(B, b)];
        prepare_pair(pa, sa, B.vel[a], wa, pb, sb, B.vel[b], wb, shape_pose(pa, sa, B.lpose, a), shape_pose(pb, sb, B.lpose, b), B.poly, B.mats, sm, vthr, r);
        store_record(C, T.max_pairs + k, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// The fallback list, solved by one whole workgroup directly on global memory after the last
// phase, colour by colour (its manifolds come from k_narrow like everybody else's, or from narrow_fallback).
__device__ __forceinline__ void fallback_pos(const TileArrays& T, const Counters* __restrict__ ctr, const BodyArrays& B,
                                             const ContactArrays& C, float max_corr) {
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap);
    if (nfb == 0) return;
    const uint32_t nc = ctr->n_fb_colors;
    for (uint32_t c = 0; c < nc; ++c) {
        const uint32_t s = ctr->fb_start[c], e = ctr->fb_start[c + 1];
        for (uint32_t k = s + threadIdx.x; k < e; k += blockDim.x) {
            uint2 ab = T.fsorted[k];
            PairRec r;
            load_record(C, T.max_pairs + k, r);
            if (r.count == 0) continue;
            ab.x = body_of(B, ab.x); ab.y = body_of(B, ab.y);   // the list names colliders, the solve moves their bodies
            float4 Da = B.dpos[ab.x], Db = B.dpos[ab.y];
            const float2 wa = B.mass[ab.x], wb = B.mass[ab.y];
            const bool moved = solve_pos_rec(r, Da, wa, Db, wb, max_corr);
            C.lam[T.max_pairs + k] = make_float2(r.l0, r.l1);
            if (moved) {
                if (wa.x > 0.0f || wa.y > 0.0f) B.dpos[ab.x] = Da;
                if (wb.x > 0.0f || wb.y > 0.0f) B.dpos[ab.y] = Db;
            }
        }
        __syncthreads();
    }
}

// (the static-friction shifts of a fallback pair are applied to the two poses at once, pair by pair: one workgroup on
// global memory has nowhere to keep them until the end of the phase)
__device__ __forceinline__ void fallback_vel(const TileArrays& T, const Counters* __restrict__ ctr, const BodyArrays& B,
                                             const ContactArrays& C, const StepParams& sp, uint32_t vstamp = 0u) {
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap);
    if (nfb == 0) return;
    const uint32_t nc = ctr->n_fb_colors;
    for (uint32_t c = 0; c < nc; ++c) {
        const uint32_t s = ctr->fb_start[c], e = ctr->fb_start[c + 1];
        for (uint32_t k = s + threadIdx.x; k < e; k += blockDim.x) {
            uint2 ab = T.fsorted[k];
            PairRec r;
            load_record(C, T.max_pairs + k, r);
            if (r.count == 0) continue;
            ab.x = body_of(B, ab.x); ab.y = body_of(B, ab.y);
            float4 va = B.vel[ab.x], vb = B.vel[ab.y];
            const float2 wa = B.mass[ab.x], wb = B.mass[ab.y];
            const bool dyna = wa.x > 0.0f || wa.y > 0.0f, dynb = wb.x > 0.0f || wb.y > 0.0f;
            bool da = false, db = false;   // derive-on-load: this pair is the first to read the body's velocity this substep
            if (vstamp) {
                if (dyna && __float_as_uint(va.w) != vstamp) {
                    float4 pa = B.pose[ab.x];
                    apply_delta(pa, B.dpos[ab.x]);
                    derive_body(pa, B.prev[ab.x], va, sp); va.w = __uint_as_float(vstamp); da = true;
                    B.pose[ab.x] = pa;
                }
                if (dynb && __float_as_uint(vb.w) != vstamp) {
                    float4 pb = B.pose[ab.y];
                    apply_delta(pb, B.dpos[ab.y]);
                    derive_body(pb, B.prev[ab.y], vb, sp); vb.w = __uint_as_float(vstamp); db = true;
                    B.pose[ab.y] = pb;
                }
            }
            float4 Sa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), Sb = Sa;
            const int ch = solve_vel_rec(r, va, wa, vb, wb, Sa, Sb, sp);
            if (ch || da || db) {
                if (dyna) B.vel[ab.x] = va;
                if (dynb) B.vel[ab.y] = vb;
            }
            if (ch & 2) {
                if (dyna) { float4 pa = B.pose[ab.x]; apply_delta(pa, Sa); B.pose[ab.x] = pa; }
                if (dynb) { float4 pb = B.pose[ab.y]; apply_delta(pb, Sb); B.pose[ab.y] = pb; }
            }
        }
        __syncthreads();
    }
}

// true for exactly one workgroup of the grid: the one that finishes last.  Everything the
// others wrote before is visible to it afterwards.
__device__ __forceinline__ bool last_workgroup(Counters* __restrict__ ctr) {
    __shared__ uint32_t s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have left before the ticket is taken
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t tk = atomicAdd(&ctr->ticket, 1u);
        s_last = (tk + 1u == gridDim.x) ? 1u : 0u;
        if (s_last) { ctr->ticket = 0u; __threadfence(); }
    }
    __syncthreads();
    const bool last = s_last != 0u;
    if (last) __threadfence();
    return last;
}

// One end of a tile pair: inside the block (LDS index) or a large non-dynamic collider outside
// it (kExt), which is read from global memory and never written.
struct PairEnds {
    uint32_t a, b;     // body ids
    uint32_t la, lb;   // local indices or kExt
    uint32_t col;
};

__device__ __forceinline__ PairEnds pair_ends(const uint4& pr) {
    PairEnds e;
    e.a = pr.x; e.b = pr.y; e.la = pr.z & 0xFFFFu; e.lb = pr.z >> 16; e.col = pr.w & 0xFFu;
    return e;
}

// Two lanes per pair (sfx_device.h solve_pos_side / solve_vel_side): lane 2k takes body a of the run's k-th pair, lane
// 2k + 1 body b.  One side of a pair as the colour loops see it: this body's LDS index + its side of the record, fetched
// from the pair list and the contact arrays one colour ahead of its use.
struct SideSlot {
    uint32_t li;     // this body's local index
    SideRec r;
    bool valid;
};

// The raw fields of one side as loaded: every field unconditionally and at once (one round trip, no load waits for
// another, nothing branches on loaded data), decoded only when the pair's colour comes up -- so the loads of colour
// c + 1 are in flight while colour c is solved.  The two lanes of a pair load the same lines (one request).
struct RawSide {
    uint4 pr;
    float4 ha, hb, he, hf, hg;   // planes A, B and (two points) E, F, G
    float4 hc;                   // velocity pass: plane C
    float2 hd;                   // velocity pass: (At0, invS) of plane D
    float2 lam;
    bool in_range;
};

template <typename V>
__device__ __forceinline__ V load_at(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off);
}

template <bool VEL>
__device__ __forceinline__ void fetch_side(const TileArrays& T, const ContactArrays& C, uint32_t slot, uint32_t side, bool in_range,
                                           RawSide& q) {
    // lanes past the end of the run load slot 0 (always there) and ignore it: no branch around the loads, so
    // nothing has to wait for them before the pair's colour comes up
    q.in_range = in_range;
    slot = in_range ? slot : 0u;
    q.lam = make_float2(0.0f, 0.0f);
    q.hc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    q.hd = make_float2(0.0f, 0.0f);
    // every array at (uniform base) + (one 32-bit byte offset per lane): global loads with a scalar base, no 64-bit address
    // arithmetic per plane in the colour step's instruction chain (slots < 2^28: sfx_upload_bodies caps a world at 16 M entries,
    // twelve pairs each).  Measured: two VGPRs fewer, the tick within +-0.1 % (profiles/r03/ab_scalar_base_fetch.log) -- the
    // address arithmetic was not what a step waits for.
    const uint32_t o16 = slot * 16u;
    q.pr = load_at<uint4>(T.tpair, o16);
    q.ha = load_at<float4>(rec_plane(C, kPlA), o16);
    q.hb = load_at<float4>(rec_plane(C, kPlB), o16);
    q.he = load_at<float4>(rec_plane(C, kPlE), o16);
    q.hf = load_at<float4>(rec_plane(C, kPlF), o16);
    q.hg = load_at<float4>(rec_plane(C, kPlG), o16);
    if (VEL) {
        q.hc = load_at<float4>(rec_plane(C, kPlC), o16);
        q.hd = load_at<float2>(rec_plane(C, kPlD), o16);
        q.lam = load_at<float2>(C.lam, slot * 8u);
    }
}

template <bool VEL>
__device__ __forceinline__ void decode_side(const RawSide& q, uint32_t side, SideSlot& p, uint32_t nb) {
    p.valid = false;
    p.r.count = 0;
    const PairEnds e = pair_ends(q.pr);
    p.li = side ? +e.lb : +e.la;
    const uint32_t bits = __float_as_uint(q.ha.z);
    const int count = (int)(bits & 0xFFu);
    if (!q.in_range || count == 0) return;
    if (e.la >= nb || e.lb >= nb) return;  // an outside body that found no LDS slot (kOvfExt was reported)
    if (VEL && !(q.lam.x + q.lam.y > 0.0f)) return;  // nothing was pushed: nothing to do in the velocity pass
    SideRec& r = p.r;
    const bool two = count == 2;   // (planes E, F, G are only written -- and meaningful -- for two points)
    r.count = count; r.block = (bits & 256u) != 0u;
    r.nx = side ? -q.ha.x : q.ha.x; r.ny = side ? -q.ha.y : q.ha.y;   // from this body to the other
    r.iK[0] = q.ha.w;
    r.c[0] = side ? +q.hb.y : +q.hb.x;   // (unary plus: a conditional of two lvalues is an lvalue -- an address select that keeps the struct in scratch memory)
    r.d0[0] = q.hb.z; r.tgt[0] = q.hb.w;
    r.ct = side ? +q.hc.y : +q.hc.x; r.mu_s = q.hc.z; r.mu_d = q.hc.w;
    // the second point's fields are only written (and meaningful) for two-point manifolds; the solves never select
    // anything computed from them otherwise.  Only its lever arm enters sums unconditionally (times a zero multiplier)
    r.c[1] = two ? (side ? +q.he.y : +q.he.x) : 0.0f;
    r.d0[1] = q.he.z; r.tgt[1] = q.he.w;
    r.i00 = q.hf.x; r.i01 = q.hf.y; r.i11 = q.hf.z; r.K01 = q.hf.w;
    r.iK[1] = q.hg.x;
    r.At[0] = q.hd.x; r.invS = q.hd.y;
    r.At[1] = q.hg.y; r.invS0 = q.hg.z; r.invS1 = q.hg.w;
    r.l0 = q.lam.x; r.l1 = q.lam.y;
    p.valid = true;
}

template <bool VEL>
__device__ __forceinline__ void fetch_side_now(const TileArrays& T, const ContactArrays& C, uint32_t slot, uint32_t side,
                                               SideSlot& p, uint32_t nb) {
    RawSide q;
    fetch_side<VEL>(T, C, slot, side, true, q);
    decode_side<VEL>(q, side, p, nb);
}

// Workgroup barrier for steps that hand data over through LDS only: waits for the LDS traffic,
// not for the global loads and stores in flight (the record prefetch of the next colour, the
// lambda stores), which only their own lane depends on.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Several phases in ONE launch (nph > 1): the workgroups take the batches of the phases
// ph0 .. ph0 + nph - 1 in that order (item I = batches of the earlier phases + index); a batch
// of a later phase waits until the batches of earlier phases whose blocks overlap its own have
// published `epoch` in T.done (agent-scope release / acquire), instead of the whole grid waiting
// for the end of a launch.  Items only ever wait for lower-numbered items and the grid is sized
// (host side) so that every workgroup is resident: the lowest unfinished item can always run.
// The wait is bounded all the same: when it gives up, kOvfStall is flagged and the batch runs
// on what it sees (the tick is then wrong, but nothing hangs).
struct PhaseItem { uint32_t ph, it; bool ok; };

__device__ __forceinline__ PhaseItem phase_item(const Counters* __restrict__ ctr, uint32_t ph0, uint32_t nph, uint32_t I) {
    PhaseItem r;
    r.ph = ph0; r.it = I; r.ok = false;
    for (uint32_t q = 0; q < nph; ++q) {
        const uint32_t nb = ctr->n_batches[ph0 + q];
        if (r.it < nb) { r.ph = ph0 + q; r.ok = true; return r; }
        r.it -= nb;
    }
    return r;
}

__device__ __forceinline__ uint32_t phase_total(const Counters* __restrict__ ctr, uint32_t ph0, uint32_t nph) {
    uint32_t t = 0;
    for (uint32_t q = 0; q < nph; ++q) t += ctr->n_batches[ph0 + q];
    return t;
}

// The batches of a launch are handed out in item order through a ticket, to workgroups that are running: the
// lowest unfinished item is therefore always held by a workgroup on the device, and since items only wait for
// lower-numbered ones the launch cannot stall on a workgroup that has not been scheduled yet -- whatever the grid
// size and however many workgroups the device holds at once.  (A static blockIdx -> item map needs every
// workgroup resident and a safety margin that, at one workgroup per CU, cost the 100 k pile a third round of
// batches.)  All threads call this; the body of the loop has a barrier between two calls.
// Flags of a tile-solve launch.  kTileStaticFirst: workgroup b starts with item b and tickets count from gridDim.x
// -- saves the first ticket's round trip (3.7 us per launch on the 100 k pile: every workgroup asks at once), but
// then the argument above needs every workgroup of the grid on the device, which the host can promise only while
// no other world runs fused launches on the same device (sfx_host.hip: live worlds per device, SFX_SHARED_DEVICE).
// kTileSlot: which of the two ticket counters this launch uses.
// kTileOnlyNear / kTileOnlyFar: the velocity sweep of a slab world in two launches (k_near_closure): the near batches from
// their list, then every other batch.
// kTileNarrow (position sweep): every batch computes the manifolds and frozen rows of its own pairs before it solves
// them -- no k_narrow launch in that substep (narrow_batch below).
// kTileNarrowAll: the batches of the first phase too (and the fallback list): no k_narrow launch at all.
constexpr int kTileFallback = 1, kTileStaticFirst = 2, kTileSlot = 4, kTileRaceSelfTest = 8, kTileOnlyNear = 16, kTileOnlyFar = 32,
              kTileNarrow = 64, kTileNarrowAll = 128;

__device__ __forceinline__ uint32_t* ticket_of(Counters* __restrict__ ctr, int flags) {
    return &ctr->item_ticket[(flags & kTileSlot) ? 1 : 0];
}

__device__ __forceinline__ uint32_t next_item(Counters* __restrict__ ctr, int flags) {
    // the next launch's counter starts at zero (that launch begins after this one has ended; the previous one, which
    // used it, has ended before this one began)
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->item_ticket[(flags & kTileSlot) ? 0 : 1] = 0u;
    if (flags & kTileStaticFirst) return blockIdx.x;
    __shared__ uint32_t s_item;
    if (threadIdx.x == 0) s_item = __hip_atomic_fetch_add(ticket_of(ctr, flags), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(s_item);   // (uniform, and the compiler is told so: the batch's descriptor and what follows from it stay in scalar registers)
}

// The ticket of the workgroup's NEXT batch is taken while the current one is solved, by the first lane of the last
// wave (the wave with the least to do in a colour step), so that its latency is off the batch's chain.  A ticket
// held ahead belongs to a running workgroup that works on a lower item: the argument above still holds.
template <int NT>
__device__ __forceinline__ uint32_t take_ahead(Counters* __restrict__ ctr, int flags) {
    uint32_t t = 0u;
    if (threadIdx.x == NT - 64)
        t = __hip_atomic_fetch_add(ticket_of(ctr, flags), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ((flags & kTileStaticFirst) ? gridDim.x : 0u);
    return t;
}

// all threads of the workgroup; tiles = the batch's tile ids
// Hand-off between the batches of one fused launch (MI355X: eight XCDs with private L2s, one L1
// per CU that other CUs' stores never refresh).  Producer: the poses / velocities a batch writes
// back are stored write-through (sc1 buffer stores), every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and one lane stores the batch's flag with
// an agent-scope atomic store.  Consumer: the first wave polls the flags of the neighbouring
// batches of earlier phases with relaxed agent-scope loads, runs ONE agent-scope acquire (drops
// this CU's L1 lines) and waits for it, then the workgroup passes a barrier and loads plainly.
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t record_rsrc(float4* base) {
    // raw buffer over 16-byte records; the range check is not used (offsets come from body ids)
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)0xFFFFFFF0u, 0x00020000);
}

template <bool WT>
__device__ __forceinline__ void store_record16(float4* base, __amdgpu_buffer_rsrc_t rs, uint32_t idx, const float4& v) {
    if constexpr (WT) {
        const v4u32 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)(idx * 16u), 0, 16 /* sc1: write-through */);
    } else {
        base[idx] = v;
    }
}

template <int NT>
__device__ __forceinline__ void wait_for_earlier_phases(const TileArrays& T, const GridParams& g, Counters* __restrict__ ctr,
                                                        const uint32_t* __restrict__ tiles, uint32_t count, uint32_t ph,
                                                        uint32_t ph0, uint32_t epoch) {
    if (threadIdx.x < 64u) {
        // a batch's bodies lie in the 2x2 tile blocks of its tiles; the batches of EARLIER phases (classes) that can have
        // written them are those of the 3x3 tiles around each tile -- the tile itself included: it has a batch per class
        const uint32_t lx = g.gxl - g.tlx;
        const bool one_row = g.tly == g.gyl;
        for (uint32_t j = threadIdx.x; j < count * 9u; j += 64u) {
            const uint32_t t = tiles[j / 9u], dir = j % 9u;
            const int dx = (int)(dir % 3u) - 1, dy = (int)(dir / 3u) - 1;
            if (one_row && dy != 0) continue;
            const int32_t tx = (int32_t)(t & ((1u << lx) - 1u)), ty = (int32_t)(t >> lx);
            const uint32_t t2 = tile_of(tx + dx, ty + dy, g);
            for (uint32_t ph2 = ph0; ph2 < ph; ++ph2) {
                const uint32_t b2 = T.tile_batch[ph2 * g.ntiles + t2];
                if (!b2) continue;  // that tile has no pairs of that class this tick
                const uint32_t* flag = &T.done[ph2 * g.pstride + (b2 - 1u)];
                uint32_t spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 400000u) { atomicOr(&ctr->overflow, kOvfStall); break; }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before the barrier opens
    }
    __syncthreads();
}

__device__ __forceinline__ void publish_batch(const TileArrays& T, const GridParams& g, uint32_t ph, uint32_t it,
                                              uint32_t epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(&T.done[ph * g.pstride + it], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exchange / compute overlap of slab worlds (SURVEY.md section 5 "overlap it with interior-constraint solving"; DESIGN.md
// "Multi-GPU / overlap").  The last kernel of a substep is the velocity sweep; what the neighbours need of it are the
// bodies on the halo send list.  k_batch_color marks every batch whose pairs touch a halo body (kBatchNear); this
// kernel -- one workgroup, once per frame -- closes the set under "waits for": a batch of phase p waits for the batches
// that hold, in an earlier phase, a unit of one of the 3 x 3 tiles around each of its own (wait_for_earlier_phases), so
// going through the phases from the last to the second, every near batch marks those.  Then it lists the near batches
// in phase order.  The velocity sweep of such a world runs as two launches: the near list (its items only ever wait for
// earlier items of the list: it is closed), after which the halo is packed and travels on a second stream, and all the
// other batches meanwhile (they find the near batches' flags set, same epoch).  Which launch runs a batch does not
// change what it computes: the states are bit-identical to the single launch (tests/test_gpu_host_api.py).
__global__ __launch_bounds__(1024) void k_near_closure(GridParams g, uint32_t nph, BatchDesc* __restrict__ batches,
                                                       const uint32_t* __restrict__ active, const uint32_t* __restrict__ tile_batch,
                                                       uint32_t* __restrict__ near_list, uint32_t near_cap, Counters* __restrict__ ctr) {
    __shared__ uint32_t s_n;
    const uint32_t lx = g.gxl - g.tlx;
    const bool one_row = g.tly == g.gyl;
    // the fallback list is solved after every batch, by the last workgroup of the sweep, and may touch halo bodies: a tick
    // that has one runs its whole sweep in the first launch (nothing is overlapped, everything is right)
    if (ctr->n_fb != 0u) { if (threadIdx.x == 0) ctr->n_near = kNoSplit; return; }
    if (threadIdx.x == 0) s_n = 0u;
    for (uint32_t ph = nph; ph-- > 1u;) {
        const uint32_t nb = min(ctr->n_batches[ph], g.pstride);
        __syncthreads();   // the marks of the later phases are in place
        for (uint32_t it = threadIdx.x; it < nb; it += blockDim.x) {
            const BatchDesc d = batches[ph * g.pstride + it];
            if (!(d.count & kBatchNear)) continue;
            const uint32_t cnt = d.count & ~kBatchNear;
            for (uint32_t j = 0; j < cnt * 9u; ++j) {
                const uint32_t t = active[d.first + j / 9u], dir = j % 9u;
                const int dx = (int)(dir % 3u) - 1, dy = (int)(dir / 3u) - 1;
                if (one_row && dy != 0) continue;
                const int32_t tx = (int32_t)(t & ((1u << lx) - 1u)), ty = (int32_t)(t >> lx);
                const uint32_t t2 = tile_of(tx + dx, ty + dy, g);
                for (uint32_t ph2 = 0; ph2 < ph; ++ph2) {
                    const uint32_t b2 = tile_batch[ph2 * g.ntiles + t2];
                    if (b2) atomicOr(&batches[ph2 * g.pstride + (b2 - 1u)].count, kBatchNear);
                }
            }
        }
    }
    for (uint32_t ph = 0; ph < nph; ++ph) {   // the list, phase by phase (inside a phase the order is free)
        const uint32_t nb = min(ctr->n_batches[ph], g.pstride);
        __syncthreads();
        for (uint32_t it = threadIdx.x; it < nb; it += blockDim.x) {
            if (!(batches[ph * g.pstride + it].count & kBatchNear)) continue;
            const uint32_t k = atomicAdd(&s_n, 1u);
            if (k < near_cap) near_list[k] = (ph << 24) | it;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) ctr->n_near = min(s_n, near_cap);
}

// In-kernel stamps of the colour loops (diagnostic builds with -DSFX_STEP_PROBE only: scripts/dev/step_probe.py reads the
// SHARES of a step's segments; the build's run time means nothing).  Scalar clock reads; workgroup 3 adds its sums to T.dbg.
#ifdef SFX_STEP_PROBE
#define SFX_STAMP(t)                                                                          \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#define SFX_STAMP_VM(t)                                                                       \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#define SFX_PROBE_DECL unsigned long long pq0 = 0, pq1 = 0, pq2 = 0, pq3 = 0, pq4 = 0, pa0 = 0, pa1 = 0, pa2 = 0, pa3 = 0, pa4 = 0, pns = 0; (void)pq0; (void)pq1; (void)pq2; (void)pq3; (void)pq4;
#define SFX_PROBE_OUT(base)                                                                   \
    if (blockIdx.x == 3u && threadIdx.x == 0u && T.dbg) {                                     \
        atomicAdd(&T.dbg[(base) + 0], pa0); atomicAdd(&T.dbg[(base) + 1], pa1); atomicAdd(&T.dbg[(base) + 2], pa2); \
        atomicAdd(&T.dbg[(base) + 3], pa3); atomicAdd(&T.dbg[(base) + 4], pa4); atomicAdd(&T.dbg[(base) + 5], pns);  \
    }
#else
#define SFX_STAMP(t)
#define SFX_STAMP_VM(t)
#define SFX_PROBE_DECL
#define SFX_PROBE_OUT(base)
#endif

// Device-side race detector ("a debug kernel that asserts no body index is touched twice per colour";
// builds with -DSFX_DEBUG_RACES: libsfx_races.so).  The colouring is what makes the solves race-free: within one colour no
// two pairs write the same body.  Here every write of a body's LDS record inside a colour step first stamps the body's
// slot with (colour + 1) by an atomic exchange; finding the same stamp already there means a second writer in that colour
// and sets overflow bit 256, which the next sfx_sync reports.  Statics are read, never written: not stamped.  The stamps
// of a batch start at zero.  kTileRaceSelfTest (host: SFX_RACE_SELFTEST=1) stamps every write twice: the report path itself.
#ifdef SFX_DEBUG_RACES
#define SFX_RACE_DECL __shared__ uint32_t s_stamp[kMaxBX];
#define SFX_RACE_CLEAR(nb) for (uint32_t l_ = threadIdx.x; l_ < (nb); l_ += blockDim.x) s_stamp[l_] = 0u;
#define SFX_RACE_WRITE(li, colour)                                                                     \
    do {                                                                                               \
        const uint32_t tag_ = (colour) + 1u;                                                           \
        if (atomicExch(&s_stamp[(li)], tag_) == tag_) atomicOr(&ctr->overflow, kOvfRace);              \
        if ((flags & kTileRaceSelfTest) && atomicExch(&s_stamp[(li)], tag_) == tag_) atomicOr(&ctr->overflow, kOvfRace); \
    } while (0)
#else
#define SFX_RACE_DECL
#define SFX_RACE_CLEAR(nb)
#define SFX_RACE_WRITE(li, colour)
#endif

// The colour loop of a batch as a flat list of STEPS: a step is one colour's run, or -- when a colour has more pairs than
// the workgroup has lane pairs -- one chunk of NT / 2 pairs of it; steps of one colour need no barrier between them.  With
// the list in LDS a lane knows the record slot of any later step, so records are fetched kDepth steps ahead: a record
// comes from HBM / Infinity Cache (written by k_narrow on other XCDs) and takes about a microsecond, a step a few hundred
// cycles.  entry = first pair of the chunk | end of the colour's run << 16 (both < 65536, as in the run table).
#ifndef SFX_DEPTH_POS
#define SFX_DEPTH_POS 1
#endif
#ifndef SFX_DEPTH_VEL
#define SFX_DEPTH_VEL 1
#endif
constexpr uint32_t kMaxSteps = SFX_MAX_COLORS + 2u * kBatchP / 128u;   // every colour once + the extra chunks of the smallest workgroup

// wave 0 of the workgroup; `runs` = the batch's 64 colour runs (first | end << 16); returns nothing: s_nsteps is set
template <int NT>
__device__ __forceinline__ void build_steps(const uint32_t* __restrict__ runs, uint32_t ncol, uint32_t* s_step, uint32_t* s_nsteps) {
    if (threadIdx.x >= 64u) return;
    const uint32_t c = threadIdx.x;
    const uint32_t v = runs[c];
    const uint32_t cs = v & 0xFFFFu, ce = v >> 16;
    const uint32_t len = (c < ncol && ce > cs) ? ce - cs : 0u;
    const uint32_t nch = (len + (uint32_t)(NT / 2) - 1u) / (uint32_t)(NT / 2);
    const uint32_t incl = wave_inclusive_scan(nch);
    const uint32_t off = incl - nch;
    for (uint32_t i = 0; i < nch; ++i)
        if (off + i < kMaxSteps) s_step[off + i] = (cs + i * (uint32_t)(NT / 2)) | (ce << 16);
    if (c == 63u) *s_nsteps = min(incl, kMaxSteps);
}

// (the integration-on-load forms of the narrowphase and of the fallback list: defined with the one-launch tick below)
template <bool WT>
__device__ __forceinline__ void narrow_batch_tick(const TileArrays& T, const BodyArrays& B, const ContactArrays& C,
                                                  uint32_t start, uint32_t np, float sm,
                                                  const StepParams& sp, uint32_t s, uint32_t nt, uint32_t tid);
__device__ __forceinline__ void narrow_fallback_tick(const TileArrays& T, const Counters* __restrict__ ctr, const BodyArrays& B,
                                                     const ContactArrays& C, float sm, const StepParams& sp, uint32_t s);
__device__ __forceinline__ void fallback_vel_tick(const TileArrays& T, const Counters* __restrict__ ctr, const BodyArrays& B,
                                                  const ContactArrays& C, const StepParams& sp, uint32_t vstamp);

// Position solve (DESIGN.md "Contract / frozen rows").  A workgroup takes one batch of tiles at a time: the displacements
// (dx, dy, dtheta) of the batch's bodies -- zero for the first phase of a substep, what earlier phases accumulated
// otherwise -- and their masses go to LDS; the colours are solved one after the other with an LDS-only barrier in
// between.  The pairs of a colour are a contiguous run of the colour-sorted list and are taken by consecutive lanes, two
// lanes per pair; every lane fetches the record of its next colour's pair while it solves the current one.  Finally the
// dynamic bodies' displacements are written back.  Poses are not touched: the displacement is applied once per body
// where its velocity is derived.  Batches of one phase have disjoint bodies.
// INTEG (DESIGN.md "Integration on load"): no k_integrate ran -- the narrowphase integrates the bodies it loads (substep `sub`
// of the tick, parameters `spi`)
template <int NT, bool INTEG = false>
__global__ __launch_bounds__(NT, (NT == 256 && SFX_TILE_MIN_WAVES < 3) ? 3 : SFX_TILE_MIN_WAVES) void k_tile_pos(uint32_t ph0, uint32_t nph, uint32_t epoch, int flags, GridParams g,
                                                  TileArrays T, Counters* __restrict__ ctr, BodyArrays B, ContactArrays C,
                                                  float max_corr, float sm, float vthr, StepParams spi, uint32_t sub) {
    __shared__ float4 s_d[kMaxBX];
    __shared__ float2 s_mass[kMaxBX];
    __shared__ uint32_t s_gid[kMaxBX];
    __shared__ float2 s_lam[kBatchP];  // the batch's multipliers, written out in one coalesced pass after the colours
    __shared__ uint32_t s_step[kMaxSteps], s_nsteps;
    SFX_RACE_DECL
    constexpr int kDepth = SFX_DEPTH_POS;
    const uint32_t tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t dpos_rs = record_rsrc(B.dpos);
    const uint32_t nitems = phase_total(ctr, ph0, nph);
    __shared__ uint32_t s_next;
    for (uint32_t item = next_item(ctr, flags); item < nitems; item = __builtin_amdgcn_readfirstlane(s_next)) {
        SFX_TT(tt0);
        const PhaseItem pi = phase_item(ctr, ph0, nph, item);
        const uint32_t ph = pi.ph, it = pi.it;
        const BatchDesc d = T.batches[ph * g.pstride + it];
        const uint32_t* tiles = T.active + d.first;   // one tile list for all phases
        const uint32_t nb = min(d.nb, kMaxBX), start = d.bstart, ncol = d.ncol, npb = min(d.np, kBatchP);
        // everything that no earlier phase of this launch writes is fetched before waiting for them: the step list, the
        // records of the first kDepth steps, the body ids, the masses
        if ((flags & kTileNarrow) && (ph != 0u || (flags & kTileNarrowAll))) {
            if constexpr (INTEG) narrow_batch_tick<false>(T, B, C, start, npb, sm, spi, sub, NT, tid);
            else narrow_batch(T, B, C, start, npb, sm, vthr, NT);
        }
        // (INTEG: the lane's index through a copy the compiler cannot see through -- what the colour loop derives from it is
        // then derived here, behind the narrowphase, which wants every register, instead of being spilled across it)
        uint32_t tl = tid;
        if constexpr (INTEG) asm volatile("" : "+v"(tl));
        const uint32_t side = tl & 1u, half = tl >> 1;   // two lanes per pair: body a, body b
        __syncthreads();  // the previous batch's write-back has read the LDS arrays
        build_steps<NT>(T.bruns + 64u * d.roff, ncol, s_step, &s_nsteps);
        for (uint32_t l = tid; l < nb; l += NT) {  // block bodies, then the large colliders outside the block its pairs touch
            const uint32_t gid = T.bbody[d.boff + l];
            s_gid[l] = gid;
            s_mass[l] = T.bmass[d.boff + l];
            if (ph == 0u) s_d[l] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // nothing has moved yet in this substep
        }
        for (uint32_t k = tid; k < npb; k += NT) s_lam[k] = make_float2(0.0f, 0.0f);
        SFX_RACE_CLEAR(nb)
        __syncthreads();
#ifdef SFX_PROBE_NOLOOP   // timing probe: a batch without its colour loop (load, hand-off and write-back only; results are wrong)
        const uint32_t ns = 0u;
#else
        const uint32_t ns = s_nsteps;
#endif
        static_assert(kDepth == 1, "the colour loop keeps the step entries of this step and the next in registers");
        RawSide q;
        SFX_PROBE_DECL
        // the step table entries of the current and the next step ride in (scalar) registers: no LDS round trip for them in a step
        uint32_t st_cur = __builtin_amdgcn_readfirstlane(s_step[0]), st_nxt = __builtin_amdgcn_readfirstlane(s_step[1]);
        fetch_side<false>(T, C, start + (st_cur & 0xFFFFu) + half, side, 0u < ns && (st_cur & 0xFFFFu) + half < (st_cur >> 16), q);
        if (nph > 1u && ph > ph0) wait_for_earlier_phases<NT>(T, g, ctr, tiles, d.count & ~kBatchNear, ph, ph0, epoch);
        SFX_TT(ttw);
        SFX_TT_ADD(6, tt0, ttw);
        if (ph != 0u)
            for (uint32_t l = tid; l < nb; l += NT) s_d[l] = B.dpos[s_gid[l]];  // (the lane that stored s_gid[l])
        __syncthreads();
        const uint32_t ahead = take_ahead<NT>(ctr, flags);  // the next batch's ticket, in flight during the colour loop
        SFX_TT(tt1);
        SFX_TT(tt2);
        uint32_t colour = 0u;   // (counts the barriers of the loop: the race detector's stamp)
        (void)colour;
        auto solve = [&](uint32_t k, SideSlot& p) __attribute__((always_inline)) {  // everything from LDS: no memory wait inside a step
            float4 D = s_d[p.li];
            const float2 w = s_mass[p.li];
#ifdef SFX_PROBE_NOSOLVE   // timing probe: the colour loop without its arithmetic (results are wrong)
            D.x += p.r.c[0] * 1e-30f;
            if (w.x > 0.0f) s_d[p.li] = D;
            return;
#endif
            if (solve_pos_side(p.r, D, w, max_corr)) {
                if (w.x > 0.0f || w.y > 0.0f) { SFX_RACE_WRITE(p.li, colour); s_d[p.li] = D; }
            }
            if (side == 0u) s_lam[k] = make_float2(p.r.l0, p.r.l1);
        };
        for (uint32_t sidx = 0; sidx < ns; ++sidx) {
            SFX_STAMP(pq0);
            const uint32_t st = st_cur, st2 = st_nxt;
            const uint32_t st3 = s_step[min(sidx + 2u, kMaxSteps - 1u)];   // (in flight during the step; first used by the next one)
            const uint32_t k = (st & 0xFFFFu) + half;
            SideSlot cur;
            SFX_STAMP_VM(pq1);
            decode_side<false>(q, side, cur, nb);
            {   // the record of the next step takes this one's registers
                const uint32_t k2 = (st2 & 0xFFFFu) + half;
                fetch_side<false>(T, C, start + k2, side, sidx + 1u < ns && k2 < (st2 >> 16), q);
            }
            SFX_STAMP(pq2);
            if (cur.valid) solve(k, cur);
            SFX_STAMP(pq3);
            if ((st & 0xFFFFu) + (uint32_t)(NT / 2) >= (st >> 16)) { lds_barrier(); ++colour; }   // the colour's last chunk
            SFX_STAMP(pq4);
#ifdef SFX_STEP_PROBE
            pa0 += pq1 - pq0; pa1 += pq2 - pq1; pa2 += pq3 - pq2; pa3 += pq4 - pq3; pns += 1;
#endif
            st_cur = st2; st_nxt = __builtin_amdgcn_readfirstlane(st3);
        }
        __syncthreads();
        SFX_PROBE_OUT(64)
        if (tid == NT - 64) s_next = ahead;  // read after the barrier that ends the write-back
        for (uint32_t k = tid; k < npb; k += NT) C.lam[start + k] = s_lam[k];
        SFX_TT(tt3);
        SFX_TT_ADD(1, tt1, tt2); SFX_TT_ADD(2, tt2, tt3); SFX_TT_ADD(5, 0ull, (unsigned long long)ncol);
        if (nph > 1u) {  // fused phases: later batches of this launch read these displacements
            for (uint32_t l = tid; l < nb; l += NT) {
                const float2 w = s_mass[l];
                if (w.x > 0.0f || w.y > 0.0f) store_record16<true>(B.dpos, dpos_rs, s_gid[l], s_d[l]);
            }
            if (ph + 1u < ph0 + nph) publish_batch(T, g, ph, it, epoch);
            else __syncthreads();
        } else {
            for (uint32_t l = tid; l < nb; l += NT) {
                const float2 w = s_mass[l];
                if (w.x > 0.0f || w.y > 0.0f) B.dpos[s_gid[l]] = s_d[l];
            }
            __syncthreads();
        }
        SFX_TT(tt4);
        SFX_TT_ADD(0, ttw, tt1); SFX_TT_ADD(3, tt0, tt4); SFX_TT_ADD(4, 0ull, 1ull); SFX_TT_ADD(7, tt3, tt4); SFX_TT_MAX(15, ttw, tt4);
    }
    // the fallback list is usually empty: then no workgroup pays for the exit ticket (fence + atomic round trip)
    if ((flags & kTileFallback) && ctr->n_fb != 0u && last_workgroup(ctr)) {
        if (flags & kTileNarrowAll) {
            if constexpr (INTEG) narrow_fallback_tick(T, ctr, B, C, sm, spi, sub);
            else narrow_fallback(T, ctr, B, C, sm, vthr);
        }
        fallback_pos(T, ctr, B, C, max_corr);
    }
}

// Velocity pass, same batches.  The bodies' velocities go to LDS; inside sfx_step (vstamp != 0) the first batch of the sweep
// to load a body applies the displacement the position sweep left for it, derives its velocity and stamps it (DESIGN.md
// "Velocities derived where they are first read").  The poses stay in the registers of the lanes that loaded them: the
// colour steps work on velocities and frozen rows only; static friction accumulates a pose shift per body in LDS, which
// is applied when the batch writes back.
// INTEG: no k_integrate ran -- the first batch to load a body integrates it, applies its displacement, derives it, and
// leaves its displacement record zero for the next substep (no start-of-substep poses in memory)
template <int NT, bool INTEG = false>
__global__ __launch_bounds__(NT, SFX_TILE_MIN_WAVES) void k_tile_vel(uint32_t ph0, uint32_t nph, uint32_t epoch, int flags, GridParams g,
                                                  TileArrays T, Counters* __restrict__ ctr, BodyArrays B, ContactArrays C,
                                                  StepParams sp, uint32_t vstamp) {
    __shared__ float4 s_vel[kMaxBX];
    __shared__ float4 s_sh[kMaxBX];    // static-friction shifts (dx, dy, dtheta, -) of this batch
    __shared__ float2 s_mass[kMaxBX];
    __shared__ uint32_t s_gid[kMaxBX];
    __shared__ uint32_t s_step[kMaxSteps], s_nsteps;
    SFX_RACE_DECL
    constexpr int kDepth = SFX_DEPTH_VEL;
    constexpr int kRounds = (int)((kMaxBX + NT - 1) / NT);   // bodies a lane loads (and writes back) per batch, at most
    const uint32_t tid = threadIdx.x;
#ifdef SFX_TILE_TIMING
    if (threadIdx.x == 0 && T.dbg) atomicMax(&T.dbg[48 + 2 * (epoch & 15u)], ~wall_clock64());   // earliest kernel entry
#endif
    const __amdgpu_buffer_rsrc_t pose_rs = record_rsrc(B.pose), vel_rs = record_rsrc(B.vel);
    // The near launch (kTileOnlyNear) runs BESIDE the far launch, on the halo stream: it takes no tickets (the far launch owns
    // the counters) -- workgroup b takes the items b, b + grid, ... of the near list, and the host keeps its grid small enough
    // for both launches to be on the device together, so the lowest unfinished item is always held by a running workgroup.
    const bool strided = (flags & kTileOnlyNear) != 0;
    bool only_near = strided;
    const bool only_far = (flags & kTileOnlyFar) != 0;
    if ((only_near || only_far) && ctr->n_near == kNoSplit) {   // this tick's sweep is not split: all of it in the near launch
        if (only_far) {   // nothing left to do -- but the ticket counter of the NEXT launch is this launch's to clear (next_item)
            if (blockIdx.x == 0 && threadIdx.x == 0) ctr->item_ticket[(flags & kTileSlot) ? 0 : 1] = 0u;
            return;
        }
        only_near = false;
    }
    if (only_near) flags &= ~kTileFallback;   // (the fallback list follows the LAST batch: the far launch has it)
    const uint32_t nitems = only_near ? min(ctr->n_near, T.near_cap) : phase_total(ctr, ph0, nph);
    __shared__ uint32_t s_next;
    for (uint32_t item = strided ? blockIdx.x : next_item(ctr, flags); item < nitems; item = __builtin_amdgcn_readfirstlane(s_next)) {
        PhaseItem pi;
        if (only_near) { const uint32_t v = T.near_list[item]; pi.ph = v >> 24; pi.it = v & 0xFFFFFFu; pi.ok = true; }
        else pi = phase_item(ctr, ph0, nph, item);
        const uint32_t ph = pi.ph, it = pi.it;
        const BatchDesc d = T.batches[ph * g.pstride + it];
        if (only_far && (d.count & kBatchNear)) {   // the near launch has solved it (and published it under this epoch)
            __syncthreads();   // every thread has read s_next
            if (tid == 0)
                s_next = __hip_atomic_fetch_add(ticket_of(ctr, flags), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ((flags & kTileStaticFirst) ? gridDim.x : 0u);
            __syncthreads();
            continue;
        }
        const uint32_t* tiles = T.active + d.first;   // one tile list for all phases
        SFX_TT(tv0);
        const uint32_t nb = min(d.nb, kMaxBX), start = d.bstart, ncol = d.ncol;
        const uint32_t side = tid & 1u, half = tid >> 1;   // two lanes per pair: body a, body b
        __syncthreads();  // the previous batch's write-back has read the LDS arrays
        build_steps<NT>(T.bruns + 64u * d.roff, ncol, s_step, &s_nsteps);   // as in k_tile_pos: what earlier phases do not write is fetched before waiting for them
        for (uint32_t l = tid; l < nb; l += NT) {
            const uint32_t gid = T.bbody[d.boff + l];
            s_gid[l] = gid;
            s_mass[l] = T.bmass[d.boff + l];
            s_sh[l] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        SFX_RACE_CLEAR(nb)
        __syncthreads();
#ifdef SFX_PROBE_NOLOOP   // timing probe: a batch without its colour loop (load, hand-off and write-back only; results are wrong)
        const uint32_t ns = 0u;
#else
        const uint32_t ns = s_nsteps;
#endif
        static_assert(kDepth == 1, "the colour loop keeps the step entries of this step and the next in registers");
        RawSide q;
        SFX_PROBE_DECL
        uint32_t st_cur = __builtin_amdgcn_readfirstlane(s_step[0]), st_nxt = __builtin_amdgcn_readfirstlane(s_step[1]);   // as in k_tile_pos
        fetch_side<true>(T, C, start + (st_cur & 0xFFFFu) + half, side, 0u < ns && (st_cur & 0xFFFFu) + half < (st_cur >> 16), q);
        if (nph > 1u && ph > ph0) wait_for_earlier_phases<NT>(T, g, ctr, tiles, d.count & ~kBatchNear, ph, ph0, epoch);
        SFX_TT(tvw);
        float4 rp[kRounds];
        uint32_t derived = 0u;
        (void)derived;
#pragma unroll
        for (int j = 0; j < kRounds; ++j) {
            const uint32_t l = tid + (uint32_t)j * NT;
            rp[j] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
            if (l < nb) {
                const uint32_t gid = s_gid[l];
                float4 ps = B.pose[gid];
                float4 vs = B.vel[gid];
                if constexpr (INTEG) {
                    const float4 dd = B.dpos[gid];   // (fetched beside the others: no load waits for the stamp)
                    const float2 w = s_mass[l];
                    if ((w.x > 0.0f || w.y > 0.0f) && __float_as_uint(vs.w) != vstamp) {
                        float4 q0;
                        integrate_body(ps, q0, vs, w, sp);
                        apply_delta(ps, dd);
                        derive_body(ps, q0, vs, sp);
                        vs.w = __uint_as_float(vstamp);
                        derived |= 1u << j;
                    }
                } else if (vstamp) {   // derive-on-load: the first batch of the sweep to read a body applies its displacement, derives its velocity and stamps it
                    const float4 q = B.prev[gid], dd = B.dpos[gid];   // (fetched beside the others: no load waits for the stamp)
                    const float2 w = s_mass[l];
                    if ((w.x > 0.0f || w.y > 0.0f) && __float_as_uint(vs.w) != vstamp) {
                        apply_delta(ps, dd);
                        derive_body(ps, q, vs, sp);
                        vs.w = __uint_as_float(vstamp);
                    }
                }
                rp[j] = ps;
                s_vel[l] = vs;
            }
        }
        __syncthreads();
        const uint32_t ahead = strided ? item + gridDim.x : take_ahead<NT>(ctr, flags);
        SFX_TT(tv1);
        uint32_t colour = 0u;
        (void)colour;
        auto solve = [&](const SideSlot& p) __attribute__((always_inline)) {
            float4 vs = s_vel[p.li], S = s_sh[p.li];
            const float2 w = s_mass[p.li];
#ifdef SFX_PROBE_NOSOLVE
            vs.x += p.r.c[0] * 1e-30f;
            if (w.x > 0.0f) s_vel[p.li] = vs;
            return;
#endif
            const int ch = solve_vel_side(p.r, vs, S, w, sp);
            if (w.x > 0.0f || w.y > 0.0f) { SFX_RACE_WRITE(p.li, colour); s_vel[p.li] = vs; if (ch & 2) s_sh[p.li] = S; }
        };
        for (uint32_t sidx = 0; sidx < ns; ++sidx) {
            SFX_STAMP(pq0);
            const uint32_t st = st_cur, st2 = st_nxt;
            const uint32_t st3 = s_step[min(sidx + 2u, kMaxSteps - 1u)];
            SideSlot cur;
            SFX_STAMP_VM(pq1);
            decode_side<true>(q, side, cur, nb);
            {
                const uint32_t k2 = (st2 & 0xFFFFu) + half;
                fetch_side<true>(T, C, start + k2, side, sidx + 1u < ns && k2 < (st2 >> 16), q);
            }
            SFX_STAMP(pq2);
            if (cur.valid) solve(cur);
            SFX_STAMP(pq3);
            if ((st & 0xFFFFu) + (uint32_t)(NT / 2) >= (st >> 16)) { lds_barrier(); ++colour; }
            SFX_STAMP(pq4);
#ifdef SFX_STEP_PROBE
            pa0 += pq1 - pq0; pa1 += pq2 - pq1; pa2 += pq3 - pq2; pa3 += pq4 - pq3; pns += 1;
#endif
            st_cur = st2; st_nxt = __builtin_amdgcn_readfirstlane(st3);
        }
        __syncthreads();
        SFX_PROBE_OUT(72)
        if (tid == NT - 64) s_next = ahead;
        SFX_TT(tv2);
        // write-back by the lanes that loaded: velocity, and the pose when it changed here (the displacement applied on load,
        // a static-friction shift).  Fused phases: later batches of this launch read these records (write-through stores)
#pragma unroll
        for (int j = 0; j < kRounds; ++j) {
            const uint32_t l = tid + (uint32_t)j * NT;
            if (l < nb) {
                const float2 w = s_mass[l];
                if (w.x > 0.0f || w.y > 0.0f) {
                    const float4 S = s_sh[l];
                    const bool shifted = S.x != 0.0f || S.y != 0.0f || S.z != 0.0f;
                    float4 ps = rp[j];
                    if (shifted) apply_delta(ps, S);
                    if (nph > 1u) {
                        store_record16<true>(B.vel, vel_rs, s_gid[l], s_vel[l]);
                        if (shifted || vstamp) store_record16<true>(B.pose, pose_rs, s_gid[l], ps);
                    } else {
                        B.vel[s_gid[l]] = s_vel[l];
                        if (shifted || vstamp) B.pose[s_gid[l]] = ps;
                    }
                    if constexpr (INTEG) {   // the next substep's sweep starts from nothing (nobody reads the record before the next launch)
                        if (derived & (1u << j)) B.dpos[s_gid[l]] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
                }
            }
        }
        if (nph > 1u && ph + 1u < ph0 + nph) publish_batch(T, g, ph, it, epoch);
        else __syncthreads();
        SFX_TT(tv3);
        SFX_TT_ADD(8, tv0, tvw); SFX_TT_ADD(9, tvw, tv1); SFX_TT_ADD(10, tv1, tv2); SFX_TT_ADD(11, tv2, tv3);
        SFX_TT_ADD(12, tv0, tv3); SFX_TT_ADD(13, 0ull, 1ull); SFX_TT_MAX(14, tvw, tv3);
#ifdef SFX_TILE_TIMING
        if (threadIdx.x == 0 && T.dbg) {
            atomicMax(&T.dbg[16 + 2 * (epoch & 15u)], ~tv0);       // earliest batch start of this launch
            atomicMax(&T.dbg[17 + 2 * (epoch & 15u)], tv3);        // latest batch end
        }
#endif
    }
    if ((flags & kTileFallback) && ctr->n_fb != 0u && last_workgroup(ctr)) {
        if constexpr (INTEG) fallback_vel_tick(T, ctr, B, C, sp, vstamp);
        else fallback_vel(T, ctr, B, C, sp, vstamp);
    }
#ifdef SFX_TILE_TIMING
    if (threadIdx.x == 0 && T.dbg) atomicMax(&T.dbg[49 + 2 * (epoch & 15u)], wall_clock64());    // latest kernel exit
#endif
}

// ---------------------------------------------------------------- all substeps of a tick in ONE launch
// DESIGN.md "The substeps of a tick as one launch".  On a world of 100 k bodies a sweep is a chain of dependent colour
// steps that keeps a wave or two per CU busy, and every launch boundary is a device-wide barrier in front of which the
// whole device waits for the slowest chain (and pays a launch's start and end).  Nothing in a substep needs that
// barrier: a batch depends on the batches around it only.  Here the position and the velocity sweeps of ALL substeps
// are the items of one launch -- item = (sweep u = 2 * substep + (0: position, 1: velocity), class, batch), in that
// order, handed out by ticket as in the launches above -- and a batch waits for exactly the batches that can have
// touched what it reads or writes: the batches of the 3 x 3 tiles around its own, of every class, in the sweep before
// (flag >= epoch0 + u) and of the earlier classes in its own sweep (flag >= epoch0 + u + 1).  A batch publishes
// epoch0 + u + 1 in its flag, the same flag in every sweep, so the flags only grow and "has finished sweep u" implies
// every sweep before it.  What used to sit between the sweeps moved into them:
//   - integration: a body's pose and velocity at the start of a substep are a function of its record at the end of the
//     substep before (integrate_body); the narrowphase computes them where it loads a body (tick_integrate) and the
//     velocity sweep where it derives one (the first batch to load a body: it integrates, applies the displacement,
//     derives, stamps and writes the body back, and zeroes its displacement for the next substep).  No k_integrate, no
//     start-of-substep poses in memory.  Bodies that are not dynamic are written by nobody inside the launch: substep s
//     sees them integrated s + 1 times from their record (their velocity is constant);
//   - the bodies no batch holds are not read by any pair: k_tick_free steps them through all substeps after the launch;
//   - the fallback list (usually empty) follows all batches of a sweep: the workgroup that finishes a sweep's last batch
//     solves it, and while there is one every batch of the next sweep waits for it.
// Everything a later item reads is stored write-through (records, multipliers, displacements, poses, velocities) and
// handed over as between the phases of the launches above.  Same functions, same order per body and pair: the states
// are bit-identical to the launch-per-stage path (tests/test_gpu_fused_tick.py).  Host side: sfx_host.hip tick_fused().
static_assert(kBatchP * sizeof(float2) <= kMaxBX * sizeof(float4), "the multipliers of a batch share the LDS of the friction shifts");

typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_rec_wt(__amdgpu_buffer_rsrc_t rs, uint32_t idx, const float4& v) {
    const v4u32 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)(idx * 16u), 0, 16 /* sc1: write-through */);
}

// store_record with write-through stores (rs: raw buffer over C.rec; the host checks that all planes lie within 4 GB)
__device__ __forceinline__ void store_record_wt(const ContactArrays& C, __amdgpu_buffer_rsrc_t rs, uint32_t slot, const PairRec& r) {
    const uint32_t n = C.nslots;
    const int count = r.count;
    const float4 pa = make_float4(r.nx, r.ny, __uint_as_float((uint32_t)count | (r.block ? 256u : 0u)), r.iK[0]);
    const float4 pb = make_float4(r.ca[0], r.cb[0], r.d0[0], r.tgt[0]), pc = make_float4(r.cta, r.ctb, r.mu_s, r.mu_d);
    const float4 pd = make_float4(r.At[0], r.invS, r.Att, 0.0f), pe = make_float4(r.ca[1], r.cb[1], r.d0[1], r.tgt[1]);
    const float4 pf = make_float4(r.i00, r.i01, r.i11, r.K01), pg = make_float4(r.iK[1], r.At[1], r.invS0, r.invS1);
    store_rec_wt(rs, kPlA * n + slot, pa);
    if (count == 0) return;
    store_rec_wt(rs, kPlB * n + slot, pb);
    store_rec_wt(rs, kPlC * n + slot, pc);
    store_rec_wt(rs, kPlD * n + slot, pd);
    if (count == 2) {
        store_rec_wt(rs, kPlE * n + slot, pe);
        store_rec_wt(rs, kPlF * n + slot, pf);
        store_rec_wt(rs, kPlG * n + slot, pg);
    }
}

// pose and velocity of a body at the start of substep `s` of the tick, from its record
__device__ __forceinline__ void tick_integrate(float4& p, float4& v, const float2& w, const StepParams& sp, uint32_t s) {
    float4 q;
    integrate_body(p, q, v, w, sp);   // a dynamic body: its record is the one of the end of the substep before
    // A body that is not dynamic is written by nobody inside the tick: its record is the one of the start of the tick,
    // and substep s sees it integrated s + 1 times (its velocity is constant).  A body at rest is a fixed point after at
    // most a rounding of its rotation -- the usual case ends here, with the one step above that changed nothing.
    if (!(w.x > 0.0f || w.y > 0.0f) && s != 0u) {
        for (uint32_t i = 0; i < s; ++i) {
            if (__float_as_uint(p.x) == __float_as_uint(q.x) && __float_as_uint(p.y) == __float_as_uint(q.y) &&
                __float_as_uint(p.z) == __float_as_uint(q.z) && __float_as_uint(p.w) == __float_as_uint(q.w)) break;
            integrate_body(p, q, v, w, sp);
        }
    }
}

// WT: the records are stored write-through (a later item of the same launch reads them)
template <bool WT>
__device__ __forceinline__ void narrow_batch_tick(const TileArrays& T, const BodyArrays& B, const ContactArrays& C,
                                                  uint32_t start, uint32_t np, float sm,
                                                  const StepParams& sp, uint32_t s, uint32_t nt, uint32_t tid) {
    const __amdgpu_buffer_rsrc_t rec_rs = record_rsrc(C.rec);
    (void)rec_rs;
    for (uint32_t k = tid; k < np; k += nt) {
        const uint4 o = T.torder[start + k];
        const bool skip = o.z == kNoPair;
        const uint32_t a = skip ? 0u : o.x, b = skip ? 0u : o.y;   // (no branch between the order entry and the gathers)
        float4 pa = B.pose[a], pb = B.pose[b], va = B.vel[a], vb = B.vel[b];
        const float4 sa = B.shape[a], sb = B.shape[b];
        const float2 wa = B.mass[a], wb = B.mass[b];   // (no further colliders in a world that takes this path: entries are bodies)
        if (skip) continue;
        tick_integrate(pa, va, wa, sp, s);
        tick_integrate(pb, vb, wb, sp, s);
        PairRec r;
        prepare_pair(pa, sa, va, wa, pb, sb, vb, wb, shape_pose(pa, sa, B.lpose, a), shape_pose(pb, sb, B.lpose, b), B.poly, B.mats, sm, sp.vthr, r);
        if constexpr (WT) store_record_wt(C, rec_rs, o.z, r);
        else store_record(C, o.z, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void narrow_fallback_tick(const TileArrays& T, const Counters* __restrict__ ctr, const BodyArrays& B,
                                                     const ContactArrays& C, float sm, const StepParams& sp, uint32_t s) {
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap) ? ctr->fb_start[SFX_MAX_COLORS] : 0u;
    for (uint32_t k = threadIdx.x; k < nfb; k += blockDim.x) {
        const uint2 ab = T.fsorted[k];
        const uint32_t a = ab.x, b = ab.y;
        PairRec r;
        float4 pa = B.pose[a], pb = B.pose[b], va = B.vel[a], vb = B.vel[b];
        const float4 sa = B.shape[a], sb = B.shape[b];
        const float2 wa = B.mass[a], wb = B.mass[b];
        tick_integrate(pa, va, wa, sp, s);
        tick_integrate(pb, vb, wb, sp, s);
        prepare_pair(pa, sa, va, wa, pb, sb, vb, wb, shape_pose(pa, sa, B.lpose, a), shape_pose(pb, sb, B.lpose, b), B.poly, B.mats, sm, sp.vthr, r);
        store_record(C, T.max_pairs + k, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// fallback_vel of the one-launch tick: a dynamic body of the list that no batch has stepped yet in this substep is
// integrated, moved and derived here, whether or not its pair touches (nobody else will)
__device__ __forceinline__ void fallback_vel_tick(const TileArrays& T, const Counters* __restrict__ ctr, const BodyArrays& B,
                                                  const ContactArrays& C, const StepParams& sp, uint32_t vstamp) {
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap);
    if (nfb == 0) return;
    const uint32_t nc = ctr->n_fb_colors;
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t c = 0; c < nc; ++c) {
        const uint32_t s = ctr->fb_start[c], e = ctr->fb_start[c + 1];
        for (uint32_t k = s + threadIdx.x; k < e; k += blockDim.x) {
            const uint2 ab = T.fsorted[k];
            PairRec r;
            load_record(C, T.max_pairs + k, r);
            float4 va = B.vel[ab.x], vb = B.vel[ab.y];
            const float2 wa = B.mass[ab.x], wb = B.mass[ab.y];
            const bool dyna = wa.x > 0.0f || wa.y > 0.0f, dynb = wb.x > 0.0f || wb.y > 0.0f;
            bool da = false, db = false;
            if (dyna && __float_as_uint(va.w) != vstamp) {
                float4 pa = B.pose[ab.x], q;
                integrate_body(pa, q, va, wa, sp);
                apply_delta(pa, B.dpos[ab.x]);
                derive_body(pa, q, va, sp); va.w = __uint_as_float(vstamp); da = true;
                B.pose[ab.x] = pa; B.dpos[ab.x] = zero;
            }
            if (dynb && __float_as_uint(vb.w) != vstamp) {
                float4 pb = B.pose[ab.y], q;
                integrate_body(pb, q, vb, wb, sp);
                apply_delta(pb, B.dpos[ab.y]);
                derive_body(pb, q, vb, sp); vb.w = __uint_as_float(vstamp); db = true;
                B.pose[ab.y] = pb; B.dpos[ab.y] = zero;
            }
            float4 Sa = zero, Sb = zero;
            const int ch = r.count ? solve_vel_rec(r, va, wa, vb, wb, Sa, Sb, sp) : 0;
            if (ch || da || db) {
                if (dyna) B.vel[ab.x] = va;
                if (dynb) B.vel[ab.y] = vb;
            }
            if (ch & 2) {
                if (dyna) { float4 pa = B.pose[ab.x]; apply_delta(pa, Sa); B.pose[ab.x] = pa; }
                if (dynb) { float4 pb = B.pose[ab.y]; apply_delta(pb, Sb); B.pose[ab.y] = pb; }
            }
        }
        __syncthreads();
    }
}

// a copy of uniform parameters that the compiler cannot see through (scalar registers): what a stage derives from them --
// broadcast copies in vector registers, products -- is then derived inside that stage instead of once per kernel, where it
// would be live across every other stage
__device__ __forceinline__ float opaque_s(float x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ StepParams opaque_params(const StepParams& sp) {
    StepParams o;
    o.h = opaque_s(sp.h); o.inv_h = opaque_s(sp.inv_h); o.inv_h2 = opaque_s(sp.inv_h2); o.gx = opaque_s(sp.gx); o.gy = opaque_s(sp.gy);
    o.vthr = opaque_s(sp.vthr); o.max_corr = opaque_s(sp.max_corr); o.damp_lin = opaque_s(sp.damp_lin); o.damp_ang = opaque_s(sp.damp_ang);
    o.flags = sp.flags;
    return o;
}

// first wave: the batches of the 3 x 3 tiles around each of this batch's tiles have published at least tgt_lo (classes
// below p_split, if `lo`) / tgt_hi (the other classes, if `hi`), and the fallback list of the sweep before is done (if
// `fb`); then one agent-scope acquire.  Always ends in a workgroup barrier.
template <int NT>
__device__ __forceinline__ void tick_wait(const TileArrays& T, const GridParams& g, Counters* __restrict__ ctr,
                                          const uint32_t* __restrict__ tiles, uint32_t count, uint32_t nph, uint32_t p_split,
                                          bool lo, uint32_t tgt_lo, bool hi, uint32_t tgt_hi, bool fb, uint32_t tgt_fb, uint32_t tid) {
    if ((lo && p_split > 0u) || hi || fb) {   // (uniform)
        if (tid < 64u) {
            const uint32_t lx = g.gxl - g.tlx;
            const bool one_row = g.tly == g.gyl;
            for (uint32_t j = tid; j < count * 9u; j += 64u) {
                const uint32_t t = tiles[j / 9u], dir = j % 9u;
                const int dx = (int)(dir % 3u) - 1, dy = (int)(dir / 3u) - 1;
                if (one_row && dy != 0) continue;
                const int32_t tx = (int32_t)(t & ((1u << lx) - 1u)), ty = (int32_t)(t >> lx);
                const uint32_t t2 = tile_of(tx + dx, ty + dy, g);
                for (uint32_t p2 = 0; p2 < nph; ++p2) {
                    const bool is_lo = p2 < p_split;
                    if (is_lo ? !lo : !hi) continue;
                    const uint32_t tgt = is_lo ? tgt_lo : tgt_hi;
                    const uint32_t b2 = T.tile_batch[p2 * g.ntiles + t2];
                    if (!b2) continue;  // that tile has no pairs of that class this tick
                    const uint32_t* flag = &T.done[p2 * g.pstride + (b2 - 1u)];
                    uint32_t spins = 0;
                    while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tgt) < 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > 400000u) { atomicOr(&ctr->overflow, kOvfStall); break; }
                    }
                }
            }
            if (fb && tid == 0u) {
                uint32_t spins = 0;
                while ((int32_t)(__hip_atomic_load(&ctr->fb_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tgt_fb) < 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 400000u) { atomicOr(&ctr->overflow, kOvfStall); break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before the barrier opens
        }
    }
    __syncthreads();
}

// the fallback list of one sweep, by the workgroup that finished the sweep's last batch (all threads).  (Measured as a
// function that is not inlined: the call costs the sweeps 105 spilled VGPRs instead of 42.)
template <bool NARROW>
__device__ __forceinline__ void tick_fallback(bool vel, const TileArrays& T, Counters* __restrict__ ctr, const BodyArrays& B,
                                                        const ContactArrays& C, float sm, const StepParams& sp, uint32_t s,
                                                        uint32_t vstamp, uint32_t e_cur, bool last) {
    if (threadIdx.x < 64u) {   // every batch of the sweep has published: their stores are in memory
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!vel) {
        if constexpr (NARROW) narrow_fallback_tick(T, ctr, B, C, sm, sp, s);
        fallback_pos(T, ctr, B, C, sp.max_corr);
    } else {
        fallback_vel_tick(T, ctr, B, C, sp, vstamp);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0u) {   // plain stores above: one release for the workgroup, then the flag
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (last) ctr->ticket = 0u;   // (the next launch of this tick counts its own batches from zero)
        __hip_atomic_store(&ctr->fb_done, e_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifndef SFX_TICK_WAVES
#define SFX_TICK_WAVES 3
#endif
// NARROW = false: the records come from a k_narrow_tick launch in front of this one (then one launch holds one substep, the
// s0-th of its tick): no narrowphase in this kernel's register allocation.
template <int NT, bool NARROW>
__global__ __launch_bounds__(NT, SFX_TICK_WAVES) void k_tile_tick(uint32_t nsub, uint32_t s0, uint32_t nph, uint32_t epoch0, uint32_t vstamp0, int flags,
                                                     GridParams g, TileArrays T, Counters* __restrict__ ctr, BodyArrays B,
                                                     ContactArrays C, StepParams sp_arg, float sm_arg) {
    __shared__ float4 s_a[kMaxBX];     // position sweep: displacements; velocity sweep: velocities
    __shared__ float4 s_b[kMaxBX];     // position sweep: the batch's multipliers (float2 per pair); velocity sweep: static-friction shifts
    __shared__ float2 s_mass[kMaxBX];
    __shared__ uint32_t s_gid[kMaxBX];
    __shared__ uint32_t s_step[kMaxSteps], s_nsteps, s_next, s_fb;
    SFX_RACE_DECL
    float2* const s_lam = reinterpret_cast<float2*>(s_b);
    constexpr int kRounds = (int)((kMaxBX + NT - 1) / NT);
    const __amdgpu_buffer_rsrc_t dpos_rs = record_rsrc(B.dpos), pose_rs = record_rsrc(B.pose), vel_rs = record_rsrc(B.vel);
    const __amdgpu_buffer_rsrc_t lam_rs = __builtin_amdgcn_make_buffer_rsrc((void*)C.lam, 0, (int)0xFFFFFFF0u, 0x00020000);
    const uint32_t S = phase_total(ctr, 0u, nph);
    const uint32_t nitems = S * 2u * nsub;
    const bool has_fb = ctr->n_fb != 0u;
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t item = next_item(ctr, flags); item < nitems; item = __builtin_amdgcn_readfirstlane(s_next)) {
        // (the lane's index through a copy the compiler cannot see through, here and again behind the narrowphase and the
        // waits: what is derived from it is then derived per item and per stage, not once per kernel and kept -- or spilled
        // and reloaded inside the colour steps -- across the narrowphase, which wants every register)
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const uint32_t u = item / S, ri = item - u * S, s = s0 + (u >> 1);
        const bool vel = (u & 1u) != 0u;
        const PhaseItem pi = phase_item(ctr, 0u, nph, ri);
        const uint32_t ph = pi.ph, it = pi.it;
        const BatchDesc d = T.batches[ph * g.pstride + it];
        const uint32_t* tiles = T.active + d.first;
        const uint32_t ntl = d.count & ~kBatchNear;
        const uint32_t nb = min(d.nb, kMaxBX), start = d.bstart, ncol = d.ncol, npb = min(d.np, kBatchP);
        const uint32_t e_prev = epoch0 + u, e_cur = e_prev + 1u;   // flags of the sweep before / of this sweep
        __syncthreads();  // the previous item's write-back has read the LDS arrays
        // what is the same all frame long -- the step list, the body ids, the masses -- before any waiting
        build_steps<NT>(T.bruns + 64u * d.roff, ncol, s_step, &s_nsteps);
        for (uint32_t l = tid; l < nb; l += NT) {
            s_gid[l] = T.bbody[d.boff + l];
            s_mass[l] = T.bmass[d.boff + l];
            if (vel) s_b[l] = zero4;
            else if (ph == 0u) s_a[l] = zero4;   // nothing has moved yet in this substep
        }
        if (!vel) for (uint32_t k = tid; k < npb; k += NT) s_lam[k] = make_float2(0.0f, 0.0f);
        SFX_RACE_CLEAR(nb)
        uint32_t colour = 0u;   // (counts the barriers of the loop: the race detector's stamp)
        (void)colour;
        if (!vel) {
            // ---- position sweep of substep s: narrowphase of the batch's pairs at the integrated poses, then the colours
            tick_wait<NT>(T, g, ctr, tiles, ntl, nph, 0u, false, 0u, u > 0u, e_prev, has_fb && u > 0u, e_prev, tid);
            const StepParams sp = opaque_params(sp_arg);
            const float sm = opaque_s(sm_arg);
            if constexpr (NARROW) {
                narrow_batch_tick<true>(T, B, C, start, npb, sm, sp, s, NT, tid);
                __syncthreads();   // the records of the other waves' pairs (same CU)
            }
            // Everything the colour loop derives from the lane's index is derived HERE, from a copy the compiler cannot see
            // through: computed once per kernel they would be live across the narrowphase, which wants every register --
            // they were spilled and reloaded (with a wait for ALL loads in flight, the record prefetch included) in every step
            uint32_t tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const uint32_t side = tid & 1u, half = tid >> 1;   // two lanes per pair: body a, body b
            RawSide q;
            const uint32_t ns = s_nsteps;
            uint32_t st_cur = __builtin_amdgcn_readfirstlane(s_step[0]), st_nxt = __builtin_amdgcn_readfirstlane(s_step[1]);
            fetch_side<false>(T, C, start + (st_cur & 0xFFFFu) + half, side, 0u < ns && (st_cur & 0xFFFFu) + half < (st_cur >> 16), q);
            if (ph != 0u) {
                tick_wait<NT>(T, g, ctr, tiles, ntl, nph, ph, true, e_cur, false, 0u, false, 0u, tid);
                for (uint32_t l = tid; l < nb; l += NT) s_a[l] = B.dpos[s_gid[l]];
                __syncthreads();
            }
            const uint32_t ahead = take_ahead<NT>(ctr, flags);
            for (uint32_t sidx = 0; sidx < ns; ++sidx) {
                const uint32_t st = st_cur, st2 = st_nxt;
                const uint32_t st3 = s_step[min(sidx + 2u, kMaxSteps - 1u)];
                const uint32_t k = (st & 0xFFFFu) + half;
                SideSlot cur;
                decode_side<false>(q, side, cur, nb);
                {
                    const uint32_t k2 = (st2 & 0xFFFFu) + half;
                    fetch_side<false>(T, C, start + k2, side, sidx + 1u < ns && k2 < (st2 >> 16), q);
                }
                if (cur.valid) {
                    float4 D = s_a[cur.li];
                    const float2 w = s_mass[cur.li];
                    if (solve_pos_side(cur.r, D, w, sp.max_corr)) {
                        if (w.x > 0.0f || w.y > 0.0f) { SFX_RACE_WRITE(cur.li, colour); s_a[cur.li] = D; }
                    }
                    if (side == 0u) s_lam[k] = make_float2(cur.r.l0, cur.r.l1);
                }
                if ((st & 0xFFFFu) + (uint32_t)(NT / 2) >= (st >> 16)) { lds_barrier(); ++colour; }
                st_cur = st2; st_nxt = __builtin_amdgcn_readfirstlane(st3);
            }
            __syncthreads();
            if (tid == NT - 64) s_next = ahead;
            for (uint32_t k = tid; k < npb; k += NT) {
                const float2 lm = s_lam[k];
                const v2u32 uu = {__float_as_uint(lm.x), __float_as_uint(lm.y)};
                __builtin_amdgcn_raw_buffer_store_b64(uu, lam_rs, (int)((start + k) * 8u), 0, 16 /* sc1 */);
            }
            for (uint32_t l = tid; l < nb; l += NT) {
                const float2 w = s_mass[l];
                if (w.x > 0.0f || w.y > 0.0f) store_record16<true>(B.dpos, dpos_rs, s_gid[l], s_a[l]);
            }
        } else {
            // ---- velocity sweep of substep s: the first batch to load a body integrates it, applies its displacement and derives it
            const uint32_t vstamp = vstamp0 + (u >> 1);
            tick_wait<NT>(T, g, ctr, tiles, ntl, nph, ph, true, e_cur, true, e_prev, has_fb, e_prev, tid);
            const StepParams sp = opaque_params(sp_arg);
            uint32_t tid = threadIdx.x;   // (as in the position sweep)
            asm volatile("" : "+v"(tid));
            const uint32_t side = tid & 1u, half = tid >> 1;
            RawSide q;
            const uint32_t ns = s_nsteps;
            uint32_t st_cur = __builtin_amdgcn_readfirstlane(s_step[0]), st_nxt = __builtin_amdgcn_readfirstlane(s_step[1]);
            fetch_side<true>(T, C, start + (st_cur & 0xFFFFu) + half, side, 0u < ns && (st_cur & 0xFFFFu) + half < (st_cur >> 16), q);
            // (no pose stays in a register across the colour loop: a body stepped here has its pose stored at once -- nobody
            // reads it before this batch has published -- and a static-friction shift, which is rare, reads it back)
#pragma unroll
            for (int j = 0; j < kRounds; ++j) {
                const uint32_t l = tid + (uint32_t)j * NT;
                if (l < nb) {
                    const uint32_t gid = s_gid[l];
                    float4 ps = B.pose[gid];
                    float4 vs = B.vel[gid];
                    const float4 dd = B.dpos[gid];   // (fetched beside the others: no load waits for the stamp)
                    const float2 w = s_mass[l];
                    if ((w.x > 0.0f || w.y > 0.0f) && __float_as_uint(vs.w) != vstamp) {
                        float4 q0;
                        integrate_body(ps, q0, vs, w, sp);
                        apply_delta(ps, dd);
                        derive_body(ps, q0, vs, sp);
                        vs.w = __uint_as_float(vstamp);
                        store_record16<true>(B.pose, pose_rs, gid, ps);
                        store_record16<true>(B.dpos, dpos_rs, gid, zero4);   // the next substep's sweep starts from nothing
                    }
                    s_a[l] = vs;
                }
            }
            __syncthreads();
            const uint32_t ahead = take_ahead<NT>(ctr, flags);
            for (uint32_t sidx = 0; sidx < ns; ++sidx) {
                const uint32_t st = st_cur, st2 = st_nxt;
                const uint32_t st3 = s_step[min(sidx + 2u, kMaxSteps - 1u)];
                SideSlot cur;
                decode_side<true>(q, side, cur, nb);
                {
                    const uint32_t k2 = (st2 & 0xFFFFu) + half;
                    fetch_side<true>(T, C, start + k2, side, sidx + 1u < ns && k2 < (st2 >> 16), q);
                }
                if (cur.valid) {
                    float4 vs = s_a[cur.li], Sh = s_b[cur.li];
                    const float2 w = s_mass[cur.li];
                    const int ch = solve_vel_side(cur.r, vs, Sh, w, sp);
                    if (w.x > 0.0f || w.y > 0.0f) { SFX_RACE_WRITE(cur.li, colour); s_a[cur.li] = vs; if (ch & 2) s_b[cur.li] = Sh; }
                }
                if ((st & 0xFFFFu) + (uint32_t)(NT / 2) >= (st >> 16)) { lds_barrier(); ++colour; }
                st_cur = st2; st_nxt = __builtin_amdgcn_readfirstlane(st3);
            }
            __syncthreads();
            if (tid == NT - 64) s_next = ahead;
#pragma unroll
            for (int j = 0; j < kRounds; ++j) {
                const uint32_t l = tid + (uint32_t)j * NT;
                if (l < nb) {
                    const float2 w = s_mass[l];
                    if (w.x > 0.0f || w.y > 0.0f) {
                        const float4 Sh = s_b[l];
                        const uint32_t gid = s_gid[l];
                        store_record16<true>(B.vel, vel_rs, gid, s_a[l]);
                        if (Sh.x != 0.0f || Sh.y != 0.0f || Sh.z != 0.0f) {
                            // (this lane's own store from before the colour loop, or an earlier class's: read past the L1)
                            const v4u32 pu = __builtin_amdgcn_raw_buffer_load_b128(pose_rs, (int)(gid * 16u), 0, 16 /* sc1 */);
                            float4 ps = make_float4(__uint_as_float(pu.x), __uint_as_float(pu.y), __uint_as_float(pu.z), __uint_as_float(pu.w));
                            apply_delta(ps, Sh);
                            store_record16<true>(B.pose, pose_rs, gid, ps);
                        }
                    }
                }
            }
        }
        publish_batch(T, g, ph, it, e_cur);
        if (has_fb) {   // the workgroup that finishes the last batch of a sweep solves the fallback list; the next sweep waits for it
            if (tid == 0u) s_fb = (atomicAdd(&ctr->ticket, 1u) + 1u == (u + 1u) * S) ? 1u : 0u;
            __syncthreads();
            if (s_fb) tick_fallback<NARROW>(vel, T, ctr, B, C, sm_arg, sp_arg, s, vstamp0 + (u >> 1), e_cur, u + 1u == 2u * nsub);
        }
    }
}

// The bodies no batch of the frame holds (and no fallback pair names) take part in nothing: after the launch above they
// are stepped through the tick's substeps here, integrate and derive in turn as k_integrate / k_derive would.  `done`:
// the stamp of the last substep's velocity sweep (a body that carries it has been stepped).  Bodies that are not dynamic
// are integrated `nsub` times (the launch above only ever read them).
__global__ __launch_bounds__(kBlock) void k_tick_free(uint32_t n, float4* __restrict__ pose, float4* __restrict__ prev,
                                                      float4* __restrict__ vel, const float2* __restrict__ mass, StepParams sp,
                                                      uint32_t nsub, uint32_t done) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float2 w = mass[i];
    float4 v = vel[i];
    const bool dyn = w.x > 0.0f || w.y > 0.0f;
    if (dyn && __float_as_uint(v.w) == done) return;
    float4 p = pose[i], q = p;
    if (dyn) {
        for (uint32_t s = 0; s < nsub; ++s) {
            integrate_body(p, q, v, w, sp);
            derive_body(p, q, v, sp);
        }
        vel[i] = v;
    } else {
        for (uint32_t s = 0; s < nsub; ++s) integrate_body(p, q, v, w, sp);
    }
    pose[i] = p;
    prev[i] = q;
}

// k_narrow with the integration on load (the one-launch tick with its narrowphase in a launch of its own, substep s of the tick)
__global__ __launch_bounds__(kBlock, SFX_NARROW_WAVES) void k_narrow_tick(TileArrays T, const Counters* __restrict__ ctr, BodyArrays B,
                                                         ContactArrays C, float sm, StepParams sp, uint32_t s) {
    const uint32_t ntp = min(ctr->tp_cursor, T.max_pairs);
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap) ? ctr->fb_start[SFX_MAX_COLORS] : 0u;
    for (uint32_t j = blockIdx.x * kBlock + threadIdx.x; j < ntp + nfb; j += gridDim.x * kBlock) {
        uint32_t a, b, slot;
        if (j < ntp) {
            const uint4 o = T.torder[j];
            slot = o.z;
            if (slot == kNoPair) continue;
            a = o.x; b = o.y;
        } else {
            const uint2 ab = T.fsorted[j - ntp];
            a = ab.x; b = ab.y; slot = T.max_pairs + (j - ntp);
        }
        PairRec r;
        float4 pa = B.pose[a], pb = B.pose[b], va = B.vel[a], vb = B.vel[b];
        const float4 sa = B.shape[a], sb = B.shape[b];
        const float2 wa = B.mass[a], wb = B.mass[b];
        tick_integrate(pa, va, wa, sp, s);
        tick_integrate(pb, vb, wb, sp, s);
        prepare_pair(pa, sa, va, wa, pb, sb, vb, wb, shape_pose(pa, sa, B.lpose, a), shape_pose(pb, sb, B.lpose, b), B.poly, B.mats, sm, sp.vthr, r);
        store_record(C, slot, r);
    }
}

// ---------------------------------------------------------------- small worlds: whole substeps in ONE workgroup
// A world of at most kSmallBodies bodies does not fill one CU, and its multi-launch substep (integrate, joint colours,
// narrowphase, fused position solve, derive, fused velocity solve) is all dispatch and hand-off: every launch loads
// the bodies, takes a step and writes them back through global memory.  Here ONE workgroup keeps every body's state
// in LDS -- indexed by body id, so that tile pairs, the fallback list and joints all address it alike -- and runs
// `nsub` substeps back to back: the same stages, the same per-pair and per-joint functions, the same batch tables,
// colour runs and orders as the launches above, with workgroup barriers where those have launch boundaries, flags or
// phases.  Results are bit-identical to the multi-launch path (tests/test_gpu_parity.py).  The manifold records still
// go through the contact arrays in global memory (written and read by this one workgroup: visible after a barrier).
// The joint and fallback functions take a BodyArrays view whose state pointers are the LDS arrays (generic pointers).
constexpr uint32_t kSmallBodies = 2048;    // LDS: 72 bytes of state per body
constexpr int kSmallThreads = 512;
constexpr uint32_t kSmallBatches = 32;     // batches of all phases together whose colour runs are tabulated in LDS

// The batches of one phase do not interact, so the workgroup solves colour c of ALL of them in one step (the launches
// above give each batch a workgroup of its own).  Per phase and colour the runs of the phase's batches are laid end to
// end: pre[c][...] holds the running sum of their lengths (one row segment per phase, a leading 0 each), first[c][b]
// the run's first pair within batch b's list, bstart[b] that list's first record slot.  Built once per launch.
struct SmallTables {
    uint16_t pre[SFX_MAX_COLORS][kSmallBatches + 4];
    uint16_t first[SFX_MAX_COLORS][kSmallBatches];
    uint32_t bstart[kSmallBatches];
    uint32_t pb0[5];      // first batch of each phase in the flattened numbering, [nph..4] = all batches
    uint32_t pncol[4];    // most colours of any batch of the phase
};

struct SmallJoints {
    uint32_t ncolors;
    uint32_t damped;
    uint32_t start[SFX_MAX_COLORS + 1];   // joint colour c = joints start[c] .. start[c + 1] of the colour-sorted arrays
};

// one batch of one phase, colour by colour, on the LDS state (the loop of k_tile_pos / k_tile_vel: two lanes per pair,
// the next colour's record in flight while this one is solved)
template <bool VEL>
__device__ __forceinline__ void small_batch(const BatchDesc& d, const TileArrays& T, const ContactArrays& C, const BodyArrays& B,
                                            float4* s_d, float4* s_vel, const float2* s_mass, uint32_t* s_cs, uint32_t* s_ce,
                                            const StepParams& sp) {
    constexpr uint32_t NT = kSmallThreads;
    const uint32_t tid = threadIdx.x, side = tid & 1u, half = tid >> 1;
    const uint32_t start = d.bstart, ncol = d.ncol;
    RawSide nxt;
    fetch_side<VEL>(T, C, start + half, side, start + half < T.max_pairs, nxt);
    __syncthreads();   // the previous batch has read its colour runs
    if (tid < SFX_MAX_COLORS) {
        const uint32_t v = T.bruns[64u * d.roff + tid];
        s_cs[tid] = v & 0xFFFFu; s_ce[tid] = v >> 16;
    }
    __syncthreads();
    nxt.in_range = ncol > 0u && half < s_ce[0];
    auto solve = [&](uint32_t k, const RawSide& q) __attribute__((always_inline)) {
        SideSlot p;
        decode_side<VEL>(q, side, p, 0xFFFFFFFFu);   // no local-index limit: the state is addressed by body id
        if (!VEL && side == 0u && q.in_range && !p.valid) C.lam[start + k] = make_float2(0.0f, 0.0f);
        if (!p.valid) return;
        const uint32_t bi = body_of(B, side ? +q.pr.y : +q.pr.x);
        float4 D = s_d[bi];   // position pass: the displacement; velocity pass: the static-friction shift of this phase
        const float2 w = s_mass[bi];
        if constexpr (VEL) {
            float4 vs = s_vel[bi];
            const int ch = solve_vel_side(p.r, vs, D, w, sp);
            if (w.x > 0.0f || w.y > 0.0f) { s_vel[bi] = vs; if (ch & 2) s_d[bi] = D; }
        } else {
            if (solve_pos_side(p.r, D, w, sp.max_corr)) {
                if (w.x > 0.0f || w.y > 0.0f) s_d[bi] = D;
            }
            if (side == 0u) C.lam[start + k] = make_float2(p.r.l0, p.r.l1);
        }
    };
    for (uint32_t c = 0; c < ncol; ++c) {
        const RawSide cur = nxt;
        const uint32_t k0 = s_cs[c] + half, ce = s_ce[c];
        {
            const uint32_t c1 = min(c + 1u, (uint32_t)SFX_MAX_COLORS - 1u);
            fetch_side<VEL>(T, C, start + s_cs[c1] + half, side, c + 1u < ncol && s_cs[c1] + half < s_ce[c1], nxt);
        }
        solve(k0, cur);
        for (uint32_t k = k0 + NT / 2; k < ce; k += NT / 2) {  // colour runs longer than half the workgroup
            RawSide more;
            fetch_side<VEL>(T, C, start + k, side, true, more);
            solve(k, more);
        }
        lds_barrier();
    }
}

// colour c of every batch of phase ph in one step, on the LDS state (two lanes per pair; the next colour's record is
// in flight while this one is solved)
template <bool VEL>
__device__ __forceinline__ void small_phase(uint32_t ph, const SmallTables& S, const TileArrays& T, const ContactArrays& C,
                                            const BodyArrays& B, float4* s_d, float4* s_vel, const float2* s_mass,
                                            const StepParams& sp) {
    constexpr uint32_t NT = kSmallThreads;
    const uint32_t tid = threadIdx.x, side = tid & 1u, half = tid >> 1;
    const uint32_t b0 = S.pb0[ph], nbat = S.pb0[ph + 1] - b0, base = b0 + ph, ncol = S.pncol[ph];
    if (nbat == 0u) return;
    // record slot of the h-th pair of colour c over the phase's batches (false past the end)
    auto locate = [&](uint32_t c, uint32_t h, uint32_t& slot) __attribute__((always_inline)) -> bool {
        const uint16_t* row = &S.pre[c][base];
        slot = 0u;
        if (h >= row[nbat]) return false;
        uint32_t lo = 0u, hi = nbat;   // largest j with row[j] <= h
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (row[mid] <= h) lo = mid; else hi = mid;
        }
        slot = S.bstart[b0 + lo] + S.first[c][b0 + lo] + (h - row[lo]);
        return true;
    };
    auto solve = [&](uint32_t slot, const RawSide& q) __attribute__((always_inline)) {
        SideSlot p;
        decode_side<VEL>(q, side, p, 0xFFFFFFFFu);   // no local-index limit: the state is addressed by body id
        if (!VEL && side == 0u && q.in_range && !p.valid) C.lam[slot] = make_float2(0.0f, 0.0f);
        if (!p.valid) return;
        const uint32_t bi = body_of(B, side ? +q.pr.y : +q.pr.x);
        float4 D = s_d[bi];   // position pass: the displacement; velocity pass: the static-friction shift of this phase
        const float2 w = s_mass[bi];
        if constexpr (VEL) {
            float4 vs = s_vel[bi];
            const int ch = solve_vel_side(p.r, vs, D, w, sp);
            if (w.x > 0.0f || w.y > 0.0f) { s_vel[bi] = vs; if (ch & 2) s_d[bi] = D; }
        } else {
            if (solve_pos_side(p.r, D, w, sp.max_corr)) {
                if (w.x > 0.0f || w.y > 0.0f) s_d[bi] = D;
            }
            if (side == 0u) C.lam[slot] = make_float2(p.r.l0, p.r.l1);
        }
    };
    RawSide nxt;
    uint32_t nslot = 0u;
    {
        const bool in = ncol > 0u && locate(0u, half, nslot);
        fetch_side<VEL>(T, C, nslot, side, in, nxt);
    }
    for (uint32_t c = 0; c < ncol; ++c) {
        const RawSide cur = nxt;
        const uint32_t cslot = nslot;
        const uint32_t total = S.pre[c][base + nbat];
        {
            const bool in = c + 1u < ncol && locate(c + 1u, half, nslot);
            fetch_side<VEL>(T, C, nslot, side, in, nxt);
        }
        solve(cslot, cur);
        for (uint32_t h = half + NT / 2; h < total; h += NT / 2) {  // colours with more pairs than half the workgroup
            uint32_t slot;
            RawSide more;
            locate(c, h, slot);
            fetch_side<VEL>(T, C, slot, side, true, more);
            solve(slot, more);
        }
        lds_barrier();
    }
}

__global__ __launch_bounds__(kSmallThreads) void k_small_substeps(uint32_t nsub, uint32_t nph, uint32_t n_entries, GridParams g,
                                                                   TileArrays T, Counters* __restrict__ ctr, BodyArrays B,
                                                                   ContactArrays C, JointArrays J, SmallJoints SJ, StepParams sp,
                                                                   float sm, uint32_t max_tabled) {
    constexpr uint32_t NT = kSmallThreads;
    __shared__ SmallTables S;
    __shared__ float4 s_pose[kSmallBodies];
    __shared__ float4 s_prev[kSmallBodies];
    __shared__ float4 s_vel[kSmallBodies];
    __shared__ float4 s_d[kSmallBodies];    // position pass: displacements since the narrowphase; velocity pass: the phase's static-friction shifts
    __shared__ float2 s_mass[kSmallBodies];
    __shared__ uint32_t s_cs[SFX_MAX_COLORS], s_ce[SFX_MAX_COLORS];
    const uint32_t tid = threadIdx.x;
    SFX_TT(ts_entry);
    const uint32_t nb = min(B.nb, kSmallBodies);
    for (uint32_t i = tid; i < nb; i += NT) {
        s_pose[i] = B.pose[i]; s_vel[i] = B.vel[i]; s_mass[i] = B.mass[i];
    }
    BodyArrays L = B;   // the same world with its state in LDS: what solve_joint, damp_joint and the fallback solves take
    L.pose = s_pose; L.prev = s_prev; L.vel = s_vel; L.mass = s_mass; L.dpos = s_d;
    const uint32_t ntp = min(ctr->tp_cursor, T.max_pairs);
    const uint32_t nfb = min(ctr->n_fb, T.fb_cap) ? ctr->fb_start[SFX_MAX_COLORS] : 0u;
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t ph = 0; ph < 4u; ++ph) {
            S.pb0[ph] = run; S.pncol[ph] = 0u;
            if (ph < nph) run += min(ctr->n_batches[ph], g.pstride);
        }
        S.pb0[4] = run;
    }
    __syncthreads();
    // (more batches than the tables hold, or pair lists past 16-bit offsets: the batches are solved one after the other)
    const uint32_t nbt = S.pb0[4];
    const bool tabled = nbt <= min(max_tabled, kSmallBatches) && ntp <= 0xFFFFu;
    if (tabled) {
        for (uint32_t t = tid; t < nbt * (uint32_t)SFX_MAX_COLORS; t += NT) {
            const uint32_t fb = t / (uint32_t)SFX_MAX_COLORS, c = t % (uint32_t)SFX_MAX_COLORS;
            uint32_t ph = 0;
            while (ph + 1u < nph && fb >= S.pb0[ph + 1]) ++ph;
            const BatchDesc d = T.batches[ph * g.pstride + (fb - S.pb0[ph])];
            const uint32_t v = T.bruns[64u * d.roff + c];
            const uint32_t cs = v & 0xFFFFu, ce = v >> 16;
            S.first[c][fb] = (uint16_t)cs;
            S.pre[c][fb + ph + 1u] = (uint16_t)((c < d.ncol && ce > cs) ? ce - cs : 0u);   // lengths first, running sums below
            if (c == 0u) { S.bstart[fb] = d.bstart; atomicMax(&S.pncol[ph], min(d.ncol, (uint32_t)SFX_MAX_COLORS)); }
        }
        __syncthreads();
        if (tid < nph * (uint32_t)SFX_MAX_COLORS) {
            const uint32_t c = tid % (uint32_t)SFX_MAX_COLORS, ph = tid / (uint32_t)SFX_MAX_COLORS;
            const uint32_t base = S.pb0[ph] + ph, nbat = S.pb0[ph + 1] - S.pb0[ph];
            uint32_t run = 0;
            S.pre[c][base] = 0;
            for (uint32_t j = 1; j <= nbat; ++j) { run += S.pre[c][base + j]; S.pre[c][base + j] = (uint16_t)run; }
        }
    }
    __syncthreads();
    SFX_TT(ts_setup);
    SFX_TT_ADD(64, ts_entry, ts_setup);
    for (uint32_t s = 0; s < nsub; ++s) {
        SFX_TT(ts0);
        for (uint32_t i = tid; i < nb; i += NT) {
            float4 p = s_pose[i], q, v = s_vel[i];
            integrate_body(p, q, v, s_mass[i], sp);
            s_pose[i] = p; s_prev[i] = q; s_vel[i] = v; s_d[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        for (uint32_t jc = 0; jc < SJ.ncolors; ++jc) {
            for (uint32_t k = SJ.start[jc] + tid; k < SJ.start[jc + 1]; k += NT) solve_joint(L, J, k, sp.inv_h2);
            __syncthreads();
        }
        SFX_TT(ts1);
        SFX_TT_ADD(65, ts0, ts1);
        // manifolds of every solver pair at these poses (k_narrow's loop; a further collider is where its body is)
        for (uint32_t j = tid; j < ntp + nfb; j += NT) {
            uint32_t a, b, slot;
            if (j < ntp) {
                const uint4 o = T.torder[j];
                slot = o.z;
                if (slot == kNoPair) continue;
                a = o.x; b = o.y;
            } else {
                const uint2 ab = T.fsorted[j - ntp];
                a = ab.x; b = ab.y; slot = T.max_pairs + (j - ntp);
            }
            const uint32_t ba = body_of(B, a), bb = body_of(B, b);
            PairRec r;
            const float4 pa = s_pose[ba], pb = s_pose[bb], sa = B.shape[a], sb = B.shape[b];
            prepare_pair(pa, sa, s_vel[ba], s_mass[ba], pb, sb, s_vel[bb], s_mass[bb], shape_pose(pa, sa, B.lpose, a), shape_pose(pb, sb, B.lpose, b), B.poly, B.mats, sm, sp.vthr, r);
            store_record(C, slot, r);
        }
        __syncthreads();   // (a workgroup barrier orders this workgroup's global stores and loads as well)
        SFX_TT(ts2);
        SFX_TT_ADD(66, ts1, ts2);
        for (uint32_t ph = 0; ph < nph; ++ph) {
            if (tabled) { small_phase<false>(ph, S, T, C, B, s_d, s_vel, s_mass, sp); continue; }
            const uint32_t nbat = S.pb0[ph + 1] - S.pb0[ph];
            for (uint32_t it = 0; it < nbat; ++it)
                small_batch<false>(T.batches[ph * g.pstride + it], T, C, B, s_d, s_vel, s_mass, s_cs, s_ce, sp);
        }
        __syncthreads();
        SFX_TT(ts3);
        SFX_TT_ADD(67, ts2, ts3);
        fallback_pos(T, ctr, L, C, sp.max_corr);
        for (uint32_t i = tid; i < nb; i += NT) {   // the displacements applied, once per body, and the velocities derived
            const float2 w = s_mass[i];
            const float4 dd = s_d[i];
            s_d[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // from here on: the static-friction shifts of a velocity phase
            if (!(w.x > 0.0f || w.y > 0.0f)) continue;
            float4 v = s_vel[i], p = s_pose[i];
            apply_delta(p, dd);
            derive_body(p, s_prev[i], v, sp);
            s_pose[i] = p; s_vel[i] = v;
        }
        __syncthreads();   // also: the multipliers the position pass stored are visible to the velocity pass
        SFX_TT(ts4);
        SFX_TT_ADD(68, ts3, ts4);
        for (uint32_t jc = 0; SJ.damped && jc < SJ.ncolors; ++jc) {
            for (uint32_t k = SJ.start[jc] + tid; k < SJ.start[jc + 1]; k += NT) damp_joint(L, J, k, sp.h);
            __syncthreads();
        }
        for (uint32_t ph = 0; ph < nph; ++ph) {
            if (tabled) small_phase<true>(ph, S, T, C, B, s_d, s_vel, s_mass, sp);
            else {
                const uint32_t nbat = S.pb0[ph + 1] - S.pb0[ph];
                for (uint32_t it = 0; it < nbat; ++it)
                    small_batch<true>(T.batches[ph * g.pstride + it], T, C, B, s_d, s_vel, s_mass, s_cs, s_ce, sp);
            }
            __syncthreads();
            for (uint32_t i = tid; i < nb; i += NT) {   // end of the phase: its static-friction shifts, once per body
                const float4 S4 = s_d[i];
                if (S4.x == 0.0f && S4.y == 0.0f && S4.z == 0.0f) continue;
                const float2 w = s_mass[i];
                s_d[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (!(w.x > 0.0f || w.y > 0.0f)) continue;
                float4 p = s_pose[i];
                apply_delta(p, S4);
                s_pose[i] = p;
            }
            __syncthreads();
        }
        __syncthreads();
        SFX_TT(ts5);
        SFX_TT_ADD(69, ts4, ts5);
        fallback_vel(T, ctr, L, C, sp);
        __syncthreads();
        SFX_TT(ts6);
        SFX_TT_ADD(70, ts5, ts6); SFX_TT_ADD(71, ts0, ts6); SFX_TT_ADD(72, 0ull, 1ull);
    }
    for (uint32_t i = tid; i < nb; i += NT) {
        B.pose[i] = s_pose[i]; B.prev[i] = s_prev[i]; B.vel[i] = s_vel[i];
    }
    for (uint32_t e = nb + tid; e < n_entries; e += NT) {   // further colliders are where their bodies are (k_mirror)
        const uint32_t b = B.cbody[e - nb];
        B.pose[e] = s_pose[b]; B.vel[e] = s_vel[b];
    }
}

__global__ __launch_bounds__(kBlock) void k_solve_joints(uint32_t start, uint32_t cnt, BodyArrays B, JointArrays J,
                                                          float inv_h2) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < cnt) solve_joint(B, J, start + k, inv_h2);
}

// joint damping of one colour (velocity pass; launched only when some joint has damping)
__global__ __launch_bounds__(kBlock) void k_damp_joints(uint32_t start, uint32_t cnt, BodyArrays B, JointArrays J, float h) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < cnt) damp_joint(B, J, start + k, h);
}

// All joint colours of a substep in ONE launch (DESIGN.md "Joints by component").  The joint graph falls into connected
// components (a chain, a ragdoll) that share no body, so the order of the solves only matters inside a component; the host
// packs whole components into groups of at most kJointGroupCap joints, and one workgroup takes a group through the colours
// in order, a workgroup barrier between two colours (the writes of one wave are the reads of another: same workgroup, so
// __syncthreads() orders them) where the launch-per-colour path has a launch boundary.  Per body the same solves in the same
// order: bit-identical.  cstart[g * (ncolors + 1) + c] .. [c + 1]: the group's joints of colour c in `index` (positions in
// the colour-sorted joint arrays).  A world with a larger component keeps the launch per colour.
constexpr uint32_t kJointGroupCap = 256;
constexpr int kJointGroupThreads = 128;

template <bool DAMP>
__global__ __launch_bounds__(kJointGroupThreads) void k_joint_groups(uint32_t ncolors, const uint32_t* __restrict__ cstart,
                                                                     const uint32_t* __restrict__ index, BodyArrays B, JointArrays J,
                                                                     float param /* inv_h2 | h */) {
    const uint32_t* cs = cstart + (size_t)blockIdx.x * (ncolors + 1u);
    // (measured and not kept: the records of a lane's joint of the NEXT colour -- index, bodies, anchors, parameters: they
    // depend on no body -- fetched while the current colour is solved: 38.8 -> 38.0 us per tick on configs[3], nothing)
    for (uint32_t c = 0; c < ncolors; ++c) {
        const uint32_t s = cs[c], e = cs[c + 1u];
        for (uint32_t i = s + threadIdx.x; i < e; i += kJointGroupThreads) {
            const uint32_t k = index[i];
            if (DAMP) damp_joint(B, J, k, param);
            else solve_joint(B, J, k, param);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- slab halos
// (pose, velocity) of the listed bodies <-> a dense [n][2] float4 buffer
__global__ __launch_bounds__(kBlock) void k_halo_pack(uint32_t n, const uint32_t* __restrict__ idx,
                                                       const float4* __restrict__ pose, const float4* __restrict__ vel,
                                                       float4* __restrict__ out) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = idx[k];
    out[2 * k] = pose[i];
    out[2 * k + 1] = vel[i];
}

__global__ __launch_bounds__(kBlock) void k_halo_unpack(uint32_t n, const uint32_t* __restrict__ idx,
                                                         const float4* __restrict__ in, float4* __restrict__ pose,
                                                         float4* __restrict__ vel) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = idx[k];
    pose[i] = in[2 * k];
    float4 v = in[2 * k + 1];
    v.w = 0.0f;   // the sender's derive stamp means nothing here (and must never look like one of this world's: "stepped already")
    vel[i] = v;
}

// x of every body at the moment the halo lists were cut (the reference of the halo guard)
__global__ __launch_bounds__(kBlock) void k_halo_snapshot(uint32_t n, const float4* __restrict__ pose, float* __restrict__ x0) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) x0[i] = pose[i].x;
}

// ---------------------------------------------------------------- reporting / parity hooks
__global__ __launch_bounds__(kBlock) void k_stats_bodies(uint32_t n, const float4* __restrict__ vel,
                                                          const float2* __restrict__ mass, Counters* __restrict__ ctr) {
    __shared__ float part[kBlock / 64];
    float ke = 0.0f;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float4 v = vel[i];
        const float2 w = mass[i];
        if (w.x > 0.0f) ke += 0.5f * (1.0f / w.x) * (v.x * v.x + v.y * v.y);
        if (w.y > 0.0f) ke += 0.5f * (1.0f / w.y) * (v.z * v.z);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ke += __shfl_down(ke, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ke;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int k = 0; k < kBlock / 64; ++k) s += part[k];
        atomicAdd(&ctr->ke, s);
    }
}

// deepest penetration at the current poses and the manifold / active contact point counts of
// the last substep, over all candidate pairs (tile pairs, then the fallback list)
__global__ __launch_bounds__(kBlock) void k_stats_pairs(const Counters* __restrict__ cin, Counters* __restrict__ ctr,
                                                         TileArrays T, BodyArrays B, ContactArrays C, float sm) {
    const uint32_t ntp = cin->n_tile_pairs, nfb = cin->fb_start[SFX_MAX_COLORS];
    uint32_t contacts = 0, mf = 0, mf2 = 0;
    float md = 0.0f;
    __shared__ uint32_t s_hist[SFX_MAX_COLORS];  // tile pairs per colour (the fallback list's are counted by k_color_global)
    if (threadIdx.x < SFX_MAX_COLORS) s_hist[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < ntp + nfb; k += gridDim.x * kBlock) {
        uint32_t a, b, slot;
        if (k < ntp) {
            const uint4 pr = T.tpair[k];
            if ((pr.w & 0xFFu) >= SFX_MAX_COLORS) continue;  // uncoloured: never solved
            atomicAdd(&s_hist[pr.w & 0xFFu], 1u);
            a = pr.x; b = pr.y; slot = k;
        } else {
            const uint2 ab = T.fsorted[k - ntp];
            a = ab.x; b = ab.y; slot = T.max_pairs + (k - ntp);
        }
        Manifold m;
        manifold(shape_pose(B.pose[a], B.shape[a], B.lpose, a), B.shape[a], shape_pose(B.pose[b], B.shape[b], B.lpose, b), B.shape[b], B.poly, sm, m);
        for (int q = 0; q < m.count; ++q) md = fmaxf(md, m.depth[q]);
        const uint32_t cnt = rec_count(rec_plane(C, kPlA)[slot]);
        if (cnt) {
            mf += 1u;
            mf2 += (cnt == 2) ? 1u : 0u;
            const float2 lv = C.lam[slot];
            contacts += (lv.x > 0.0f) ? 1u : 0u;
            contacts += (cnt == 2 && lv.y > 0.0f) ? 1u : 0u;
        }
    }
    if (contacts) atomicAdd(&ctr->n_contacts, contacts);
    if (mf) atomicAdd(&ctr->n_manifolds, mf);
    if (mf2) atomicAdd(&ctr->n_manifolds2, mf2);
    if (md > 0.0f) atomicMax(&ctr->max_depth_bits, __float_as_uint(md));
    __syncthreads();
    if (threadIdx.x < SFX_MAX_COLORS && s_hist[threadIdx.x]) atomicAdd(&ctr->color_count[threadIdx.x], s_hist[threadIdx.x]);
}

__global__ __launch_bounds__(kBlock) void k_contacts_debug(uint32_t np, const uint2* __restrict__ pairs, BodyArrays B,
                                                            float sm, sfx_contact* __restrict__ out) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= np) return;
    const uint2 ab = pairs[k];
    Manifold m;
    manifold(shape_pose(B.pose[ab.x], B.shape[ab.x], B.lpose, ab.x), B.shape[ab.x], shape_pose(B.pose[ab.y], B.shape[ab.y], B.lpose, ab.y),
             B.shape[ab.y], B.poly, sm, m);
    sfx_contact c;
    c.body_a = ab.x; c.body_b = ab.y; c.count = (uint32_t)m.count;
    c.normal[0] = m.count ? m.nx : 0.0f; c.normal[1] = m.count ? m.ny : 0.0f;
    for (int q = 0; q < 2; ++q) {
        const bool on = q < m.count;
        c.depth[q] = on ? m.depth[q] : 0.0f;
        c.point_a[q][0] = on ? m.p1x[q] : 0.0f; c.point_a[q][1] = on ? m.p1y[q] : 0.0f;
        c.point_b[q][0] = on ? m.p2x[q] : 0.0f; c.point_b[q][1] = on ? m.p2y[q] : 0.0f;
    }
    out[k] = c;
}

// ---------------------------------------------------------------- spatial queries
// DESIGN.md "Contract / spatial queries".  The query index is the broadphase's counting sort run
// on the colliders' tight AABBs at the current poses (k_aabb_count with margin, pad and dt = 0,
// row-major buckets; k_scan_*, k_scatter, k_finalize_slots) into buffers of its own, plus the
// bounds of the small colliders' boxes (k_query_bounds), which keep rays and boxes of any length
// to a bounded walk.  Every kernel enumerates a superset of the colliders that can pass the
// per-collider test and applies that test exactly as the oracle does.
struct QueryIndex {
    const uint32_t* cell_start;
    const uint32_t* sorted_id;
    const float4* sorted_aabb;
    const int2* sorted_cell;
    const Counters* ctr;      // n_large
    const uint32_t* bounds;   // ordered keys, see fkey(): [~min x, ~min y, max x, max y], 0 = nothing yet
    uint32_t gxl, gyl, ncell, large_cap;
    float inv_cell, cell;
};

// monotone map float -> uint32 (and back), so that integer atomicMax orders floats
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ __launch_bounds__(kBlock) void k_query_bounds(const uint32_t* __restrict__ cell_start, uint32_t ncell,
                                                          const float4* __restrict__ sorted_aabb,
                                                          uint32_t* __restrict__ bounds) {
    const uint32_t ns = cell_start[ncell];
    uint32_t k0 = 0u, k1 = 0u, k2 = 0u, k3 = 0u;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < ns; p += gridDim.x * kBlock) {
        const float4 A = sorted_aabb[p];
        k0 = max(k0, ~fkey(A.x)); k1 = max(k1, ~fkey(A.y)); k2 = max(k2, fkey(A.z)); k3 = max(k3, fkey(A.w));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        k0 = max(k0, (uint32_t)__shfl_xor((int)k0, off, 64)); k1 = max(k1, (uint32_t)__shfl_xor((int)k1, off, 64));
        k2 = max(k2, (uint32_t)__shfl_xor((int)k2, off, 64)); k3 = max(k3, (uint32_t)__shfl_xor((int)k3, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && k2) {
        atomicMax(&bounds[0], k0); atomicMax(&bounds[1], k1); atomicMax(&bounds[2], k2); atomicMax(&bounds[3], k3);
    }
}

// clamps to the cell range of the small colliders before converting (rays may be infinite)
__device__ __forceinline__ int32_t cell_clamped(float v_cells, float lo_cells, float hi_cells) {
    const float f = floorf(fminf(fmaxf(v_cells, lo_cells - 2.0f), hi_cells + 2.0f));
    return (int32_t)f;
}

__device__ __forceinline__ uint32_t query_bucket(const QueryIndex& Q, int32_t X, int32_t Y) {
    return (((uint32_t)Y & ((1u << Q.gyl) - 1u)) << Q.gxl) | ((uint32_t)X & ((1u << Q.gxl) - 1u));
}

// one point per lane: the four home cells a small box holding the point can have, then the large list
__global__ __launch_bounds__(kBlock) void k_query_points(uint32_t nq, const float2* __restrict__ pts, uint32_t mask,
                                                          QueryIndex Q, BodyArrays B, uint32_t* __restrict__ body,
                                                          uint32_t* __restrict__ count) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= nq) return;
    const float2 p = pts[k];
    const uint32_t ns = Q.cell_start[Q.ncell], nl = min(Q.ctr->n_large, Q.large_cap);
    uint32_t lowest = SFX_NO_BODY, cnt = 0u;
    auto try_slot = [&](uint32_t slot) {
        const uint32_t i = Q.sorted_id[slot];
        if (!(B.filt[i].x & mask)) return;
        const float4 A = Q.sorted_aabb[slot];
        if (p.x < A.x || p.x > A.z || p.y < A.y || p.y > A.w) return;
        if (!collider_contains(shape_pose(B.pose[i], B.shape[i], B.lpose, i), B.shape[i], B.poly, p.x, p.y)) return;
        lowest = min(lowest, i);
        ++cnt;
    };
    if (ns) {
        const int32_t cx = cell_coord(p.x, Q.inv_cell), cy = cell_coord(p.y, Q.inv_cell);
        for (int32_t Y = cy - 1; Y <= cy; ++Y)
            for (int32_t X = cx - 1; X <= cx; ++X) {
                const uint32_t b = query_bucket(Q, X, Y);
                for (uint32_t s = Q.cell_start[b], e = Q.cell_start[b + 1]; s < e; ++s) {
                    const int2 c = Q.sorted_cell[s];
                    if (c.x == X && c.y == Y) try_slot(s);
                }
            }
    }
    for (uint32_t l = 0; l < nl; ++l) try_slot(ns + l);
    body[k] = lowest == SFX_NO_BODY ? lowest : body_of(B, lowest);   // the body of the lowest collider id
    if (count) count[k] = cnt;
}

// one box per wave: the lanes share the home cells the box reaches, then the large list; hits are
// appended (query << 32 | body) in no particular order, n_hits counts all of them
__global__ __launch_bounds__(kBlock) void k_query_aabbs(uint32_t nq, const float4* __restrict__ boxes, uint32_t mask,
                                                         QueryIndex Q, BodyArrays B, unsigned long long* __restrict__ hits,
                                                         uint32_t cap, uint32_t* __restrict__ n_hits, int name_bodies) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
    const uint32_t ns = Q.cell_start[Q.ncell], nl = min(Q.ctr->n_large, Q.large_cap);
    const uint32_t GX = 1u << Q.gxl, GY = 1u << Q.gyl;
    for (uint32_t k = wave; k < nq; k += nwaves) {
        const float4 Bx = boxes[k];
        auto try_slot = [&](uint32_t slot) {
            const uint32_t i = Q.sorted_id[slot];
            if (!(B.filt[i].x & mask)) return;
            if (!aabb_overlap(Bx, Q.sorted_aabb[slot])) return;
            const uint32_t idx = atomicAdd(n_hits, 1u);
            if (idx < cap) hits[idx] = ((unsigned long long)k << 32) | (name_bodies ? body_of(B, i) : i);
        };
        if (ns && Bx.x <= Bx.z && Bx.y <= Bx.w) {
            const float lox = fkey_inv(~Q.bounds[0]) * Q.inv_cell, hix = fkey_inv(Q.bounds[2]) * Q.inv_cell;
            const float loy = fkey_inv(~Q.bounds[1]) * Q.inv_cell, hiy = fkey_inv(Q.bounds[3]) * Q.inv_cell;
            const int32_t x0 = cell_clamped(Bx.x * Q.inv_cell, lox, hix) - 1, x1 = cell_clamped(Bx.z * Q.inv_cell, lox, hix);
            const int32_t y0 = cell_clamped(Bx.y * Q.inv_cell, loy, hiy) - 1, y1 = cell_clamped(Bx.w * Q.inv_cell, loy, hiy);
            const uint32_t nx = min((uint32_t)(x1 - x0 + 1), GX), ny = min((uint32_t)(y1 - y0 + 1), GY);
            for (uint32_t c = (uint32_t)lane; c < nx * ny; c += 64u) {
                const uint32_t by = c / nx, bx = c - by * nx;
                const uint32_t b = query_bucket(Q, x0 + (int32_t)bx, y0 + (int32_t)by);
                for (uint32_t s = Q.cell_start[b], e = Q.cell_start[b + 1]; s < e; ++s) {
                    const int2 cc = Q.sorted_cell[s];
                    if (cc.x >= x0 && cc.x <= x1 && cc.y >= y0 && cc.y <= y1) try_slot(s);
                }
            }
        }
        for (uint32_t l = (uint32_t)lane; l < nl; l += 64u) try_slot(ns + l);
    }
}

// one ray per lane: the large list, then columns of home cells along the ray's major axis (a
// collider homed in column X lies within [X, X + 1.9) cells of that axis), nearest column first,
// until the best hit lies before the next column
__global__ __launch_bounds__(kBlock) void k_query_rays(uint32_t nq, const float4* __restrict__ rays,
                                                        const float* __restrict__ max_t, uint32_t mask, QueryIndex Q,
                                                        BodyArrays B, sfx_ray_hit* __restrict__ out) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= nq) return;
    const float4 ray = rays[k];
    const uint32_t ns = Q.cell_start[Q.ncell], nl = min(Q.ctr->n_large, Q.large_cap);
    uint32_t bbody = SFX_NO_BODY;
    float bt = 0.0f, bnx = 0.0f, bny = 0.0f;
    float mt = max_t[k];
    if (mt >= 0.0f) {
        mt = fminf(mt, 3.402823466e+38f);
        auto try_slot = [&](uint32_t slot) {
            const uint32_t i = Q.sorted_id[slot];
            if (!(B.filt[i].x & mask)) return;
            if (!ray_box(Q.sorted_aabb[slot], ray.x, ray.y, ray.z, ray.w, mt)) return;
            float t, nx, ny;
            if (!collider_ray(shape_pose(B.pose[i], B.shape[i], B.lpose, i), B.shape[i], B.poly, ray.x, ray.y, ray.z, ray.w, t, nx, ny)) return;
            if (!(t <= mt)) return;
            if (bbody == SFX_NO_BODY || t < bt || (t == bt && i < bbody)) { bbody = i; bt = t; bnx = nx; bny = ny; }
        };
        for (uint32_t l = 0; l < nl; ++l) try_slot(ns + l);
        if (ns) {
            const bool My = fabsf(ray.w) > fabsf(ray.z);
            const float oM = My ? ray.y : ray.x, dM = My ? ray.w : ray.z, om = My ? ray.x : ray.y, dm = My ? ray.z : ray.w;
            const float inv = Q.inv_cell, cell = Q.cell;
            const float blx = fkey_inv(~Q.bounds[0]) * inv, bhx = fkey_inv(Q.bounds[2]) * inv;
            const float bly = fkey_inv(~Q.bounds[1]) * inv, bhy = fkey_inv(Q.bounds[3]) * inv;
            const float bloM = My ? bly : blx, bhiM = My ? bhy : bhx, blom = My ? blx : bly, bhim = My ? bhx : bhy;
            const float a = oM * inv, b = (oM + mt * dM) * inv;
            const int32_t X0 = cell_clamped(fminf(a, b) - 0.05f, bloM, bhiM) - 1;
            const int32_t X1 = cell_clamped(fmaxf(a, b) + 0.05f, bloM, bhiM);
            const int32_t step = dM < 0.0f ? -1 : 1;
            for (int32_t X = step > 0 ? X0 : X1; X >= X0 && X <= X1; X += step) {
                float ta = -INFINITY, tb = INFINITY;
                if (dM != 0.0f) {
                    ta = (((float)X - 0.05f) * cell - oM) / dM;
                    tb = (((float)X + 2.0f) * cell - oM) / dM;
                    if (ta > tb) { const float s = ta; ta = tb; tb = s; }
                }
                if (bbody != SFX_NO_BODY && bt < ta) break;  // nothing further on can be closer
                const float tin = fminf(fmaxf(ta, 0.0f), mt), tout = fmaxf(fminf(tb, mt), 0.0f);
                const float ya = (om + tin * dm) * inv, yb = (om + tout * dm) * inv;
                const int32_t Y0 = cell_clamped(fminf(ya, yb) - 0.05f, blom, bhim) - 1;
                const int32_t Y1 = cell_clamped(fmaxf(ya, yb) + 0.05f, blom, bhim);
                for (int32_t Y = Y0; Y <= Y1; ++Y) {
                    const int32_t hx = My ? Y : X, hy = My ? X : Y;
                    const uint32_t bk = query_bucket(Q, hx, hy);
                    for (uint32_t s = Q.cell_start[bk], e = Q.cell_start[bk + 1]; s < e; ++s) {
                        const int2 c = Q.sorted_cell[s];
                        if (c.x == hx && c.y == hy) try_slot(s);
                    }
                }
            }
        }
    }
    sfx_ray_hit h;
    h.body = bbody == SFX_NO_BODY ? bbody : body_of(B, bbody); h.t = bt; h.normal[0] = bnx; h.normal[1] = bny;
    out[k] = h;
}

}  // namespace sfx
