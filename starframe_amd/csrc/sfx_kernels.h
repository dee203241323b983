// sfx_kernels.h -- the HIP kernels of the physics tick (gfx950, wave64).
//
// K1  k_integrate          substep integration over the body records (streaming)
// K2  k_aabb_count         world AABB + home cell + per-cell histogram (atomics)
// K3b k_scan_*             exclusive prefix scan, block scan in LDS with wave shuffles
// K3c k_scatter / k_finalize_slots   counting sort into cell order, ascending id per cell
// K3d k_pairs<FILL>        candidate pairs, wave ballot compaction, deterministic order
// K5  k_color_*            edge colouring of the candidate-pair graph
// K6  k_solve_pos*         fused narrowphase + XPBD manifold block solve, one launch/colour
// K7  k_derive             velocities from pose deltas
// K8  k_solve_vel*         restitution + friction per manifold, one launch per colour
//
// All of it is HBM/latency-bound gather-scatter work: no MFMA anywhere.
#pragma once

#include "sfx_device.h"

namespace sfx {

constexpr int kBlock = 256;
constexpr int kScanItems = 16;                      // items per thread in the block scan
constexpr int kScanTile = kBlock * kScanItems;      // 4096 items per block
constexpr uint32_t kLeftCap = 4096;                 // leftover edges the serial finish takes
constexpr uint32_t kOvfPairs = 1u, kOvfColors = 2u, kOvfLeftover = 4u, kOvfLarge = 8u;

// device-resident counters of one tick (one allocation, zeroed per tick where noted)
struct Counters {
    uint32_t n_small;       // written by the cell scan
    uint32_t n_large;       // zeroed per tick
    uint32_t n_pairs;       // written after the pair scan (clamped to capacity)
    uint32_t overflow;      // zeroed per tick, kOvf* bits
    uint32_t n_colors;
    uint32_t n_leftover;    // zeroed per tick
    uint32_t n_contacts;    // stats kernel
    uint32_t max_depth_bits;
    float ke;
    uint32_t n_manifolds, n_manifolds2;
    uint32_t pad[5];
    uint32_t color_count[SFX_MAX_COLORS];      // zeroed per tick
    uint32_t color_start[SFX_MAX_COLORS + 1];
    uint32_t color_fill[SFX_MAX_COLORS];       // zeroed per tick
};

struct GridParams {
    float inv_cell;
    float small_lim;   // 0.9 * cell
    uint32_t gxl, gyl; // log2 grid dims
    uint32_t ncell;
};

__device__ __forceinline__ uint32_t bucket_of(int32_t cx, int32_t cy, const GridParams& g) {
    return (((uint32_t)cy & ((1u << g.gyl) - 1u)) << g.gxl) | ((uint32_t)cx & ((1u << g.gxl) - 1u));
}

// ---------------------------------------------------------------- K1 / K7 streaming
__global__ __launch_bounds__(kBlock) void k_integrate(uint32_t n, float4* __restrict__ pose, float4* __restrict__ prev,
                                                      float4* __restrict__ vel, const float2* __restrict__ mass,
                                                      StepParams sp) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 p = pose[i];
    float4 v = vel[i];
    const float2 w = mass[i];
    prev[i] = p;
    if (w.x > 0.0f) {
        v.x = v.x + sp.gx * sp.h;
        v.y = v.y + sp.gy * sp.h;
    }
    if (w.x > 0.0f || w.y > 0.0f) {
        v.x = v.x * sp.damp_lin;
        v.y = v.y * sp.damp_lin;
        v.z = v.z * sp.damp_ang;
        vel[i] = v;
    }
    p.x = p.x + v.x * sp.h;
    p.y = p.y + v.y * sp.h;
    rotate_by(p.z, p.w, v.z * sp.h);
    pose[i] = p;
}

__global__ __launch_bounds__(kBlock) void k_derive(uint32_t n, const float4* __restrict__ pose,
                                                    const float4* __restrict__ prev, float4* __restrict__ vel,
                                                    const float2* __restrict__ mass, StepParams sp) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float2 w = mass[i];
    if (!(w.x > 0.0f || w.y > 0.0f)) return;
    const float4 p = pose[i], q = prev[i];
    float4 v = vel[i];
    v.x = (p.x - q.x) * sp.inv_h;
    v.y = (p.y - q.y) * sp.inv_h;
    float dc = p.z * q.z + p.w * q.w;
    float ds = p.w * q.z - p.z * q.w;
    if (dc > 1e-6f) v.z = (ds / dc) * sp.inv_h;
    vel[i] = v;
}

// ---------------------------------------------------------------- K2 AABB + histogram
__global__ __launch_bounds__(kBlock) void k_aabb_count(uint32_t n, const float4* __restrict__ pose,
                                                        const float4* __restrict__ vel, const float4* __restrict__ shape,
                                                        const float4* __restrict__ poly, float margin, float vel_pad_max,
                                                        float dt, GridParams g, float4* __restrict__ aabb,
                                                        uint32_t* __restrict__ bucket, int2* __restrict__ cellxy,
                                                        uint32_t* __restrict__ cell_count, uint32_t* __restrict__ large_tmp,
                                                        uint32_t large_cap, Counters* __restrict__ ctr) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 sh = shape[i];
    const float4 A = compute_aabb(pose[i], vel[i], sh, poly, margin, vel_pad_max, dt);
    aabb[i] = A;
    uint32_t bk = SFX_NO_BUCKET;
    if (shape_type(sh) != SFX_SHAPE_NONE) {
        if ((A.z - A.x) <= g.small_lim && (A.w - A.y) <= g.small_lim) {
            const int32_t cx = cell_coord(A.x, g.inv_cell), cy = cell_coord(A.y, g.inv_cell);
            bk = bucket_of(cx, cy, g);
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
__global__ __launch_bounds__(kBlock) void k_scan_local(uint32_t* __restrict__ data, uint32_t count,
                                                        uint32_t* __restrict__ sums, uint32_t* __restrict__ copy,
                                                        int single) {
    __shared__ uint32_t lds[kBlock / 64 + 1];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
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
    if (threadIdx.x == 0 && !single) sums[blockIdx.x] = total;
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

// exclusive scan of up to 4096 tile totals by one block of 1024 threads
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t nsums) {
    __shared__ uint32_t lds[1024 / 64 + 1];
    uint32_t v[4];
    const uint32_t base = threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (base + k < nsums) ? sums[base + k] : 0u;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { uint32_t t = v[k]; v[k] = sum; sum += t; }
    uint32_t total;
    const uint32_t pre = block_exclusive_scan<1024>(sum, lds, &total);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < nsums) sums[base + k] = v[k] + pre;
}

__global__ __launch_bounds__(kBlock) void k_scan_add(uint32_t* __restrict__ data, uint32_t count,
                                                      const uint32_t* __restrict__ sums, uint32_t* __restrict__ copy) {
    const uint32_t add = sums[blockIdx.x];
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
                                                            int2* __restrict__ sorted_cell, Counters* __restrict__ ctr) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t ns = cell_start[g.ncell];
    const uint32_t nl = min(ctr->n_large, large_cap);
    if (p == 0) ctr->n_small = ns;
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
    } else if (p - ns < nl) {
        const uint32_t i = large_tmp[p - ns];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < nl; ++q) rank += (large_tmp[q] < i) ? 1u : 0u;
        const uint32_t dst = ns + rank;
        sorted_id[dst] = i;
        sorted_aabb[dst] = aabb[i];
        sorted_cell[dst] = make_int2(0, 0);
    }
}

// ---------------------------------------------------------------- K3d candidate pairs
// Eight lanes share one slot (eight slots per 64-lane wave): the candidates of a slot -- the
// entries of its own cell behind it, the four forward neighbour cells, then every large
// collider -- are spread over the eight lanes, tested, and the survivors compacted with one
// wave-wide __ballot + popcount per step, so a slot's pairs land contiguously and in
// candidate order.  Two passes (count, then fill behind an exclusive scan of the per-slot
// counts) give every pair a fixed place: the list order is a pure function of the slot
// order, with no atomics and no capacity races.
constexpr int kGroup = 8;                 // lanes per slot
constexpr int kSlotsPerWave = 64 / kGroup;

template <bool FILL>
__global__ __launch_bounds__(kBlock) void k_pairs(uint32_t n, GridParams g, const uint32_t* __restrict__ cell_start,
                                                   const uint32_t* __restrict__ sorted_id,
                                                   const float4* __restrict__ sorted_aabb,
                                                   const int2* __restrict__ sorted_cell, const uint4* __restrict__ filt,
                                                   uint32_t* __restrict__ slot_cnt /* count: out, fill: scanned in */,
                                                   uint2* __restrict__ pairs, uint32_t max_pairs,
                                                   const Counters* __restrict__ ctr, uint32_t large_cap) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & (kGroup - 1);          // lane inside the group
    const int gshift = lane & ~(kGroup - 1);     // first lane of the group
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * kBlock) >> 6;
    const uint32_t ns = cell_start[g.ncell];
    const uint32_t nl = min(ctr->n_large, large_cap);
    const uint32_t nact = ns + nl;
    const uint64_t below = (1ull << gl) - 1ull;  // group-local lanes before this one
    // slots [0, n] are covered so that the count array is fully written (n + 1 entries scanned)
    for (uint32_t p0 = wave * kSlotsPerWave; p0 <= n; p0 += nwaves * kSlotsPerWave) {
        const uint32_t p = p0 + (uint32_t)(lane / kGroup);
        const bool live = p < nact;
        uint32_t i = 0;
        float4 Ai = make_float4(0, 0, 0, 0);
        uint4 fi = make_uint4(0, 0, 0, 0);
        int2 c = make_int2(0, 0);
        if (live) { i = sorted_id[p]; Ai = sorted_aabb[p]; fi = filt[i]; c = sorted_cell[p]; }
        const uint32_t base = (FILL && live) ? slot_cnt[p] : 0u;
        uint32_t cnt = 0;
        // six candidate ranges per slot, walked in lockstep by the eight groups of the wave
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            uint32_t qs = 0, qe = 0;
            int32_t ncx = 0, ncy = 0;
            bool check_cell = true;
            if (live && p < ns) {
                if (r == 0) {
                    ncx = c.x; ncy = c.y;
                    qs = p + 1; qe = cell_start[bucket_of(ncx, ncy, g) + 1];
                } else if (r < 5) {
                    const int dx = (r == 1) ? 1 : (r == 2) ? -1 : (r == 3) ? 0 : 1;
                    const int dy = (r == 1) ? 0 : 1;
                    ncx = c.x + dx; ncy = c.y + dy;
                    const uint32_t bk = bucket_of(ncx, ncy, g);
                    qs = cell_start[bk]; qe = cell_start[bk + 1];
                } else {
                    qs = ns; qe = nact; check_cell = false;
                }
            } else if (live && r == 5) {
                qs = p + 1; qe = nact; check_cell = false;   // large slot: later larges only
            }
            for (uint32_t q0 = qs; __any(q0 < qe); q0 += kGroup) {
                const uint32_t q = q0 + (uint32_t)gl;
                bool hit = false;
                uint32_t j = 0;
                if (q < qe) {
                    bool ok = true;
                    if (check_cell) { const int2 cq = sorted_cell[q]; ok = (cq.x == ncx && cq.y == ncy); }
                    if (ok && aabb_overlap(Ai, sorted_aabb[q])) {
                        j = sorted_id[q];
                        hit = pair_filter(fi, filt[j]);
                    }
                }
                const uint64_t bal = (__ballot(hit) >> gshift) & ((1ull << kGroup) - 1ull);
                if (FILL && hit) {
                    const uint32_t dst = base + cnt + (uint32_t)__popcll(bal & below);
                    if (dst < max_pairs) pairs[dst] = make_uint2(min(i, j), max(i, j));
                }
                cnt += (uint32_t)__popcll(bal);
            }
        }
        if (!FILL && gl == 0 && p <= n) slot_cnt[p] = cnt;
    }
}

// after the scan of the slot counts: publish the pair count, flag capacity overflow
__global__ void k_pairs_finish(uint32_t n, const uint32_t* __restrict__ slot_off, uint32_t max_pairs,
                               Counters* __restrict__ ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t total = slot_off[n];
        if (total > max_pairs) { atomicOr(&ctr->overflow, kOvfPairs); total = max_pairs; }
        ctr->n_pairs = total;
    }
}

// ---------------------------------------------------------------- K5 colouring
// DESIGN.md "Contract / colouring"; the CPU oracle restates the same rounds sequentially.
struct ColorArrays {
    const uint2* pairs;            // (a, b), SFX_WORLD_BODY allowed (joints)
    const uint4* filt;             // .w = dynamic
    unsigned long long* used;      // per body: colours taken
    uint32_t* deg;                 // per body: uncoloured incident edges
    unsigned long long* claim;     // per body x 8: lowest priority seen
    uint8_t* color;                // per edge
    uint8_t* prop;                 // per edge: proposal of this round
    uint32_t* left_list;           // uncoloured edges after the last round
    const uint32_t* n_edges;       // device-side edge count
    Counters* ctr;
};

__device__ __forceinline__ bool edge_dyn(const ColorArrays& A, uint32_t body) {
    return body != SFX_WORLD_BODY && A.filt[body].w != 0u;
}

__global__ __launch_bounds__(kBlock) void k_color_init(ColorArrays A) {
    const uint32_t ne = *A.n_edges;
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < ne; e += gridDim.x * kBlock) {
        const uint2 ab = A.pairs[e];
        A.color[e] = SFX_NO_COLOR;
        if (edge_dyn(A, ab.x)) atomicAdd(&A.deg[ab.x], 1u);
        if (edge_dyn(A, ab.y)) atomicAdd(&A.deg[ab.y], 1u);
    }
}

__device__ __forceinline__ unsigned long long edge_prio(uint32_t e, uint32_t r, uint32_t& h) {
    h = mix32(e * 0x9E3779B1u + r * 0x7F4A7C15u + 0x165667B1u);
    return ((unsigned long long)(63u - r) << 58) | ((unsigned long long)(h >> 6) << 32) | e;
}

__global__ __launch_bounds__(kBlock) void k_color_propose(ColorArrays A, uint32_t r) {
    const uint32_t ne = *A.n_edges;
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < ne; e += gridDim.x * kBlock) {
        if (A.color[e] != SFX_NO_COLOR) continue;
        const uint2 ab = A.pairs[e];
        const bool da = edge_dyn(A, ab.x), db = edge_dyn(A, ab.y);
        const unsigned long long ma = da ? A.used[ab.x] : 0ull, mb = db ? A.used[ab.y] : 0ull;
        const unsigned long long freem = ~(ma | mb);
        if (freem == 0ull) { A.prop[e] = SFX_NO_COLOR; atomicOr(&A.ctr->overflow, kOvfColors); continue; }
        uint32_t dg = 1;
        if (da) dg = max(dg, A.deg[ab.x]);
        if (db) dg = max(dg, A.deg[ab.y]);
        dg = min(dg, 8u);
        dg = min(dg, (uint32_t)__popcll(freem));
        uint32_t h;
        const unsigned long long p = edge_prio(e, r, h);
        const uint32_t k = mix32(h ^ 0xA5A5A5A5u) % dg;
        const uint32_t c = kth_set_bit(freem, k);
        A.prop[e] = (uint8_t)c;
        if (da) atomicMin(&A.claim[(size_t)ab.x * 8 + (c & 7u)], p);
        if (db) atomicMin(&A.claim[(size_t)ab.y * 8 + (c & 7u)], p);
    }
}

__global__ __launch_bounds__(kBlock) void k_color_resolve(ColorArrays A, uint32_t r, int last) {
    const uint32_t ne = *A.n_edges;
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < ne; e += gridDim.x * kBlock) {
        if (A.color[e] != SFX_NO_COLOR) continue;
        const uint32_t c = A.prop[e];
        bool win = false;
        if (c != SFX_NO_COLOR) {
            const uint2 ab = A.pairs[e];
            const bool da = edge_dyn(A, ab.x), db = edge_dyn(A, ab.y);
            uint32_t h;
            const unsigned long long p = edge_prio(e, r, h);
            win = (!da || A.claim[(size_t)ab.x * 8 + (c & 7u)] == p) &&
                  (!db || A.claim[(size_t)ab.y * 8 + (c & 7u)] == p);
            if (win) {
                A.color[e] = (uint8_t)c;
                if (da) { atomicOr(&A.used[ab.x], 1ull << c); atomicSub(&A.deg[ab.x], 1u); }
                if (db) { atomicOr(&A.used[ab.y], 1ull << c); atomicSub(&A.deg[ab.y], 1u); }
            }
        }
        if (!win && last) {
            const uint32_t l = atomicAdd(&A.ctr->n_leftover, 1u);
            if (l < kLeftCap) A.left_list[l] = e;
        }
    }
}

// serial first-fit finish over the (few) edges the rounds left uncoloured, ascending edge id
__global__ __launch_bounds__(kBlock) void k_color_finish(ColorArrays A) {
    __shared__ uint32_t raw[kLeftCap];
    __shared__ uint32_t srt[kLeftCap];
    uint32_t nl = A.ctr->n_leftover;
    if (nl == 0) return;
    if (nl > kLeftCap) {
        if (threadIdx.x == 0) atomicOr(&A.ctr->overflow, kOvfLeftover);
        nl = kLeftCap;
    }
    for (uint32_t t = threadIdx.x; t < nl; t += kBlock) raw[t] = A.left_list[t];
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nl; t += kBlock) {
        const uint32_t mine = raw[t];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < nl; ++q) rank += (raw[q] < mine) ? 1u : 0u;
        srt[rank] = mine;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t t = 0; t < nl; ++t) {
            const uint32_t e = srt[t];
            const uint2 ab = A.pairs[e];
            const bool da = edge_dyn(A, ab.x), db = edge_dyn(A, ab.y);
            const unsigned long long freem = ~((da ? A.used[ab.x] : 0ull) | (db ? A.used[ab.y] : 0ull));
            if (freem == 0ull) { atomicOr(&A.ctr->overflow, kOvfColors); continue; }  // dropped from the solve
            const uint32_t c = (uint32_t)__builtin_ctzll(freem);
            A.color[e] = (uint8_t)c;
            if (da) A.used[ab.x] |= 1ull << c;
            if (db) A.used[ab.y] |= 1ull << c;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_color_hist(ColorArrays A) {
    __shared__ uint32_t hist[SFX_MAX_COLORS];
    if (threadIdx.x < SFX_MAX_COLORS) hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t ne = *A.n_edges;
    for (uint32_t e = blockIdx.x * kBlock + threadIdx.x; e < ne; e += gridDim.x * kBlock) {
        const uint32_t c = A.color[e];
        if (c < SFX_MAX_COLORS) atomicAdd(&hist[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < SFX_MAX_COLORS && hist[threadIdx.x]) atomicAdd(&A.ctr->color_count[threadIdx.x], hist[threadIdx.x]);
}

__global__ void k_color_scan(Counters* __restrict__ ctr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t run = 0, nc = 0;
    for (uint32_t c = 0; c < SFX_MAX_COLORS; ++c) {
        ctr->color_start[c] = run;
        const uint32_t k = ctr->color_count[c];
        if (k) nc = c + 1;
        run += k;
    }
    ctr->color_start[SFX_MAX_COLORS] = run;
    ctr->n_colors = nc;
}

// colour-major reorder.  A block reserves one contiguous run per colour (LDS histogram, one
// global atomic per colour and block), so pairs that were neighbours in the spatially sorted
// list stay neighbours inside their colour: the per-colour gathers keep their L2 locality.
__global__ __launch_bounds__(kBlock) void k_color_scatter(ColorArrays A, uint2* __restrict__ cpairs) {
    __shared__ uint32_t cnt[SFX_MAX_COLORS];
    __shared__ uint32_t base[SFX_MAX_COLORS];
    const uint32_t ne = *A.n_edges;
    const uint32_t ntiles = (ne + kBlock - 1) / kBlock;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x < SFX_MAX_COLORS) cnt[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t e = tile * kBlock + threadIdx.x;
        uint32_t c = SFX_NO_COLOR, rank = 0;
        uint2 ab = make_uint2(0, 0);
        if (e < ne) {
            c = A.color[e];
            ab = A.pairs[e];
            if (c < SFX_MAX_COLORS) rank = atomicAdd(&cnt[c], 1u);
        }
        __syncthreads();
        if (threadIdx.x < SFX_MAX_COLORS) {
            const uint32_t k = cnt[threadIdx.x];
            base[threadIdx.x] = k ? A.ctr->color_start[threadIdx.x] + atomicAdd(&A.ctr->color_fill[threadIdx.x], k) : 0u;
        }
        __syncthreads();
        if (e < ne && c < SFX_MAX_COLORS) cpairs[base[c] + rank] = ab;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- K6 / K8 solves
// One launch per colour: inside a colour no dynamic body appears twice, so every thread
// owns the bodies of its pair.  Colours >= first_tail share one single-workgroup kernel that
// walks them in order with a workgroup barrier in between (the tail colours are tiny).
__global__ __launch_bounds__(kBlock) void k_solve_pos(uint32_t color, const Counters* __restrict__ ctr,
                                                       const uint2* __restrict__ cpairs, BodyArrays B, ContactArrays C,
                                                       float max_corr) {
    const uint32_t start = ctr->color_start[color], cnt = ctr->color_count[color];
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < cnt; k += gridDim.x * kBlock) {
        const uint2 ab = cpairs[start + k];
        solve_pos_pair(B, C, start + k, ab.x, ab.y, max_corr);
    }
}

__global__ __launch_bounds__(kBlock) void k_solve_pos_tail(uint32_t first, const Counters* __restrict__ ctr,
                                                          const uint2* __restrict__ cpairs, BodyArrays B, ContactArrays C,
                                                          float max_corr) {
    const uint32_t nc = ctr->n_colors;
    for (uint32_t color = first; color < nc; ++color) {
        const uint32_t start = ctr->color_start[color], cnt = ctr->color_count[color];
        for (uint32_t k = threadIdx.x; k < cnt; k += kBlock) {
            const uint2 ab = cpairs[start + k];
            solve_pos_pair(B, C, start + k, ab.x, ab.y, max_corr);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void k_solve_vel(uint32_t color, const Counters* __restrict__ ctr,
                                                       const uint2* __restrict__ cpairs, BodyArrays B, ContactArrays C,
                                                       StepParams sp) {
    const uint32_t start = ctr->color_start[color], cnt = ctr->color_count[color];
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < cnt; k += gridDim.x * kBlock) {
        const uint2 ab = cpairs[start + k];
        solve_vel_pair(B, C, start + k, ab.x, ab.y, sp);
    }
}

__global__ __launch_bounds__(kBlock) void k_solve_vel_tail(uint32_t first, const Counters* __restrict__ ctr,
                                                          const uint2* __restrict__ cpairs, BodyArrays B, ContactArrays C,
                                                          StepParams sp) {
    const uint32_t nc = ctr->n_colors;
    for (uint32_t color = first; color < nc; ++color) {
        const uint32_t start = ctr->color_start[color], cnt = ctr->color_count[color];
        for (uint32_t k = threadIdx.x; k < cnt; k += kBlock) {
            const uint2 ab = cpairs[start + k];
            solve_vel_pair(B, C, start + k, ab.x, ab.y, sp);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void k_solve_joints(uint32_t start, uint32_t cnt, BodyArrays B, JointArrays J,
                                                          float inv_h2) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < cnt) solve_joint(B, J, start + k, inv_h2);
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

// deepest penetration at the current poses and the number of active contact points of the
// last substep, over all candidate pairs
__global__ __launch_bounds__(kBlock) void k_stats_pairs(const Counters* __restrict__ cin, Counters* __restrict__ ctr,
                                                         const uint2* __restrict__ cpairs, BodyArrays B, ContactArrays C) {
    const uint32_t np = cin->color_start[SFX_MAX_COLORS];  // coloured pairs (all, unless colours overflowed)
    uint32_t contacts = 0, mf = 0, mf2 = 0;
    float md = 0.0f;
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < np; k += gridDim.x * kBlock) {
        const uint2 ab = cpairs[k];
        Manifold m;
        manifold(B.pose[ab.x], B.shape[ab.x], B.pose[ab.y], B.shape[ab.y], B.poly, m);
        for (int q = 0; q < m.count; ++q) md = fmaxf(md, m.depth[q]);
        const uint32_t cnt = __float_as_uint(C.misc[k].y);
        if (cnt) {
            mf += 1u;
            mf2 += (cnt == 2) ? 1u : 0u;
            const float4 lv = C.lamv[k];
            contacts += (lv.x > 0.0f) ? 1u : 0u;
            contacts += (cnt == 2 && lv.y > 0.0f) ? 1u : 0u;
        }
    }
    if (contacts) atomicAdd(&ctr->n_contacts, contacts);
    if (mf) atomicAdd(&ctr->n_manifolds, mf);
    if (mf2) atomicAdd(&ctr->n_manifolds2, mf2);
    if (md > 0.0f) atomicMax(&ctr->max_depth_bits, __float_as_uint(md));
}

__global__ __launch_bounds__(kBlock) void k_contacts_debug(uint32_t np, const uint2* __restrict__ pairs, BodyArrays B,
                                                            sfx_contact* __restrict__ out) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= np) return;
    const uint2 ab = pairs[k];
    Manifold m;
    manifold(B.pose[ab.x], B.shape[ab.x], B.pose[ab.y], B.shape[ab.y], B.poly, m);
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

}  // namespace sfx
