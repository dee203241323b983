// sfx_device.h -- device-side math, narrowphase and per-manifold XPBD solvers (gfx950).
//
// Written for the wave64 / HBM-bound regime of MI355X: every routine here works on
// registers only; the kernels in sfx_kernels.h do the (16-byte, lane-contiguous where the
// data layout allows) loads and stores around them.  Arithmetic follows DESIGN.md
// "Contract" operation by operation and the library is built with -ffp-contract=off, so
// the CPU oracle (oracle/sfx_oracle.c, an independent restatement) can be compared
// bit-for-bit where the contract demands exactness.
//
// Paths restated from: Mueller et al. 2020 (XPBD rigid bodies) Alg. 2, sec. 3.3-3.6; the
// reference implementation (starframe, Rust) is not available -- see DESIGN.md.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sfx.h"

#define SFX_EPS_LEN 1e-9f
#define SFX_NO_COLOR 0xFFu
#define SFX_PRUNED_COLOR 0xFEu  // candidate pair dropped by the reach test: never solved this tick
#define SFX_NO_BUCKET 0xFFFFFFFFu

namespace sfx {

// ---- body state records (all 16-byte aligned, one record per body, SoA of records) ----
// pose  : float4 (x, y, cos, sin)        prev : the pose at the start of the substep
// vel   : float4 (vx, vy, omega, 0)      mass : float2 (inv_mass, inv_inertia)
// shape : float4 (bits: type | material << 8, a, b, r); POLY keeps offset/count in a/b bits
// filt  : uint4  (category, mask, group, dynamic)

struct BodyArrays {
    float4* pose;
    float4* prev;
    float4* vel;
    float4* dpos;        // (dx, dy, dtheta, -): what the position sweep of this substep has accumulated per body (DESIGN.md
                         // "Contract / frozen rows"); zeroed by the integration, applied once where the velocity is derived
    const float2* mass;
    const float4* shape;
    const float4* lpose; // local pose of the shape in the body frame (ox, oy, cos, sin); read only where shape_local() says so
    const uint4* filt;
    const float4* poly;  // (vx, vy, nx, ny) per pool vertex
    const float4* mats;  // (restitution, mu_s, mu_d, 0)
    // Several colliders per body (DESIGN.md "Contract / several colliders per body"): every array above has one element
    // per ENTRY -- the nb bodies first (entry k < nb = body k with its own collider), then one entry per further
    // collider, which MIRRORS the pose and velocity of its body (k_mirror) and has no mass.
    const uint32_t* cbody;  // [entries - nb] body of entry nb + k
    uint32_t nb;
};

// the body an entry belongs to (solves write bodies, never mirrors)
__device__ __forceinline__ uint32_t body_of(const BodyArrays& B, uint32_t e) { return e < B.nb ? e : B.cbody[e - B.nb]; }

// filt.w: bit 0 the entry's body is dynamic; bit 1 a further collider of a dynamic body (its pairs go to the fallback
// list: they are solved on the body, through body_of); bits 8.. the body id (colliders of one body never pair)
constexpr uint32_t kFiltDynamic = 1u, kFiltFurther = 2u, kFiltHalo = 4u;   // kFiltHalo: the body is on a slab halo list (sent or received)

// shape.x bits: type (8) | material (23) << 8 | has a local pose << 31
__device__ __forceinline__ uint32_t shape_type(const float4& s) { return __float_as_uint(s.x) & 0xFFu; }
__device__ __forceinline__ uint32_t shape_material(const float4& s) { return (__float_as_uint(s.x) >> 8) & 0x7FFFFFu; }
__device__ __forceinline__ bool shape_local(const float4& s) { return (__float_as_uint(s.x) >> 31) != 0u; }

// DESIGN.md "Contract / collider pose": where the shape is = body pose o local pose.  A shape without a local pose has
// the body's pose bit for bit (callers branch on shape_local and do not even load lp then).
__device__ __forceinline__ float4 collider_pose(const float4& pose, const float4& lp) {
    return make_float4(pose.x + (pose.z * lp.x - pose.w * lp.y), pose.y + (pose.w * lp.x + pose.z * lp.y),
                       pose.z * lp.z - pose.w * lp.w, pose.w * lp.z + pose.z * lp.w);
}

__device__ __forceinline__ float4 shape_pose(const float4& pose, const float4& shape, const float4* __restrict__ lpose, uint32_t i) {
    return shape_local(shape) ? collider_pose(pose, lpose[i]) : pose;
}

// how far the collider may travel per axis before the next broadphase: the body's speed plus, for a shape off the body
// origin, what the spin adds at that distance (lp is read only for such shapes)
__device__ __forceinline__ void motion_pads(const float4& vel, const float4& shape, const float4* __restrict__ lpose, uint32_t i,
                                            float margin, float vel_pad_max, float dt, float& padx, float& pady) {
    float sx = fabsf(vel.x), sy = fabsf(vel.y);
    if (shape_local(shape)) {
        const float4 lp = lpose[i];
        const float swing = fabsf(vel.z) * sqrtf(lp.x * lp.x + lp.y * lp.y);
        sx = sx + swing; sy = sy + swing;
    }
    padx = margin + fminf(sx * dt, vel_pad_max);
    pady = margin + fminf(sy * dt, vel_pad_max);
}

__device__ __forceinline__ void rotate_by(float& c, float& s, float dth) {
    float nc = c - dth * s;
    float ns = s + dth * c;
    float inv = 1.0f / sqrtf(nc * nc + ns * ns);
    c = nc * inv;
    s = ns * inv;
}

// ---- AABB (DESIGN.md "Contract / AABB") --------------------------------------------
__device__ __forceinline__ float4 compute_aabb(const float4& pose /* of the shape */, const float4& shape,
                                               const float4* __restrict__ poly, float padx, float pady) {
    const float px = pose.x, py = pose.y, rc = pose.z, rs = pose.w;
    float minx, maxx, miny, maxy;
    const uint32_t type = shape_type(shape);
    if (type == SFX_SHAPE_CIRCLE) {
        minx = -shape.w; maxx = shape.w; miny = -shape.w; maxy = shape.w;
    } else if (type == SFX_SHAPE_RECT) {
        float ex = (fabsf(rc * shape.y) + fabsf(rs * shape.z)) + shape.w;
        float ey = (fabsf(rs * shape.y) + fabsf(rc * shape.z)) + shape.w;
        minx = -ex; maxx = ex; miny = -ey; maxy = ey;
    } else if (type == SFX_SHAPE_CAPSULE) {
        float ex = fabsf(rc * shape.y) + shape.w;
        float ey = fabsf(rs * shape.y) + shape.w;
        minx = -ex; maxx = ex; miny = -ey; maxy = ey;
    } else if (type == SFX_SHAPE_POLY) {
        const uint32_t off = __float_as_uint(shape.y), cnt = __float_as_uint(shape.z);
        minx = miny = INFINITY; maxx = maxy = -INFINITY;
        for (uint32_t k = 0; k < cnt; ++k) {
            float4 v = poly[off + k];
            float wx = rc * v.x - rs * v.y, wy = rs * v.x + rc * v.y;
            minx = fminf(minx, wx); maxx = fmaxf(maxx, wx);
            miny = fminf(miny, wy); maxy = fmaxf(maxy, wy);
        }
        minx -= shape.w; maxx += shape.w; miny -= shape.w; maxy += shape.w;
    } else {
        return make_float4(INFINITY, INFINITY, -INFINITY, -INFINITY);
    }
    return make_float4((px + minx) - padx, (py + miny) - pady, (px + maxx) + padx, (py + maxy) + pady);
}

__device__ __forceinline__ bool aabb_overlap(const float4& A, const float4& B) {
    return A.x <= B.z && B.x <= A.z && A.y <= B.w && B.y <= A.w;
}

// ---- reach test (DESIGN.md "Contract / solver pairs") --------------------------------
// A gap along any axis is a lower bound of the distance between two convex shapes; a pair
// whose gap exceeds what the two bodies may move in the tick (the AABB pads) plus the
// speculative margin cannot touch before the next broadphase.
__device__ __forceinline__ void support_extents(const float4& pose, const float4& shape, const float4* __restrict__ poly,
                                                float nx, float ny, float& hp, float& hm) {
    const float rc = pose.z, rs = pose.w;
    const uint32_t type = shape_type(shape);
    if (type == SFX_SHAPE_CIRCLE) {
        hp = hm = shape.w;
    } else if (type == SFX_SHAPE_RECT) {
        const float ax = nx * rc + ny * rs, ay = ny * rc - nx * rs;
        hp = hm = (fabsf(ax) * shape.y + fabsf(ay) * shape.z) + shape.w;
    } else if (type == SFX_SHAPE_CAPSULE) {
        const float ax = nx * rc + ny * rs;
        hp = hm = fabsf(ax) * shape.y + shape.w;
    } else if (type == SFX_SHAPE_POLY) {
        const uint32_t off = __float_as_uint(shape.y), cnt = __float_as_uint(shape.z);
        float mx = -INFINITY, mn = INFINITY;
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4 v = poly[off + k];
            const float wx = rc * v.x - rs * v.y, wy = rs * v.x + rc * v.y;
            const float pr = nx * wx + ny * wy;
            mx = fmaxf(mx, pr); mn = fminf(mn, pr);
        }
        hp = mx + shape.w; hm = shape.w - mn;
    } else {
        hp = hm = INFINITY;
    }
}

__device__ __forceinline__ float axis_gap(const float4& pa, const float4& sa, const float4& pb, const float4& sb,
                                          const float4* __restrict__ poly, float nx, float ny, float dx, float dy) {
    float hap, ham, hbp, hbm;
    support_extents(pa, sa, poly, nx, ny, hap, ham);
    support_extents(pb, sb, poly, nx, ny, hbp, hbm);
    const float t = nx * dx + ny * dy;
    return t >= 0.0f ? t - (hap + hbm) : (-t) - (ham + hbp);
}

// pa, pb: the SHAPES' poses; ra, rb: the two colliders' reaches, sqrt(padx^2 + pady^2) of motion_pads
__device__ __forceinline__ bool pair_in_reach(const float4& pa, const float4& sa, float ra, const float4& pb,
                                              const float4& sb, float rb, const float4* __restrict__ poly, float sm) {
    const float dx = pb.x - pa.x, dy = pb.y - pa.y;
    const float d2 = dx * dx + dy * dy;
    float gap = -INFINITY;
    if (d2 > 1e-12f) {
        const float d = sqrtf(d2);
        gap = axis_gap(pa, sa, pb, sb, poly, dx / d, dy / d, dx, dy);
    }
    const uint32_t ta = shape_type(sa), tb = shape_type(sb);
    if (ta == SFX_SHAPE_RECT || ta == SFX_SHAPE_CAPSULE) {
        gap = fmaxf(gap, axis_gap(pa, sa, pb, sb, poly, pa.z, pa.w, dx, dy));
        gap = fmaxf(gap, axis_gap(pa, sa, pb, sb, poly, -pa.w, pa.z, dx, dy));
    }
    if (tb == SFX_SHAPE_RECT || tb == SFX_SHAPE_CAPSULE) {
        gap = fmaxf(gap, axis_gap(pa, sa, pb, sb, poly, pb.z, pb.w, dx, dy));
        gap = fmaxf(gap, axis_gap(pa, sa, pb, sb, poly, -pb.w, pb.z, dx, dy));
    }
    const float thr = (ra + rb) + sm;
    return !(gap > thr);
}

__device__ __forceinline__ bool pair_filter(const uint4& a, const uint4& b) {
    if (!((a.w | b.w) & kFiltDynamic)) return false;
    if ((a.w >> 8) == (b.w >> 8)) return false;   // two colliders of one body
    if (!(a.x & b.y) || !(b.x & a.y)) return false;
    if (a.z != 0 && a.z == b.z) return false;
    return true;
}

__device__ __forceinline__ int32_t cell_coord(float lo, float inv_cell) {
    float f = floorf(lo * inv_cell);
    if (!(f > -1073741824.0f)) f = -1073741824.0f;
    if (f > 1073741824.0f) f = 1073741824.0f;
    return (int32_t)f;
}

// ---- colouring hash (DESIGN.md "Contract / colouring") ------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t edge_hash(uint32_t u, uint32_t v, uint32_t r) {
    return mix32((u * 0x9E3779B1u) ^ mix32(v + r * 0x7F4A7C15u + 0x165667B1u));
}

// 6 bits of (63 - round) | 10 bits of hash | 24 + 24 bits of key: unique per edge and round,
// independent of where the edge sits in any list
__device__ __forceinline__ unsigned long long edge_prio(uint32_t u, uint32_t v, uint32_t r, uint32_t h) {
    return ((unsigned long long)(63u - r) << 58) | ((unsigned long long)(h >> 22) << 48) |
           ((unsigned long long)(u & 0xFFFFFFu) << 24) | (unsigned long long)(v & 0xFFFFFFu);
}

__device__ __forceinline__ uint32_t kth_set_bit(uint64_t m, uint32_t k) {
    for (uint32_t i = 0; i < k; ++i) m &= m - 1;
    return (uint32_t)__builtin_ctzll(m);
}

// ---- narrowphase ---------------------------------------------------------------------
// Every shape is a convex core polygon of 1..MAXV vertices plus a radius.  Arrays are only
// ever indexed by unrolled loop counters so they live in VGPRs (runtime-indexed arrays
// would go to scratch memory).

template <int MAXV>
struct Core {
    int n;
    float vx[MAXV], vy[MAXV], nx[MAXV], ny[MAXV];
    float r;
};

struct Manifold {
    int count;
    float nx, ny;
    float p1x[2], p1y[2], p2x[2], p2y[2], depth[2];
};

template <int N>
__device__ __forceinline__ float pick(const float (&a)[N], int k) {
    float v = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) v = (i == k) ? a[i] : v;
    return v;
}

template <int MAXV>
__device__ __forceinline__ void build_core(const float4& pose, const float4& shape,
                                           const float4* __restrict__ poly, Core<MAXV>& c) {
    const float px = pose.x, py = pose.y, rc = pose.z, rs = pose.w;
    const uint32_t type = shape_type(shape);
    c.r = shape.w;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) { c.vx[k] = 0.0f; c.vy[k] = 0.0f; c.nx[k] = 0.0f; c.ny[k] = 0.0f; }
    if (type == SFX_SHAPE_CIRCLE) {
        c.n = 1; c.vx[0] = px; c.vy[0] = py;
    } else if (type == SFX_SHAPE_RECT) {
        float ax = rc * shape.y, ay = rs * shape.y;
        float bx = -rs * shape.z, by = rc * shape.z;
        c.n = 4;
        c.vx[0] = (px - ax) - bx; c.vy[0] = (py - ay) - by;
        c.vx[1] = (px + ax) - bx; c.vy[1] = (py + ay) - by;
        c.vx[2] = (px + ax) + bx; c.vy[2] = (py + ay) + by;
        c.vx[3] = (px - ax) + bx; c.vy[3] = (py - ay) + by;
        c.nx[0] = rs;  c.ny[0] = -rc;
        c.nx[1] = rc;  c.ny[1] = rs;
        c.nx[2] = -rs; c.ny[2] = rc;
        c.nx[3] = -rc; c.ny[3] = -rs;
    } else if (type == SFX_SHAPE_CAPSULE) {
        float ax = rc * shape.y, ay = rs * shape.y;
        c.n = 2;
        c.vx[0] = px - ax; c.vy[0] = py - ay;
        c.vx[1] = px + ax; c.vy[1] = py + ay;
        c.nx[0] = rs;  c.ny[0] = -rc;
        c.nx[1] = -rs; c.ny[1] = rc;
    } else if (type == SFX_SHAPE_POLY) {
        if constexpr (MAXV >= 8) {
            const uint32_t off = __float_as_uint(shape.y);
            c.n = (int)__float_as_uint(shape.z);
#pragma unroll
            for (int k = 0; k < MAXV; ++k) {
                if (k < c.n) {
                    float4 v = poly[off + k];
                    c.vx[k] = px + (rc * v.x - rs * v.y); c.vy[k] = py + (rs * v.x + rc * v.y);
                    c.nx[k] = rc * v.z - rs * v.w;        c.ny[k] = rs * v.z + rc * v.w;
                }
            }
        } else {
            c.n = 0;
        }
    } else {
        c.n = 0;
    }
}

__device__ __forceinline__ void cc_manifold(float ax, float ay, float ar, float bx, float by, float br, float sm, Manifold& m) {
    float dx = bx - ax, dy = by - ay;
    float d2 = dx * dx + dy * dy;
    float rsum = ar + br;
    float reach = rsum + sm;
    m.count = 0;
    if (d2 > reach * reach) return;
    float dist = sqrtf(d2);
    float nx = 0.0f, ny = 1.0f;
    if (dist > SFX_EPS_LEN) { nx = dx / dist; ny = dy / dist; }
    float depth = rsum - dist;
    if (!(depth > -sm)) return;
    m.count = 1; m.nx = nx; m.ny = ny; m.depth[0] = depth;
    m.p1x[0] = ax + nx * ar; m.p1y[0] = ay + ny * ar;
    m.p2x[0] = bx - nx * br; m.p2y[0] = by - ny * br;
}

// closest point on the boundary of P (n >= 2) to (cx, cy); first minimum wins
template <int MAXV>
__device__ __forceinline__ void closest_on_poly(const Core<MAXV>& P, float cx, float cy, float& qx, float& qy, float& d2out) {
    float best = INFINITY, bx = 0.0f, by = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (k < P.n) {
            const bool wrap = (k + 1 == P.n) || (k + 1 == MAXV);
            float x1 = wrap ? P.vx[0] : P.vx[(k + 1) % MAXV];
            float y1 = wrap ? P.vy[0] : P.vy[(k + 1) % MAXV];
            float ex = x1 - P.vx[k], ey = y1 - P.vy[k];
            float wx = cx - P.vx[k], wy = cy - P.vy[k];
            float ee = ex * ex + ey * ey;
            float t = 0.0f;
            if (ee > 0.0f) t = (wx * ex + wy * ey) / ee;
            t = fminf(fmaxf(t, 0.0f), 1.0f);
            float x = P.vx[k] + t * ex, y = P.vy[k] + t * ey;
            float ddx = cx - x, ddy = cy - y;
            float d2 = ddx * ddx + ddy * ddy;
            if (d2 < best) { best = d2; bx = x; by = y; }
        }
    }
    qx = bx; qy = by; d2out = best;
}

// polygon P (n >= 2) against a circle: normal from P to the circle
template <int MAXV>
__device__ __forceinline__ void pc_manifold(const Core<MAXV>& P, float cx, float cy, float cr, float sm, Manifold& m) {
    float rsum = P.r + cr;
    float reach = rsum + sm;
    float sep = -INFINITY, snx = 0.0f, sny = 0.0f;
    m.count = 0;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (k < P.n) {
            float s = P.nx[k] * (cx - P.vx[k]) + P.ny[k] * (cy - P.vy[k]);
            if (s > sep) { sep = s; snx = P.nx[k]; sny = P.ny[k]; }
        }
    }
    if (sep > reach) return;
    float nx, ny, qx, qy, depth;
    if (P.n == 2 || sep > 0.0f) {
        float d2;
        closest_on_poly<MAXV>(P, cx, cy, qx, qy, d2);
        if (d2 > reach * reach) return;
        float dist = sqrtf(d2);
        if (dist > SFX_EPS_LEN) { nx = (cx - qx) / dist; ny = (cy - qy) / dist; }
        else { nx = snx; ny = sny; }
        depth = rsum - dist;
    } else {
        nx = snx; ny = sny;
        qx = cx - nx * sep; qy = cy - ny * sep;
        depth = rsum - sep;
    }
    if (!(depth > -sm)) return;
    m.count = 1; m.nx = nx; m.ny = ny; m.depth[0] = depth;
    m.p1x[0] = qx + nx * P.r; m.p1y[0] = qy + ny * P.r;
    m.p2x[0] = cx - nx * cr;  m.p2y[0] = cy - ny * cr;
}

// Largest separation of Q from the faces of P (first maximum wins).  The winning face is
// returned by value (normal, its two end points), never as an index: an index would have to be
// looked up in the register arrays again, which the compiler turns into scratch memory.
template <int MAXV>
__device__ __forceinline__ float sat_sep(const Core<MAXV>& P, const Core<MAXV>& Q, float& fnx, float& fny,
                                         float& f1x, float& f1y, float& f2x, float& f2y) {
    float best = -INFINITY;
    fnx = 0.0f; fny = 0.0f; f1x = 0.0f; f1y = 0.0f; f2x = 0.0f; f2y = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (k < P.n) {
            float s = INFINITY;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                if (j < Q.n) {
                    float d = P.nx[k] * (Q.vx[j] - P.vx[k]) + P.ny[k] * (Q.vy[j] - P.vy[k]);
                    if (d < s) s = d;
                }
            }
            const bool wrap = (k + 1 == P.n) || (k + 1 == MAXV);
            const float x1 = wrap ? P.vx[0] : P.vx[(k + 1) % MAXV];
            const float y1 = wrap ? P.vy[0] : P.vy[(k + 1) % MAXV];
            if (s > best) {
                best = s;
                fnx = P.nx[k]; fny = P.ny[k];
                f1x = P.vx[k]; f1y = P.vy[k];
                f2x = x1; f2y = y1;
            }
        }
    }
    return best;
}

// polygon against polygon (both n >= 2); normal from A to B
template <int MAXV>
__device__ __forceinline__ void pp_manifold(const Core<MAXV>& A, const Core<MAXV>& B, float sm, Manifold& m) {
    const float rsum = A.r + B.r;
    const float reach = rsum + sm;
    m.count = 0;
    float anx, any, a1x, a1y, a2x, a2y, bnx, bny, b1x, b1y, b2x, b2y;
    float sa = sat_sep<MAXV>(A, B, anx, any, a1x, a1y, a2x, a2y);
    if (sa > reach) return;
    float sb = sat_sep<MAXV>(B, A, bnx, bny, b1x, b1y, b2x, b2y);
    if (sb > reach) return;
    const bool flip = sb > sa;
    // reference face (by value) and incident polygon I, selected element-wise
    const float nrx = flip ? bnx : anx, nry = flip ? bny : any;
    const float r1x = flip ? b1x : a1x, r1y = flip ? b1y : a1y, r2x = flip ? b2x : a2x, r2y = flip ? b2y : a2y;
    Core<MAXV> I;
    I.n = flip ? A.n : B.n;
    I.r = flip ? A.r : B.r;
    const float rr_sel = flip ? B.r : A.r;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        I.vx[k] = flip ? A.vx[k] : B.vx[k]; I.vy[k] = flip ? A.vy[k] : B.vy[k];
        I.nx[k] = flip ? A.nx[k] : B.nx[k]; I.ny[k] = flip ? A.ny[k] : B.ny[k];
    }
    // incident edge: most anti-parallel normal, first minimum, again tracked by value
    float bestd = INFINITY;
    float i1x = 0.0f, i1y = 0.0f, i2x = 0.0f, i2y = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        if (j < I.n) {
            float d = I.nx[j] * nrx + I.ny[j] * nry;
            const bool wrap = (j + 1 == I.n) || (j + 1 == MAXV);
            const float x1 = wrap ? I.vx[0] : I.vx[(j + 1) % MAXV];
            const float y1 = wrap ? I.vy[0] : I.vy[(j + 1) % MAXV];
            if (d < bestd) { bestd = d; i1x = I.vx[j]; i1y = I.vy[j]; i2x = x1; i2y = y1; }
        }
    }
    const float tx = -nry, ty = nrx;
    const float L = (r2x - r1x) * tx + (r2y - r1y) * ty;
    const float u1 = (i1x - r1x) * tx + (i1y - r1y) * ty;
    const float u2 = (i2x - r1x) * tx + (i2y - r1y) * ty;
    const float du = u2 - u1;
    float ta = 0.0f, tb = 1.0f;
    bool empty = false;
    if (du > 0.0f) {
        ta = fmaxf(ta, (0.0f - u1) / du);
        tb = fminf(tb, (L - u1) / du);
    } else if (du < 0.0f) {
        tb = fminf(tb, (0.0f - u1) / du);
        ta = fmaxf(ta, (L - u1) / du);
    } else if (u1 < 0.0f || u1 > L) {
        empty = true;
    }
    if (ta > tb) empty = true;
    const float rr = rr_sel, ri = I.r;
    float nx, ny;
    float prx[2], pry[2], pix[2], piy[2], dep[2];
    int cnt = 0;
    if (!empty) {
        const float ex = i2x - i1x, ey = i2y - i1y;
        const int np = (tb > ta) ? 2 : 1;
        nx = nrx; ny = nry;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k < np) {
                const float tk = k == 0 ? ta : tb;
                float cx = i1x + tk * ex, cy = i1y + tk * ey;
                float s = (cx - r1x) * nrx + (cy - r1y) * nry;
                float d = rsum - s;
                if (d > -sm) {
                    float fx = (cx - nrx * s) + nrx * rr, fy = (cy - nry * s) + nry * rr;
                    float gx = cx - nrx * ri, gy = cy - nry * ri;
                    if (cnt == 0) { prx[0] = fx; pry[0] = fy; pix[0] = gx; piy[0] = gy; dep[0] = d; }
                    else { prx[1] = fx; pry[1] = fy; pix[1] = gx; piy[1] = gy; dep[1] = d; }
                    ++cnt;
                }
            }
        }
    } else {
        const bool lowend = (u1 + u2 < 0.0f);
        float vx = lowend ? r1x : r2x, vy = lowend ? r1y : r2y;
        float qx, qy, d2;
        closest_on_poly<MAXV>(I, vx, vy, qx, qy, d2);
        if (d2 > reach * reach) return;
        float dist = sqrtf(d2);
        if (dist > SFX_EPS_LEN) { nx = (qx - vx) / dist; ny = (qy - vy) / dist; }
        else { nx = nrx; ny = nry; }
        float d = rsum - dist;
        if (d > -sm) {
            prx[0] = vx + nx * rr; pry[0] = vy + ny * rr;
            pix[0] = qx - nx * ri; piy[0] = qy - ny * ri;
            dep[0] = d;
            cnt = 1;
        }
    }
    if (cnt == 0) return;
    m.count = cnt;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < cnt) {
            m.depth[k] = dep[k];
            m.p1x[k] = flip ? pix[k] : prx[k]; m.p1y[k] = flip ? piy[k] : pry[k];
            m.p2x[k] = flip ? prx[k] : pix[k]; m.p2y[k] = flip ? pry[k] : piy[k];
        }
    }
    m.nx = flip ? -nx : nx;
    m.ny = flip ? -ny : ny;
}

template <int MAXV>
__device__ __forceinline__ void manifold_poly(const float4& pa, const float4& sa, const float4& pb, const float4& sb,
                                              const float4* __restrict__ poly, float sm, Manifold& m) {
    const uint32_t ta = shape_type(sa), tb = shape_type(sb);
    if (tb == SFX_SHAPE_CIRCLE) {
        Core<MAXV> A;
        build_core<MAXV>(pa, sa, poly, A);
        if (A.n >= 2) pc_manifold<MAXV>(A, pb.x, pb.y, sb.w, sm, m);
    } else if (ta == SFX_SHAPE_CIRCLE) {
        Core<MAXV> B;
        build_core<MAXV>(pb, sb, poly, B);
        if (B.n >= 2) {
            // solved as B against the circle, then turned round in place
            pc_manifold<MAXV>(B, pa.x, pa.y, sa.w, sm, m);
            if (m.count) {
                m.nx = -m.nx; m.ny = -m.ny;
                const float tx = m.p1x[0], ty = m.p1y[0];
                m.p1x[0] = m.p2x[0]; m.p1y[0] = m.p2y[0];
                m.p2x[0] = tx; m.p2y[0] = ty;
            }
        }
    } else {
        Core<MAXV> A, B;
        build_core<MAXV>(pa, sa, poly, A);
        build_core<MAXV>(pb, sb, poly, B);
        if (A.n >= 2 && B.n >= 2) pp_manifold<MAXV>(A, B, sm, m);
    }
}

// manifold of two posed shapes, normal from a to b; points up to `sm` apart are kept with
// their negative depth (speculative margin)
__device__ __forceinline__ void manifold(const float4& pa, const float4& sa, const float4& pb, const float4& sb,
                                         const float4* __restrict__ poly, float sm, Manifold& m) {
    const uint32_t ta = shape_type(sa), tb = shape_type(sb);
    m.count = 0;
    if (ta == SFX_SHAPE_NONE || tb == SFX_SHAPE_NONE) return;
    if (ta == SFX_SHAPE_CIRCLE && tb == SFX_SHAPE_CIRCLE) {
        cc_manifold(pa.x, pa.y, sa.w, pb.x, pb.y, sb.w, sm, m);
    } else if (ta == SFX_SHAPE_POLY || tb == SFX_SHAPE_POLY) {
        manifold_poly<8>(pa, sa, pb, sb, poly, sm, m);
    } else {
        manifold_poly<4>(pa, sa, pb, sb, poly, sm, m);
    }
}

// ---- per-manifold solver records -----------------------------------------------------
// DESIGN.md "Contract / frozen rows": everything of a pair's constraint rows that depends on geometry and mass only is
// computed ONCE per substep by the narrowphase (prepare_pair, one pair per lane, the stage that fills the device) and
// kept in the pair's record: the lever arms of the normal rows and of the tangent row as they act on each side, the
// inverse of the 2x2 effective mass, the tangent row's couplings and the reciprocals of its Schur complements, the
// restitution targets.  The colour steps of the solves -- one wave's dependent chain each, the part of the tick that
// does not parallelise -- are then a few dozen multiply-adds per pair on a displacement (position pass) or a velocity
// (velocity pass) held in LDS: no anchor is rotated, nothing is divided and no rotation is renormalised inside them.
//
// One record = seven 16-byte words per pair slot (colour-sorted order), kept as seven PLANES: word i of slot s is
// rec[i * nslots + s].  The colour loops take the pairs of a colour with consecutive lanes, so every load of a plane is
// one contiguous run of 16-byte words (a record kept as one 128-byte block made each of the seven loads of a wave touch
// 32 different lines, and the colour steps of both solves were bound by the address pipeline: measured in DESIGN.md):
//   A (nx, ny, count | block << 8 as bits, 1 / K00)    B (ca0, cb0, d0_0, tgt0)    C (cta, ctb, mu_s, mu_d)    D (At0, invS, Att, -)
//   two-point manifolds only:  E (ca1, cb1, d0_1, tgt1)    F (i00, i01, i11, K01)    G (1 / K11, At1, invS0, invS1)
// The position pass reads A, B and E, F, G; the velocity pass all seven.
struct ContactArrays {
    float4* rec;
    float2* lam;     // (lambda0, lambda1) of this substep: written by the position solve, read by the velocity pass
    uint32_t nslots; // plane stride
};
constexpr uint32_t kRecPlanes = 7;
enum { kPlA = 0, kPlB, kPlC, kPlD, kPlE, kPlF, kPlG };

// What the narrowphase of one candidate pair leaves for the solves, in registers (whole pair, one lane).
struct PairRec {
    int count;             // 0, 1 or 2 points
    bool block;            // count == 2 and the two normal rows are solved as a block
    float nx, ny;          // from a to b
    float mu_s, mu_d;
    float ca[2], cb[2];    // lever arms of the normal rows AS THEY ACT on each side: ca = ra x n, cb = -(rb x n)
    float d0[2];           // depth at the narrowphase poses
    float tgt[2];          // normal velocity the velocity pass asks for: max(-e vn_pre, 0)
    float iK[2];           // 1 / K00, 1 / K11
    float i00, i01, i11, K01;
    float cta, ctb;        // lever arms of the tangent row t = (-ny, nx) at the mean anchor, as they act
    float Att, At[2];
    float invS, invS0, invS1;
    float l0, l1;          // normal lambdas of this substep
};

// y = K^-1 r restricted to the active normal rows (inverse from the record: nothing is divided).  Written as selects
// between the candidates, not as branches: the colour steps are one wave's instruction chain, and a divergent branch
// costs more scalar bookkeeping than the few multiplies it skips.  (Fields of the second point are only meaningful when
// count == 2; what is computed from them otherwise is never selected.)
__device__ __forceinline__ void kinv_rec(int count, bool block, float i00, float i01, float i11, float iK0, float iK1,
                                         bool act0, bool act1, float r0, float r1, float& y0, float& y1) {
    const float yb0 = i00 * r0 - i01 * r1, yb1 = i11 * r1 - i01 * r0;
    const float ys0 = r0 * iK0, ys1 = r1 * iK1;
    const float yh0 = 0.5f * ys0, yh1 = 0.5f * ys1;
    const bool both = count == 2 && act0 && act1;
    const float b0 = block ? yb0 : yh0, b1 = block ? yb1 : yh1;
    const float s0 = act0 ? ys0 : 0.0f;
    const float s1 = (!act0 && count == 2 && act1) ? ys1 : 0.0f;
    y0 = both ? b0 : s0;
    y1 = both ? b1 : s1;
}

__device__ __forceinline__ void lcp2_rec(bool block, float i00, float i01, float i11, float iK0, float iK1, float K01,
                                         float d0, float d1, float& l0, float& l1) {
    const float ab = i00 * d0 - i01 * d1, bb = i11 * d1 - i01 * d0;
    const bool okb = block && ab >= 0.0f && bb >= 0.0f;
    const float a1 = d0 * iK0;
    const bool ok1 = a1 >= 0.0f && d1 - K01 * a1 <= 0.0f;
    const float b2 = d1 * iK1;
    const bool ok2 = b2 >= 0.0f && d0 - K01 * b2 <= 0.0f;
    const bool ge = d0 >= d1;
    const float f0 = ge ? fmaxf(d0, 0.0f) * iK0 : 0.0f, f1 = ge ? 0.0f : fmaxf(d1, 0.0f) * iK1;
    const float t0 = ok2 ? 0.0f : f0, t1 = ok2 ? b2 : f1;
    const float u0 = ok1 ? a1 : t0, u1 = ok1 ? 0.0f : t1;
    l0 = okb ? ab : u0;
    l1 = okb ? bb : u1;
}

__device__ __forceinline__ float inv_pos(float S) { return S > 0.0f ? 1.0f / S : 0.0f; }

// Narrowphase of one solver pair at the poses the substep's integration and joints left (DESIGN.md "Contract / sweep
// order") and the pair's frozen rows.  pa / pb, va / vb: pose and velocity of the two ENTRIES (colliders; a further
// collider mirrors its body); wa / wb: the masses of their BODIES; ca / cb: the poses of the two shapes.
__device__ __forceinline__ void prepare_pair(const float4& pa, const float4& sa, const float4& va, const float2& wa,
                                             const float4& pb, const float4& sb, const float4& vb, const float2& wb,
                                             const float4& ca /* pose of a's shape */, const float4& cb,
                                             const float4* __restrict__ poly, const float4* __restrict__ mats,
                                             float sm, float vthr, PairRec& r) {
    Manifold m;
    manifold(ca, sa, cb, sb, poly, sm, m);
    r.count = m.count;
    r.l0 = 0.0f; r.l1 = 0.0f;
    r.block = false;
    r.nx = 0.0f; r.ny = 0.0f; r.mu_s = 0.0f; r.mu_d = 0.0f;
    r.ca[0] = r.ca[1] = r.cb[0] = r.cb[1] = 0.0f;
    r.d0[0] = r.d0[1] = r.tgt[0] = r.tgt[1] = 0.0f;
    r.iK[0] = r.iK[1] = 0.0f; r.i00 = r.i01 = r.i11 = r.K01 = 0.0f;
    r.cta = r.ctb = r.Att = r.At[0] = r.At[1] = 0.0f;
    r.invS = r.invS0 = r.invS1 = 0.0f;
    if (m.count == 0) return;
    const float4 ma = mats[shape_material(sa)], mb = mats[shape_material(sb)];
    r.mu_s = 0.5f * (ma.y + mb.y);
    r.mu_d = 0.5f * (ma.z + mb.z);
    const float rest = fmaxf(ma.x, mb.x);
    const float nx = m.nx, ny = m.ny;
    r.nx = nx; r.ny = ny;
    const float ima = wa.x, iia = wa.y, imb = wb.x, iib = wb.y;
    const float ims = ima + imb;
    float rax[2], ray[2], rbx[2], rby[2], cna[2], cnb[2];
    rax[1] = ray[1] = rbx[1] = rby[1] = 0.0f; cna[1] = cnb[1] = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < m.count) {
            rax[k] = m.p1x[k] - pa.x; ray[k] = m.p1y[k] - pa.y;
            rbx[k] = m.p2x[k] - pb.x; rby[k] = m.p2y[k] - pb.y;
            const float vax = va.x - va.z * ray[k], vay = va.y + va.z * rax[k];
            const float vbx = vb.x - vb.z * rby[k], vby = vb.y + vb.z * rbx[k];
            const float vpre = (vbx - vax) * nx + (vby - vay) * ny;
            const float e = fabsf(vpre) <= vthr ? 0.0f : rest;
            r.tgt[k] = fmaxf(-e * vpre, 0.0f);
            r.d0[k] = (m.p1x[k] - m.p2x[k]) * nx + (m.p1y[k] - m.p2y[k]) * ny;
            cna[k] = rax[k] * ny - ray[k] * nx;
            cnb[k] = rbx[k] * ny - rby[k] * nx;
        }
    }
    const float K00 = ims + (iia * (cna[0] * cna[0]) + iib * (cnb[0] * cnb[0]));
    if (!(K00 > 0.0f)) { r.count = 0; return; }
    r.iK[0] = 1.0f / K00;
    if (r.count == 2) {
        const float K11 = ims + (iia * (cna[1] * cna[1]) + iib * (cnb[1] * cnb[1]));
        const float K01 = ims + (iia * (cna[0] * cna[1]) + iib * (cnb[0] * cnb[1]));
        if (!(K11 > 0.0f)) {
            r.count = 1;
        } else {
            r.iK[1] = 1.0f / K11; r.K01 = K01;
            const float det = K00 * K11 - K01 * K01;
            if (det > 1e-6f * (K00 * K11)) {
                const float inv = 1.0f / det;
                r.block = true; r.i00 = K11 * inv; r.i01 = K01 * inv; r.i11 = K00 * inv;
            }
        }
    }
    if (r.count == 1) { cna[1] = 0.0f; cnb[1] = 0.0f; r.d0[1] = 0.0f; r.tgt[1] = 0.0f; }
    r.ca[0] = cna[0]; r.ca[1] = cna[1]; r.cb[0] = -cnb[0]; r.cb[1] = -cnb[1];
    float max_, may_, mbx_, mby_;
    if (r.count == 2) {
        max_ = 0.5f * (rax[0] + rax[1]); may_ = 0.5f * (ray[0] + ray[1]);
        mbx_ = 0.5f * (rbx[0] + rbx[1]); mby_ = 0.5f * (rby[0] + rby[1]);
    } else { max_ = rax[0]; may_ = ray[0]; mbx_ = rbx[0]; mby_ = rby[0]; }
    const float tx = -ny, ty = nx;
    const float cta = max_ * ty - may_ * tx, ctb = mbx_ * ty - mby_ * tx;
    r.cta = cta; r.ctb = -ctb;
    const float Att = ims + (iia * (cta * cta) + iib * (ctb * ctb));
    const float At0 = iia * (cta * cna[0]) + iib * (ctb * cnb[0]);
    const float At1 = iia * (cta * cna[1]) + iib * (ctb * cnb[1]);
    r.Att = Att; r.At[0] = At0; r.At[1] = At1;
    float y0, y1;
    kinv_rec(r.count, r.block, r.i00, r.i01, r.i11, r.iK[0], r.iK[1], true, r.count == 2, At0, At1, y0, y1);
    r.invS = inv_pos(Att - (At0 * y0 + At1 * y1));
    if (r.count == 2) {
        r.invS0 = inv_pos(Att - At0 * (At0 * r.iK[0]));
        r.invS1 = inv_pos(Att - At1 * (At1 * r.iK[1]));
    }
}

// Position solve of one prepared pair on the two bodies' displacements Da, Db = (dx, dy, dtheta, -) accumulated since
// the narrowphase: the depth of a point is its depth there plus what the displacements change along its frozen row;
// the manifold's normal constraints are solved as a block; the displacements grow.  Returns true when they changed.
// Mueller et al. 2020 sec. 3.5.
__device__ __forceinline__ bool solve_pos_rec(PairRec& r, float4& Da, const float2& wa, float4& Db, const float2& wb,
                                              float max_corr) {
    if (r.count == 0) return false;
    const float nx = r.nx, ny = r.ny;
    const float la = Da.x * nx + Da.y * ny, lb = -(Db.x * nx + Db.y * ny);
    float d[2];
    d[1] = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < r.count) {
            const float ga = la + Da.z * r.ca[k], gb = lb + Db.z * r.cb[k];
            d[k] = r.d0[k] + (ga + gb);
            d[k] = fminf(d[k], max_corr);  // bounded depenetration per substep
        }
    }
    float l0 = 0.0f, l1 = 0.0f;
    if (r.count == 2) lcp2_rec(r.block, r.i00, r.i01, r.i11, r.iK[0], r.iK[1], r.K01, d[0], d[1], l0, l1);
    else l0 = fmaxf(d[0], 0.0f) * r.iK[0];
    r.l0 = l0; r.l1 = l1;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return false;
    Da.x = Da.x - (nx * lt) * wa.x;
    Da.y = Da.y - (ny * lt) * wa.x;
    Da.z = Da.z - wa.y * (r.ca[0] * l0 + r.ca[1] * l1);
    Db.x = Db.x + (nx * lt) * wb.x;
    Db.y = Db.y + (ny * lt) * wb.x;
    Db.z = Db.z - wb.y * (r.cb[0] * l0 + r.cb[1] * l1);
    return true;
}

struct StepParams {
    float h, inv_h, inv_h2;
    float gx, gy;
    float vthr;      // restitution threshold velocity: cfg.restitution_threshold * |g| * h
    float max_corr;  // cfg.max_depenetration_velocity * h
    float damp_lin, damp_ang;  // 1 / (1 + h * damping)
    uint32_t flags;
};

// a displacement (dx, dy, dtheta) applied to a pose: the translation added, the rotation through rotate_by; a body that
// was not turned keeps the bits of its rotation
__device__ __forceinline__ void apply_delta(float4& p, const float4& d) {
    p.x = p.x + d.x;
    p.y = p.y + d.y;
    if (d.z != 0.0f) rotate_by(p.z, p.w, d.z);
}

// Velocity pass of one prepared pair on the frozen rows: restitution block, then friction along t = (-ny, nx) with
// compensating normal impulses; static friction also shifts the poses: the shift accumulates in Sa, Sb = (dx, dy,
// dtheta, -) and is applied by the caller (at the end of the phase; the fallback list applies it pair by pair).
// Returns bit 0 when the velocities changed and bit 1 when the shifts did.  Mueller et al. 2020 sec. 3.6.
__device__ __forceinline__ int solve_vel_rec(const PairRec& r, float4& va, const float2& wa, float4& vb, const float2& wb,
                                             float4& Sa, float4& Sb, const StepParams& sp) {
    const int count = r.count;
    if (count == 0) return 0;
    const float l0 = r.l0, l1 = count == 2 ? r.l1 : 0.0f;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return 0;
    const float nx = r.nx, ny = r.ny;
    const float ima = wa.x, iia = wa.y, imb = wb.x, iib = wb.y;
    const bool act0 = l0 > 0.0f, act1 = count == 2 && l1 > 0.0f;
    {
        const float la = va.x * nx + va.y * ny, lb = -(vb.x * nx + vb.y * ny);
        float dv[2];
        dv[0] = dv[1] = 0.0f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k < count) {
                const float ua = la + va.z * r.ca[k], ub = lb + vb.z * r.cb[k];
                dv[k] = r.tgt[k] + (ua + ub);
            }
        }
        float p0, p1;
        kinv_rec(count, r.block, r.i00, r.i01, r.i11, r.iK[0], r.iK[1], act0, act1, dv[0], dv[1], p0, p1);
        const float pt = p0 + p1;
        va.x = va.x - (nx * pt) * ima;
        va.y = va.y - (ny * pt) * ima;
        va.z = va.z - iia * (r.ca[0] * p0 + r.ca[1] * p1);
        vb.x = vb.x + (nx * pt) * imb;
        vb.y = vb.y + (ny * pt) * imb;
        vb.z = vb.z - iib * (r.cb[0] * p0 + r.cb[1] * p1);
    }
    if (sp.flags & SFX_CFG_NO_FRICTION) return 1;
    const float tx = -ny, ty = nx;
    const float wta = (va.x * tx + va.y * ty) + va.z * r.cta;
    const float wtb = -(vb.x * tx + vb.y * ty) + vb.z * r.ctb;
    const float vt = -(wta + wtb);
    if (!(fabsf(vt) > SFX_EPS_LEN)) return 1;
    float y0, y1;
    kinv_rec(count, r.block, r.i00, r.i01, r.i11, r.iK[0], r.iK[1], act0, act1, r.At[0], r.At[1], y0, y1);
    float invS = r.invS1;   // (assignments, not a conditional between members: that becomes an address select and the record goes to scratch memory)
    if (act0) invS = r.invS0;
    if (act0 && (act1 || count == 1)) invS = r.invS;
    const float need = vt * invS;
    if (need == 0.0f) return 1;
    const bool is_static = fabsf(need) * sp.h < r.mu_s * lt;
    const float pt = is_static ? -need : -copysignf(fminf(r.mu_d * lt * sp.inv_h, fabsf(need)), need);
    const float q0 = -(y0 * pt), q1 = -(y1 * pt);
    const float jx = tx * pt + nx * (q0 + q1), jy = ty * pt + ny * (q0 + q1);
    const float ja = (r.cta * pt + r.ca[0] * q0) + r.ca[1] * q1;
    const float jb = (r.ctb * pt + r.cb[0] * q0) + r.cb[1] * q1;
    va.x = va.x - jx * ima;
    va.y = va.y - jy * ima;
    va.z = va.z - iia * ja;
    vb.x = vb.x + jx * imb;
    vb.y = vb.y + jy * imb;
    vb.z = vb.z - iib * jb;
    if (!is_static) return 1;
    Sa.x = Sa.x - (jx * sp.h) * ima;
    Sa.y = Sa.y - (jy * sp.h) * ima;
    Sa.z = Sa.z - iia * (ja * sp.h);
    Sb.x = Sb.x + (jx * sp.h) * imb;
    Sb.y = Sb.y + (jy * sp.h) * imb;
    Sb.z = Sb.z - iib * (jb * sp.h);
    return 3;
}

// ---- the same two solves with TWO LANES PER PAIR (the tile kernels): lane 2k holds body a of a pair, lane 2k + 1 body b.
// A lane computes what belongs to its own body (its projections on the frozen rows, the update of its displacement or
// velocity) and takes the other body's projections from its neighbour lane (one DPP move each); what belongs to the pair
// (depths, the 2x2 solve, friction) both lanes compute, on identical inputs, to identical bits.  With the normal SIGNED
// per lane (n for a, -n for b) and the lever arms stored as they act, the code is the same on both sides and equals
// solve_pos_rec / solve_vel_rec bit for bit: negations are exact, sums of two are commutative, and rounding is symmetric
// in the sign.
__device__ __forceinline__ float other_side(float x) {   // the neighbour lane's value (lanes 2k <-> 2k + 1: quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}

// One side of a prepared pair, in registers.
struct SideRec {
    int count;             // 0, 1 or 2 points
    bool block;
    float nx, ny;          // the normal as this side sees it: from this body to the other
    float mu_s, mu_d;
    float c[2];            // this side's lever arms of the normal rows, as they act
    float ct;              // ... of the tangent row
    float d0[2], tgt[2];
    float iK[2], i00, i01, i11, K01;
    float At[2];
    float invS, invS0, invS1;
    float l0, l1;
};

// returns true when the displacement changed; r.l0 / r.l1 = the multipliers (the same on both lanes)
__device__ __forceinline__ bool solve_pos_side(SideRec& r, float4& D, const float2& w, float max_corr) {
    const float nx = r.nx, ny = r.ny;
    const float l = D.x * nx + D.y * ny;
    const bool two = r.count == 2;
    const float g0 = l + D.z * r.c[0], g1 = l + D.z * r.c[1];
    const float d0 = fminf(r.d0[0] + (g0 + other_side(g0)), max_corr);
    const float d1 = fminf(r.d0[1] + (g1 + other_side(g1)), max_corr);   // (meaningful for two points only)
    float m0, m1;
    lcp2_rec(r.block, r.i00, r.i01, r.i11, r.iK[0], r.iK[1], r.K01, d0, d1, m0, m1);
    const float l0 = two ? m0 : fmaxf(d0, 0.0f) * r.iK[0];
    const float l1 = two ? m1 : 0.0f;
    r.l0 = l0; r.l1 = l1;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return false;
    D.x = D.x - (nx * lt) * w.x;
    D.y = D.y - (ny * lt) * w.x;
    D.z = D.z - w.y * (r.c[0] * l0 + r.c[1] * l1);
    return true;
}

// bit 0: the velocity changed, bit 1: the shift did (static friction)
__device__ __forceinline__ int solve_vel_side(const SideRec& r, float4& v, float4& S, const float2& w, const StepParams& sp) {
    const int count = r.count;
    const float l0 = r.l0, l1 = count == 2 ? r.l1 : 0.0f;
    const float lt = l0 + l1;
    const float nx = r.nx, ny = r.ny;
    const float im = w.x, ii = w.y;
    const bool act0 = l0 > 0.0f, act1 = count == 2 && l1 > 0.0f;
    {
        const float l = v.x * nx + v.y * ny;
        const float u0 = l + v.z * r.c[0], u1 = l + v.z * r.c[1];
        const float dv0 = r.tgt[0] + (u0 + other_side(u0));
        const float dv1 = r.tgt[1] + (u1 + other_side(u1));   // (meaningful for two points only: kinv_rec selects)
        float p0, p1;
        kinv_rec(count, r.block, r.i00, r.i01, r.i11, r.iK[0], r.iK[1], act0, act1, dv0, dv1, p0, p1);
        const float pt = p0 + p1;
        v.x = v.x - (nx * pt) * im;
        v.y = v.y - (ny * pt) * im;
        v.z = v.z - ii * (r.c[0] * p0 + r.c[1] * p1);
    }
    if (sp.flags & SFX_CFG_NO_FRICTION) return 1;
    const float tx = -ny, ty = nx;
    const float wt = (v.x * tx + v.y * ty) + v.z * r.ct;
    const float vt = -(wt + other_side(wt));
    if (!(fabsf(vt) > SFX_EPS_LEN)) return 1;
    float y0, y1;
    kinv_rec(count, r.block, r.i00, r.i01, r.i11, r.iK[0], r.iK[1], act0, act1, r.At[0], r.At[1], y0, y1);
    float invS = r.invS1;   // (assignments, not a conditional between members: that becomes an address select and the record goes to scratch memory)
    if (act0) invS = r.invS0;
    if (act0 && (act1 || count == 1)) invS = r.invS;
    const float need = vt * invS;
    if (need == 0.0f) return 1;
    const bool is_static = fabsf(need) * sp.h < r.mu_s * lt;
    const float pt = is_static ? -need : -copysignf(fminf(r.mu_d * lt * sp.inv_h, fabsf(need)), need);
    const float q0 = -(y0 * pt), q1 = -(y1 * pt);
    const float jx = tx * pt + nx * (q0 + q1), jy = ty * pt + ny * (q0 + q1);
    const float jr = (r.ct * pt + r.c[0] * q0) + r.c[1] * q1;
    v.x = v.x - jx * im;
    v.y = v.y - jy * im;
    v.z = v.z - ii * jr;
    if (!is_static) return 1;
    S.x = S.x - (jx * sp.h) * im;
    S.y = S.y - (jy * sp.h) * im;
    S.z = S.z - ii * (jr * sp.h);
    return 3;
}

__device__ __forceinline__ void anchor_world(const float4& pose, float lx, float ly, float& rx, float& ry) {
    rx = pose.z * lx - pose.w * ly;
    ry = pose.w * lx + pose.z * ly;
}

// joints: distance constraints (Mueller et al. 2020 sec. 3.3, XPBD compliance) and angle constraints
// (sec. 3.4.1 "joint limits", in 2D a scalar), lambda restarts at 0
#define SFX_JOINT_ANGLE 3u  // kind of an angle constraint (0..2 = SFX_LIMIT_* of a distance constraint)
struct JointArrays {
    const uint2* bodies;    // (a, b), SFX_WORLD_BODY for the world
    const float4* anchors;  // distance: (a.x, a.y, b.x, b.y); angle: (min, max, 0, 0)
    const float4* params;   // (rest_length, compliance, kind as bits, damping)
};

// atan2 with the same operations on both sides (Cephes atanf: two range reductions and a degree-4
// polynomial in x^2, ~2 ulp), so that the oracle and the device agree bit for bit
__device__ __forceinline__ float atan_pos(float x /* >= 0 */) {
    float y = 0.0f;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    const float z = x * x;
    const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z;
    return y + (p * x + x);
}

__device__ __forceinline__ float atan2_det(float y, float x) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    const float ax = fabsf(x), ay = fabsf(y);
    float r;
    if (ax >= ay) r = atan_pos(ay / ax);                       // [0, pi/4]
    else r = 1.5707963267948966f - atan_pos(ax / ay);          // (pi/4, pi/2]
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return y < 0.0f ? -r : r;
}

__device__ __forceinline__ void apply_pos(float4& pose, const float2& w, float rx, float ry, float px, float py, float sign) {
    if (w.x > 0.0f) {
        pose.x = pose.x + sign * (px * w.x);
        pose.y = pose.y + sign * (py * w.x);
    }
    if (w.y > 0.0f) {
        float dth = sign * (w.y * (rx * py - ry * px));
        rotate_by(pose.z, pose.w, dth);
    }
}

__device__ __forceinline__ float gen_inv_mass(const float2& w, float rx, float ry, float nx, float ny) {
    float rn = rx * ny - ry * nx;
    return w.x + w.y * (rn * rn);
}

// one joint from its three records (k_joint_groups fetches them a colour ahead), and from its index
__device__ __forceinline__ void solve_joint_rec(const BodyArrays& B, const uint2& ab, const float4& an, const float4& pr, float inv_h2);
__device__ __forceinline__ void solve_joint(const BodyArrays& B, const JointArrays& J, uint32_t k, float inv_h2) {
    solve_joint_rec(B, J.bodies[k], J.anchors[k], J.params[k], inv_h2);
}
__device__ __forceinline__ void solve_joint_rec(const BodyArrays& B, const uint2& ab, const float4& an, const float4& pr, float inv_h2) {
    const uint32_t a = ab.x, b = ab.y;
    float rax = 0.0f, ray = 0.0f, rbx = 0.0f, rby = 0.0f, pax, pay, pbx, pby;
    float4 pa = make_float4(0, 0, 1, 0), pb = make_float4(0, 0, 1, 0);
    float2 wa = make_float2(0, 0), wb = make_float2(0, 0);
    if (__float_as_uint(pr.z) == SFX_JOINT_ANGLE) {
        if (a != SFX_WORLD_BODY) { pa = B.pose[a]; wa = B.mass[a]; }
        if (b != SFX_WORLD_BODY) { pb = B.pose[b]; wb = B.mass[b]; }
        // relative rotation b * conj(a) and its angle
        const float qc = pb.z * pa.z + pb.w * pa.w, qs = pb.w * pa.z - pb.z * pa.w;
        const float phi = atan2_det(qs, qc);
        const float Ca = phi - fminf(fmaxf(phi, an.x), an.y);
        if (Ca == 0.0f) return;
        const float wsa = (wa.y + wb.y) + pr.y * inv_h2;
        if (!(wsa > 0.0f)) return;
        const float dl = Ca / wsa;
        if (wa.y > 0.0f) { rotate_by(pa.z, pa.w, wa.y * dl); B.pose[a] = pa; }
        if (wb.y > 0.0f) { rotate_by(pb.z, pb.w, -(wb.y * dl)); B.pose[b] = pb; }
        return;
    }
    if (a != SFX_WORLD_BODY) {
        pa = B.pose[a]; wa = B.mass[a];
        anchor_world(pa, an.x, an.y, rax, ray);
        pax = pa.x + rax; pay = pa.y + ray;
    } else { pax = an.x; pay = an.y; }
    if (b != SFX_WORLD_BODY) {
        pb = B.pose[b]; wb = B.mass[b];
        anchor_world(pb, an.z, an.w, rbx, rby);
        pbx = pb.x + rbx; pby = pb.y + rby;
    } else { pbx = an.z; pby = an.w; }
    float dx = pbx - pax, dy = pby - pay;
    float len = sqrtf(dx * dx + dy * dy);
    if (!(len > SFX_EPS_LEN)) return;
    float nx = dx / len, ny = dy / len;
    float Cc = len - pr.x;
    const uint32_t limit = __float_as_uint(pr.z);  // one-sided links: a rope only pulls, a strut only pushes
    if (limit == SFX_LIMIT_ROPE && !(Cc > 0.0f)) return;
    if (limit == SFX_LIMIT_STRUT && !(Cc < 0.0f)) return;
    float wsum = pr.y * inv_h2;
    if (a != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(wa, rax, ray, nx, ny);
    if (b != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(wb, rbx, rby, nx, ny);
    if (!(wsum > 0.0f)) return;
    float corr = Cc / wsum;
    if (a != SFX_WORLD_BODY && (wa.x > 0.0f || wa.y > 0.0f)) {
        apply_pos(pa, wa, rax, ray, nx * corr, ny * corr, 1.0f);
        B.pose[a] = pa;
    }
    if (b != SFX_WORLD_BODY && (wb.x > 0.0f || wb.y > 0.0f)) {
        apply_pos(pb, wb, rbx, rby, nx * corr, ny * corr, -1.0f);
        B.pose[b] = pb;
    }
}

// joint damping in the velocity pass (Mueller et al. 2020 sec. 3.3.1 / 3.6): removes the fraction
// min(damping * h, 1) of the relative velocity the joint sees
__device__ __forceinline__ void damp_joint(const BodyArrays& B, const JointArrays& J, uint32_t k, float h) {
    const float4 pr = J.params[k];
    if (!(pr.w > 0.0f)) return;
    const uint2 ab = J.bodies[k];
    const uint32_t a = ab.x, b = ab.y;
    const float frac = fminf(pr.w * h, 1.0f);
    float4 pa = make_float4(0, 0, 1, 0), pb = make_float4(0, 0, 1, 0), va = make_float4(0, 0, 0, 0), vb = make_float4(0, 0, 0, 0);
    float2 wa = make_float2(0, 0), wb = make_float2(0, 0);
    if (a != SFX_WORLD_BODY) { pa = B.pose[a]; va = B.vel[a]; wa = B.mass[a]; }
    if (b != SFX_WORLD_BODY) { pb = B.pose[b]; vb = B.vel[b]; wb = B.mass[b]; }
    if (__float_as_uint(pr.z) == SFX_JOINT_ANGLE) {
        const float wsa = wa.y + wb.y;
        if (!(wsa > 0.0f)) return;
        const float imp = ((vb.z - va.z) * frac) / wsa;
        if (wa.y > 0.0f) { va.z = va.z + wa.y * imp; B.vel[a] = va; }
        if (wb.y > 0.0f) { vb.z = vb.z - wb.y * imp; B.vel[b] = vb; }
        return;
    }
    const float4 an = J.anchors[k];
    float rax = 0.0f, ray = 0.0f, rbx = 0.0f, rby = 0.0f, pax = an.x, pay = an.y, pbx = an.z, pby = an.w;
    if (a != SFX_WORLD_BODY) { anchor_world(pa, an.x, an.y, rax, ray); pax = pa.x + rax; pay = pa.y + ray; }
    if (b != SFX_WORLD_BODY) { anchor_world(pb, an.z, an.w, rbx, rby); pbx = pb.x + rbx; pby = pb.y + rby; }
    // relative velocity of the anchors, b minus a
    float dvx = (vb.x - vb.z * rby) - (va.x - va.z * ray);
    float dvy = (vb.y + vb.z * rbx) - (va.y + va.z * rax);
    if (pr.x != 0.0f) {  // a link: only the rate of change of its length
        const float dx = pbx - pax, dy = pby - pay;
        const float len = sqrtf(dx * dx + dy * dy);
        if (!(len > SFX_EPS_LEN)) return;
        const float ux = dx / len, uy = dy / len;
        const float rate = dvx * ux + dvy * uy;
        dvx = ux * rate; dvy = uy * rate;
    }
    const float sp = sqrtf(dvx * dvx + dvy * dvy);
    if (!(sp > SFX_EPS_LEN)) return;
    const float nx = dvx / sp, ny = dvy / sp;
    float wsum = 0.0f;
    if (a != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(wa, rax, ray, nx, ny);
    if (b != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(wb, rbx, rby, nx, ny);
    if (!(wsum > 0.0f)) return;
    const float imp = (sp * frac) / wsum;  // along n on a, against it on b
    const float jx = nx * imp, jy = ny * imp;
    if (a != SFX_WORLD_BODY && (wa.x > 0.0f || wa.y > 0.0f)) {
        va.x = va.x + jx * wa.x; va.y = va.y + jy * wa.x;
        va.z = va.z + wa.y * (rax * jy - ray * jx);
        B.vel[a] = va;
    }
    if (b != SFX_WORLD_BODY && (wb.x > 0.0f || wb.y > 0.0f)) {
        vb.x = vb.x - jx * wb.x; vb.y = vb.y - jy * wb.x;
        vb.z = vb.z - wb.y * (rbx * jy - rby * jx);
        B.vel[b] = vb;
    }
}

// ---- spatial queries (DESIGN.md "Contract / spatial queries") -----------------------------
// Same operations in the same order as oracle/sfx_oracle.c (shape_contains, ray_disc, ray_shape,
// ray_box): the results are compared bit for bit.
template <int MAXV>
__device__ __forceinline__ bool shape_contains(const Core<MAXV>& c, float px, float py) {
    if (c.n <= 0) return false;
    if (c.n == 1) {
        const float dx = px - c.vx[0], dy = py - c.vy[0];
        return dx * dx + dy * dy <= c.r * c.r;
    }
    if (c.n >= 3) {
        float sep = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            if (k < c.n) {
                const float s = c.nx[k] * (px - c.vx[k]) + c.ny[k] * (py - c.vy[k]);
                if (s > sep) sep = s;
            }
        }
        if (sep <= 0.0f) return true;
    }
    float qx, qy, d2;
    closest_on_poly<MAXV>(c, px, py, qx, qy, d2);
    return d2 <= c.r * c.r;
}

// first entry of the ray into the disc (cx, cy, r), origin outside it
__device__ __forceinline__ bool ray_disc(float ox, float oy, float dx, float dy, float cx, float cy, float r, float& t,
                                         float& nx, float& ny) {
    if (!(r > 0.0f)) return false;
    const float mx = ox - cx, my = oy - cy;
    const float a = dx * dx + dy * dy;
    const float b = mx * dx + my * dy;
    const float c = (mx * mx + my * my) - r * r;
    if (!(a > 0.0f) || !(b < 0.0f)) return false;
    const float disc = b * b - a * c;
    if (!(disc >= 0.0f)) return false;
    const float tt = (-b - sqrtf(disc)) / a;
    if (!(tt >= 0.0f)) return false;
    t = tt;
    nx = (mx + tt * dx) / r;
    ny = (my + tt * dy) / r;
    return true;
}

// first entry of the ray into the posed shape, origin outside it; first minimum wins (faces in
// order, then vertex discs in order)
template <int MAXV>
__device__ __forceinline__ bool ray_shape(const Core<MAXV>& c, float ox, float oy, float dx, float dy, float& t, float& nx,
                                          float& ny) {
    if (c.n <= 0) return false;
    if (c.n == 1) return ray_disc(ox, oy, dx, dy, c.vx[0], c.vy[0], c.r, t, nx, ny);
    float best = INFINITY, bnx = 0.0f, bny = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if (k < c.n) {
            const bool wrap = (k + 1 == c.n) || (k + 1 == MAXV);
            const float x1 = wrap ? c.vx[0] : c.vx[(k + 1) % MAXV];
            const float y1 = wrap ? c.vy[0] : c.vy[(k + 1) % MAXV];
            const float denom = dx * c.nx[k] + dy * c.ny[k];
            if (denom < 0.0f) {
                const float p0x = c.vx[k] + c.r * c.nx[k], p0y = c.vy[k] + c.r * c.ny[k];
                const float num = (p0x - ox) * c.nx[k] + (p0y - oy) * c.ny[k];
                const float tt = num / denom;
                if (tt >= 0.0f && tt < best) {
                    const float hx = ox + tt * dx, hy = oy + tt * dy;
                    const float ex = x1 - c.vx[k], ey = y1 - c.vy[k];
                    const float u = (hx - p0x) * ex + (hy - p0y) * ey;
                    const float ee = ex * ex + ey * ey;
                    if (u >= 0.0f && u <= ee) { best = tt; bnx = c.nx[k]; bny = c.ny[k]; }
                }
            }
        }
    }
    if (c.r > 0.0f) {
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            if (k < c.n) {
                float tt, mx, my;
                if (ray_disc(ox, oy, dx, dy, c.vx[k], c.vy[k], c.r, tt, mx, my) && tt < best) { best = tt; bnx = mx; bny = my; }
            }
        }
    }
    if (!(best < INFINITY)) return false;
    t = best; nx = bnx; ny = bny;
    return true;
}

__device__ __forceinline__ bool ray_box(const float4& A, float ox, float oy, float dx, float dy, float max_t) {
    float t0 = 0.0f, t1 = max_t;
    if (dx != 0.0f) {
        float ta = (A.x - ox) / dx, tb = (A.z - ox) / dx;
        if (ta > tb) { const float s = ta; ta = tb; tb = s; }
        t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
    } else if (ox < A.x || ox > A.z) return false;
    if (dy != 0.0f) {
        float ta = (A.y - oy) / dy, tb = (A.w - oy) / dy;
        if (ta > tb) { const float s = ta; ta = tb; tb = s; }
        t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
    } else if (oy < A.y || oy > A.w) return false;
    return t0 <= t1;
}

// point / ray against one posed collider of any type
__device__ __forceinline__ bool collider_contains(const float4& pose, const float4& shape, const float4* __restrict__ poly,
                                                  float px, float py) {
    if (shape_type(shape) == SFX_SHAPE_POLY) {
        Core<8> c;
        build_core<8>(pose, shape, poly, c);
        return shape_contains<8>(c, px, py);
    }
    Core<4> c;
    build_core<4>(pose, shape, poly, c);
    return shape_contains<4>(c, px, py);
}

// origin inside: t = 0, normal (0, 0); else the first entry, if any
__device__ __forceinline__ bool collider_ray(const float4& pose, const float4& shape, const float4* __restrict__ poly,
                                             float ox, float oy, float dx, float dy, float& t, float& nx, float& ny) {
    t = 0.0f; nx = 0.0f; ny = 0.0f;
    if (shape_type(shape) == SFX_SHAPE_POLY) {
        Core<8> c;
        build_core<8>(pose, shape, poly, c);
        if (shape_contains<8>(c, ox, oy)) return true;
        return ray_shape<8>(c, ox, oy, dx, dy, t, nx, ny);
    }
    Core<4> c;
    build_core<4>(pose, shape, poly, c);
    if (shape_contains<4>(c, ox, oy)) return true;
    return ray_shape<4>(c, ox, oy, dx, dy, t, nx, ny);
}

}  // namespace sfx
