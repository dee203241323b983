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
constexpr uint32_t kFiltDynamic = 1u, kFiltFurther = 2u;

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
// What the narrowphase leaves for the solves, one record per solver pair slot (colour-sorted order): a
// whole 64-byte line per slot, so that the narrowphase (which visits the slots in class order, i.e.
// scattered) writes full lines and the colour loops (consecutive lanes, consecutive slots) read them
// coalesced.
struct ContactArrays {
    float4* rec;     // one 64-byte line per slot: [4k] (nx, ny, mu_s, mu_d), [4k+1] point 0 anchors (ra.x, ra.y, rb.x, rb.y)
                     // body-local, [4k+2] point 1 anchors, [4k+3] (restitution, count as bits, vn_pre0, vn_pre1)
    float2* lam;     // (lambda0, lambda1) of this substep: written by the position solve, read by the velocity pass
};

__device__ __forceinline__ void anchor_world(const float4& pose, float lx, float ly, float& rx, float& ry) {
    rx = pose.z * lx - pose.w * ly;
    ry = pose.w * lx + pose.z * ly;
}

__device__ __forceinline__ void solve_lcp2(float K00, float K01, float K11, float d0, float d1, float& l0, float& l1) {
    float det = K00 * K11 - K01 * K01;
    if (det > 1e-6f * (K00 * K11)) {
        float inv = 1.0f / det;
        float a = (K11 * d0 - K01 * d1) * inv;
        float b = (K00 * d1 - K01 * d0) * inv;
        if (a >= 0.0f && b >= 0.0f) { l0 = a; l1 = b; return; }
    }
    float a = d0 / K00;
    if (a >= 0.0f && d1 - K01 * a <= 0.0f) { l0 = a; l1 = 0.0f; return; }
    float b = d1 / K11;
    if (b >= 0.0f && d0 - K01 * b <= 0.0f) { l0 = 0.0f; l1 = b; return; }
    if (d0 >= d1) { l0 = fmaxf(d0, 0.0f) / K00; l1 = 0.0f; }
    else { l0 = 0.0f; l1 = fmaxf(d1, 0.0f) / K11; }
}

// What the narrowphase of one candidate pair leaves for the solves, in registers.
struct PairRec {
    int count;             // 0, 1 or 2 points
    float nx, ny;          // from a to b
    float mu_s, mu_d, rest;
    float la[2][2], lb[2][2];  // body-local anchors
    float vpre[2];         // normal relative velocity before the position solve
    float l0, l1;          // normal lambdas of this substep
};

// Narrowphase of one candidate pair at the poses of the batch start (DESIGN.md "Contract /
// batches"): manifold, body-local anchors, pre-solve normal velocities, combined material.
__device__ __forceinline__ void prepare_pair(const float4& pa, const float4& sa, const float4& va,
                                             const float4& pb, const float4& sb, const float4& vb,
                                             const float4& ca /* pose of a's shape */, const float4& cb,
                                             const float4* __restrict__ poly, const float4* __restrict__ mats,
                                             float sm, PairRec& r) {
    Manifold m;
    manifold(ca, sa, cb, sb, poly, sm, m);
    r.count = m.count;
    r.l0 = 0.0f; r.l1 = 0.0f;
    r.nx = 0.0f; r.ny = 0.0f; r.mu_s = 0.0f; r.mu_d = 0.0f; r.rest = 0.0f;
    r.la[0][0] = r.la[0][1] = r.lb[0][0] = r.lb[0][1] = 0.0f;
    r.la[1][0] = r.la[1][1] = r.lb[1][0] = r.lb[1][1] = 0.0f;
    r.vpre[0] = r.vpre[1] = 0.0f;
    if (m.count == 0) return;
    const float4 ma = mats[shape_material(sa)], mb = mats[shape_material(sb)];
    r.mu_s = 0.5f * (ma.y + mb.y);
    r.mu_d = 0.5f * (ma.z + mb.z);
    r.rest = fmaxf(ma.x, mb.x);
    r.nx = m.nx; r.ny = m.ny;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < m.count) {
            float ax = m.p1x[k] - pa.x, ay = m.p1y[k] - pa.y;
            float bx = m.p2x[k] - pb.x, by = m.p2y[k] - pb.y;
            r.la[k][0] = pa.z * ax + pa.w * ay; r.la[k][1] = pa.z * ay - pa.w * ax;
            r.lb[k][0] = pb.z * bx + pb.w * by; r.lb[k][1] = pb.z * by - pb.w * bx;
            float rax, ray, rbx, rby;
            anchor_world(pa, r.la[k][0], r.la[k][1], rax, ray);
            anchor_world(pb, r.lb[k][0], r.lb[k][1], rbx, rby);
            float vax = va.x - va.z * ray, vay = va.y + va.z * rax;
            float vbx = vb.x - vb.z * rby, vby = vb.y + vb.z * rbx;
            r.vpre[k] = (vbx - vax) * m.nx + (vby - vay) * m.ny;
        }
    }
}

// Position solve of one prepared pair at the current poses: penetration re-evaluated from the
// anchors, the manifold's normal constraints solved as a block, poses updated in registers.
// Returns true when the poses changed.  Mueller et al. 2020 sec. 3.5.
__device__ __forceinline__ bool solve_pos_rec(PairRec& r, float4& pa, const float2& wa, float4& pb, const float2& wb,
                                              float max_corr) {
    if (r.count == 0) return false;
    const float nx = r.nx, ny = r.ny;
    const float ima = wa.x, iia = wa.y, imb = wb.x, iib = wb.y;
    float d[2], cna[2], cnb[2];
    d[1] = 0.0f; cna[1] = 0.0f; cnb[1] = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < r.count) {
            float rax, ray, rbx, rby;
            anchor_world(pa, r.la[k][0], r.la[k][1], rax, ray);
            anchor_world(pb, r.lb[k][0], r.lb[k][1], rbx, rby);
            d[k] = ((pa.x + rax) - (pb.x + rbx)) * nx + ((pa.y + ray) - (pb.y + rby)) * ny;
            d[k] = fminf(d[k], max_corr);  // bounded depenetration per substep
            cna[k] = rax * ny - ray * nx;
            cnb[k] = rbx * ny - rby * nx;
        }
    }
    float l0 = 0.0f, l1 = 0.0f;
    const float ims = ima + imb;
    // effective masses: the sum of the two inverse masses plus the SUM of the two bodies' angular terms (a sum of two is the
    // same whichever body comes first: solve_pos_side computes it on either side of the pair)
    const float K00 = ims + (iia * (cna[0] * cna[0]) + iib * (cnb[0] * cnb[0]));
    if (!(K00 > 0.0f)) return false;
    if (r.count == 2) {
        float K11 = ims + (iia * (cna[1] * cna[1]) + iib * (cnb[1] * cnb[1]));
        float K01 = ims + (iia * (cna[0] * cna[1]) + iib * (cnb[0] * cnb[1]));
        solve_lcp2(K00, K01, K11, d[0], d[1], l0, l1);
    } else {
        l0 = fmaxf(d[0], 0.0f) / K00;
    }
    r.l0 = l0; r.l1 = l1;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return false;
    if (ima > 0.0f) {
        pa.x = pa.x - (nx * lt) * ima;
        pa.y = pa.y - (ny * lt) * ima;
    }
    if (iia > 0.0f) rotate_by(pa.z, pa.w, -(iia * (cna[0] * l0 + cna[1] * l1)));
    if (imb > 0.0f) {
        pb.x = pb.x + (nx * lt) * imb;
        pb.y = pb.y + (ny * lt) * imb;
    }
    if (iib > 0.0f) rotate_by(pb.z, pb.w, iib * (cnb[0] * l0 + cnb[1] * l1));
    return true;
}

__device__ __forceinline__ void kinv_apply(float K0, float K1, float K01, bool act0, bool act1, int count,
                                           float r0, float r1, float& y0, float& y1) {
    y0 = 0.0f; y1 = 0.0f;
    if (count == 2 && act0 && act1) {
        float det = K0 * K1 - K01 * K01;
        if (det > 1e-6f * (K0 * K1)) {
            float inv = 1.0f / det;
            y0 = (K1 * r0 - K01 * r1) * inv;
            y1 = (K0 * r1 - K01 * r0) * inv;
        } else {
            y0 = 0.5f * (r0 / K0); y1 = 0.5f * (r1 / K1);
        }
    } else if (act0) {
        y0 = r0 / K0;
    } else if (count == 2 && act1) {
        y1 = r1 / K1;
    }
}

struct StepParams {
    float h, inv_h, inv_h2;
    float gx, gy;
    float vthr;      // restitution threshold velocity: cfg.restitution_threshold * |g| * h
    float max_corr;  // cfg.max_depenetration_velocity * h
    float damp_lin, damp_ang;  // 1 / (1 + h * damping)
    uint32_t flags;
};

// Velocity pass of one prepared pair: restitution block, then friction with compensating
// normal impulses (static friction also shifts the poses).  Works on registers; returns bit 0
// when the velocities changed and bit 1 when the poses changed.  Mueller et al. 2020 sec. 3.6.
__device__ __forceinline__ int solve_vel_rec(const PairRec& r, float4& pa, float4& va, const float2& wa,
                                             float4& pb, float4& vb, const float2& wb, const StepParams& sp) {
    const int count = r.count;
    if (count == 0) return 0;
    const float l0 = r.l0, l1 = count == 2 ? r.l1 : 0.0f;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return 0;
    const float nx = r.nx, ny = r.ny, mu_s = r.mu_s, mu_d = r.mu_d, rest = r.rest;
    const float ima = wa.x, iia = wa.y, imb = wb.x, iib = wb.y;
    const float ims = ima + imb;
    float rax[2], ray[2], rbx[2], rby[2], dv[2], cna[2], cnb[2], K[2];
    bool act[2];
    act[1] = false; dv[1] = 0.0f; cna[1] = 0.0f; cnb[1] = 0.0f; K[1] = 1.0f;
    rax[1] = ray[1] = rbx[1] = rby[1] = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < count) {
            const float lam = k == 0 ? r.l0 : r.l1;
            const float vpre = r.vpre[k];
            anchor_world(pa, r.la[k][0], r.la[k][1], rax[k], ray[k]);
            anchor_world(pb, r.lb[k][0], r.lb[k][1], rbx[k], rby[k]);
            float vax = va.x - va.z * ray[k], vay = va.y + va.z * rax[k];
            float vbx = vb.x - vb.z * rby[k], vby = vb.y + vb.z * rbx[k];
            float vn = (vbx - vax) * nx + (vby - vay) * ny;
            float e = fabsf(vpre) <= sp.vthr ? 0.0f : rest;
            dv[k] = fmaxf(-e * vpre, 0.0f) - vn;
            cna[k] = rax[k] * ny - ray[k] * nx;
            cnb[k] = rbx[k] * ny - rby[k] * nx;
            K[k] = ims + (iia * (cna[k] * cna[k]) + iib * (cnb[k] * cnb[k]));
            act[k] = lam > 0.0f && K[k] > 0.0f;
        }
    }
    float K01 = 0.0f;
    if (count == 2) K01 = ims + (iia * (cna[0] * cna[1]) + iib * (cnb[0] * cnb[1]));
    float p0, p1;
    kinv_apply(K[0], K[1], K01, act[0], act[1], count, dv[0], dv[1], p0, p1);
    {
        float pt = p0 + p1;
        float cra = cna[0] * p0 + cna[1] * p1, crb = cnb[0] * p0 + cnb[1] * p1;
        va.x = va.x - (nx * pt) * ima;
        va.y = va.y - (ny * pt) * ima;
        va.z = va.z - iia * cra;
        vb.x = vb.x + (nx * pt) * imb;
        vb.y = vb.y + (ny * pt) * imb;
        vb.z = vb.z + iib * crb;
    }
    bool shift = false;
    if (!(sp.flags & SFX_CFG_NO_FRICTION)) {
        float max_, may_, mbx_, mby_;
        if (count == 2) {
            max_ = 0.5f * (rax[0] + rax[1]); may_ = 0.5f * (ray[0] + ray[1]);
            mbx_ = 0.5f * (rbx[0] + rbx[1]); mby_ = 0.5f * (rby[0] + rby[1]);
        } else { max_ = rax[0]; may_ = ray[0]; mbx_ = rbx[0]; mby_ = rby[0]; }
        float vax = va.x - va.z * may_, vay = va.y + va.z * max_;
        float vbx = vb.x - vb.z * mby_, vby = vb.y + vb.z * mbx_;
        float vx = vbx - vax, vy = vby - vay;
        float vn = vx * nx + vy * ny;
        float vtx = vx - nx * vn, vty = vy - ny * vn;
        float vt = sqrtf(vtx * vtx + vty * vty);
        if (vt > SFX_EPS_LEN) {
            float tx = vtx / vt, ty = vty / vt;
            float cta = max_ * ty - may_ * tx, ctb = mbx_ * ty - mby_ * tx;
            float Att = ims + (iia * (cta * cta) + iib * (ctb * ctb));
            float At0 = iia * (cta * cna[0]) + iib * (ctb * cnb[0]);
            float At1 = iia * (cta * cna[1]) + iib * (ctb * cnb[1]);
            float y0, y1;
            kinv_apply(K[0], K[1], K01, act[0], act[1], count, At0, At1, y0, y1);
            float S = Att - (At0 * y0 + At1 * y1);
            if (S > 0.0f) {
                float need = vt / S;
                bool is_static = need * sp.h < mu_s * lt;
                float pt = is_static ? -need : -fminf(mu_d * lt * sp.inv_h, need);
                float q0 = -(y0 * pt), q1 = -(y1 * pt);
                float jx = tx * pt + nx * (q0 + q1), jy = ty * pt + ny * (q0 + q1);
                float ja = (cta * pt + cna[0] * q0) + cna[1] * q1;
                float jb = (ctb * pt + cnb[0] * q0) + cnb[1] * q1;
                va.x = va.x - jx * ima;
                va.y = va.y - jy * ima;
                va.z = va.z - iia * ja;
                vb.x = vb.x + jx * imb;
                vb.y = vb.y + jy * imb;
                vb.z = vb.z + iib * jb;
                if (is_static) {
                    shift = true;
                    if (ima > 0.0f) {
                        pa.x = pa.x - (jx * sp.h) * ima;
                        pa.y = pa.y - (jy * sp.h) * ima;
                    }
                    if (iia > 0.0f) rotate_by(pa.z, pa.w, -(iia * (ja * sp.h)));
                    if (imb > 0.0f) {
                        pb.x = pb.x + (jx * sp.h) * imb;
                        pb.y = pb.y + (jy * sp.h) * imb;
                    }
                    if (iib > 0.0f) rotate_by(pb.z, pb.w, iib * (jb * sp.h));
                }
            }
        }
    }
    return shift ? 3 : 1;
}

// ---- the same two solves with TWO LANES PER PAIR (the tile kernels): lane 2k holds body a of a pair, lane 2k + 1 body b.
// A lane computes what belongs to its own body (anchors, lever arms, angular terms, the update of its pose and velocity)
// and takes the other body's few numbers from its neighbour lane (one DPP move each); what belongs to the pair (depths,
// the 2x2 solve, friction) both lanes compute, on identical inputs, to identical bits.  With the normal SIGNED per lane
// (n for a, -n for b) the code is the same on both sides and equals solve_pos_rec / solve_vel_rec bit for bit: negations
// are exact, sums of two are commutative, and rounding is symmetric in the sign.
__device__ __forceinline__ float other_side(float x) {   // the neighbour lane's value (lanes 2k <-> 2k + 1: quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false));
}

// One side of a prepared pair, in registers.
struct SideRec {
    int count;             // 0, 1 or 2 points
    float nx, ny;          // the normal as this side sees it: from this body to the other
    float mu_s, mu_d, rest;
    float l[2][2];         // this body's local anchors
    float vpre[2];
    float l0, l1;
};

// returns true when the pose changed; r.l0 / r.l1 = the multipliers (the same on both lanes)
__device__ __forceinline__ bool solve_pos_side(SideRec& r, float4& p, const float2& w, float max_corr) {
    const float nx = r.nx, ny = r.ny, im = w.x, ii = w.y;
    float d[2], cn[2];
    d[1] = 0.0f; cn[1] = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < r.count) {
            float rx, ry;
            anchor_world(p, r.l[k][0], r.l[k][1], rx, ry);
            const float qx = p.x + rx, qy = p.y + ry;
            const float ox = other_side(qx), oy = other_side(qy);
            d[k] = (qx - ox) * nx + (qy - oy) * ny;
            d[k] = fminf(d[k], max_corr);
            cn[k] = rx * ny - ry * nx;
        }
    }
    float l0 = 0.0f, l1 = 0.0f;
    const float ims = im + other_side(im);
    const float t00 = ii * (cn[0] * cn[0]);
    const float K00 = ims + (t00 + other_side(t00));
    if (!(K00 > 0.0f)) return false;
    if (r.count == 2) {
        const float t11 = ii * (cn[1] * cn[1]), t01 = ii * (cn[0] * cn[1]);
        const float K11 = ims + (t11 + other_side(t11));
        const float K01 = ims + (t01 + other_side(t01));
        solve_lcp2(K00, K01, K11, d[0], d[1], l0, l1);
    } else {
        l0 = fmaxf(d[0], 0.0f) / K00;
    }
    r.l0 = l0; r.l1 = l1;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return false;
    if (im > 0.0f) {
        p.x = p.x - (nx * lt) * im;
        p.y = p.y - (ny * lt) * im;
    }
    if (ii > 0.0f) rotate_by(p.z, p.w, -(ii * (cn[0] * l0 + cn[1] * l1)));
    return true;
}

// bit 0: the velocity changed, bit 1: the pose changed (static friction)
__device__ __forceinline__ int solve_vel_side(const SideRec& r, float4& p, float4& v, const float2& w, const StepParams& sp) {
    const int count = r.count;
    const float l0 = r.l0, l1 = count == 2 ? r.l1 : 0.0f;
    const float lt = l0 + l1;
    const float nx = r.nx, ny = r.ny, mu_s = r.mu_s, mu_d = r.mu_d, rest = r.rest;
    const float im = w.x, ii = w.y;
    const float ims = im + other_side(im);
    float rx[2], ry[2], dv[2], cn[2], K[2];
    bool act[2];
    act[1] = false; dv[1] = 0.0f; cn[1] = 0.0f; K[1] = 1.0f;
    rx[1] = ry[1] = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k < count) {
            const float lam = k == 0 ? r.l0 : r.l1;
            const float vpre = r.vpre[k];
            anchor_world(p, r.l[k][0], r.l[k][1], rx[k], ry[k]);
            const float ux = v.x - v.z * ry[k], uy = v.y + v.z * rx[k];
            const float ox = other_side(ux), oy = other_side(uy);
            const float vn = (ox - ux) * nx + (oy - uy) * ny;
            const float e = fabsf(vpre) <= sp.vthr ? 0.0f : rest;
            dv[k] = fmaxf(-e * vpre, 0.0f) - vn;
            cn[k] = rx[k] * ny - ry[k] * nx;
            const float tkk = ii * (cn[k] * cn[k]);
            K[k] = ims + (tkk + other_side(tkk));
            act[k] = lam > 0.0f && K[k] > 0.0f;
        }
    }
    float K01 = 0.0f;
    if (count == 2) { const float t01 = ii * (cn[0] * cn[1]); K01 = ims + (t01 + other_side(t01)); }
    float p0, p1;
    kinv_apply(K[0], K[1], K01, act[0], act[1], count, dv[0], dv[1], p0, p1);
    {
        const float pt = p0 + p1;
        const float cr = cn[0] * p0 + cn[1] * p1;
        v.x = v.x - (nx * pt) * im;
        v.y = v.y - (ny * pt) * im;
        v.z = v.z - ii * cr;
    }
    bool shift = false;
    if (!(sp.flags & SFX_CFG_NO_FRICTION)) {
        float mx, my;
        if (count == 2) { mx = 0.5f * (rx[0] + rx[1]); my = 0.5f * (ry[0] + ry[1]); }
        else { mx = rx[0]; my = ry[0]; }
        const float ux = v.x - v.z * my, uy = v.y + v.z * mx;
        const float vx = other_side(ux) - ux, vy = other_side(uy) - uy;
        const float vn = vx * nx + vy * ny;
        const float vtx = vx - nx * vn, vty = vy - ny * vn;
        const float vt = sqrtf(vtx * vtx + vty * vty);
        if (vt > SFX_EPS_LEN) {
            const float tx = vtx / vt, ty = vty / vt;
            const float ct = mx * ty - my * tx;
            const float att = ii * (ct * ct), a0 = ii * (ct * cn[0]), a1 = ii * (ct * cn[1]);
            const float Att = ims + (att + other_side(att));
            const float At0 = a0 + other_side(a0);
            const float At1 = a1 + other_side(a1);
            float y0, y1;
            kinv_apply(K[0], K[1], K01, act[0], act[1], count, At0, At1, y0, y1);
            const float S = Att - (At0 * y0 + At1 * y1);
            if (S > 0.0f) {
                const float need = vt / S;
                const bool is_static = need * sp.h < mu_s * lt;
                const float pt = is_static ? -need : -fminf(mu_d * lt * sp.inv_h, need);
                const float q0 = -(y0 * pt), q1 = -(y1 * pt);
                const float jx = tx * pt + nx * (q0 + q1), jy = ty * pt + ny * (q0 + q1);
                const float jr = (ct * pt + cn[0] * q0) + cn[1] * q1;
                v.x = v.x - jx * im;
                v.y = v.y - jy * im;
                v.z = v.z - ii * jr;
                if (is_static) {
                    shift = true;
                    if (im > 0.0f) {
                        p.x = p.x - (jx * sp.h) * im;
                        p.y = p.y - (jy * sp.h) * im;
                    }
                    if (ii > 0.0f) rotate_by(p.z, p.w, -(ii * (jr * sp.h)));
                }
            }
        }
    }
    return shift ? 3 : 1;
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

__device__ __forceinline__ void solve_joint(const BodyArrays& B, const JointArrays& J, uint32_t k, float inv_h2) {
    const uint2 ab = J.bodies[k];
    const float4 an = J.anchors[k];
    const float4 pr = J.params[k];
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
