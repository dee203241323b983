/*
 * sfx_oracle.c -- CPU oracle for the libsfx physics tick.  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED -- see sfx_oracle.h for why and for what this restates.
 *
 * Sources restated (no reference file:line exists; the reference source is absent):
 *   [XPBD-RB]  Mueller, Macklin, Chentanez, Jeschke, Kim: "Detailed Rigid Body Simulation
 *              with Extended Position Based Dynamics", SCA 2020.  Alg. 2 (substep loop),
 *              sec. 3.3 (positional constraint), 3.5 (contacts + static friction),
 *              3.6 (velocity pass: dynamic friction, restitution).
 *   [XPBD]     Macklin, Mueller, Chentanez: "XPBD", MIG 2016 (compliance term).
 *   [SPEC]     /root/repo/BASELINE.json:5 (kernel list, tolerances), SURVEY.md 8a-8d.
 *   [DESIGN]   DESIGN.md "Contract" (AABB formula, candidate set definition, grid
 *              enumeration order, colouring algorithm, narrowphase conventions).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include "sfx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS_LEN 1e-9f
#define NO_COLOR 0xFFu
#define PRUNED_COLOR 0xFEu /* candidate pair dropped by the reach test: never solved this tick */
#define SFO_FALLBACK 0xFFFFFFFFu

typedef struct { float x, y; } v2;

typedef struct {
    int n;
    float vx[SFX_MAX_POLY_VERTS], vy[SFX_MAX_POLY_VERTS];
    float nx[SFX_MAX_POLY_VERTS], ny[SFX_MAX_POLY_VERTS];
    float r;
} core_t;

/* per candidate pair solver record ([DESIGN] DESIGN.md "Contract / frozen rows"): everything of the pair's constraint
 * rows that depends on geometry and mass only, computed ONCE per substep by the narrowphase at the poses it sees
 * (prepare_pair), so that the solves are a handful of multiply-adds per pair: no anchor is rotated, nothing is divided
 * and no rotation is renormalised inside the sweeps */
typedef struct {
    uint32_t count;     /* 0, 1 or 2 points */
    uint32_t block;     /* count == 2 and both normal rows are solved as a block (K not near-singular) */
    float nx, ny;       /* from a to b */
    float mu_s, mu_d;
    float ca[2], cb[2]; /* generalised lever arms of the normal rows AS THEY ACT on each side: ca = ra x n, cb = -(rb x n) */
    float d0[2];        /* depth at the narrowphase poses */
    float tgt[2];       /* normal velocity the velocity pass asks for: max(-e vn_pre, 0) */
    float iK[2];        /* 1 / K00, 1 / K11 */
    float i00, i01, i11, K01; /* K^-1 (with i01 = K01 / det) and K01 */
    float cta, ctb;     /* lever arms of the tangent row t = (-ny, nx) at the mean anchor, as they act: ctb = -(mb x t) */
    float Att, At[2];
    float invS, invS0, invS1; /* 1 / Schur complement of the tangent row: every counted point active / only point 0 /
                                 only point 1 (0 where it is not positive: no friction) */
    float lam[2];       /* normal lambda of this substep */
} pair_rec;

struct sfo_world {
    sfx_config cfg;
    float cell, inv_cell;
    uint32_t gxl, gyl;
    /* [DESIGN] DESIGN.md "Contract / several colliders per body".  n counts ENTRIES: the nb bodies (entry k < nb is
     * body k with its first collider), then one entry per further collider.  Every per-entry array below has n
     * elements; an entry e >= nb MIRRORS the pose and velocity of its body cbody[e] (mirror_entries) and has no mass. */
    uint32_t n, nb;
    uint32_t* cbody; /* [n] the body an entry belongs to; identity for e < nb */
    float *pos, *rot, *vel, *angvel, *inv_mass, *inv_inertia, *prev_pos, *prev_rot;
    sfx_collider* col;
    float* lrot;   /* (cos, sin) of every collider's local_angle, computed once at upload */
    float *poly, *poly_nrm;
    uint32_t n_poly;
    sfx_material* mats;
    uint32_t n_mats;
    /* joints */
    uint32_t m;
    sfx_distance_constraint* joints; /* m_dist of them; the joint list is these, then the angle constraints */
    sfx_angle_constraint* angles;
    uint32_t m_dist, m_angle;
    uint32_t* joint_perm; /* colour-major order */
    uint8_t* joint_color;
    uint32_t n_joint_colors;
    /* per tick */
    float* aabb;
    uint32_t n_small, n_large;
    uint32_t max_pairs, n_pairs;
    uint32_t *pa, *pb; /* emission order */
    int32_t *ccx, *ccy;  /* home cell of every small collider (last broadphase) */
    uint8_t* is_small;   /* sorted into the grid (else: large list or no collider) */
    uint32_t tlx, tly;   /* log2 of the tile width / height in cells */
    uint8_t* color;      /* colour of the pair inside its batch (tile or fallback list) */
    uint32_t* owner;     /* wrapped owner tile of the pair, SFO_FALLBACK for the fallback list */
    uint8_t* pclass;     /* class of a tile-owned pair: 0 interior, 1 x seam, 2 y seam, 3 corner */
    uint32_t n_colors, n_leftover, overflow;
    uint32_t n_tiles, n_fallback, max_tile_pairs, max_tile_bodies, n_solver_pairs;
    uint32_t color_count[SFX_MAX_COLORS];
    uint32_t* order;   /* processing order (indices into pa/pb) */
    uint32_t* batch_end; /* order[] is cut into batches: batch k = [batch_end[k-1], batch_end[k]) */
    uint32_t n_batches, n_order;
    int order_valid;
    int tick_open, tick_order;
    uint32_t tick_nbatch;
    uint32_t phase_end[5]; /* batches of phase ph (4 = the fallback list) = [phase_end[ph-1], phase_end[ph]): the units of one
                            * phase share no dynamic body (the OpenMP build solves them side by side) */
    float tick_h;
    float* dpos;   /* [3 nb] (dx, dy, dtheta) the position sweep of this substep has accumulated per body; applied once, before the derive */
    float* dshift; /* [3 nb] static-friction pose shifts of the running velocity phase; applied at the end of the phase */
    pair_rec* rec;
    uint32_t n_contacts, n_manifolds, n_manifolds2;
    float max_depth;
};

/* ------------------------------------------------------------------ config */

static void resolve_defaults(sfx_config* c) {
    if (c->substeps == 0) c->substeps = 1;
    if (c->color_rounds == 0) c->color_rounds = 12;
    if (c->color_rounds > 60) c->color_rounds = 60;
    /* tiles are at least two cells wide and high: the seams of one class then never share a body (tile order below) */
    if (c->tile_cells == 0 || c->tile_cells == SFX_TILE_WHOLE_GRID) { /* automatic / one tile: resolve_grid */ }
    else if (c->tile_cells >= 8) c->tile_cells = 8;
    else if (c->tile_cells >= 4) c->tile_cells = 4;
    else c->tile_cells = 2;
    if (c->restitution_threshold == 0.0f) c->restitution_threshold = 2.0f;
    if (c->vel_pad_max == 0.0f) c->vel_pad_max = 0.03f;
    if (!(c->max_depenetration_velocity > 0.0f)) c->max_depenetration_velocity = INFINITY;
    if (c->vel_pad_max < 0.0f) c->vel_pad_max = 0.0f;
    if (c->speculative_margin == 0.0f) c->speculative_margin = 0.010f;
    if (c->speculative_margin < 0.0f) c->speculative_margin = 0.0f;
}

sfo_world* sfo_create(const sfx_config* cfg) {
    sfo_world* w = (sfo_world*)calloc(1, sizeof(sfo_world));
    if (!w) return NULL;
    w->cfg = *cfg;
    resolve_defaults(&w->cfg);
    return w;
}

void sfo_destroy(sfo_world* w) {
    if (!w) return;
    free(w->pos); free(w->rot); free(w->vel); free(w->angvel); free(w->inv_mass);
    free(w->inv_inertia); free(w->prev_pos); free(w->prev_rot); free(w->col); free(w->lrot); free(w->poly);
    free(w->poly_nrm); free(w->mats); free(w->joints); free(w->angles); free(w->joint_perm); free(w->joint_color);
    free(w->aabb); free(w->pa); free(w->pb); free(w->color); free(w->order); free(w->rec);
    free(w->ccx); free(w->ccy); free(w->is_small); free(w->owner); free(w->pclass); free(w->batch_end);
    free(w->cbody); free(w->dpos); free(w->dshift);
    free(w);
}

static float* dupf(const float* src, size_t n) {
    float* p = (float*)malloc((n ? n : 1) * sizeof(float));
    if (p && src && n) memcpy(p, src, n * sizeof(float));
    return p;
}

/* entries beyond the bodies take the pose and the velocity of their body (and never a mass): after an upload, after
 * every integration (+ joints), at the end of every substep */
static void mirror_entries(sfo_world* w) {
    for (uint32_t e = w->nb; e < w->n; ++e) {
        uint32_t b = w->cbody[e];
        w->pos[2 * e] = w->pos[2 * b]; w->pos[2 * e + 1] = w->pos[2 * b + 1];
        w->rot[2 * e] = w->rot[2 * b]; w->rot[2 * e + 1] = w->rot[2 * b + 1];
        w->vel[2 * e] = w->vel[2 * b]; w->vel[2 * e + 1] = w->vel[2 * b + 1];
        w->angvel[e] = w->angvel[b];
    }
}

static float* growf(float* p, size_t n_old, size_t n_new) {
    float* q = (float*)realloc(p, (n_new ? n_new : 1) * sizeof(float));
    if (q && n_new > n_old) memset(q + n_old, 0, (n_new - n_old) * sizeof(float));
    return q;
}

/* the per-entry arrays get n_new elements; the bodies' part is kept */
static void resize_entries(sfo_world* w, uint32_t n_new) {
    uint32_t n_old = w->n;
    w->pos = growf(w->pos, 2u * n_old, 2u * n_new); w->rot = growf(w->rot, 2u * n_old, 2u * n_new);
    w->vel = growf(w->vel, 2u * n_old, 2u * n_new); w->angvel = growf(w->angvel, n_old, n_new);
    w->inv_mass = growf(w->inv_mass, n_old, n_new); w->inv_inertia = growf(w->inv_inertia, n_old, n_new);
    w->prev_pos = growf(w->prev_pos, 2u * n_old, 2u * n_new); w->prev_rot = growf(w->prev_rot, 2u * n_old, 2u * n_new);
    w->cbody = (uint32_t*)realloc(w->cbody, (n_new ? n_new : 1) * sizeof(uint32_t));
    for (uint32_t e = w->nb; e < n_new; ++e) { w->inv_mass[e] = 0.0f; w->inv_inertia[e] = 0.0f; }
    w->n = n_new;
}

int sfo_upload_bodies(sfo_world* w, uint32_t n, const float* pos, const float* rot,
                      const float* vel, const float* angvel, const float* inv_mass,
                      const float* inv_inertia) {
    if (w->col && n == w->nb) { /* same bodies, new state: colliders (and their entries) stay */
        memcpy(w->pos, pos, 2u * n * sizeof(float)); memcpy(w->rot, rot, 2u * n * sizeof(float));
        memcpy(w->vel, vel, 2u * n * sizeof(float)); memcpy(w->angvel, angvel, n * sizeof(float));
        memcpy(w->inv_mass, inv_mass, n * sizeof(float)); memcpy(w->inv_inertia, inv_inertia, n * sizeof(float));
        memcpy(w->prev_pos, pos, 2u * n * sizeof(float)); memcpy(w->prev_rot, rot, 2u * n * sizeof(float));
        mirror_entries(w);
        return SFX_OK;
    }
    free(w->pos); free(w->rot); free(w->vel); free(w->angvel); free(w->inv_mass);
    free(w->inv_inertia); free(w->prev_pos); free(w->prev_rot); free(w->cbody);
    free(w->col); w->col = NULL; /* another body count: the colliders are uploaded again */
    w->n = w->nb = n;
    w->pos = dupf(pos, 2u * n); w->rot = dupf(rot, 2u * n); w->vel = dupf(vel, 2u * n);
    w->angvel = dupf(angvel, n); w->inv_mass = dupf(inv_mass, n);
    w->inv_inertia = dupf(inv_inertia, n);
    w->prev_pos = dupf(pos, 2u * n); w->prev_rot = dupf(rot, 2u * n);
    w->cbody = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i) w->cbody[i] = i;
    return SFX_OK;
}

static int is_dynamic(const sfo_world* w, uint32_t i) {
    return w->inv_mass[i] > 0.0f || w->inv_inertia[i] > 0.0f;
}

/* of an entry: is the body it belongs to dynamic (an entry beyond the bodies has no mass of its own) */
static int owner_dynamic(const sfo_world* w, uint32_t e) { return is_dynamic(w, w->cbody[e]); }

/* a further collider of a DYNAMIC body: its pairs are solved on the body, not on the entry (build_order: tile_end_ok) */
static int dynamic_secondary(const sfo_world* w, uint32_t e) { return e >= w->nb && owner_dynamic(w, e); }

/* [DESIGN] bounding diameter of a shape, used only for the automatic cell size */
static float shape_diameter(const sfo_world* w, const sfx_collider* c) {
    switch (c->type) {
    case SFX_SHAPE_CIRCLE: return 2.0f * c->p[2];
    case SFX_SHAPE_RECT: return 2.0f * (sqrtf(c->p[0] * c->p[0] + c->p[1] * c->p[1]) + c->p[2]);
    case SFX_SHAPE_CAPSULE: return 2.0f * (c->p[0] + c->p[2]);
    case SFX_SHAPE_POLY: {
        float m = 0.0f;
        for (uint32_t k = 0; k < c->poly_count; ++k) {
            float x = w->poly[2 * (c->poly_offset + k)], y = w->poly[2 * (c->poly_offset + k) + 1];
            float d = sqrtf(x * x + y * y);
            if (d > m) m = d;
        }
        return 2.0f * (m + c->p[2]);
    }
    default: return 0.0f;
    }
}

static int32_t cell_coord(float lo, float inv_cell) {
    float f = floorf(lo * inv_cell);
    if (!(f > -1073741824.0f)) f = -1073741824.0f;
    if (f > 1073741824.0f) f = 1073741824.0f;
    return (int32_t)f;
}

static uint32_t ceil_log2(uint32_t v) {
    uint32_t l = 0;
    while ((1u << l) < v && l < 31) ++l;
    return l;
}

/* [DESIGN] DESIGN.md "Contract / grid": cell = 1.25 * (largest dynamic diameter + 2 (margin + vel_pad_max));
 * grid = smallest square power-of-two torus with >= 2 cells per body, (4 tiles)..4096 per side. */
static void resolve_grid(sfo_world* w) {
    float cell = w->cfg.cell_size;
    if (!(cell > 0.0f)) {
        float dmax = 0.0f;
        for (uint32_t i = 0; i < w->n; ++i) {
            if (!owner_dynamic(w, i) || w->col[i].type == SFX_SHAPE_NONE) continue;
            float d = shape_diameter(w, &w->col[i]);
            if (d > dmax) dmax = d;
        }
        cell = dmax > 0.0f ? 1.25f * (dmax + 2.0f * (w->cfg.aabb_margin + w->cfg.vel_pad_max)) : 1.0f;
    }
    w->cell = cell;
    w->inv_cell = 1.0f / cell;
    uint32_t gx = w->cfg.grid_log2[0], gy = w->cfg.grid_log2[1];
    if (gx == 0 || gy == 0) {
        uint32_t side = (uint32_t)ceil(sqrt(2.0 * (double)(w->n ? w->n : 1)));
        uint32_t l = ceil_log2(side);
        if (l > 12) l = 12;
        gx = gy = l;
        /* [DESIGN] the 2^(2l) buckets take the shape of the scene: a wide, shallow pile on a square torus wraps
         * round it and stacks several cell columns into one (denser tiles, for nothing).  Extent of the dynamic
         * colliders' centres at upload, in cells, rounded up to powers of two; then at least 2^(2l) buckets (the
         * shorter side grows) and at most four times as many (the longer side shrinks). */
        float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
        for (uint32_t i = 0; i < w->n; ++i) {
            if (!owner_dynamic(w, i) || w->col[i].type == SFX_SHAPE_NONE) continue;
            float x = w->pos[2 * i], y = w->pos[2 * i + 1];
            if (x < xmin) xmin = x;
            if (x > xmax) xmax = x;
            if (y < ymin) ymin = y;
            if (y > ymax) ymax = y;
        }
        if (xmax >= xmin && ymax >= ymin) {
            float ex = floorf((xmax - xmin) * w->inv_cell) + 3.0f, ey = floorf((ymax - ymin) * w->inv_cell) + 3.0f;
            if (!(ex < 8192.0f)) ex = 8192.0f;
            if (!(ey < 8192.0f)) ey = 8192.0f;
            uint32_t bx = ceil_log2((uint32_t)ex), by = ceil_log2((uint32_t)ey), tot = 2u * l;
            if (bx < 2) bx = 2;
            if (bx > 12) bx = 12;
            if (by < 2) by = 2;
            if (by > 12) by = 12;
            while (bx + by < tot) { if (by <= bx && by < 12) ++by; else if (bx < 12) ++bx; else break; }
            while (bx + by > tot + 2u) { if (bx >= by && bx > 2) --bx; else if (by > 2) --by; else break; }
            gx = bx; gy = by;
        }
    }
    /* at least 4 tiles per side, so that the 2x2 tile blocks of one phase never meet on the torus */
    w->tlx = w->cfg.tile_cells == 8 ? 3u : w->cfg.tile_cells == 4 ? 2u : 1u;
    if (gx < w->tlx + 2) gx = w->tlx + 2;
    if (gy < 3) gy = 3;
    /* tile height: as wide by default; a power of two; the grid height at most (= strips) */
    w->tly = w->tlx;
    if (w->cfg.tile_cells_y) {
        w->tly = 1;
        while (w->tly < 31 && (2u << w->tly) <= w->cfg.tile_cells_y) w->tly++;
    } else {
        /* automatic: colliders per tile column (centre x, wrapped like the cells) at upload time */
        uint32_t ncol = 1u << (gx - w->tlx), worst = 0;
        uint32_t* cnt = (uint32_t*)calloc(ncol, sizeof(uint32_t));
        for (uint32_t i = 0; i < w->n; ++i) {
            if (w->col[i].type == SFX_SHAPE_NONE) continue;
            int32_t cx = cell_coord(w->pos[2 * i], w->inv_cell);
            cnt[((uint32_t)cx >> w->tlx) & (ncol - 1u)]++;   /* two's complement: >> then & wraps negatives too */
        }
        for (uint32_t c = 0; c < ncol; ++c) {
            uint32_t two = cnt[c] + cnt[(c + 1u) & (ncol - 1u)];
            if (two > worst) worst = two;
        }
        free(cnt);
        if (worst <= 7u * SFX_TILE_MAX_BODIES / 8u) w->tly = gy;
    }
    if (w->tly > gy) w->tly = gy;
    if (w->tly < gy && gy < w->tly + 2) w->tly = gy; /* fewer than four tile rows: one row (strips) */
    if (gx > 12) gx = 12;
    if (gy > 12) gy = 12;
    /* [DESIGN] one tile for the whole grid (DESIGN.md "Contract / tiles"): asked for, or automatic for small worlds --
     * every pair is then an interior pair of the one unit: one colouring, no seam phase */
    /* (automatic only while the colliders that may be too large for a base cell -- the ones a tile's batch keeps in its
     * SFX_ONE_TILE_LARGE outside-body slots -- all fit one batch: a level of many platforms stays tiled) */
    uint32_t nshape = 0, nbig = 0;
    for (uint32_t i = 0; i < w->n; ++i) {
        if (w->col[i].type == SFX_SHAPE_NONE) continue;
        nshape++;
        if (!(shape_diameter(w, &w->col[i]) + 2.0f * (w->cfg.aabb_margin + w->cfg.vel_pad_max) <= 0.9f * cell)) nbig++;
    }
    if (w->cfg.tile_cells == SFX_TILE_WHOLE_GRID ||
        (w->cfg.tile_cells == 0 && w->cfg.tile_cells_y == 0 && nshape <= SFX_ONE_TILE_COLLIDERS && nbig <= SFX_ONE_TILE_LARGE)) {
        w->tlx = gx; w->tly = gy;
    }
    w->gxl = gx; w->gyl = gy;
}

int sfo_upload_colliders(sfo_world* w, uint32_t n, const sfx_collider* colliders,
                         const float* poly_verts, uint32_t n_poly_verts) {
    /* collider k < nb is body k's first; a collider k >= nb belongs to colliders[k].body */
    if (n < w->nb) return SFX_ERR_INVALID;
    for (uint32_t k = w->nb; k < n; ++k)
        if (colliders[k].body >= w->nb) return SFX_ERR_INVALID;
    resize_entries(w, n);
    for (uint32_t k = 0; k < n; ++k) w->cbody[k] = k < w->nb ? k : colliders[k].body;
    mirror_entries(w);
    free(w->col); free(w->poly); free(w->poly_nrm); free(w->lrot);
    w->col = (sfx_collider*)malloc((n ? n : 1) * sizeof(sfx_collider));
    memcpy(w->col, colliders, n * sizeof(sfx_collider));
    w->lrot = (float*)malloc((n ? n : 1) * 2 * sizeof(float));
    for (uint32_t i = 0; i < n; ++i) { /* the only trigonometry of the contract: once, on the host, at upload */
        w->lrot[2 * i] = w->col[i].local_angle == 0.0f ? 1.0f : cosf(w->col[i].local_angle);
        w->lrot[2 * i + 1] = w->col[i].local_angle == 0.0f ? 0.0f : sinf(w->col[i].local_angle);
    }
    w->n_poly = n_poly_verts;
    w->poly = dupf(poly_verts, 2u * n_poly_verts);
    w->poly_nrm = (float*)calloc(2u * n_poly_verts + 1, sizeof(float));
    for (uint32_t i = 0; i < n; ++i) {
        const sfx_collider* c = &w->col[i];
        if (c->type != SFX_SHAPE_POLY) continue;
        if (c->poly_count < 3 || c->poly_count > SFX_MAX_POLY_VERTS ||
            c->poly_offset + c->poly_count > n_poly_verts) return SFX_ERR_INVALID;
        for (uint32_t k = 0; k < c->poly_count; ++k) {
            uint32_t k0 = c->poly_offset + k, k1 = c->poly_offset + (k + 1) % c->poly_count;
            float ex = w->poly[2 * k1] - w->poly[2 * k0], ey = w->poly[2 * k1 + 1] - w->poly[2 * k0 + 1];
            float len = sqrtf(ex * ex + ey * ey);
            w->poly_nrm[2 * k0] = ey / len;
            w->poly_nrm[2 * k0 + 1] = -ex / len;
        }
    }
    resolve_grid(w);
    free(w->aabb); free(w->pa); free(w->pb); free(w->color); free(w->order); free(w->rec);
    free(w->ccx); free(w->ccy); free(w->is_small); free(w->owner); free(w->pclass); free(w->batch_end);
    w->max_pairs = w->cfg.max_pairs ? w->cfg.max_pairs : 12u * n;
    if (w->max_pairs < 1024u) w->max_pairs = 1024u;
    w->aabb = (float*)calloc(4u * n + 4, sizeof(float));
    w->pa = (uint32_t*)malloc(w->max_pairs * sizeof(uint32_t));
    w->pb = (uint32_t*)malloc(w->max_pairs * sizeof(uint32_t));
    w->color = (uint8_t*)malloc(w->max_pairs);
    w->order = (uint32_t*)malloc(w->max_pairs * sizeof(uint32_t));
    w->rec = (pair_rec*)calloc(w->max_pairs, sizeof(pair_rec));
    free(w->dpos); free(w->dshift);
    w->dpos = (float*)calloc(3u * (size_t)w->nb + 3, sizeof(float));
    w->dshift = (float*)calloc(3u * (size_t)w->nb + 3, sizeof(float));
    w->owner = (uint32_t*)malloc(w->max_pairs * sizeof(uint32_t));
    w->pclass = (uint8_t*)calloc(w->max_pairs, 1);
    w->batch_end = (uint32_t*)malloc(((size_t)w->max_pairs + 2) * sizeof(uint32_t));
    w->ccx = (int32_t*)calloc(n ? n : 1, sizeof(int32_t));
    w->ccy = (int32_t*)calloc(n ? n : 1, sizeof(int32_t));
    w->is_small = (uint8_t*)calloc(n ? n : 1, 1);
    w->n_pairs = 0; w->order_valid = 0;
    return SFX_OK;
}

int sfo_upload_materials(sfo_world* w, uint32_t n, const sfx_material* materials) {
    free(w->mats);
    w->mats = (sfx_material*)malloc((n ? n : 1) * sizeof(sfx_material));
    memcpy(w->mats, materials, n * sizeof(sfx_material));
    w->n_mats = n;
    return SFX_OK;
}

void sfo_resolved_config(const sfo_world* w, sfx_config* out) {
    *out = w->cfg;
    out->cell_size = w->cell;
    out->grid_log2[0] = w->gxl;
    out->grid_log2[1] = w->gyl;
    out->max_pairs = w->max_pairs;
    out->tile_cells = (w->tlx == w->gxl && w->tly == w->gyl) ? SFX_TILE_WHOLE_GRID : 1u << w->tlx;
    out->tile_cells_y = 1u << w->tly;
}

/* ------------------------------------------------------------------ colouring
 * [DESIGN] DESIGN.md "Contract / colouring".  Edge colouring of a constraint graph: two
 * constraints that share a dynamic body never get the same colour.  `rounds` parallel rounds
 * (each uncoloured edge proposes the lowest colour free at both ends; per (body, colour mod 4)
 * the lowest hashed priority of the round wins; an edge is coloured when it wins at both
 * dynamic ends), then a serial first-fit finish in ascending key order.
 * Priorities and hashes depend only on the edge's key (u, v) -- the two body ids of a
 * candidate pair, or the index of a joint -- never on its place in a list, so the result is
 * the same for every enumeration order of the same edge set. */

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

static uint32_t edge_hash(uint32_t u, uint32_t v, uint32_t r) {
    return mix32((u * 0x9E3779B1u) ^ mix32(v + r * 0x7F4A7C15u + 0x165667B1u));
}

/* 6 bits of (63 - round) | 10 bits of hash | 24 + 24 bits of key: unique per edge and round */
static uint64_t edge_prio(uint32_t u, uint32_t v, uint32_t r, uint32_t h) {
    return ((uint64_t)(63u - r) << 58) | ((uint64_t)(h >> 22) << 48) |
           ((uint64_t)(u & 0xFFFFFFu) << 24) | (uint64_t)(v & 0xFFFFFFu);
}

typedef struct {
    uint64_t* used;  /* [n_bodies] colours taken, all 0 between calls   */
    uint64_t* claim; /* [n_bodies][4], all ~0 between calls             */
} color_scratch;

static int color_scratch_init(color_scratch* cs, uint32_t n_bodies) {
    size_t n = n_bodies ? n_bodies : 1;
    cs->used = (uint64_t*)calloc(n, sizeof(uint64_t));
    cs->claim = (uint64_t*)malloc(n * 4 * sizeof(uint64_t));
    if (!cs->used || !cs->claim) return -1;
    memset(cs->claim, 0xFF, n * 4 * sizeof(uint64_t));
    return 0;
}

static void color_scratch_free(color_scratch* cs) { free(cs->used); free(cs->claim); }

#define SFO_LEFT_CAP 4096u /* = kLeftCap of the device code */
typedef struct { uint64_t key; uint32_t idx; } key_idx;
static int cmp_key_idx(const void* a, const void* b) {
    const key_idx *x = (const key_idx*)a, *y = (const key_idx*)b;
    return x->key < y->key ? -1 : x->key > y->key;
}

/* Colours the ne edges idx[0..ne) of (ea, eb); keys (ku[e], kv[e]).  dyn[] says which bodies
 * are dynamic (SFX_WORLD_BODY ends never are).  Returns the number of colours; *leftover
 * counts the edges the serial finish took; *overflow is set when an edge found no free colour
 * (it stays NO_COLOR and is dropped from the solve).  The scratch tables are left clean. */
static uint32_t color_edges(color_scratch* cs, uint32_t ne, const uint32_t* idx, const uint32_t* ea,
                            const uint32_t* eb, const uint32_t* ku, const uint32_t* kv, const uint8_t* dyn,
                            uint32_t rounds, uint8_t* color, uint32_t* leftover, uint32_t* overflow) {
    uint64_t *used = cs->used, *claim = cs->claim;
    uint8_t* prop = (uint8_t*)malloc(ne ? ne : 1);
    uint64_t* prio = (uint64_t*)malloc((ne ? ne : 1) * sizeof(uint64_t));
#define DYN(x) ((x) != SFX_WORLD_BODY && dyn[(x)])
    for (uint32_t q = 0; q < ne; ++q) {
        uint32_t e = idx ? idx[q] : q;
        color[e] = NO_COLOR;
    }
    for (uint32_t r = 0; r < rounds; ++r) {
        for (uint32_t q = 0; q < ne; ++q) { /* propose */
            uint32_t e = idx ? idx[q] : q;
            if (color[e] != NO_COLOR) continue;
            uint32_t a = ea[e], b = eb[e];
            int da = DYN(a), db = DYN(b);
            uint64_t freem = ~((da ? used[a] : 0) | (db ? used[b] : 0));
            if (freem == 0) { prop[q] = NO_COLOR; *overflow = 1; continue; }
            uint32_t c = (uint32_t)__builtin_ctzll(freem); /* first fit: the lowest colour free at both ends */
            uint32_t h = edge_hash(ku[e], kv[e], r);
            prop[q] = (uint8_t)c;
            uint64_t p = edge_prio(ku[e], kv[e], r, h);
            prio[q] = p;
            if (da && p < claim[(size_t)a * 4 + (c & 3)]) claim[(size_t)a * 4 + (c & 3)] = p;
            if (db && p < claim[(size_t)b * 4 + (c & 3)]) claim[(size_t)b * 4 + (c & 3)] = p;
        }
        for (uint32_t q = 0; q < ne; ++q) { /* resolve */
            uint32_t e = idx ? idx[q] : q;
            if (color[e] != NO_COLOR || prop[q] == NO_COLOR) continue;
            uint32_t a = ea[e], b = eb[e], c = prop[q];
            int da = DYN(a), db = DYN(b);
            int win = (!da || claim[(size_t)a * 4 + (c & 3)] == prio[q]) &&
                      (!db || claim[(size_t)b * 4 + (c & 3)] == prio[q]);
            if (!win) continue;
            color[e] = (uint8_t)c;
            if (da) used[a] |= 1ull << c;
            if (db) used[b] |= 1ull << c;
        }
    }
    /* serial first-fit finish, ascending key */
    uint32_t nl = 0;
    key_idx* left = (key_idx*)malloc((ne ? ne : 1) * sizeof(key_idx));
    for (uint32_t q = 0; q < ne; ++q) {
        uint32_t e = idx ? idx[q] : q;
        if (color[e] != NO_COLOR) continue;
        left[nl].key = ((uint64_t)(ku[e] & 0xFFFFFFu) << 24) | (kv[e] & 0xFFFFFFu);
        left[nl].idx = e;
        ++nl;
    }
    qsort(left, nl, sizeof(key_idx), cmp_key_idx);
    if (nl > SFO_LEFT_CAP) *overflow = 1; /* [DESIGN] the serial finish of one list takes at most this many edges (the device's
                                            * table); more is a reported capacity overflow, like a 65th colour */
    for (uint32_t q = 0; q < nl; ++q) {
        uint32_t e = left[q].idx, a = ea[e], b = eb[e];
        int da = DYN(a), db = DYN(b);
        uint64_t freem = ~((da ? used[a] : 0) | (db ? used[b] : 0));
        if (freem == 0) { *overflow = 1; continue; }
        uint32_t c = (uint32_t)__builtin_ctzll(freem);
        color[e] = (uint8_t)c;
        if (da) used[a] |= 1ull << c;
        if (db) used[b] |= 1ull << c;
    }
    *leftover += nl;
    uint32_t nc = 0;
    for (uint32_t q = 0; q < ne; ++q) {
        uint32_t e = idx ? idx[q] : q, a = ea[e], b = eb[e];
        if (color[e] != NO_COLOR && color[e] + 1u > nc) nc = color[e] + 1u;
        if (DYN(a)) { used[a] = 0; memset(&claim[(size_t)a * 4], 0xFF, 4 * sizeof(uint64_t)); }
        if (DYN(b)) { used[b] = 0; memset(&claim[(size_t)b * 4], 0xFF, 4 * sizeof(uint64_t)); }
    }
#undef DYN
    free(prop); free(prio); free(left);
    return nc;
}

/* colours the joint list = distance constraints, then angle constraints (sfx.h), keyed by the
 * index in that list */
static int rebuild_joints(sfo_world* w) {
    uint32_t md = w->m_dist, m = md + w->m_angle;
    free(w->joint_perm); free(w->joint_color);
    w->m = m;
    w->joint_perm = (uint32_t*)malloc((m ? m : 1) * sizeof(uint32_t));
    w->joint_color = (uint8_t*)malloc(m ? m : 1);
    uint32_t* ea = (uint32_t*)malloc((m ? m : 1) * sizeof(uint32_t));
    uint32_t* eb = (uint32_t*)malloc((m ? m : 1) * sizeof(uint32_t));
    uint32_t* ku = (uint32_t*)malloc((m ? m : 1) * sizeof(uint32_t));
    uint32_t* kv = (uint32_t*)malloc((m ? m : 1) * sizeof(uint32_t));
    uint8_t* dyn = (uint8_t*)malloc(w->n ? w->n : 1);
    for (uint32_t i = 0; i < w->n; ++i) dyn[i] = (uint8_t)is_dynamic(w, i);
    for (uint32_t k = 0; k < m; ++k) {
        ea[k] = k < md ? w->joints[k].body_a : w->angles[k - md].body_a;
        eb[k] = k < md ? w->joints[k].body_b : w->angles[k - md].body_b;
        ku[k] = k >> 24; kv[k] = k & 0xFFFFFFu; /* joints are keyed by their index */
    }
    uint32_t left = 0, ovf = 0;
    color_scratch cs;
    color_scratch_init(&cs, w->n);
    w->n_joint_colors = color_edges(&cs, m, NULL, ea, eb, ku, kv, dyn, w->cfg.color_rounds, w->joint_color, &left, &ovf);
    color_scratch_free(&cs);
    /* colour-major, stable */
    uint32_t pos = 0;
    for (uint32_t col = 0; col < w->n_joint_colors; ++col)
        for (uint32_t k = 0; k < m; ++k)
            if (w->joint_color[k] == col) w->joint_perm[pos++] = k;
    for (uint32_t k = 0; k < m; ++k) /* uncoloured (colour overflow): listed last, never solved */
        if (w->joint_color[k] == NO_COLOR) w->joint_perm[pos++] = k;
    free(ea); free(eb); free(ku); free(kv); free(dyn);
    return ovf ? SFX_ERR_CAPACITY : SFX_OK;
}

static int body_ok(const sfo_world* w, uint32_t b) { return b == SFX_WORLD_BODY || b < w->nb; }

int sfo_upload_constraints(sfo_world* w, uint32_t m, const sfx_distance_constraint* c) {
    for (uint32_t k = 0; k < m; ++k) {
        if (!body_ok(w, c[k].body_a) || !body_ok(w, c[k].body_b) || c[k].limit > SFX_LIMIT_STRUT) return SFX_ERR_INVALID;
        if (!(c[k].damping >= 0.0f) || (c[k].damping > 0.0f && c[k].limit != SFX_LIMIT_NONE)) return SFX_ERR_INVALID;
    }
    free(w->joints);
    w->m_dist = m;
    w->joints = (sfx_distance_constraint*)malloc((m ? m : 1) * sizeof(*c));
    memcpy(w->joints, c, m * sizeof(*c));
    return rebuild_joints(w);
}

int sfo_upload_angle_constraints(sfo_world* w, uint32_t m, const sfx_angle_constraint* c) {
    const float pi = 3.14159265358979323846f;
    for (uint32_t k = 0; k < m; ++k) {
        if (!body_ok(w, c[k].body_a) || !body_ok(w, c[k].body_b)) return SFX_ERR_INVALID;
        if (!(c[k].min_angle > -pi && c[k].min_angle <= c[k].max_angle && c[k].max_angle < pi)) return SFX_ERR_INVALID;
        if (!(c[k].damping >= 0.0f) || !(c[k].compliance >= 0.0f)) return SFX_ERR_INVALID;
    }
    free(w->angles);
    w->m_angle = m;
    w->angles = (sfx_angle_constraint*)malloc((m ? m : 1) * sizeof(*c));
    memcpy(w->angles, c, m * sizeof(*c));
    return rebuild_joints(w);
}

/* ------------------------------------------------------------------ collider pose
 * [DESIGN] DESIGN.md "Contract / collider pose" (SURVEY.md section 8a row a4 "local offset").  The shape of body i
 * sits at local_pos, turned by local_angle, in the body's frame.  A collider without a local pose (both zero)
 * has the body's pose, bit for bit. */
static int has_local(const sfo_world* w, uint32_t i) {
    const sfx_collider* c = &w->col[i];
    return c->local_pos[0] != 0.0f || c->local_pos[1] != 0.0f || c->local_angle != 0.0f;
}

static void collider_pose(const sfo_world* w, uint32_t i, float* px, float* py, float* rc, float* rs) {
    float bx = w->pos[2 * i], by = w->pos[2 * i + 1], bc = w->rot[2 * i], bs = w->rot[2 * i + 1];
    if (!has_local(w, i)) { *px = bx; *py = by; *rc = bc; *rs = bs; return; }
    float ox = w->col[i].local_pos[0], oy = w->col[i].local_pos[1], lc = w->lrot[2 * i], ls = w->lrot[2 * i + 1];
    *px = bx + (bc * ox - bs * oy);
    *py = by + (bs * ox + bc * oy);
    *rc = bc * lc - bs * ls;
    *rs = bs * lc + bc * ls;
}

/* how far the collider may travel per axis before the next broadphase: the body's speed, plus -- for a shape off the
 * body origin -- what the spin adds at that distance */
static void motion_pads(const sfo_world* w, uint32_t i, float dt, float margin, float vel_pad_max, float* padx, float* pady) {
    float sx = fabsf(w->vel[2 * i]), sy = fabsf(w->vel[2 * i + 1]);
    if (has_local(w, i)) {
        float ox = w->col[i].local_pos[0], oy = w->col[i].local_pos[1];
        float swing = fabsf(w->angvel[i]) * sqrtf(ox * ox + oy * oy);
        sx = sx + swing; sy = sy + swing;
    }
    *padx = margin + fminf(sx * dt, vel_pad_max);
    *pady = margin + fminf(sy * dt, vel_pad_max);
}

/* ------------------------------------------------------------------ AABB
 * [DESIGN] DESIGN.md "Contract / AABB".  Operation order is part of the contract. */
static void compute_aabb_ex(const sfo_world* w, uint32_t i, float dt, float margin, float vel_pad_max, float* out) {
    const sfx_collider* c = &w->col[i];
    float px, py, rc, rs;
    collider_pose(w, i, &px, &py, &rc, &rs);
    float minx, maxx, miny, maxy;
    switch (c->type) {
    case SFX_SHAPE_CIRCLE:
        minx = -c->p[2]; maxx = c->p[2]; miny = -c->p[2]; maxy = c->p[2];
        break;
    case SFX_SHAPE_RECT: {
        float ex = (fabsf(rc * c->p[0]) + fabsf(rs * c->p[1])) + c->p[2];
        float ey = (fabsf(rs * c->p[0]) + fabsf(rc * c->p[1])) + c->p[2];
        minx = -ex; maxx = ex; miny = -ey; maxy = ey;
        break;
    }
    case SFX_SHAPE_CAPSULE: {
        float ex = fabsf(rc * c->p[0]) + c->p[2];
        float ey = fabsf(rs * c->p[0]) + c->p[2];
        minx = -ex; maxx = ex; miny = -ey; maxy = ey;
        break;
    }
    case SFX_SHAPE_POLY: {
        minx = miny = INFINITY; maxx = maxy = -INFINITY;
        for (uint32_t k = 0; k < c->poly_count; ++k) {
            float lx = w->poly[2 * (c->poly_offset + k)], ly = w->poly[2 * (c->poly_offset + k) + 1];
            float wx = rc * lx - rs * ly, wy = rs * lx + rc * ly;
            minx = fminf(minx, wx); maxx = fmaxf(maxx, wx);
            miny = fminf(miny, wy); maxy = fmaxf(maxy, wy);
        }
        minx -= c->p[2]; maxx += c->p[2]; miny -= c->p[2]; maxy += c->p[2];
        break;
    }
    default: /* no collider: empty box */
        out[0] = out[1] = INFINITY; out[2] = out[3] = -INFINITY;
        return;
    }
    float padx, pady;
    motion_pads(w, i, dt, margin, vel_pad_max, &padx, &pady);
    out[0] = (px + minx) - padx;
    out[1] = (py + miny) - pady;
    out[2] = (px + maxx) + padx;
    out[3] = (py + maxy) + pady;
}

static void compute_aabb(const sfo_world* w, uint32_t i, float dt, float* out) {
    compute_aabb_ex(w, i, dt, w->cfg.aabb_margin, w->cfg.vel_pad_max, out);
}

static int pair_filter(const sfo_world* w, uint32_t i, uint32_t j) {
    if (!owner_dynamic(w, i) && !owner_dynamic(w, j)) return 0;
    if (w->cbody[i] == w->cbody[j]) return 0; /* two colliders of one body */
    const sfx_collider *a = &w->col[i], *b = &w->col[j];
    if (!(a->category & b->mask) || !(b->category & a->mask)) return 0;
    if (a->group != 0 && a->group == b->group) return 0;
    return 1;
}

static int aabb_overlap(const float* A, const float* B) {
    return A[0] <= B[2] && B[0] <= A[2] && A[1] <= B[3] && B[1] <= A[3];
}

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}

static uint64_t pair_key(uint32_t i, uint32_t j) {
    uint32_t a = i < j ? i : j, b = i < j ? j : i;
    return ((uint64_t)a << 32) | b;
}

size_t sfo_pairs_brute(const sfo_world* w, uint64_t* keys, size_t cap) {
    size_t n = 0;
    for (uint32_t i = 0; i < w->n; ++i) {
        if (w->col[i].type == SFX_SHAPE_NONE) continue;
        for (uint32_t j = i + 1; j < w->n; ++j) {
            if (w->col[j].type == SFX_SHAPE_NONE) continue;
            if (aabb_overlap(&w->aabb[4 * i], &w->aabb[4 * j]) && pair_filter(w, i, j)) {
                if (n < cap) keys[n] = pair_key(i, j);
                ++n;
            }
        }
    }
    return n; /* already ascending */
}

typedef struct { float lo; uint32_t id; } sweep_ent;
static int cmp_sweep(const void* a, const void* b) {
    const sweep_ent *x = (const sweep_ent*)a, *y = (const sweep_ent*)b;
    if (x->lo < y->lo) return -1;
    if (x->lo > y->lo) return 1;
    return x->id < y->id ? -1 : x->id > y->id;
}

size_t sfo_pairs_sweep(const sfo_world* w, uint64_t* keys, size_t cap) {
    sweep_ent* s = (sweep_ent*)malloc((w->n ? w->n : 1) * sizeof(sweep_ent));
    uint32_t ns = 0;
    for (uint32_t i = 0; i < w->n; ++i)
        if (w->col[i].type != SFX_SHAPE_NONE) { s[ns].lo = w->aabb[4 * i]; s[ns].id = i; ++ns; }
    qsort(s, ns, sizeof(sweep_ent), cmp_sweep);
    size_t n = 0;
    for (uint32_t p = 0; p < ns; ++p) {
        uint32_t i = s[p].id;
        float hix = w->aabb[4 * i + 2];
        for (uint32_t q = p + 1; q < ns && s[q].lo <= hix; ++q) {
            uint32_t j = s[q].id;
            if (aabb_overlap(&w->aabb[4 * i], &w->aabb[4 * j]) && pair_filter(w, i, j)) {
                if (n < cap) keys[n] = pair_key(i, j);
                ++n;
            }
        }
    }
    free(s);
    qsort(keys, n < cap ? n : cap, sizeof(uint64_t), cmp_u64);
    return n;
}

/* [DESIGN] DESIGN.md "Contract / grid": home cell = cell of the AABB's low corner on a
 * 2^gx x 2^gy torus; small colliders (extent <= 0.9 cell) are counting-sorted by bucket
 * (ascending id inside a bucket), large ones are kept in an ascending-id list.  Emission:
 * per small slot the own cell (later slots only), then the forward neighbours
 * (1,0) (-1,1) (0,1) (1,1), then every large; per large slot every later large. */

static void emit_pair(sfo_world* w, uint32_t i, uint32_t j) {
    if (!aabb_overlap(&w->aabb[4 * i], &w->aabb[4 * j]) || !pair_filter(w, i, j)) return;
    if (w->n_pairs < w->max_pairs) {
        w->pa[w->n_pairs] = i < j ? i : j;
        w->pb[w->n_pairs] = i < j ? j : i;
    } else {
        w->overflow = 1;
    }
    w->n_pairs++;
}

/* [DESIGN] DESIGN.md "Contract / grid", two levels (upstream's broadphase is a hierarchical grid: SURVEY.md 8a row a7).
 * Level 0: cells of the base size, for the SMALL colliders (extent <= 0.9 cell on both axes).  Level 1: cells eight
 * times as wide on the same torus, for the MEDIUM ones (not small, extent <= 0.9 of such a cell); what is larger still
 * stays on the flat list.  Every collider is sorted into one bucket: the home cell of its AABB's low corner at its own
 * level.  Emission per small slot: its own cell (later slots), the forward neighbours (1,0) (-1,1) (0,1) (1,1), the
 * medium colliders homed in the level-1 cells [c1(lo) - 1, c1(hi)] on both axes (an overlapping medium's home cell lies
 * there: its box is less than 1.9 cells wide), then the flat list; per medium slot the same forward scheme on level 1,
 * then the flat list; per listed collider every later one.  SFX_CFG_ONE_LEVEL puts the medium ones on the flat list. */
static void grid_pairs(sfo_world* w) {
    uint32_t n = w->n, GX = 1u << w->gxl, GY = 1u << w->gyl, ncell = GX * GY;
    const int two = !(w->cfg.flags & SFX_CFG_ONE_LEVEL);
    uint32_t gxl1 = w->gxl - 3u, gyl1 = w->gyl - 3u, GX1 = 1u << gxl1, GY1 = 1u << gyl1, ncell1 = two ? GX1 * GY1 : 0u;
    float inv1 = w->inv_cell * 0.125f;
    uint32_t* start = (uint32_t*)calloc((size_t)ncell + ncell1 + 1, sizeof(uint32_t));
    uint32_t* bucket = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    int32_t *ccx = w->ccx, *ccy = w->ccy;
    uint32_t* slot_id = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* large = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t ns = 0, nm = 0, nl = 0;
    float lim = 0.9f * w->cell, lim1 = 0.9f * (8.0f * w->cell);
    for (uint32_t i = 0; i < n; ++i) {
        bucket[i] = 0xFFFFFFFFu;
        w->is_small[i] = 0;
        if (w->col[i].type == SFX_SHAPE_NONE) continue;
        const float* A = &w->aabb[4 * i];
        if ((A[2] - A[0]) <= lim && (A[3] - A[1]) <= lim) {
            ccx[i] = cell_coord(A[0], w->inv_cell);
            ccy[i] = cell_coord(A[1], w->inv_cell);
            bucket[i] = (((uint32_t)ccy[i] & (GY - 1)) << w->gxl) | ((uint32_t)ccx[i] & (GX - 1));
            start[bucket[i] + 1]++;
            w->is_small[i] = 1;
            ++ns;
        } else if (two && (A[2] - A[0]) <= lim1 && (A[3] - A[1]) <= lim1) {
            ccx[i] = cell_coord(A[0], inv1);      /* home cell at level 1 */
            ccy[i] = cell_coord(A[1], inv1);
            bucket[i] = ncell + ((((uint32_t)ccy[i] & (GY1 - 1)) << gxl1) | ((uint32_t)ccx[i] & (GX1 - 1)));
            start[bucket[i] + 1]++;
            ++nm;
        } else {
            large[nl++] = i;
        }
    }
    for (uint32_t c = 0; c < ncell + ncell1; ++c) start[c + 1] += start[c];
    uint32_t* fill = (uint32_t*)malloc(((size_t)ncell + ncell1 + 1) * sizeof(uint32_t));
    memcpy(fill, start, ((size_t)ncell + ncell1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i)
        if (bucket[i] != 0xFFFFFFFFu) slot_id[fill[bucket[i]]++] = i;   /* small slots [0, ns), medium slots [ns, ns + nm) */
    w->n_small = ns; w->n_large = nm + nl;
    w->n_pairs = 0; w->overflow = 0;
    static const int off[5][2] = {{0, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};
    for (uint32_t p = 0; p < ns + nm; ++p) {
        uint32_t i = slot_id[p];
        const int lvl = p >= ns;
        for (int o = 0; o < 5; ++o) {
            int32_t nx = ccx[i] + off[o][0], ny = ccy[i] + off[o][1];
            uint32_t nb = lvl ? ncell + ((((uint32_t)ny & (GY1 - 1)) << gxl1) | ((uint32_t)nx & (GX1 - 1)))
                              : (((uint32_t)ny & (GY - 1)) << w->gxl) | ((uint32_t)nx & (GX - 1));
            for (uint32_t q = start[nb]; q < start[nb + 1]; ++q) {
                uint32_t j = slot_id[q];
                if (ccx[j] != nx || ccy[j] != ny) continue;
                if (o == 0 && q <= p) continue;
                emit_pair(w, i, j);
            }
        }
        if (!lvl && nm) { /* a small collider against the medium ones around it */
            const float* A = &w->aabb[4 * i];
            int32_t x0 = cell_coord(A[0], inv1) - 1, x1 = cell_coord(A[2], inv1);
            int32_t y0 = cell_coord(A[1], inv1) - 1, y1 = cell_coord(A[3], inv1);
            for (int32_t Y = y0; Y <= y1; ++Y)
                for (int32_t X = x0; X <= x1; ++X) {
                    uint32_t nb = ncell + ((((uint32_t)Y & (GY1 - 1)) << gxl1) | ((uint32_t)X & (GX1 - 1)));
                    for (uint32_t q = start[nb]; q < start[nb + 1]; ++q) {
                        uint32_t j = slot_id[q];
                        if (ccx[j] == X && ccy[j] == Y) emit_pair(w, i, j);
                    }
                }
        }
        for (uint32_t l = 0; l < nl; ++l) emit_pair(w, i, large[l]);
    }
    for (uint32_t l = 0; l < nl; ++l)
        for (uint32_t l2 = l + 1; l2 < nl; ++l2) emit_pair(w, large[l], large[l2]);
    free(start); free(bucket); free(slot_id); free(large); free(fill);
}

/* ------------------------------------------------------------------ reach test
 * [DESIGN] DESIGN.md "Contract / solver pairs".  The padded AABBs say how far a body may move in
 * the tick (per axis: aabb_margin + min(|v| dt, vel_pad_max)); a candidate pair whose shapes are
 * further apart than the two reaches plus the speculative margin cannot touch before the next
 * broadphase and is left out of the solve.  The distance is bounded from below by the gap along a
 * few axes (the centre line and the local axes of rectangles and capsules): a gap along ANY axis
 * is a lower bound of the distance, so the test never drops a pair that could touch. */
static void support_extents(const sfo_world* w, uint32_t i, float nx, float ny, float* hp, float* hm) {
    const sfx_collider* c = &w->col[i];
    float cpx, cpy, rc, rs;
    collider_pose(w, i, &cpx, &cpy, &rc, &rs);
    (void)cpx; (void)cpy;
    switch (c->type) {
    case SFX_SHAPE_CIRCLE:
        *hp = *hm = c->p[2];
        break;
    case SFX_SHAPE_RECT: {
        float ax = nx * rc + ny * rs, ay = ny * rc - nx * rs;
        *hp = *hm = (fabsf(ax) * c->p[0] + fabsf(ay) * c->p[1]) + c->p[2];
        break;
    }
    case SFX_SHAPE_CAPSULE: {
        float ax = nx * rc + ny * rs;
        *hp = *hm = fabsf(ax) * c->p[0] + c->p[2];
        break;
    }
    case SFX_SHAPE_POLY: {
        float mx = -INFINITY, mn = INFINITY;
        for (uint32_t k = 0; k < c->poly_count; ++k) {
            float lx = w->poly[2 * (c->poly_offset + k)], ly = w->poly[2 * (c->poly_offset + k) + 1];
            float wx = rc * lx - rs * ly, wy = rs * lx + rc * ly;
            float pr = nx * wx + ny * wy;
            mx = fmaxf(mx, pr); mn = fminf(mn, pr);
        }
        *hp = mx + c->p[2]; *hm = c->p[2] - mn;
        break;
    }
    default:
        *hp = *hm = INFINITY;
    }
}

static float axis_gap(const sfo_world* w, uint32_t a, uint32_t b, float nx, float ny, float dx, float dy) {
    float hap, ham, hbp, hbm;
    support_extents(w, a, nx, ny, &hap, &ham);
    support_extents(w, b, nx, ny, &hbp, &hbm);
    float t = nx * dx + ny * dy;
    return t >= 0.0f ? t - (hap + hbm) : (-t) - (ham + hbp);
}

static float reach_of(const sfo_world* w, uint32_t i, float dt) {
    float px, py;
    motion_pads(w, i, dt, w->cfg.aabb_margin, w->cfg.vel_pad_max, &px, &py);
    return sqrtf(px * px + py * py);
}

static int pair_in_reach(const sfo_world* w, uint32_t a, uint32_t b, float dt) {
    float ax, ay, ac, as, bx, by, bc, bs; /* the shapes' poses: the gaps are measured between the shapes */
    collider_pose(w, a, &ax, &ay, &ac, &as);
    collider_pose(w, b, &bx, &by, &bc, &bs);
    float dx = bx - ax, dy = by - ay;
    float d2 = dx * dx + dy * dy;
    float gap = -INFINITY;
    if (d2 > 1e-12f) {
        float d = sqrtf(d2);
        gap = axis_gap(w, a, b, dx / d, dy / d, dx, dy);
    }
    for (int k = 0; k < 2; ++k) {
        uint32_t i = k ? b : a;
        uint32_t t = w->col[i].type;
        if (t != SFX_SHAPE_RECT && t != SFX_SHAPE_CAPSULE) continue;
        float rc = k ? bc : ac, rs = k ? bs : as;
        gap = fmaxf(gap, axis_gap(w, a, b, rc, rs, dx, dy));
        gap = fmaxf(gap, axis_gap(w, a, b, -rs, rc, dx, dy));
    }
    float thr = (reach_of(w, a, dt) + reach_of(w, b, dt)) + w->cfg.speculative_margin;
    return !(gap > thr);
}

/* ------------------------------------------------------------------ tile order
 * [DESIGN] DESIGN.md "Contract / tiles".  Tiles are 2^tlx x 2^tly blocks of grid cells (true
 * cell coordinates, wrapped onto the torus like the cells), at least two cells wide and high.
 * A candidate pair of two small colliders is owned by the tile (min tx, min ty) of its two home
 * tiles; a pair of a small collider and a large non-dynamic one by the small one's tile;
 * everything else, and every pair of a tile that owns more than SFX_TILE_MAX_PAIRS pairs or
 * whose 2x2 tile block holds more than SFX_TILE_MAX_BODIES small colliders, goes to the
 * fallback list.  The pairs of a tile fall into four CLASSES by where their two home tiles lie:
 *   0 interior   both in the tile itself (and the pairs with a large non-dynamic collider)
 *   1 x seam     they differ in x only       2 y seam     they differ in y only
 *   3 corner     they differ in both
 * Overlapping small colliders have home cells at most one apart, so a seam pair joins the last
 * cell column (row) of one tile with the first of the next: with tiles of two cells or more, the
 * units (tile, class) of ONE class never share a body.  The four classes are the four phases of
 * a sweep -- interior first, the heavy one; the seams are short -- and every unit, like the
 * fallback list, is edge-coloured on its own.  With one tile row (strips) classes 2 and 3 are
 * empty. */
#define SFO_UNIT(cls, tile) (((uint32_t)(cls) << 28) | (tile))

static uint32_t wrapped_tile(const sfo_world* w, int32_t tx, int32_t ty) {
    uint32_t lx = w->gxl - w->tlx, ly = w->gyl - w->tly;
    return (((uint32_t)ty & ((1u << ly) - 1u)) << lx) | ((uint32_t)tx & ((1u << lx) - 1u));
}

static int32_t tile_coord(int32_t c, uint32_t tl) { return c >= 0 ? (int32_t)((uint32_t)c >> tl) : -(int32_t)(((uint32_t)(-(c + 1)) >> tl) + 1u); }

/* [DESIGN] May the end `e` of a candidate pair stand in a tile's pair list?  Always, unless it is a further collider of
 * a dynamic body: the tile solves keep one state per BODY, at the place of its first collider (entry = body id), so
 * the further collider must be small and lie in the home tile of that first collider, which must be small too; and where
 * the pair crosses a tile border in x (ddx) / in y (ddy) the two must share the cell column / row -- the body then takes
 * part in exactly the seam units its first collider alone could take part in, and the units of one class still share
 * no body (see "tile order" above).  Everything else of such a body goes to the fallback list. */
static int tile_end_ok(const sfo_world* w, uint32_t e, int ddx, int ddy) {
    if (!dynamic_secondary(w, e)) return 1;
    const uint32_t b = w->cbody[e];
    if (!w->is_small[e] || !w->is_small[b]) return 0;
    if (tile_coord(w->ccx[e], w->tlx) != tile_coord(w->ccx[b], w->tlx) || tile_coord(w->ccy[e], w->tly) != tile_coord(w->ccy[b], w->tly)) return 0;
    if (ddx && w->ccx[e] != w->ccx[b]) return 0;
    if (ddy && w->ccy[e] != w->ccy[b]) return 0;
    return 1;
}

static void build_order(sfo_world* w, float frame_dt) {
    uint32_t np = w->n_pairs < w->max_pairs ? w->n_pairs : w->max_pairs;
    uint32_t lx = w->gxl - w->tlx, ly = w->gyl - w->tly, ntiles = 1u << (lx + ly);
    uint32_t* tbodies = (uint32_t*)calloc(ntiles, sizeof(uint32_t));
    uint32_t* tcnt = (uint32_t*)calloc(ntiles, sizeof(uint32_t));
    uint8_t* tover = (uint8_t*)calloc(ntiles, 1);
    uint8_t* dyn = (uint8_t*)malloc(w->n ? w->n : 1);
    for (uint32_t i = 0; i < w->n; ++i) {
        dyn[i] = (uint8_t)is_dynamic(w, i);
        if (w->is_small[i]) tbodies[wrapped_tile(w, tile_coord(w->ccx[i], w->tlx), tile_coord(w->ccy[i], w->tly))]++;
    }
    for (uint32_t e = 0; e < np; ++e) {
        uint32_t a = w->pa[e], b = w->pb[e], own = SFO_FALLBACK;
        int32_t txa = tile_coord(w->ccx[a], w->tlx), tya = tile_coord(w->ccy[a], w->tly);
        int32_t txb = tile_coord(w->ccx[b], w->tlx), tyb = tile_coord(w->ccy[b], w->tly);
        int ddx = 0, ddy = 0;
        if (w->is_small[a] && w->is_small[b]) {
            /* (wrapped tile coordinates: with one tile column / row every home tile is the same one) */
            ddx = (((uint32_t)txa ^ (uint32_t)txb) & ((1u << lx) - 1u)) != 0;
            ddy = (((uint32_t)tya ^ (uint32_t)tyb) & ((1u << ly) - 1u)) != 0;
        }
        if (!tile_end_ok(w, a, ddx, ddy) || !tile_end_ok(w, b, ddx, ddy)) own = SFO_FALLBACK;
        else if (w->is_small[a] && w->is_small[b]) own = wrapped_tile(w, txa < txb ? txa : txb, tya < tyb ? tya : tyb);
        else if (w->is_small[a] && !dyn[b]) own = wrapped_tile(w, txa, tya);
        else if (w->is_small[b] && !dyn[a]) own = wrapped_tile(w, txb, tyb);
        w->owner[e] = own;
        w->pclass[e] = (uint8_t)(own == SFO_FALLBACK ? 0 : ddx ? (ddy ? 3 : 1) : (ddy ? 2 : 0));
        if (own != SFO_FALLBACK) tcnt[own]++;
    }
    w->max_tile_pairs = 0; w->max_tile_bodies = 0; w->n_tiles = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        if (!tcnt[t]) continue;
        uint32_t tx = t & ((1u << lx) - 1u), ty = t >> lx, nb = 0;
        for (int dy = 0; dy < (ly ? 2 : 1); ++dy) /* one tile row (strips): the block is two tiles, not four */
            for (int dx = 0; dx < (lx ? 2 : 1); ++dx) nb += tbodies[wrapped_tile(w, (int32_t)tx + dx, (int32_t)ty + dy)];
        if (tcnt[t] > SFX_TILE_MAX_PAIRS || nb > SFX_TILE_MAX_BODIES) { tover[t] = 1; continue; }
        if (tcnt[t] > w->max_tile_pairs) w->max_tile_pairs = tcnt[t];
        if (nb > w->max_tile_bodies) w->max_tile_bodies = nb;
        w->n_tiles++;
    }
    /* stable counting sort of the pair indices by owner tile; overflowing tiles join the fallback */
    uint32_t* tstart = (uint32_t*)calloc((size_t)ntiles + 2, sizeof(uint32_t));
    for (uint32_t e = 0; e < np; ++e) {
        if (w->owner[e] != SFO_FALLBACK && tover[w->owner[e]]) w->owner[e] = SFO_FALLBACK;
        tstart[(w->owner[e] == SFO_FALLBACK ? ntiles : w->owner[e]) + 1]++;
    }
    for (uint32_t t = 0; t <= ntiles; ++t) tstart[t + 1] += tstart[t];
    uint32_t* bytile = (uint32_t*)malloc((np ? np : 1) * sizeof(uint32_t));
    uint32_t* fill = (uint32_t*)malloc(((size_t)ntiles + 1) * sizeof(uint32_t));
    memcpy(fill, tstart, ((size_t)ntiles + 1) * sizeof(uint32_t));
    for (uint32_t e = 0; e < np; ++e) bytile[fill[w->owner[e] == SFO_FALLBACK ? ntiles : w->owner[e]]++] = e;
    w->n_fallback = tstart[ntiles + 1] - tstart[ntiles];

    color_scratch cs;
    color_scratch_init(&cs, w->n);
    memset(w->color_count, 0, sizeof(w->color_count));
    w->n_colors = 0; w->n_leftover = 0; w->n_batches = 0; w->n_order = 0;
    uint32_t ovf = 0;
    uint32_t* kept = (uint32_t*)malloc((np ? np : 1) * sizeof(uint32_t));
    /* the colouring's conflicts are between BODIES (the ends of an edge are the bodies its two colliders belong to);
     * its keys stay the pair's two collider ids, unique per edge */
    uint32_t* ba = (uint32_t*)malloc((np ? np : 1) * sizeof(uint32_t));
    uint32_t* bb = (uint32_t*)malloc((np ? np : 1) * sizeof(uint32_t));
    for (uint32_t e = 0; e < np; ++e) { ba[e] = w->cbody[w->pa[e]]; bb[e] = w->cbody[w->pb[e]]; }
    /* batches: phase by phase, inside a phase tile by tile (independent of each other), the
     * fallback list last; inside a batch colour-major.  The units of one phase share no dynamic body, so their reach
     * tests and colourings do not touch each other's scratch entries: the OpenMP build runs them side by side; the
     * order is then written unit by unit, serially, exactly as the serial build does. */
    uint32_t* unk = (uint32_t*)calloc((size_t)ntiles + 1, sizeof(uint32_t)); /* pairs kept per unit of the phase */
    uint32_t* unc = (uint32_t*)calloc((size_t)ntiles + 1, sizeof(uint32_t)); /* its colours */
    for (uint32_t ph = 0; ph <= 4; ++ph) {
        const uint32_t first = ph < 4 ? 0u : ntiles, last = ph < 4 ? ntiles : ntiles + 1u;
        uint32_t left_ph = 0, ovf_ph = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : left_ph) reduction(| : ovf_ph) if (last - first > 16)
#endif
        for (uint32_t t = first; t < last; ++t) {
            uint32_t s0 = tstart[t], cnt = tstart[t + 1] - s0;
            unk[t] = 0;
            if (!cnt) continue;
            const uint32_t* idx = &bytile[s0];
            if (ph < 4) { /* tile-owned pairs: only those within reach are coloured and solved */
                uint32_t nk = 0;
                for (uint32_t q = 0; q < cnt; ++q) {
                    uint32_t e = bytile[s0 + q];
                    if (w->pclass[e] != ph) continue; /* the unit (tile t, class ph) */
                    if (pair_in_reach(w, w->pa[e], w->pb[e], frame_dt)) kept[s0 + nk++] = e;
                    else w->color[e] = PRUNED_COLOR;
                }
                idx = &kept[s0]; cnt = nk;
                if (!cnt) continue;
            }
            uint32_t lo = 0, ov = 0;
            unc[t] = color_edges(&cs, cnt, idx, ba, bb, w->pa, w->pb, dyn, w->cfg.color_rounds, w->color, &lo, &ov);
            unk[t] = cnt;
            left_ph += lo; ovf_ph |= ov;
        }
        w->n_leftover += left_ph; ovf |= ovf_ph;
        for (uint32_t t = first; t < last; ++t) {
            const uint32_t cnt = unk[t], nc = unc[t];
            if (!cnt) continue;
            const uint32_t* idx = ph < 4 ? &kept[tstart[t]] : &bytile[tstart[t]];
            if (nc > w->n_colors) w->n_colors = nc;
            for (uint32_t c = 0; c < nc; ++c)
                for (uint32_t q = 0; q < cnt; ++q) {
                    uint32_t e = idx[q];
                    if (w->color[e] == c) { w->order[w->n_order++] = e; w->color_count[c]++; }
                }
            w->batch_end[w->n_batches++] = w->n_order;
        }
        w->phase_end[ph] = w->n_batches;
    }
    free(unk); free(unc);
    color_scratch_free(&cs);
    if (ovf) w->overflow = 1;
    w->order_valid = 1;
    w->n_solver_pairs = w->n_order;
    free(tbodies); free(tcnt); free(tover); free(dyn); free(tstart); free(bytile); free(fill); free(kept);
    free(ba); free(bb);
}

static void broadphase_core(sfo_world* w, float frame_dt) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (w->n > 4096)
#endif
    for (uint32_t i = 0; i < w->n; ++i) compute_aabb(w, i, frame_dt, &w->aabb[4 * i]);
    grid_pairs(w);
    w->order_valid = 0;
}

int sfo_broadphase(sfo_world* w, float frame_dt) {
    if (!w->col) return SFX_ERR_INVALID;
    broadphase_core(w, frame_dt);
    build_order(w, frame_dt);
    return w->overflow ? SFX_ERR_CAPACITY : SFX_OK;
}

int sfo_aabbs(const sfo_world* w, float* out) {
    memcpy(out, w->aabb, 4u * w->n * sizeof(float));
    return SFX_OK;
}

size_t sfo_pairs_grid(const sfo_world* w, uint64_t* keys, size_t cap) {
    uint32_t np = w->n_pairs < w->max_pairs ? w->n_pairs : w->max_pairs;
    size_t k = 0;
    for (uint32_t e = 0; e < np && k < cap; ++e) keys[k++] = ((uint64_t)w->pa[e] << 32) | w->pb[e];
    qsort(keys, k, sizeof(uint64_t), cmp_u64);
    return w->n_pairs;
}

size_t sfo_pair_colors(const sfo_world* w, uint64_t* keys, uint8_t* colors, uint32_t* tiles, size_t cap) {
    uint32_t np = w->n_pairs < w->max_pairs ? w->n_pairs : w->max_pairs;
    if (!w->order_valid) return 0;
    for (uint32_t e = 0; e < np && e < cap; ++e) {
        keys[e] = ((uint64_t)w->pa[e] << 32) | w->pb[e];
        colors[e] = w->color[e];
        if (tiles) tiles[e] = w->owner[e] == SFO_FALLBACK ? SFO_FALLBACK : SFO_UNIT(w->pclass[e], w->owner[e]);
    }
    return np;
}

size_t sfo_joint_order(const sfo_world* w, uint32_t* perm, uint8_t* colors, size_t cap) {
    for (uint32_t k = 0; k < w->m && k < cap; ++k) {
        perm[k] = w->joint_perm[k];
        colors[k] = w->joint_color[w->joint_perm[k]];
    }
    return w->m;
}

/* ------------------------------------------------------------------ narrowphase
 * [DESIGN] DESIGN.md "Contract / narrowphase".  Every shape is a convex core polygon of
 * 1..8 vertices plus a radius; contacts come from SAT + reference-face clipping on the
 * cores (closest features when the clip is empty), with the radii subtracted. */

static void build_core(const sfo_world* w, uint32_t i, core_t* c) {
    const sfx_collider* col = &w->col[i];
    float px, py, rc, rs;
    collider_pose(w, i, &px, &py, &rc, &rs);
    c->r = col->p[2];
    switch (col->type) {
    case SFX_SHAPE_CIRCLE:
        c->n = 1; c->vx[0] = px; c->vy[0] = py; c->nx[0] = 0.0f; c->ny[0] = 0.0f;
        break;
    case SFX_SHAPE_RECT: {
        float ax = rc * col->p[0], ay = rs * col->p[0];
        float bx = -rs * col->p[1], by = rc * col->p[1];
        c->n = 4;
        c->vx[0] = (px - ax) - bx; c->vy[0] = (py - ay) - by;
        c->vx[1] = (px + ax) - bx; c->vy[1] = (py + ay) - by;
        c->vx[2] = (px + ax) + bx; c->vy[2] = (py + ay) + by;
        c->vx[3] = (px - ax) + bx; c->vy[3] = (py - ay) + by;
        c->nx[0] = rs;  c->ny[0] = -rc;
        c->nx[1] = rc;  c->ny[1] = rs;
        c->nx[2] = -rs; c->ny[2] = rc;
        c->nx[3] = -rc; c->ny[3] = -rs;
        break;
    }
    case SFX_SHAPE_CAPSULE: {
        float ax = rc * col->p[0], ay = rs * col->p[0];
        c->n = 2;
        c->vx[0] = px - ax; c->vy[0] = py - ay;
        c->vx[1] = px + ax; c->vy[1] = py + ay;
        c->nx[0] = rs;  c->ny[0] = -rc;
        c->nx[1] = -rs; c->ny[1] = rc;
        break;
    }
    case SFX_SHAPE_POLY:
        c->n = (int)col->poly_count;
        for (int k = 0; k < c->n; ++k) {
            float lx = w->poly[2 * (col->poly_offset + k)], ly = w->poly[2 * (col->poly_offset + k) + 1];
            float mx = w->poly_nrm[2 * (col->poly_offset + k)], my = w->poly_nrm[2 * (col->poly_offset + k) + 1];
            c->vx[k] = px + (rc * lx - rs * ly); c->vy[k] = py + (rs * lx + rc * ly);
            c->nx[k] = rc * mx - rs * my;        c->ny[k] = rs * mx + rc * my;
        }
        break;
    default:
        c->n = 0;
    }
}

typedef struct {
    int count;
    float nx, ny;         /* from first core to second core */
    float p1x[2], p1y[2]; /* surface points on the first shape */
    float p2x[2], p2y[2]; /* surface points on the second shape */
    float depth[2];
} manifold_t;

static void cc_manifold(const core_t* A, const core_t* B, float sm, manifold_t* m) {
    float dx = B->vx[0] - A->vx[0], dy = B->vy[0] - A->vy[0];
    float d2 = dx * dx + dy * dy;
    float rsum = A->r + B->r;
    float reach = rsum + sm;
    m->count = 0;
    if (d2 > reach * reach) return;
    float dist = sqrtf(d2);
    float nx = 0.0f, ny = 1.0f;
    if (dist > EPS_LEN) { nx = dx / dist; ny = dy / dist; }
    float depth = rsum - dist;
    if (!(depth > -sm)) return;
    m->count = 1; m->nx = nx; m->ny = ny; m->depth[0] = depth;
    m->p1x[0] = A->vx[0] + nx * A->r; m->p1y[0] = A->vy[0] + ny * A->r;
    m->p2x[0] = B->vx[0] - nx * B->r; m->p2y[0] = B->vy[0] - ny * B->r;
}

/* closest point on polygon boundary P (n >= 2) to point (cx, cy); first minimum wins */
static void closest_on_poly(const core_t* P, float cx, float cy, float* qx, float* qy, float* d2out) {
    float best = INFINITY, bx = 0.0f, by = 0.0f;
    for (int k = 0; k < P->n; ++k) {
        int k1 = (k + 1 == P->n) ? 0 : k + 1;
        float ex = P->vx[k1] - P->vx[k], ey = P->vy[k1] - P->vy[k];
        float wx = cx - P->vx[k], wy = cy - P->vy[k];
        float ee = ex * ex + ey * ey;
        float t = 0.0f;
        if (ee > 0.0f) t = (wx * ex + wy * ey) / ee;
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        float x = P->vx[k] + t * ex, y = P->vy[k] + t * ey;
        float ddx = cx - x, ddy = cy - y;
        float d2 = ddx * ddx + ddy * ddy;
        if (d2 < best) { best = d2; bx = x; by = y; }
    }
    *qx = bx; *qy = by; *d2out = best;
}

/* polygon P (n >= 2) against circle C: normal points from P to C */
static void pc_manifold(const core_t* P, const core_t* C, float sm, manifold_t* m) {
    float cx = C->vx[0], cy = C->vy[0];
    float rsum = P->r + C->r;
    float reach = rsum + sm;
    float sep = -INFINITY; int ks = 0;
    m->count = 0;
    for (int k = 0; k < P->n; ++k) {
        float s = P->nx[k] * (cx - P->vx[k]) + P->ny[k] * (cy - P->vy[k]);
        if (s > sep) { sep = s; ks = k; }
    }
    if (sep > reach) return;
    float nx, ny, qx, qy, depth;
    if (P->n == 2 || sep > 0.0f) {
        float d2;
        closest_on_poly(P, cx, cy, &qx, &qy, &d2);
        if (d2 > reach * reach) return;
        float dist = sqrtf(d2);
        if (dist > EPS_LEN) { nx = (cx - qx) / dist; ny = (cy - qy) / dist; }
        else { nx = P->nx[ks]; ny = P->ny[ks]; }
        depth = rsum - dist;
    } else {
        nx = P->nx[ks]; ny = P->ny[ks];
        qx = cx - nx * sep; qy = cy - ny * sep;
        depth = rsum - sep;
    }
    if (!(depth > -sm)) return;
    m->count = 1; m->nx = nx; m->ny = ny; m->depth[0] = depth;
    m->p1x[0] = qx + nx * P->r; m->p1y[0] = qy + ny * P->r;
    m->p2x[0] = cx - nx * C->r; m->p2y[0] = cy - ny * C->r;
}

static float sat_sep(const core_t* P, const core_t* Q, int* kbest) {
    float best = -INFINITY; int bk = 0;
    for (int k = 0; k < P->n; ++k) {
        float s = INFINITY;
        for (int j = 0; j < Q->n; ++j) {
            float d = P->nx[k] * (Q->vx[j] - P->vx[k]) + P->ny[k] * (Q->vy[j] - P->vy[k]);
            if (d < s) s = d;
        }
        if (s > best) { best = s; bk = k; }
    }
    *kbest = bk;
    return best;
}

/* polygon (n >= 2) against polygon (n >= 2); normal from A to B */
static void pp_manifold(const core_t* A, const core_t* B, float sm, manifold_t* m) {
    int ka, kb;
    float rsum = A->r + B->r;
    float reach = rsum + sm;
    m->count = 0;
    float sa = sat_sep(A, B, &ka);
    if (sa > reach) return;
    float sb = sat_sep(B, A, &kb);
    if (sb > reach) return;
    int flip = sb > sa;
    const core_t* R = flip ? B : A; /* reference */
    const core_t* I = flip ? A : B; /* incident  */
    int kr = flip ? kb : ka;
    int kr1 = (kr + 1 == R->n) ? 0 : kr + 1;
    float nrx = R->nx[kr], nry = R->ny[kr];
    float r1x = R->vx[kr], r1y = R->vy[kr], r2x = R->vx[kr1], r2y = R->vy[kr1];
    /* incident edge: most anti-parallel normal, first minimum */
    float bestd = INFINITY; int ki = 0;
    for (int j = 0; j < I->n; ++j) {
        float d = I->nx[j] * nrx + I->ny[j] * nry;
        if (d < bestd) { bestd = d; ki = j; }
    }
    int ki1 = (ki + 1 == I->n) ? 0 : ki + 1;
    float i1x = I->vx[ki], i1y = I->vy[ki], i2x = I->vx[ki1], i2y = I->vy[ki1];
    float tx = -nry, ty = nrx;
    float L = (r2x - r1x) * tx + (r2y - r1y) * ty;
    float u1 = (i1x - r1x) * tx + (i1y - r1y) * ty;
    float u2 = (i2x - r1x) * tx + (i2y - r1y) * ty;
    float du = u2 - u1;
    float ta = 0.0f, tb = 1.0f;
    int empty = 0;
    if (du > 0.0f) {
        ta = fmaxf(ta, (0.0f - u1) / du);
        tb = fminf(tb, (L - u1) / du);
    } else if (du < 0.0f) {
        tb = fminf(tb, (0.0f - u1) / du);
        ta = fmaxf(ta, (L - u1) / du);
    } else if (u1 < 0.0f || u1 > L) {
        empty = 1;
    }
    if (ta > tb) empty = 1;
    float rr = R->r, ri = I->r;
    float nx, ny;           /* from reference to incident */
    float prx[2], pry[2], pix[2], piy[2], dep[2];
    int cnt = 0;
    if (!empty) {
        float ex = i2x - i1x, ey = i2y - i1y;
        float tt[2]; tt[0] = ta; tt[1] = tb;
        int np = (tb > ta) ? 2 : 1;
        nx = nrx; ny = nry;
        for (int k = 0; k < np; ++k) {
            float cx = i1x + tt[k] * ex, cy = i1y + tt[k] * ey;
            float s = (cx - r1x) * nrx + (cy - r1y) * nry;
            float d = rsum - s;
            if (!(d > -sm)) continue;
            /* foot on the reference edge pushed out by its radius / incident point pulled in */
            prx[cnt] = (cx - nrx * s) + nrx * rr; pry[cnt] = (cy - nry * s) + nry * rr;
            pix[cnt] = cx - nrx * ri;             piy[cnt] = cy - nry * ri;
            dep[cnt] = d;
            ++cnt;
        }
    } else {
        /* incident edge lies past one end of the reference edge: vertex against polygon */
        float vx = (u1 + u2 < 0.0f) ? r1x : r2x, vy = (u1 + u2 < 0.0f) ? r1y : r2y;
        float qx, qy, d2;
        closest_on_poly(I, vx, vy, &qx, &qy, &d2);
        if (d2 > reach * reach) return;
        float dist = sqrtf(d2);
        if (dist > EPS_LEN) { nx = (qx - vx) / dist; ny = (qy - vy) / dist; }
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
    m->count = cnt;
    for (int k = 0; k < cnt; ++k) {
        m->depth[k] = dep[k];
        if (!flip) { m->p1x[k] = prx[k]; m->p1y[k] = pry[k]; m->p2x[k] = pix[k]; m->p2y[k] = piy[k]; }
        else       { m->p1x[k] = pix[k]; m->p1y[k] = piy[k]; m->p2x[k] = prx[k]; m->p2y[k] = pry[k]; }
    }
    if (!flip) { m->nx = nx; m->ny = ny; } else { m->nx = -nx; m->ny = -ny; }
}

/* manifold of bodies a, b (normal from a to b); points closer than the speculative margin
 * are kept with their (negative) depth */
static void manifold(const sfo_world* w, uint32_t a, uint32_t b, manifold_t* m) {
    core_t A, B;
    float sm = w->cfg.speculative_margin;
    build_core(w, a, &A);
    build_core(w, b, &B);
    m->count = 0;
    if (A.n == 0 || B.n == 0) return;
    if (A.n == 1 && B.n == 1) { cc_manifold(&A, &B, sm, m); return; }
    if (B.n == 1) { pc_manifold(&A, &B, sm, m); return; }
    if (A.n == 1) {
        manifold_t t;
        pc_manifold(&B, &A, sm, &t);
        m->count = t.count;
        if (t.count) {
            m->nx = -t.nx; m->ny = -t.ny; m->depth[0] = t.depth[0];
            m->p1x[0] = t.p2x[0]; m->p1y[0] = t.p2y[0];
            m->p2x[0] = t.p1x[0]; m->p2y[0] = t.p1y[0];
        }
        return;
    }
    pp_manifold(&A, &B, sm, m);
}

static void fill_contact(uint32_t a, uint32_t b, const manifold_t* m, sfx_contact* c) {
    memset(c, 0, sizeof(*c));
    c->body_a = a; c->body_b = b; c->count = (uint32_t)m->count;
    if (!m->count) return;
    c->normal[0] = m->nx; c->normal[1] = m->ny;
    for (int k = 0; k < m->count; ++k) {
        c->depth[k] = m->depth[k];
        c->point_a[k][0] = m->p1x[k]; c->point_a[k][1] = m->p1y[k];
        c->point_b[k][0] = m->p2x[k]; c->point_b[k][1] = m->p2y[k];
    }
}

int sfo_contact_pair(const sfo_world* w, uint32_t a, uint32_t b, sfx_contact* out) {
    manifold_t m;
    if (a >= w->n || b >= w->n) return SFX_ERR_INVALID;
    manifold(w, a, b, &m);
    fill_contact(a, b, &m, out);
    return SFX_OK;
}

size_t sfo_contacts(const sfo_world* w, sfx_contact* out, size_t cap) {
    uint32_t np = w->n_pairs < w->max_pairs ? w->n_pairs : w->max_pairs;
    uint64_t* keys = (uint64_t*)malloc((np ? np : 1) * sizeof(uint64_t));
    for (uint32_t e = 0; e < np; ++e) keys[e] = ((uint64_t)w->pa[e] << 32) | w->pb[e];
    qsort(keys, np, sizeof(uint64_t), cmp_u64);
    for (uint32_t e = 0; e < np && e < cap; ++e) {
        manifold_t m;
        uint32_t a = (uint32_t)(keys[e] >> 32), b = (uint32_t)keys[e];
        manifold(w, a, b, &m);
        fill_contact(a, b, &m, &out[e]);
    }
    free(keys);
    return np;
}

/* ------------------------------------------------------------------ solver */

static void rotate_by(float* c, float* s, float dth) {
    float nc = *c - dth * *s;
    float ns = *s + dth * *c;
    float inv = 1.0f / sqrtf(nc * nc + ns * ns);
    *c = nc * inv; *s = ns * inv;
}

/* [XPBD-RB] Alg. 2 lines 3-9 */
static void integrate(sfo_world* w, float h) {
    float gx = w->cfg.gravity[0], gy = w->cfg.gravity[1];
    float dl = 1.0f / (1.0f + h * w->cfg.linear_damping), da = 1.0f / (1.0f + h * w->cfg.angular_damping);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (w->nb > 4096)
#endif
    for (uint32_t i = 0; i < w->nb; ++i) {
        w->prev_pos[2 * i] = w->pos[2 * i]; w->prev_pos[2 * i + 1] = w->pos[2 * i + 1];
        w->prev_rot[2 * i] = w->rot[2 * i]; w->prev_rot[2 * i + 1] = w->rot[2 * i + 1];
        w->dpos[3 * i] = 0.0f; w->dpos[3 * i + 1] = 0.0f; w->dpos[3 * i + 2] = 0.0f;
        if (w->inv_mass[i] > 0.0f) {
            w->vel[2 * i] = w->vel[2 * i] + gx * h;
            w->vel[2 * i + 1] = w->vel[2 * i + 1] + gy * h;
        }
        if (is_dynamic(w, i)) {
            w->vel[2 * i] = w->vel[2 * i] * dl;
            w->vel[2 * i + 1] = w->vel[2 * i + 1] * dl;
            w->angvel[i] = w->angvel[i] * da;
        }
        w->pos[2 * i] = w->pos[2 * i] + w->vel[2 * i] * h;
        w->pos[2 * i + 1] = w->pos[2 * i + 1] + w->vel[2 * i + 1] * h;
        rotate_by(&w->rot[2 * i], &w->rot[2 * i + 1], w->angvel[i] * h);
    }
}

/* [DESIGN] "Contract / frozen rows": the solves accumulate a displacement (dx, dy, dtheta) per body; it is applied
 * once -- the translation added, the rotation through rotate_by -- and a body that was not moved keeps its bits */
static void apply_delta(sfo_world* w, uint32_t i, const float* d) {
    w->pos[2 * i] = w->pos[2 * i] + d[0];
    w->pos[2 * i + 1] = w->pos[2 * i + 1] + d[1];
    if (d[2] != 0.0f) rotate_by(&w->rot[2 * i], &w->rot[2 * i + 1], d[2]);
}

/* [XPBD-RB] Alg. 2 lines 12-16; omega is the exact inverse of rotate_by: tan(dtheta)/h */
static void derive_velocities(sfo_world* w, float inv_h) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (w->nb > 4096)
#endif
    for (uint32_t i = 0; i < w->nb; ++i) {
        if (!is_dynamic(w, i)) continue;
        apply_delta(w, i, &w->dpos[3 * i]); /* what the position sweep accumulated, once per body and substep */
        w->vel[2 * i] = (w->pos[2 * i] - w->prev_pos[2 * i]) * inv_h;
        w->vel[2 * i + 1] = (w->pos[2 * i + 1] - w->prev_pos[2 * i + 1]) * inv_h;
        float c = w->rot[2 * i], s = w->rot[2 * i + 1];
        float pc = w->prev_rot[2 * i], ps = w->prev_rot[2 * i + 1];
        float dc = c * pc + s * ps;
        float ds = s * pc - c * ps;
        if (dc > 1e-6f) {
            /* [DESIGN] tan(dtheta) has a pole at a quarter turn: what the substep's corrections add to the angular
             * velocity is bounded by SFX_MAX_TAN_CORRECTION / h (a free spin, whose omega the substep does not
             * change, is never touched) */
            float wn = (ds / dc) * inv_h;
            float dw = wn - w->angvel[i], lim = SFX_MAX_TAN_CORRECTION * inv_h;
            if (dw > lim) wn = w->angvel[i] + lim;
            if (dw < -lim) wn = w->angvel[i] - lim;
            w->angvel[i] = wn;
        }
    }
}

/* apply a positional impulse p (world) at world offset r from the centre of body i, scaled by sign */
static void apply_pos(sfo_world* w, uint32_t i, float rx, float ry, float px, float py, float sign) {
    float im = w->inv_mass[i], ii = w->inv_inertia[i];
    if (im > 0.0f) {
        w->pos[2 * i] = w->pos[2 * i] + sign * (px * im);
        w->pos[2 * i + 1] = w->pos[2 * i + 1] + sign * (py * im);
    }
    if (ii > 0.0f) {
        float dth = sign * (ii * (rx * py - ry * px));
        rotate_by(&w->rot[2 * i], &w->rot[2 * i + 1], dth);
    }
}

static float gen_inv_mass(const sfo_world* w, uint32_t i, float rx, float ry, float nx, float ny) {
    float rn = rx * ny - ry * nx;
    return w->inv_mass[i] + w->inv_inertia[i] * (rn * rn);
}

/* [DESIGN] DESIGN.md "Contract / manifold block solve": the two normal constraints of a
 * manifold are solved together as a 2x2 LCP (lambda >= 0, no residual penetration), which
 * removes the point-order bias of a sequential solve that makes box stacks creep.
 * The inverse of K comes from the record (prepare_pair): nothing is divided here. */
static void solve_lcp2(const pair_rec* r, float d0, float d1, float* l0, float* l1) {
    if (r->block) {
        float a = r->i00 * d0 - r->i01 * d1;
        float b = r->i11 * d1 - r->i01 * d0;
        if (a >= 0.0f && b >= 0.0f) { *l0 = a; *l1 = b; return; }
    }
    float a = d0 * r->iK[0];
    if (a >= 0.0f && d1 - r->K01 * a <= 0.0f) { *l0 = a; *l1 = 0.0f; return; }
    float b = d1 * r->iK[1];
    if (b >= 0.0f && d0 - r->K01 * b <= 0.0f) { *l0 = 0.0f; *l1 = b; return; }
    if (d0 >= d1) { *l0 = fmaxf(d0, 0.0f) * r->iK[0]; *l1 = 0.0f; }
    else { *l0 = 0.0f; *l1 = fmaxf(d1, 0.0f) * r->iK[1]; }
}

/* y = K^-1 r restricted to the active normal constraints of a manifold (inverse from the record) */
static void kinv_apply(const pair_rec* rec, int act0, int act1, float r0, float r1, float* y0, float* y1) {
    *y0 = 0.0f; *y1 = 0.0f;
    if (rec->count == 2 && act0 && act1) {
        if (rec->block) {
            *y0 = rec->i00 * r0 - rec->i01 * r1;
            *y1 = rec->i11 * r1 - rec->i01 * r0;
        } else {
            *y0 = 0.5f * (r0 * rec->iK[0]); *y1 = 0.5f * (r1 * rec->iK[1]);
        }
    } else if (act0) {
        *y0 = r0 * rec->iK[0];
    } else if (rec->count == 2 && act1) {
        *y1 = r1 * rec->iK[1];
    }
}

static float inv_pos(float S) { return S > 0.0f ? 1.0f / S : 0.0f; }

/* [XPBD-RB] sec. 3.5, first half: narrowphase of one solver pair (entries a, b: colliders, which mirror their
 * bodies) at the poses the substep's integration and joints left, and -- [DESIGN] "Contract / frozen rows" --
 * everything of the pair's constraint rows that depends on geometry and mass only: the lever arms of the normal rows
 * and of the tangent row about the two BODIES' origins, the 2x2 effective mass of the normal rows and its inverse, the
 * tangent row's couplings and Schur complements, the restitution targets.  The rows stay frozen for the substep (the
 * classic linearisation of impulse solvers; first order in the substep's corrections). */
static void prepare_pair(sfo_world* w, uint32_t a, uint32_t b, pair_rec* rec) {
    manifold_t m;
    manifold(w, a, b, &m);
    rec->count = (uint32_t)m.count;
    rec->lam[0] = 0.0f; rec->lam[1] = 0.0f;
    if (!m.count) return;
    const sfx_material* ma = &w->mats[w->col[a].material];
    const sfx_material* mb = &w->mats[w->col[b].material];
    rec->mu_s = 0.5f * (ma->static_friction + mb->static_friction);
    rec->mu_d = 0.5f * (ma->dynamic_friction + mb->dynamic_friction);
    const float rest = fmaxf(ma->restitution, mb->restitution);
    const float gl = sqrtf(w->cfg.gravity[0] * w->cfg.gravity[0] + w->cfg.gravity[1] * w->cfg.gravity[1]);
    const float vthr = w->cfg.restitution_threshold * gl * w->tick_h;
    const float nx = m.nx, ny = m.ny;
    rec->nx = nx; rec->ny = ny;
    const uint32_t ba = w->cbody[a], bb = w->cbody[b];   /* the masses are the bodies' (entries beyond the bodies have none) */
    const float ima = w->inv_mass[ba], iia = w->inv_inertia[ba], imb = w->inv_mass[bb], iib = w->inv_inertia[bb];
    const float ims = ima + imb;
    float rax[2] = {0, 0}, ray[2] = {0, 0}, rbx[2] = {0, 0}, rby[2] = {0, 0}, cna[2] = {0, 0}, cnb[2] = {0, 0};
    rec->d0[1] = 0.0f; rec->tgt[1] = 0.0f;
    for (int k = 0; k < m.count; ++k) {
        rax[k] = m.p1x[k] - w->pos[2 * a]; ray[k] = m.p1y[k] - w->pos[2 * a + 1];
        rbx[k] = m.p2x[k] - w->pos[2 * b]; rby[k] = m.p2y[k] - w->pos[2 * b + 1];
        /* pre-solve normal velocity of b relative to a (velocities are still the integrated ones) */
        float vax = w->vel[2 * a] - w->angvel[a] * ray[k], vay = w->vel[2 * a + 1] + w->angvel[a] * rax[k];
        float vbx = w->vel[2 * b] - w->angvel[b] * rby[k], vby = w->vel[2 * b + 1] + w->angvel[b] * rbx[k];
        float vpre = (vbx - vax) * nx + (vby - vay) * ny;
        float e = fabsf(vpre) <= vthr ? 0.0f : rest;
        rec->tgt[k] = fmaxf(-e * vpre, 0.0f);
        rec->d0[k] = (m.p1x[k] - m.p2x[k]) * nx + (m.p1y[k] - m.p2y[k]) * ny;
        cna[k] = rax[k] * ny - ray[k] * nx;
        cnb[k] = rbx[k] * ny - rby[k] * nx;
    }
    /* effective masses: the sum of the inverse masses plus the sum of the two bodies' angular terms */
    const float K00 = ims + (iia * (cna[0] * cna[0]) + iib * (cnb[0] * cnb[0]));
    if (!(K00 > 0.0f)) { rec->count = 0; return; }
    rec->iK[0] = 1.0f / K00; rec->iK[1] = 0.0f;
    rec->block = 0; rec->i00 = rec->i01 = rec->i11 = rec->K01 = 0.0f;
    if (rec->count == 2) {
        const float K11 = ims + (iia * (cna[1] * cna[1]) + iib * (cnb[1] * cnb[1]));
        const float K01 = ims + (iia * (cna[0] * cna[1]) + iib * (cnb[0] * cnb[1]));
        if (!(K11 > 0.0f)) {
            rec->count = 1;
        } else {
            rec->iK[1] = 1.0f / K11; rec->K01 = K01;
            const float det = K00 * K11 - K01 * K01;
            if (det > 1e-6f * (K00 * K11)) {
                const float inv = 1.0f / det;
                rec->block = 1; rec->i00 = K11 * inv; rec->i01 = K01 * inv; rec->i11 = K00 * inv;
            }
        }
    }
    if (rec->count == 1) { cna[1] = 0.0f; cnb[1] = 0.0f; rec->d0[1] = 0.0f; rec->tgt[1] = 0.0f; }
    rec->ca[0] = cna[0]; rec->ca[1] = cna[1]; rec->cb[0] = -cnb[0]; rec->cb[1] = -cnb[1];
    /* the tangent row: t = (-ny, nx) at the mean anchor */
    float max_, may_, mbx_, mby_;
    if (rec->count == 2) {
        max_ = 0.5f * (rax[0] + rax[1]); may_ = 0.5f * (ray[0] + ray[1]);
        mbx_ = 0.5f * (rbx[0] + rbx[1]); mby_ = 0.5f * (rby[0] + rby[1]);
    } else { max_ = rax[0]; may_ = ray[0]; mbx_ = rbx[0]; mby_ = rby[0]; }
    const float tx = -ny, ty = nx;
    const float cta = max_ * ty - may_ * tx, ctb = mbx_ * ty - mby_ * tx;
    rec->cta = cta; rec->ctb = -ctb;
    const float Att = ims + (iia * (cta * cta) + iib * (ctb * ctb));
    const float At0 = iia * (cta * cna[0]) + iib * (ctb * cnb[0]);
    const float At1 = iia * (cta * cna[1]) + iib * (ctb * cnb[1]);
    rec->Att = Att; rec->At[0] = At0; rec->At[1] = At1;
    float y0, y1;
    kinv_apply(rec, 1, rec->count == 2, At0, At1, &y0, &y1);
    rec->invS = inv_pos(Att - (At0 * y0 + At1 * y1));
    rec->invS0 = 0.0f; rec->invS1 = 0.0f;
    if (rec->count == 2) {
        rec->invS0 = inv_pos(Att - At0 * (At0 * rec->iK[0]));
        rec->invS1 = inv_pos(Att - At1 * (At1 * rec->iK[1]));
    }
}

/* [XPBD-RB] sec. 3.5, second half, on the frozen rows: the depth of a point is its depth at the narrowphase plus
 * what the displacements accumulated since (dpos) change along its row; the manifold's normal constraints are solved
 * as a block; the displacements of the two bodies grow.  Friction is not applied here but in the velocity pass. */
static void solve_pos_pair(sfo_world* w, uint32_t a, uint32_t b, pair_rec* rec, float max_corr) {
    int count = (int)rec->count;
    if (!count) return;
    const float nx = rec->nx, ny = rec->ny;
    const float ima = w->inv_mass[a], iia = w->inv_inertia[a], imb = w->inv_mass[b], iib = w->inv_inertia[b];
    float* Da = &w->dpos[3 * a];
    float* Db = &w->dpos[3 * b];
    const float la = Da[0] * nx + Da[1] * ny, lb = -(Db[0] * nx + Db[1] * ny);
    float d[2] = {0.0f, 0.0f};
    for (int k = 0; k < count; ++k) {
        const float ga = la + Da[2] * rec->ca[k], gb = lb + Db[2] * rec->cb[k];
        d[k] = rec->d0[k] + (ga + gb);
        d[k] = fminf(d[k], max_corr); /* [DESIGN] bounded depenetration per substep */
    }
    float l0 = 0.0f, l1 = 0.0f;
    if (count == 2) solve_lcp2(rec, d[0], d[1], &l0, &l1);
    else l0 = fmaxf(d[0], 0.0f) * rec->iK[0];
    rec->lam[0] = l0; rec->lam[1] = l1;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return;
    /* both normal impulses at once: a moves against n, b along n */
    Da[0] = Da[0] - (nx * lt) * ima;
    Da[1] = Da[1] - (ny * lt) * ima;
    Da[2] = Da[2] - iia * (rec->ca[0] * l0 + rec->ca[1] * l1);
    Db[0] = Db[0] + (nx * lt) * imb;
    Db[1] = Db[1] + (ny * lt) * imb;
    Db[2] = Db[2] - iib * (rec->cb[0] * l0 + rec->cb[1] * l1);
}

/* [XPBD-RB] sec. 3.6, as a per-manifold block ([DESIGN] DESIGN.md "Contract / velocity pass"), on the frozen rows:
 *  1. restitution (eq. 34): the normal velocities of the active points are replaced by
 *     max(-e vn_pre, 0), both points solved together;
 *  2. friction once per manifold along t = (-ny, nx) at the mean anchor, with compensating normal impulses so
 *     that step 1 stays satisfied: static (eq. 27-28, applied to velocity AND pose, i.e. the
 *     tangential displacement of this substep is undone; the pose part accumulates in dshift and is applied at the end
 *     of the phase) when the stopping impulse fits the static cone, else dynamic (eq. 30). */
static void solve_vel_pair(sfo_world* w, uint32_t a, uint32_t b, const pair_rec* rec, float h, float inv_h) {
    const float nx = rec->nx, ny = rec->ny;
    const int count = (int)rec->count;
    const float l0 = rec->lam[0], l1 = count == 2 ? rec->lam[1] : 0.0f;
    const float lt = l0 + l1;
    if (!(lt > 0.0f)) return;
    const float ima = w->inv_mass[a], iia = w->inv_inertia[a], imb = w->inv_mass[b], iib = w->inv_inertia[b];
    const int act0 = l0 > 0.0f, act1 = count == 2 && l1 > 0.0f;
    {
        const float la = w->vel[2 * a] * nx + w->vel[2 * a + 1] * ny, lb = -(w->vel[2 * b] * nx + w->vel[2 * b + 1] * ny);
        float dv[2] = {0.0f, 0.0f};
        for (int k = 0; k < count; ++k) {
            const float ua = la + w->angvel[a] * rec->ca[k], ub = lb + w->angvel[b] * rec->cb[k];
            dv[k] = rec->tgt[k] + (ua + ub);   /* target minus vn, vn = -(ua + ub) */
        }
        float p0, p1;
        kinv_apply(rec, act0, act1, dv[0], dv[1], &p0, &p1);
        const float pt = p0 + p1;
        w->vel[2 * a] = w->vel[2 * a] - (nx * pt) * ima;
        w->vel[2 * a + 1] = w->vel[2 * a + 1] - (ny * pt) * ima;
        w->angvel[a] = w->angvel[a] - iia * (rec->ca[0] * p0 + rec->ca[1] * p1);
        w->vel[2 * b] = w->vel[2 * b] + (nx * pt) * imb;
        w->vel[2 * b + 1] = w->vel[2 * b + 1] + (ny * pt) * imb;
        w->angvel[b] = w->angvel[b] - iib * (rec->cb[0] * p0 + rec->cb[1] * p1);
    }
    if (w->cfg.flags & SFX_CFG_NO_FRICTION) return;
    const float tx = -ny, ty = nx;
    const float wa = (w->vel[2 * a] * tx + w->vel[2 * a + 1] * ty) + w->angvel[a] * rec->cta;
    const float wb = -(w->vel[2 * b] * tx + w->vel[2 * b + 1] * ty) + w->angvel[b] * rec->ctb;
    const float vt = -(wa + wb);   /* of b relative to a along t, signed */
    if (!(fabsf(vt) > EPS_LEN)) return;
    float y0, y1;
    kinv_apply(rec, act0, act1, rec->At[0], rec->At[1], &y0, &y1);
    const float invS = (act0 && (act1 || count == 1)) ? rec->invS : (act0 ? rec->invS0 : rec->invS1);
    const float need = vt * invS;
    if (need == 0.0f) return;
    const int is_static = fabsf(need) * h < rec->mu_s * lt;
    const float pt = is_static ? -need : -copysignf(fminf(rec->mu_d * lt * inv_h, fabsf(need)), need);
    const float q0 = -(y0 * pt), q1 = -(y1 * pt);
    /* total impulse on b (a gets the opposite): pt t + (q0 + q1) n, with the torques of the frozen rows */
    const float jx = tx * pt + nx * (q0 + q1), jy = ty * pt + ny * (q0 + q1);
    const float ja = (rec->cta * pt + rec->ca[0] * q0) + rec->ca[1] * q1;
    const float jb = (rec->ctb * pt + rec->cb[0] * q0) + rec->cb[1] * q1;
    w->vel[2 * a] = w->vel[2 * a] - jx * ima;
    w->vel[2 * a + 1] = w->vel[2 * a + 1] - jy * ima;
    w->angvel[a] = w->angvel[a] - iia * ja;
    w->vel[2 * b] = w->vel[2 * b] + jx * imb;
    w->vel[2 * b + 1] = w->vel[2 * b + 1] + jy * imb;
    w->angvel[b] = w->angvel[b] - iib * jb;
    if (is_static) {
        float* Sa = &w->dshift[3 * a];
        float* Sb = &w->dshift[3 * b];
        Sa[0] = Sa[0] - (jx * h) * ima;
        Sa[1] = Sa[1] - (jy * h) * ima;
        Sa[2] = Sa[2] - iia * (ja * h);
        Sb[0] = Sb[0] + (jx * h) * imb;
        Sb[1] = Sb[1] + (jy * h) * imb;
        Sb[2] = Sb[2] - iib * (jb * h);
    }
}

/* end of a velocity phase: the static-friction shifts of the bodies of pair e are applied and cleared */
static void apply_shift(sfo_world* w, uint32_t i) {
    float* S = &w->dshift[3 * i];
    if (S[0] == 0.0f && S[1] == 0.0f && S[2] == 0.0f) return;
    if (is_dynamic(w, i)) apply_delta(w, i, S);
    S[0] = 0.0f; S[1] = 0.0f; S[2] = 0.0f;
}

/* [XPBD-RB] sec. 3.3 / [XPBD] eq. 18: distance constraint with compliance, lambda restarts
 * at 0 every substep (one iteration per substep) */
static void solve_joint(sfo_world* w, const sfx_distance_constraint* j, float inv_h2) {
    uint32_t a = j->body_a, b = j->body_b;
    float rax = 0, ray = 0, rbx = 0, rby = 0, pax, pay, pbx, pby;
    if (a != SFX_WORLD_BODY) {
        float c = w->rot[2 * a], s = w->rot[2 * a + 1];
        rax = c * j->anchor_a[0] - s * j->anchor_a[1]; ray = s * j->anchor_a[0] + c * j->anchor_a[1];
        pax = w->pos[2 * a] + rax; pay = w->pos[2 * a + 1] + ray;
    } else { pax = j->anchor_a[0]; pay = j->anchor_a[1]; }
    if (b != SFX_WORLD_BODY) {
        float c = w->rot[2 * b], s = w->rot[2 * b + 1];
        rbx = c * j->anchor_b[0] - s * j->anchor_b[1]; rby = s * j->anchor_b[0] + c * j->anchor_b[1];
        pbx = w->pos[2 * b] + rbx; pby = w->pos[2 * b + 1] + rby;
    } else { pbx = j->anchor_b[0]; pby = j->anchor_b[1]; }
    float dx = pbx - pax, dy = pby - pay;
    float len = sqrtf(dx * dx + dy * dy);
    if (!(len > EPS_LEN)) return;
    float nx = dx / len, ny = dy / len;
    float C = len - j->rest_length;
    /* one-sided links (sfx.h SFX_LIMIT_*): a rope only pulls, a strut only pushes */
    if (j->limit == SFX_LIMIT_ROPE && !(C > 0.0f)) return;
    if (j->limit == SFX_LIMIT_STRUT && !(C < 0.0f)) return;
    float wsum = j->compliance * inv_h2;
    if (a != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(w, a, rax, ray, nx, ny);
    if (b != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(w, b, rbx, rby, nx, ny);
    if (!(wsum > 0.0f)) return;
    float corr = C / wsum;
    if (a != SFX_WORLD_BODY) apply_pos(w, a, rax, ray, nx * corr, ny * corr, 1.0f);
    if (b != SFX_WORLD_BODY) apply_pos(w, b, rbx, rby, nx * corr, ny * corr, -1.0f);
}

/* atan2 from two range reductions and the Cephes atanf polynomial (about 2 ulp): the same
 * operations as the device code, so both sides get the same bits */
static float atan_pos(float x) {
    float y = 0.0f;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    float z = x * x;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z;
    return y + (p * x + x);
}

static float atan2_det(float y, float x) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    float ax = fabsf(x), ay = fabsf(y), r;
    if (ax >= ay) r = atan_pos(ay / ax);
    else r = 1.5707963267948966f - atan_pos(ax / ay);
    if (x < 0.0f) r = 3.14159265358979323846f - r;
    return y < 0.0f ? -r : r;
}

float sfo_debug_atan2(float y, float x) { return atan2_det(y, x); } /* test hook */

/* [XPBD-RB] sec. 3.4.1 joint limits, in 2D a scalar: the relative angle of b against a is pushed
 * back into [min, max]; XPBD compliance as for the distance constraint */
static void solve_angle(sfo_world* w, const sfx_angle_constraint* j, float inv_h2) {
    uint32_t a = j->body_a, b = j->body_b;
    float ca = 1.0f, sa = 0.0f, cb = 1.0f, sb = 0.0f, iia = 0.0f, iib = 0.0f;
    if (a != SFX_WORLD_BODY) { ca = w->rot[2 * a]; sa = w->rot[2 * a + 1]; iia = w->inv_inertia[a]; }
    if (b != SFX_WORLD_BODY) { cb = w->rot[2 * b]; sb = w->rot[2 * b + 1]; iib = w->inv_inertia[b]; }
    float qc = cb * ca + sb * sa, qs = sb * ca - cb * sa;
    float phi = atan2_det(qs, qc);
    float C = phi - fminf(fmaxf(phi, j->min_angle), j->max_angle);
    if (C == 0.0f) return;
    float wsum = (iia + iib) + j->compliance * inv_h2;
    if (!(wsum > 0.0f)) return;
    float dl = C / wsum;
    if (iia > 0.0f) rotate_by(&w->rot[2 * a], &w->rot[2 * a + 1], iia * dl);
    if (iib > 0.0f) rotate_by(&w->rot[2 * b], &w->rot[2 * b + 1], -(iib * dl));
}

/* [XPBD-RB] sec. 3.3.1 / 3.6 joint damping in the velocity pass: the fraction min(damping h, 1) of
 * the relative velocity the joint sees is removed by a pair of opposite impulses */
static void damp_angle(sfo_world* w, const sfx_angle_constraint* j, float h) {
    if (!(j->damping > 0.0f)) return;
    uint32_t a = j->body_a, b = j->body_b;
    float frac = fminf(j->damping * h, 1.0f);
    float iia = a != SFX_WORLD_BODY ? w->inv_inertia[a] : 0.0f, iib = b != SFX_WORLD_BODY ? w->inv_inertia[b] : 0.0f;
    float oa = a != SFX_WORLD_BODY ? w->angvel[a] : 0.0f, ob = b != SFX_WORLD_BODY ? w->angvel[b] : 0.0f;
    float wsum = iia + iib;
    if (!(wsum > 0.0f)) return;
    float imp = ((ob - oa) * frac) / wsum;
    if (iia > 0.0f) w->angvel[a] = oa + iia * imp;
    if (iib > 0.0f) w->angvel[b] = ob - iib * imp;
}

static void damp_joint(sfo_world* w, const sfx_distance_constraint* j, float h) {
    if (!(j->damping > 0.0f)) return;
    uint32_t a = j->body_a, b = j->body_b;
    float frac = fminf(j->damping * h, 1.0f);
    float rax = 0, ray = 0, rbx = 0, rby = 0, pax = j->anchor_a[0], pay = j->anchor_a[1], pbx = j->anchor_b[0], pby = j->anchor_b[1];
    float vax = 0, vay = 0, oa = 0, vbx = 0, vby = 0, ob = 0;
    if (a != SFX_WORLD_BODY) {
        float c = w->rot[2 * a], s = w->rot[2 * a + 1];
        rax = c * j->anchor_a[0] - s * j->anchor_a[1]; ray = s * j->anchor_a[0] + c * j->anchor_a[1];
        pax = w->pos[2 * a] + rax; pay = w->pos[2 * a + 1] + ray;
        vax = w->vel[2 * a]; vay = w->vel[2 * a + 1]; oa = w->angvel[a];
    }
    if (b != SFX_WORLD_BODY) {
        float c = w->rot[2 * b], s = w->rot[2 * b + 1];
        rbx = c * j->anchor_b[0] - s * j->anchor_b[1]; rby = s * j->anchor_b[0] + c * j->anchor_b[1];
        pbx = w->pos[2 * b] + rbx; pby = w->pos[2 * b + 1] + rby;
        vbx = w->vel[2 * b]; vby = w->vel[2 * b + 1]; ob = w->angvel[b];
    }
    float dvx = (vbx - ob * rby) - (vax - oa * ray);
    float dvy = (vby + ob * rbx) - (vay + oa * rax);
    if (j->rest_length != 0.0f) { /* a link: only the rate of change of its length */
        float dx = pbx - pax, dy = pby - pay;
        float len = sqrtf(dx * dx + dy * dy);
        if (!(len > EPS_LEN)) return;
        float ux = dx / len, uy = dy / len;
        float rate = dvx * ux + dvy * uy;
        dvx = ux * rate; dvy = uy * rate;
    }
    float sp = sqrtf(dvx * dvx + dvy * dvy);
    if (!(sp > EPS_LEN)) return;
    float nx = dvx / sp, ny = dvy / sp;
    float wsum = 0.0f;
    if (a != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(w, a, rax, ray, nx, ny);
    if (b != SFX_WORLD_BODY) wsum = wsum + gen_inv_mass(w, b, rbx, rby, nx, ny);
    if (!(wsum > 0.0f)) return;
    float imp = (sp * frac) / wsum;
    float jx = nx * imp, jy = ny * imp;
    if (a != SFX_WORLD_BODY && (w->inv_mass[a] > 0.0f || w->inv_inertia[a] > 0.0f)) {
        w->vel[2 * a] = vax + jx * w->inv_mass[a]; w->vel[2 * a + 1] = vay + jy * w->inv_mass[a];
        w->angvel[a] = oa + w->inv_inertia[a] * (rax * jy - ray * jx);
    }
    if (b != SFX_WORLD_BODY && (w->inv_mass[b] > 0.0f || w->inv_inertia[b] > 0.0f)) {
        w->vel[2 * b] = vbx - jx * w->inv_mass[b]; w->vel[2 * b + 1] = vby - jy * w->inv_mass[b];
        w->angvel[b] = ob - w->inv_inertia[b] * (rbx * jy - rby * jx);
    }
}

/* the tick in pieces, mirroring sfx_tick_begin / sfx_tick_substep (slab drivers swap halo
 * bodies between substeps) */
int sfo_tick_begin(sfo_world* w, float frame_dt, int order) {
    if (!w->col || !w->mats) return SFX_ERR_INVALID;
    broadphase_core(w, frame_dt); /* capacity overflow is reported by sfo_stats, the tick goes on */
    uint32_t np = w->n_pairs < w->max_pairs ? w->n_pairs : w->max_pairs;
    for (uint32_t e = 0; e < np; ++e) w->rec[e].count = 0;
    w->tick_nbatch = 1;
    if (order == SFO_ORDER_SEQUENTIAL) {
        /* one batch, ascending pair key: what a single-threaded engine does */
        key_idx* ki = (key_idx*)malloc((np ? np : 1) * sizeof(key_idx));
        for (uint32_t e = 0; e < np; ++e) { ki[e].key = ((uint64_t)w->pa[e] << 32) | w->pb[e]; ki[e].idx = e; }
        qsort(ki, np, sizeof(key_idx), cmp_key_idx);
        uint32_t nk = 0; /* the reach test (DESIGN.md "Contract / solver pairs") applies to every order */
        for (uint32_t e = 0; e < np; ++e)
            if (pair_in_reach(w, w->pa[ki[e].idx], w->pb[ki[e].idx], frame_dt)) w->order[nk++] = ki[e].idx;
        free(ki);
        w->batch_end[0] = nk;
        w->n_solver_pairs = nk;
    } else {
        build_order(w, frame_dt);
        w->tick_nbatch = w->n_batches;
    }
    w->tick_h = frame_dt / (float)w->cfg.substeps;
    w->tick_order = order;
    w->tick_open = 1;
    return SFX_OK;
}

int sfo_tick_substep(sfo_world* w) {
    if (!w->tick_open) return SFX_ERR_INVALID;
    int order = w->tick_order;
    uint32_t nbatch = w->tick_nbatch;
    float h = w->tick_h;
    float inv_h = 1.0f / h;
    float inv_h2 = inv_h * inv_h;
    float max_corr = w->cfg.max_depenetration_velocity * h;
    integrate(w, h);
    for (uint32_t k = 0; k < w->m; ++k) {
        uint32_t jk = order == SFO_ORDER_SEQUENTIAL ? k : w->joint_perm[k];
        if (order != SFO_ORDER_SEQUENTIAL && w->joint_color[jk] == NO_COLOR) continue;
        if (jk < w->m_dist) solve_joint(w, &w->joints[jk], inv_h2);
        else solve_angle(w, &w->angles[jk - w->m_dist], inv_h2);
    }
    /* [DESIGN] DESIGN.md "Contract / sweep order": the manifolds of ALL solver pairs are computed
     * first, at the poses the substep's integration (and the joints) left -- Mueller et al. 2020
     * Alg. 2 "collect collision pairs" per substep -- then the solves run in the chosen order */
    uint32_t n_ord = nbatch ? w->batch_end[nbatch - 1] : 0;
    /* the manifolds are taken between COLLIDERS (entries, which mirror their bodies), the solves act on the BODIES */
    mirror_entries(w);
    /* (OpenMP build, libsfx_oracle_omp.so: every manifold is independent; in the coloured order the units of one phase
     * share no dynamic body, so they run side by side, each in its own order -- the same bits as the serial build) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n_ord > 2048)
#endif
    for (uint32_t k = 0; k < n_ord; ++k) prepare_pair(w, w->pa[w->order[k]], w->pb[w->order[k]], &w->rec[w->order[k]]);
    const int phased = order != SFO_ORDER_SEQUENTIAL;
    for (uint32_t ph = 0; ph < (phased ? 5u : 1u); ++ph) {
        const uint32_t b0 = phased ? (ph ? w->phase_end[ph - 1] : 0u) : 0u, b1 = phased ? w->phase_end[ph] : nbatch;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) if (b1 - b0 > 16)
#endif
        for (uint32_t bt = b0; bt < b1; ++bt)
            for (uint32_t k = bt ? w->batch_end[bt - 1] : 0u; k < w->batch_end[bt]; ++k)
                solve_pos_pair(w, w->cbody[w->pa[w->order[k]]], w->cbody[w->pb[w->order[k]]], &w->rec[w->order[k]], max_corr);
    }
    derive_velocities(w, inv_h);
    for (uint32_t k = 0; k < w->m; ++k) { /* joint damping, in the joints' solve order */
        uint32_t jk = order == SFO_ORDER_SEQUENTIAL ? k : w->joint_perm[k];
        if (order != SFO_ORDER_SEQUENTIAL && w->joint_color[jk] == NO_COLOR) continue;
        if (jk < w->m_dist) damp_joint(w, &w->joints[jk], h);
        else damp_angle(w, &w->angles[jk - w->m_dist], h);
    }
    for (uint32_t ph = 0; ph < (phased ? 5u : 1u); ++ph) {
        const uint32_t b0 = phased ? (ph ? w->phase_end[ph - 1] : 0u) : 0u, b1 = phased ? w->phase_end[ph] : nbatch;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) if (b1 - b0 > 16)
#endif
        for (uint32_t bt = b0; bt < b1; ++bt)
            for (uint32_t k = bt ? w->batch_end[bt - 1] : 0u; k < w->batch_end[bt]; ++k) {
                uint32_t e = w->order[k];
                if (!w->rec[e].count) continue;
                solve_vel_pair(w, w->cbody[w->pa[e]], w->cbody[w->pb[e]], &w->rec[e], h, inv_h);
                if (phased && ph == 4u) { /* the fallback list (one workgroup on global memory) applies its shifts pair by pair */
                    apply_shift(w, w->cbody[w->pa[e]]);
                    apply_shift(w, w->cbody[w->pb[e]]);
                }
            }
        /* the phase's static-friction shifts, once per body (the batches of a phase share no dynamic body) */
        for (uint32_t k = b0 ? w->batch_end[b0 - 1] : 0u; k < (b1 ? w->batch_end[b1 - 1] : 0u); ++k) {
            uint32_t e = w->order[k];
            if (!w->rec[e].count) continue;
            apply_shift(w, w->cbody[w->pa[e]]);
            apply_shift(w, w->cbody[w->pb[e]]);
        }
    }
    mirror_entries(w);
    /* stats of the last substep, measured on the records */
    uint32_t np = w->n_pairs < w->max_pairs ? w->n_pairs : w->max_pairs;
    w->n_contacts = 0; w->max_depth = 0.0f; w->n_manifolds = 0; w->n_manifolds2 = 0;
    for (uint32_t e = 0; e < np; ++e) {
        if (w->rec[e].count) w->n_manifolds++;
        if (w->rec[e].count == 2) w->n_manifolds2++;
        for (uint32_t k = 0; k < w->rec[e].count; ++k)
            if (w->rec[e].lam[k] > 0.0f) w->n_contacts++;
    }
    return SFX_OK;
}

/* threads the solves use: 1 in the serial build (libsfx_oracle.so), omp_get_max_threads() in libsfx_oracle_omp.so */
int sfo_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int sfo_step(sfo_world* w, float frame_dt, int order) {
    int rc = sfo_tick_begin(w, frame_dt, order);
    for (uint32_t s = 0; rc == SFX_OK && s < w->cfg.substeps; ++s) rc = sfo_tick_substep(w);
    w->tick_open = 0;
    return rc;
}

/* (pose, velocity) of the listed bodies as [n][8] = (x, y, cos, sin, vx, vy, omega, 0) */
int sfo_halo_get(const sfo_world* w, const uint32_t* idx, uint32_t n, float* out) {
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t i = idx[k];
        if (i >= w->nb) return SFX_ERR_INVALID;
        out[8 * k] = w->pos[2 * i]; out[8 * k + 1] = w->pos[2 * i + 1];
        out[8 * k + 2] = w->rot[2 * i]; out[8 * k + 3] = w->rot[2 * i + 1];
        out[8 * k + 4] = w->vel[2 * i]; out[8 * k + 5] = w->vel[2 * i + 1];
        out[8 * k + 6] = w->angvel[i]; out[8 * k + 7] = 0.0f;
    }
    return SFX_OK;
}

int sfo_halo_put(sfo_world* w, const uint32_t* idx, uint32_t n, const float* in) {
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t i = idx[k];
        if (i >= w->nb) return SFX_ERR_INVALID;
        w->pos[2 * i] = in[8 * k]; w->pos[2 * i + 1] = in[8 * k + 1];
        w->rot[2 * i] = in[8 * k + 2]; w->rot[2 * i + 1] = in[8 * k + 3];
        w->vel[2 * i] = in[8 * k + 4]; w->vel[2 * i + 1] = in[8 * k + 5];
        w->angvel[i] = in[8 * k + 6];
    }
    mirror_entries(w);
    return SFX_OK;
}

int sfo_download_poses(const sfo_world* w, float* pos, float* rot) {
    if (pos) memcpy(pos, w->pos, 2u * w->nb * sizeof(float));
    if (rot) memcpy(rot, w->rot, 2u * w->nb * sizeof(float));
    return SFX_OK;
}

int sfo_download_velocities(const sfo_world* w, float* vel, float* angvel) {
    if (vel) memcpy(vel, w->vel, 2u * w->nb * sizeof(float));
    if (angvel) memcpy(angvel, w->angvel, w->nb * sizeof(float));
    return SFX_OK;
}

int sfo_stats(const sfo_world* w, sfx_frame_stats* out) {
    memset(out, 0, sizeof(*out));
    out->n_bodies = w->nb; out->n_small = w->n_small; out->n_large = w->n_large;
    out->n_pairs = w->n_pairs; out->n_colors = w->n_colors; out->n_leftover = w->n_leftover;
    out->n_contacts = w->n_contacts; out->n_manifolds = w->n_manifolds; out->n_manifolds2 = w->n_manifolds2;
    out->n_joint_colors = w->n_joint_colors;
    out->overflow = w->overflow;
    out->n_tiles = w->n_tiles; out->n_fallback = w->n_fallback;
    out->max_tile_pairs = w->max_tile_pairs; out->max_tile_bodies = w->max_tile_bodies;
    out->n_solver_pairs = w->n_solver_pairs;
    memcpy(out->color_count, w->color_count, sizeof(w->color_count));
    /* deepest penetration at the current poses */
    float md = 0.0f;
    uint32_t np = w->n_pairs < w->max_pairs ? w->n_pairs : w->max_pairs;
    for (uint32_t e = 0; e < np; ++e) {
        manifold_t m;
        manifold(w, w->pa[e], w->pb[e], &m);
        for (int k = 0; k < m.count; ++k) if (m.depth[k] > md) md = m.depth[k];
    }
    out->max_depth = md;
    double ke = 0.0;
    for (uint32_t i = 0; i < w->nb; ++i) {
        if (w->inv_mass[i] > 0.0f)
            ke += 0.5 * (1.0 / w->inv_mass[i]) * ((double)w->vel[2 * i] * w->vel[2 * i] + (double)w->vel[2 * i + 1] * w->vel[2 * i + 1]);
        if (w->inv_inertia[i] > 0.0f)
            ke += 0.5 * (1.0 / w->inv_inertia[i]) * (double)w->angvel[i] * w->angvel[i];
    }
    out->kinetic_energy = (float)ke;
    return SFX_OK;
}

/* ------------------------------------------------------------------ spatial queries
 * [DESIGN] DESIGN.md "Contract / spatial queries" (no reference
 * source to follow).  Queries see the colliders at their current poses through their TIGHT
 * AABBs (the contract's AABB with margin 0 and no velocity pad).  Per collider the tests are
 *   point : box contains the point (closed) and the shape contains it (circle: d^2 <= r^2; core
 *           of >= 3 vertices: no face separates, else distance to the core boundary <= r);
 *   box   : closed-interval overlap of the two boxes;
 *   ray   : slab test of the box over [0, max_t], then the origin inside the shape (t = 0) or the
 *           first entry into the union of the core's offset faces and (r > 0) vertex discs.
 * Each query comes in two enumerations that must agree exactly: brute force over all colliders,
 * and the grid index (small colliders by home cell of the box's low corner, large ones listed). */

typedef struct {
    float* box;       /* [n][4] tight AABBs */
    uint32_t* start;  /* [ncell + 1] */
    uint32_t* slot;   /* small ids, bucket-major, ascending id inside a bucket */
    int32_t *cx, *cy; /* home cell per body */
    uint32_t* large;
    uint32_t ns, nl;
    float blo[2], bhi[2]; /* bounds of the small colliders' boxes */
} qindex;

static void qindex_free(qindex* q) {
    free(q->box); free(q->start); free(q->slot); free(q->cx); free(q->cy); free(q->large);
}

static void qindex_build(const sfo_world* w, qindex* q, int with_grid) {
    uint32_t n = w->n, GX = 1u << w->gxl, GY = 1u << w->gyl, ncell = GX * GY;
    memset(q, 0, sizeof(*q));
    q->box = (float*)malloc((n ? n : 1) * 4 * sizeof(float));
    for (uint32_t i = 0; i < n; ++i) compute_aabb_ex(w, i, 0.0f, 0.0f, 0.0f, &q->box[4 * i]);
    if (!with_grid) return;
    q->start = (uint32_t*)calloc((size_t)ncell + 1, sizeof(uint32_t));
    q->slot = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    q->cx = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    q->cy = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    q->large = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t* bucket = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    float lim = 0.9f * w->cell;
    q->blo[0] = q->blo[1] = INFINITY; q->bhi[0] = q->bhi[1] = -INFINITY;
    for (uint32_t i = 0; i < n; ++i) {
        bucket[i] = 0xFFFFFFFFu;
        if (w->col[i].type == SFX_SHAPE_NONE) continue;
        const float* A = &q->box[4 * i];
        if ((A[2] - A[0]) <= lim && (A[3] - A[1]) <= lim) {
            q->cx[i] = cell_coord(A[0], w->inv_cell);
            q->cy[i] = cell_coord(A[1], w->inv_cell);
            bucket[i] = (((uint32_t)q->cy[i] & (GY - 1)) << w->gxl) | ((uint32_t)q->cx[i] & (GX - 1));
            q->start[bucket[i] + 1]++;
            q->ns++;
            q->blo[0] = fminf(q->blo[0], A[0]); q->blo[1] = fminf(q->blo[1], A[1]);
            q->bhi[0] = fmaxf(q->bhi[0], A[2]); q->bhi[1] = fmaxf(q->bhi[1], A[3]);
        } else {
            q->large[q->nl++] = i;
        }
    }
    for (uint32_t c = 0; c < ncell; ++c) q->start[c + 1] += q->start[c];
    uint32_t* fill = (uint32_t*)malloc((size_t)ncell * sizeof(uint32_t));
    memcpy(fill, q->start, (size_t)ncell * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; ++i)
        if (bucket[i] != 0xFFFFFFFFu) q->slot[fill[bucket[i]]++] = i;
    free(fill); free(bucket);
}

static int shape_contains(const core_t* c, float px, float py) {
    if (c->n <= 0) return 0;
    if (c->n == 1) {
        float dx = px - c->vx[0], dy = py - c->vy[0];
        return dx * dx + dy * dy <= c->r * c->r;
    }
    if (c->n >= 3) {
        float sep = -INFINITY;
        for (int k = 0; k < c->n; ++k) {
            float s = c->nx[k] * (px - c->vx[k]) + c->ny[k] * (py - c->vy[k]);
            if (s > sep) sep = s;
        }
        if (sep <= 0.0f) return 1;
    }
    float qx, qy, d2;
    closest_on_poly(c, px, py, &qx, &qy, &d2);
    return d2 <= c->r * c->r;
}

/* first entry of the ray into the disc (cx, cy, r), origin outside it */
static int ray_disc(float ox, float oy, float dx, float dy, float cx, float cy, float r, float* t, float* nx, float* ny) {
    if (!(r > 0.0f)) return 0;
    float mx = ox - cx, my = oy - cy;
    float a = dx * dx + dy * dy;
    float b = mx * dx + my * dy;
    float c = (mx * mx + my * my) - r * r;
    if (!(a > 0.0f) || !(b < 0.0f)) return 0;
    float disc = b * b - a * c;
    if (!(disc >= 0.0f)) return 0;
    float tt = (-b - sqrtf(disc)) / a;
    if (!(tt >= 0.0f)) return 0;
    *t = tt;
    *nx = (mx + tt * dx) / r;
    *ny = (my + tt * dy) / r;
    return 1;
}

/* first entry of the ray into the posed shape, origin outside it; first minimum wins
 * (faces in order, then vertex discs in order) */
static int ray_shape(const core_t* c, float ox, float oy, float dx, float dy, float* t, float* nx, float* ny) {
    if (c->n <= 0) return 0;
    if (c->n == 1) return ray_disc(ox, oy, dx, dy, c->vx[0], c->vy[0], c->r, t, nx, ny);
    float best = INFINITY, bnx = 0.0f, bny = 0.0f;
    for (int k = 0; k < c->n; ++k) {
        int k1 = (k + 1 == c->n) ? 0 : k + 1;
        float denom = dx * c->nx[k] + dy * c->ny[k];
        if (denom < 0.0f) {
            float p0x = c->vx[k] + c->r * c->nx[k], p0y = c->vy[k] + c->r * c->ny[k];
            float num = (p0x - ox) * c->nx[k] + (p0y - oy) * c->ny[k];
            float tt = num / denom;
            if (tt >= 0.0f && tt < best) {
                float hx = ox + tt * dx, hy = oy + tt * dy;
                float ex = c->vx[k1] - c->vx[k], ey = c->vy[k1] - c->vy[k];
                float u = (hx - p0x) * ex + (hy - p0y) * ey;
                float ee = ex * ex + ey * ey;
                if (u >= 0.0f && u <= ee) { best = tt; bnx = c->nx[k]; bny = c->ny[k]; }
            }
        }
    }
    if (c->r > 0.0f) {
        for (int k = 0; k < c->n; ++k) {
            float tt, mx, my;
            if (ray_disc(ox, oy, dx, dy, c->vx[k], c->vy[k], c->r, &tt, &mx, &my) && tt < best) {
                best = tt; bnx = mx; bny = my;
            }
        }
    }
    if (!(best < INFINITY)) return 0;
    *t = best; *nx = bnx; *ny = bny;
    return 1;
}

static int ray_box(const float* A, float ox, float oy, float dx, float dy, float max_t) {
    float t0 = 0.0f, t1 = max_t;
    if (dx != 0.0f) {
        float ta = (A[0] - ox) / dx, tb = (A[2] - ox) / dx;
        if (ta > tb) { float s = ta; ta = tb; tb = s; }
        t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
    } else if (ox < A[0] || ox > A[2]) return 0;
    if (dy != 0.0f) {
        float ta = (A[1] - oy) / dy, tb = (A[3] - oy) / dy;
        if (ta > tb) { float s = ta; ta = tb; tb = s; }
        t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
    } else if (oy < A[1] || oy > A[3]) return 0;
    return t0 <= t1;
}

typedef struct { uint32_t body; float t, nx, ny; } ray_best;

static void ray_try(const sfo_world* w, const qindex* q, uint32_t i, uint32_t mask, const float* ray, float max_t,
                    ray_best* best) {
    if (!(w->col[i].category & mask)) return;
    if (!ray_box(&q->box[4 * i], ray[0], ray[1], ray[2], ray[3], max_t)) return;
    core_t c;
    build_core(w, i, &c);
    float t = 0.0f, nx = 0.0f, ny = 0.0f;
    if (!shape_contains(&c, ray[0], ray[1])) {
        if (!ray_shape(&c, ray[0], ray[1], ray[2], ray[3], &t, &nx, &ny)) return;
        if (!(t <= max_t)) return;
    }
    if (best->body == SFX_NO_BODY || t < best->t || (t == best->t && i < best->body)) {
        best->body = i; best->t = t; best->nx = nx; best->ny = ny; /* the collider; its body is reported */
    }
}

static void point_try(const sfo_world* w, const qindex* q, uint32_t i, uint32_t mask, float px, float py,
                      uint32_t* lowest, uint32_t* count) {
    if (!(w->col[i].category & mask)) return;
    const float* A = &q->box[4 * i];
    if (px < A[0] || px > A[2] || py < A[1] || py > A[3]) return;
    core_t c;
    build_core(w, i, &c);
    if (!shape_contains(&c, px, py)) return;
    if (i < *lowest) *lowest = i;
    ++*count;
}

/* clamps to the cell range of the small colliders before converting (rays may be infinite) */
static int32_t cell_clamped(float v_cells, float lo_cells, float hi_cells) {
    float f = floorf(fminf(fmaxf(v_cells, lo_cells - 2.0f), hi_cells + 2.0f));
    return (int32_t)f;
}

int sfo_query_points(const sfo_world* w, uint32_t n, const float* points, uint32_t mask, uint32_t* body,
                     uint32_t* count, int brute) {
    if (!w->col) return SFX_ERR_INVALID;
    qindex q;
    qindex_build(w, &q, !brute);
    uint32_t GX = 1u << w->gxl, GY = 1u << w->gyl;
    for (uint32_t k = 0; k < n; ++k) {
        float px = points[2 * k], py = points[2 * k + 1];
        uint32_t lowest = SFX_NO_BODY, cnt = 0;
        if (brute) {
            for (uint32_t i = 0; i < w->n; ++i)
                if (w->col[i].type != SFX_SHAPE_NONE) point_try(w, &q, i, mask, px, py, &lowest, &cnt);
        } else {
            /* a small box holding the point has its low corner in the point's cell or the one before */
            int32_t cx = cell_coord(px, w->inv_cell), cy = cell_coord(py, w->inv_cell);
            for (int32_t Y = cy - 1; Y <= cy; ++Y)
                for (int32_t X = cx - 1; X <= cx; ++X) {
                    uint32_t b = (((uint32_t)Y & (GY - 1)) << w->gxl) | ((uint32_t)X & (GX - 1));
                    for (uint32_t s = q.start[b]; s < q.start[b + 1]; ++s) {
                        uint32_t i = q.slot[s];
                        if (q.cx[i] == X && q.cy[i] == Y) point_try(w, &q, i, mask, px, py, &lowest, &cnt);
                    }
                }
            for (uint32_t l = 0; l < q.nl; ++l) point_try(w, &q, q.large[l], mask, px, py, &lowest, &cnt);
        }
        body[k] = lowest == SFX_NO_BODY ? lowest : w->cbody[lowest]; /* the body of the lowest collider id */
        if (count) count[k] = cnt;
    }
    qindex_free(&q);
    return SFX_OK;
}

size_t sfo_query_aabbs(const sfo_world* w, uint32_t n, const float* boxes, uint32_t mask, uint64_t* hits, size_t cap,
                       int brute) {
    if (!w->col) return 0;
    qindex q;
    qindex_build(w, &q, !brute);
    uint32_t GX = 1u << w->gxl, GY = 1u << w->gyl;
    size_t nh = 0;
#define BOX_TRY(i)                                                                              \
    do {                                                                                        \
        if ((w->col[(i)].category & mask) && aabb_overlap(B, &q.box[4 * (i)])) {                \
            if (nh < cap) hits[nh] = ((uint64_t)k << 32) | w->cbody[(i)];                       \
            ++nh;                                                                               \
        }                                                                                       \
    } while (0)
    for (uint32_t k = 0; k < n; ++k) {
        const float* B = &boxes[4 * k];
        if (brute) {
            for (uint32_t i = 0; i < w->n; ++i)
                if (w->col[i].type != SFX_SHAPE_NONE) BOX_TRY(i);
            continue;
        }
        if (q.ns && B[0] <= B[2] && B[1] <= B[3]) {
            float lox = q.blo[0] * w->inv_cell, hix = q.bhi[0] * w->inv_cell;
            float loy = q.blo[1] * w->inv_cell, hiy = q.bhi[1] * w->inv_cell;
            int32_t x0 = cell_clamped(B[0] * w->inv_cell, lox, hix) - 1, x1 = cell_clamped(B[2] * w->inv_cell, lox, hix);
            int32_t y0 = cell_clamped(B[1] * w->inv_cell, loy, hiy) - 1, y1 = cell_clamped(B[3] * w->inv_cell, loy, hiy);
            uint32_t nx = (uint32_t)(x1 - x0 + 1), ny = (uint32_t)(y1 - y0 + 1);
            if (nx > GX) nx = GX;
            if (ny > GY) ny = GY;
            for (uint32_t by = 0; by < ny; ++by)
                for (uint32_t bx = 0; bx < nx; ++bx) {
                    uint32_t b = ((((uint32_t)y0 + by) & (GY - 1)) << w->gxl) | (((uint32_t)x0 + bx) & (GX - 1));
                    for (uint32_t s = q.start[b]; s < q.start[b + 1]; ++s) {
                        uint32_t i = q.slot[s];
                        if (q.cx[i] >= x0 && q.cx[i] <= x1 && q.cy[i] >= y0 && q.cy[i] <= y1) BOX_TRY(i);
                    }
                }
        }
        for (uint32_t l = 0; l < q.nl; ++l) BOX_TRY(q.large[l]);
    }
#undef BOX_TRY
    qsort(hits, nh < cap ? nh : cap, sizeof(uint64_t), cmp_u64);
    qindex_free(&q);
    if (nh <= cap && w->n > w->nb) { /* a body is named once per query, however many of its colliders the box meets */
        size_t m = 0;
        for (size_t h = 0; h < nh; ++h)
            if (m == 0 || hits[h] != hits[m - 1]) hits[m++] = hits[h];
        nh = m;
    }
    return nh;
}

int sfo_query_rays(const sfo_world* w, uint32_t n, const float* rays, const float* max_t, uint32_t mask,
                   sfx_ray_hit* out, int brute) {
    if (!w->col) return SFX_ERR_INVALID;
    qindex q;
    qindex_build(w, &q, !brute);
    uint32_t GX = 1u << w->gxl, GY = 1u << w->gyl;
    for (uint32_t k = 0; k < n; ++k) {
        const float* ray = &rays[4 * k];
        ray_best best = {SFX_NO_BODY, 0.0f, 0.0f, 0.0f};
        float mt = max_t[k];
        if (mt >= 0.0f) {
            mt = fminf(mt, FLT_MAX);
            if (brute) {
                for (uint32_t i = 0; i < w->n; ++i)
                    if (w->col[i].type != SFX_SHAPE_NONE) ray_try(w, &q, i, mask, ray, mt, &best);
            } else {
                for (uint32_t l = 0; l < q.nl; ++l) ray_try(w, &q, q.large[l], mask, ray, mt, &best);
                if (q.ns) {
                    /* columns of home cells along the major axis; a collider homed in column X lies
                     * within [X, X + 1.9) cells of that axis */
                    const int M = fabsf(ray[3]) > fabsf(ray[2]) ? 1 : 0, m = 1 - M;
                    const float oM = ray[M], dM = ray[2 + M], om = ray[m], dm = ray[2 + m];
                    const float inv = w->inv_cell, cell = w->cell;
                    const float bloM = q.blo[M] * inv, bhiM = q.bhi[M] * inv, blom = q.blo[m] * inv, bhim = q.bhi[m] * inv;
                    const float a = oM * inv, b = (oM + mt * dM) * inv;
                    const int32_t X0 = cell_clamped(fminf(a, b) - 0.05f, bloM, bhiM) - 1;
                    const int32_t X1 = cell_clamped(fmaxf(a, b) + 0.05f, bloM, bhiM);
                    const int32_t step = dM < 0.0f ? -1 : 1;
                    for (int32_t X = step > 0 ? X0 : X1; X >= X0 && X <= X1; X += step) {
                        float ta = -INFINITY, tb = INFINITY;
                        if (dM != 0.0f) {
                            ta = (((float)X - 0.05f) * cell - oM) / dM;
                            tb = (((float)X + 2.0f) * cell - oM) / dM;
                            if (ta > tb) { float s = ta; ta = tb; tb = s; }
                        }
                        if (best.body != SFX_NO_BODY && best.t < ta) break; /* nothing further on can be closer */
                        const float tin = fminf(fmaxf(ta, 0.0f), mt), tout = fmaxf(fminf(tb, mt), 0.0f);
                        const float ya = (om + tin * dm) * inv, yb = (om + tout * dm) * inv;
                        const int32_t Y0 = cell_clamped(fminf(ya, yb) - 0.05f, blom, bhim) - 1;
                        const int32_t Y1 = cell_clamped(fmaxf(ya, yb) + 0.05f, blom, bhim);
                        for (int32_t Y = Y0; Y <= Y1; ++Y) {
                            const int32_t hx = M ? Y : X, hy = M ? X : Y;
                            uint32_t bk = (((uint32_t)hy & (GY - 1)) << w->gxl) | ((uint32_t)hx & (GX - 1));
                            for (uint32_t s = q.start[bk]; s < q.start[bk + 1]; ++s) {
                                uint32_t i = q.slot[s];
                                if (q.cx[i] == hx && q.cy[i] == hy) ray_try(w, &q, i, mask, ray, mt, &best);
                            }
                        }
                    }
                }
            }
        }
        out[k].body = best.body == SFX_NO_BODY ? best.body : w->cbody[best.body]; out[k].t = best.t; out[k].normal[0] = best.nx; out[k].normal[1] = best.ny;
    }
    qindex_free(&q);
    return SFX_OK;
}
