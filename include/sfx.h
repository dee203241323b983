/*
 * sfx.h -- C ABI of libsfx: the MI355X (gfx950) 2D XPBD physics tick.
 *
 * This is the drop-in boundary for starframe's per-frame physics tick
 * (SURVEY.md section 8b).  The reference (m0lentum/starframe, Rust) has NO FFI
 * or plugin interface for this path: physics is ordinary Rust called directly,
 * and its source is not present in /root/reference (README.md:1-3 is a redirect
 * stub).  Every entry point below therefore cites the *contract line* it
 * implements (BASELINE.json:5 "north_star") instead of a reference file:line;
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; all array arguments are caller-owned host
 *    memory borrowed for the duration of the call;
 *  - every function returns SFX_OK (0) or a negative sfx_status, never throws,
 *    never aborts; sfx_last_error() gives a human readable reason;
 *  - a world is not thread-safe (the reference tick takes `&mut self`);
 *  - scalars are f32, rotations are unit complex numbers (cos, sin), units are
 *    metres / seconds / radians.
 */
#ifndef SFX_H
#define SFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFX_ABI_VERSION 12u

typedef enum sfx_status {
    SFX_OK = 0,
    SFX_ERR_INVALID = -1,     /* bad argument / call order                      */
    SFX_ERR_UNSUPPORTED = -2, /* outside the documented limits                  */
    SFX_ERR_DEVICE = -3,      /* HIP / RCCL failure (message in sfx_last_error) */
    SFX_ERR_CAPACITY = -4,    /* a device buffer overflowed (pairs, colours)    */
    SFX_ERR_NOMEM = -5
} sfx_status;

/* collider shapes: BASELINE.json:5 "circle, rectangle and rounded-polygon/capsule" */
typedef enum sfx_shape_type {
    SFX_SHAPE_NONE = 0,    /* body has no collider                                     */
    SFX_SHAPE_CIRCLE = 1,  /* p[2] = radius                                            */
    SFX_SHAPE_RECT = 2,    /* p[0],p[1] = half extents, p[2] = rounding radius (>= 0)  */
    SFX_SHAPE_CAPSULE = 3, /* p[0] = half length along local x, p[2] = radius          */
    SFX_SHAPE_POLY = 4     /* convex CCW polygon, 3..8 verts in the pool, p[2] = rounding */
} sfx_shape_type;

#define SFX_MAX_POLY_VERTS 8u
#define SFX_WORLD_BODY 0xFFFFFFFFu /* constraint end attached to the world */
#define SFX_MAX_COLORS 64u
/* the velocity derive (omega = tan(dtheta) / h) lets one substep's corrections change a body's angular velocity by
 * at most this / h (DESIGN.md "Order"): tan has a pole at a quarter turn */
#define SFX_MAX_TAN_CORRECTION 0.25f
/* solver tiles (DESIGN.md "Contract / tiles"): a tile that owns more candidate pairs, or whose
 * 2x2 tile block holds more grid-sorted colliders, hands its pairs to the fallback list */
#define SFX_TILE_MAX_PAIRS 2048u
#define SFX_TILE_MAX_BODIES 1024u
#define SFX_TILE_WHOLE_GRID 0xFFFFu  /* sfx_config.tile_cells: one tile for the whole grid */
#define SFX_ONE_TILE_COLLIDERS 512u   /* automatic tiling: worlds up to this many colliders are one tile ...            */
#define SFX_ONE_TILE_LARGE 64u        /* ... unless more than this many of them may be too large for a base grid cell */
#define SFX_FALLBACK_TILE 0xFFFFFFFFu

/* physics config struct (BASELINE.json:5 "physics config") */
typedef struct sfx_config {
    uint32_t struct_size;   /* = sizeof(sfx_config), for ABI growth                  */
    int32_t device;         /* HIP device ordinal                                    */
    uint32_t substeps;      /* substeps per tick (>= 1)                              */
    float gravity[2];
    float aabb_margin;      /* fixed AABB padding (m)                                 */
    float vel_pad_max;      /* AABBs also grow by min(|v| * dt, vel_pad_max) per axis;
                               0 = 0.03 m; negative = no velocity padding              */
    float cell_size;        /* broadphase cell edge (m); 0 = derive from shapes      */
    uint32_t grid_log2[2];  /* wrapped grid is 2^gx by 2^gy cells; 0 = auto          */
    uint32_t max_pairs;     /* candidate pair capacity; 0 = 12 * bodies              */
    uint32_t color_rounds;  /* parallel colouring rounds before the serial finish; 0 = 12 */
    uint32_t tile_cells;    /* solver tile width in grid cells: 2, 4 or 8 (a tile is at least two cells wide
                               and high: DESIGN.md "Tiles, classes, phases"), or SFX_TILE_WHOLE_GRID: ONE
                               tile, the whole grid -- every pair is an interior pair, one unit, one
                               colouring, no seam phases.  0 = automatic (with tile_cells_y == 0): the whole
                               grid for a world of at most SFX_ONE_TILE_COLLIDERS colliders at upload, of
                               which at most SFX_ONE_TILE_LARGE larger than a grid cell; else 2          */
    uint32_t tile_cells_y;  /* solver tile height in grid cells, a power of two >= 2; anything >= the
                               grid height makes full-height strips: then a sweep has two
                               phases (interior pairs, seam pairs) instead of four.  0 = automatic,
                               decided when the colliders are uploaded: strips if no two neighbouring
                               tile columns hold more than 7/8 SFX_TILE_MAX_BODIES colliders at that
                               time (wide, shallow scenes), else as high as wide                   */
    float restitution_threshold; /* restitution is dropped when |vn_pre| <= this * |g| * h; 0 = 2.0 */
    float max_depenetration_velocity; /* > 0: a contact corrects at most this * h of penetration per
                               substep (m/s).  0 (default) or negative = unlimited, as published:
                               a clamp bounds the load a contact can carry -- at 3 m/s and 4
                               substeps 73 bodies -- and deeper piles collapse       */
    float speculative_margin; /* the narrowphase keeps contact points up to this far apart
                               (negative depth; they only push once they penetrate);
                               0 = 0.010 m (contacts that open and close inside a sweep are otherwise missed:
                               measured in DESIGN.md "Jitter"); negative = none          */
    float linear_damping;   /* 1/s: v *= 1 / (1 + h * c) every substep; 0 = none        */
    float angular_damping;  /* 1/s, same for the angular velocity                       */
    uint32_t flags;         /* SFX_CFG_* */
} sfx_config;

#define SFX_CFG_MULTI_LAUNCH 1u  /* never take the one-workgroup path.  Default: a world that is ONE tile (tile_cells:
                                    automatic up to SFX_ONE_TILE_COLLIDERS colliders) runs its substeps in ONE
                                    workgroup with the bodies in LDS -- one launch per sfx_step, one per
                                    sfx_tick_substep -- with bit-identical results (scheduling only)       */
#define SFX_CFG_ONE_WORKGROUP 8u /* a world of at most 2048 bodies takes the one-workgroup path whatever its tiling
                                    (slower than the default for tiled worlds above a few hundred bodies: measured
                                    in DESIGN.md)                                                                   */
#define SFX_CFG_NO_FRICTION 2u   /* skip static + dynamic friction (debug)          */
#define SFX_CFG_ONE_LEVEL 4u     /* broadphase grid without its second level: every collider too large for a base
                                    cell goes on the flat list (tested against everything)                     */

/* collider component (BASELINE.json:5 "body/collider/material components"; SURVEY.md section 8a row a4: shape,
 * local offset, layer mask).  The shape sits at a LOCAL POSE in the body's frame: local_pos (metres) and
 * local_angle (radians); both 0 = centred on the body origin.  Mass and inertia are the body's (the caller's: a
 * shape off the centre of mass has the inertia of the parallel axis theorem); AABB, narrowphase and queries use
 * the shape where it is, contact anchors are taken about the body origin.
 * SEVERAL COLLIDERS PER BODY: sfx_upload_colliders takes n >= body count colliders.  Collider k < bodies is the
 * first collider of body k (SFX_SHAPE_NONE: none; its `body` field is ignored); every collider k >= bodies is a
 * further collider of body colliders[k].body.  Colliders of one body never collide with each other.  The further
 * colliders of a non-dynamic body (level geometry) cost what any static collider costs; those of a DYNAMIC body are
 * solved on the fallback list (one workgroup, global memory): meant for a few compound bodies, not for thousands
 * (DESIGN.md "Several colliders per body"). */
typedef struct sfx_collider {
    uint32_t type;      /* sfx_shape_type                                   */
    uint32_t material;  /* index into the material table                    */
    uint32_t category;  /* layer bits this collider is in                   */
    uint32_t mask;      /* layer bits it collides with                      */
    uint32_t group;     /* != 0: never collides with the same group         */
    float p[3];         /* shape parameters, see sfx_shape_type             */
    uint32_t poly_offset; /* first vertex in the pool (SFX_SHAPE_POLY)      */
    uint32_t poly_count;  /* vertex count (SFX_SHAPE_POLY)                  */
    float local_pos[2];   /* shape origin in the body frame                 */
    float local_angle;    /* shape rotation in the body frame (radians)     */
    uint32_t body;        /* colliders beyond the body count: the body this one belongs to */
} sfx_collider;

typedef struct sfx_material {
    float restitution;
    float static_friction;
    float dynamic_friction;
} sfx_material;

/* distance constraint between two anchors (BASELINE.json:10 "distance-constrained links";
 * SURVEY.md section 8f rank 2 "attachment with offset anchors, compliance + damping, limits").
 * rest_length = 0 makes it an attachment (the two anchors are held together: a pin / revolute
 * joint).  limit makes it one-sided (a rope link only pulls). */
#define SFX_LIMIT_NONE 0u   /* holds the rest length both ways                           */
#define SFX_LIMIT_ROPE 1u   /* acts only while the anchors are further apart than rest   */
#define SFX_LIMIT_STRUT 2u  /* acts only while they are closer than rest                 */
typedef struct sfx_distance_constraint {
    uint32_t body_a;     /* body index or SFX_WORLD_BODY                         */
    uint32_t body_b;
    float anchor_a[2];   /* body-local anchor (world position for SFX_WORLD_BODY) */
    float anchor_b[2];
    float rest_length;
    float compliance;    /* XPBD alpha (m/N); 0 = rigid                           */
    uint32_t limit;      /* SFX_LIMIT_*                                           */
    float damping;       /* 1/s, >= 0: every substep the velocity pass removes the fraction
                            min(damping * h, 1) of the anchors' relative velocity -- along the
                            link for a link, all of it for an attachment (rest_length == 0)
                            (Mueller et al. 2020 sec. 3.3.1).  Two-sided constraints only:
                            must be 0 with SFX_LIMIT_ROPE / SFX_LIMIT_STRUT               */
} sfx_distance_constraint;

/* angle constraint (SURVEY.md section 8f rank 2 "angular limits"): keeps the relative angle
 * phi = angle(b) - angle(a), measured in (-pi, pi], inside [min_angle, max_angle].  min == max
 * holds one relative angle (with compliance: a torsion spring); together with an attachment it
 * makes a hinge with stops.  damping acts on the relative angular velocity at all times (a hinge
 * damper). */
typedef struct sfx_angle_constraint {
    uint32_t body_a;     /* body index or SFX_WORLD_BODY (angle 0)                */
    uint32_t body_b;
    float min_angle;     /* radians, -pi < min_angle <= max_angle < pi            */
    float max_angle;
    float compliance;    /* rad / (N m); 0 = rigid                                */
    float damping;       /* 1/s, >= 0                                             */
} sfx_angle_constraint;

/* one narrowphase manifold, as returned by sfx_debug_contacts */
typedef struct sfx_contact {
    uint32_t body_a, body_b; /* a < b                                        */
    uint32_t count;          /* 0, 1 or 2 points                             */
    float normal[2];         /* unit, points from a to b                     */
    float depth[2];          /* penetration depth per point (> 0)            */
    float point_a[2][2];     /* world point on a's surface, per point        */
    float point_b[2][2];     /* world point on b's surface, per point        */
} sfx_contact;

typedef struct sfx_frame_stats {
    uint32_t n_bodies;
    uint32_t n_small;        /* colliders sorted into the grid               */
    uint32_t n_large;        /* colliders tested against everything          */
    uint32_t n_pairs;        /* candidate pairs of the last tick (AABB overlap) */
    uint32_t n_colors;       /* most colours used by any tile / the fallback list */
    uint32_t n_leftover;     /* pairs coloured by the serial finishes        */
    uint32_t n_manifolds;    /* pairs that touched in the last substep       */
    uint32_t n_manifolds2;   /* ... of which with two contact points         */
    uint32_t n_contacts;     /* active contact points of the last substep    */
    uint32_t n_joint_colors;
    uint32_t overflow;       /* != 0: a capacity was exceeded this tick      */
    uint32_t launches;       /* kernel launches issued by the last tick      */
    uint32_t n_tiles;        /* tiles that own pairs (solved in LDS)         */
    uint32_t n_fallback;     /* pairs on the fallback list                   */
    uint32_t max_tile_pairs; /* most pairs owned by one tile                 */
    uint32_t max_tile_bodies;/* most grid-sorted colliders in one tile block */
    uint32_t n_solver_pairs; /* candidate pairs within reach (DESIGN.md "Contract / solver pairs"):
                                the ones the substeps run the narrowphase and the solves on */
    float max_depth;         /* deepest penetration seen in the last substep */
    float kinetic_energy;    /* sum 1/2 m v^2 + 1/2 I w^2 over dynamic bodies */
    uint32_t color_count[SFX_MAX_COLORS]; /* pairs per colour, summed over tiles + fallback */
} sfx_frame_stats;

/* closest hit of one ray, as returned by sfx_query_rays */
#define SFX_NO_BODY 0xFFFFFFFFu
typedef struct sfx_ray_hit {
    uint32_t body;   /* SFX_NO_BODY: nothing hit within max_t                        */
    float t;         /* hit point = origin + t * direction (0 when the origin is inside) */
    float normal[2]; /* outward unit normal of the surface hit; (0, 0) when t == 0 inside */
} sfx_ray_hit;

/* per kernel-class device time, filled while profiling is on */
typedef enum sfx_kernel_class {
    SFX_K_AABB_COUNT = 0, SFX_K_SCAN, SFX_K_SCATTER, SFX_K_PAIR_COUNT, SFX_K_PAIR_FILL,
    SFX_K_TILE_LIST, SFX_K_COLOR, SFX_K_INTEGRATE, SFX_K_SOLVE_JOINTS, SFX_K_SOLVE_POS,
    SFX_K_DERIVE_VEL, SFX_K_SOLVE_VEL, SFX_K_STATS, SFX_K_HALO, SFX_K_QUERY, SFX_K_NARROW,
    SFX_K_TICK /* all substeps of a tick in one launch: narrowphase, position and velocity sweeps (sfx_step) */, SFX_K_CLASSES
} sfx_kernel_class;

typedef struct sfx_kernel_time {
    double total_ms;   /* summed device time between the bracketing events */
    uint64_t launches;
    uint64_t items;    /* bodies / pairs / constraints processed            */
} sfx_kernel_time;

typedef struct sfx_world sfx_world;

/* ---- lifetime ------------------------------------------------------------ */
/* "physics config" -> world.  Fails with SFX_ERR_DEVICE when no HIP device is usable:
 * there is no CPU fallback behind this ABI. */
int sfx_create(const sfx_config* cfg, sfx_world** out);
void sfx_destroy(sfx_world* w); /* NULL is a no-op */
uint32_t sfx_abi_version(void);
/* the first 16 hex digits of the SHA-256 of the sources the library was compiled from (starframe_amd/build.py): what
 * tells a shipped libsfx.so from one that lags behind its tree */
const char* sfx_build_id(void);
void sfx_default_config(sfx_config* cfg);

/* ---- scene upload (BASELINE.json:5 "body/collider/material components") --- */
/* SoA body state; pos/rot/vel are [n][2], the rest [n].  inv_mass == 0 and
 * inv_inertia == 0 marks a static (or, with a velocity, kinematic) body. */
int sfx_upload_bodies(sfx_world* w, uint32_t n, const float* pos, const float* rot,
                      const float* vel, const float* angvel, const float* inv_mass,
                      const float* inv_inertia);
/* n >= the body count (see sfx_collider: the first `bodies` colliders are the bodies' own, the rest name their
 * body); bodies + further colliders <= 16 M.  poly_verts is the [n_poly_verts][2] vertex pool. */
int sfx_upload_colliders(sfx_world* w, uint32_t n, const sfx_collider* colliders,
                         const float* poly_verts, uint32_t n_poly_verts);
int sfx_upload_materials(sfx_world* w, uint32_t n, const sfx_material* materials);
/* the two constraint arrays are independent: uploading one keeps the other.  Together they form
 * one joint list (distance constraints first, then angle constraints), coloured at upload and
 * solved colour by colour before the contacts of every substep */
int sfx_upload_constraints(sfx_world* w, uint32_t m, const sfx_distance_constraint* c);
int sfx_upload_angle_constraints(sfx_world* w, uint32_t m, const sfx_angle_constraint* c);

/* ---- the tick (BASELINE.json:5 "the per-frame tick entry point") ---------- */
/* One frame: broadphase + colouring once, then `substeps` x (integrate, narrowphase, position
 * solve, velocity derive, velocity pass).  Asynchronous on the world's
 * stream; sfx_sync / any download waits for it. */
int sfx_step(sfx_world* w, float frame_dt);
/* Waits for everything enqueued.  Capacity overflows are detected on the device and kept in a sticky word until a
 * call reports them: sfx_sync returns SFX_ERR_CAPACITY -- once -- when ANY tick since the last sfx_sync exceeded a
 * capacity (candidate pairs, colours, fallback list, a fused launch that gave up waiting ...), however many ticks ago,
 * and clears the condition.  sfx_step / sfx_tick_begin refuse to start a tick (SFX_ERR_CAPACITY, state untouched)
 * once the host has seen such a bit (through its asynchronous counter read-back, a few ticks late) and until sfx_sync
 * has reported it; sfx_frame_stats.overflow shows the bits without clearing them. */
int sfx_sync(sfx_world* w);

/* ---- the tick in pieces + slab halos (BASELINE.json:5 "slab domain decomposition,
 * exchanging boundary bodies each substep") ------------------------------------------------
 * sfx_step == sfx_tick_begin, then `substeps` x sfx_tick_substep.  A slab driver (one world per
 * GPU, starframe_amd/slab.py) runs the pieces itself and swaps halo bodies between substeps:
 *   sfx_halo_set      the local bodies this world sends (its own, near a slab boundary) and the
 *                     ghost bodies it receives (copies of the neighbours' boundary bodies);
 *   sfx_halo_pack     gathers (pose, velocity) = 8 floats per send body into the send buffer;
 *   sfx_halo_unpack   scatters the receive buffer over the ghosts' poses and velocities;
 *   sfx_halo_buffers  the two device buffers (library-owned, sized by sfx_halo_set), which the
 *                     driver hands to RCCL send/recv on the world's stream;
 *   sfx_halo_download / _upload   the same through host memory (tests, gloo transport);
 *   sfx_halo_copy     device-to-device between two worlds of one process (logical slabs).
 * Everything is enqueued on the world's stream; sfx_set_stream makes that an existing HIP
 * stream (e.g. the one RCCL work is ordered against). */
int sfx_tick_begin(sfx_world* w, float frame_dt);   /* broadphase + tiles + colouring */
int sfx_tick_substep(sfx_world* w);                 /* one substep of the tick begun */
int sfx_set_stream(sfx_world* w, void* hip_stream); /* NULL = back to the world's own stream */
int sfx_get_stream(sfx_world* w, void** hip_stream); /* the hipStream_t everything is enqueued on (what RCCL work is ordered against) */
int sfx_halo_counts(sfx_world* w, uint32_t* n_send, uint32_t* n_recv); /* what sfx_halo_set was given */
int sfx_halo_set(sfx_world* w, const uint32_t* send_idx, uint32_t n_send, const uint32_t* recv_idx, uint32_t n_recv);
/* Halo lists are cut for the bodies' positions at sfx_halo_set: a body that was not ghosted lay more than the halo
 * width from the boundary.  sfx_halo_guard(w, d), d > 0, makes every later tick check (in its first kernel, no extra
 * launch) that no body has moved further than d on x since sfx_halo_set; when one has, overflow bit 128 is set and
 * the next sfx_sync / sfx_step returns SFX_ERR_CAPACITY -- cut the slabs again -- instead of seam contacts going
 * missing silently.  The slab drivers set d = halo / 4.  0 switches the guard off. */
int sfx_halo_guard(sfx_world* w, float max_move);
int sfx_halo_pack(sfx_world* w);
int sfx_halo_unpack(sfx_world* w);
int sfx_halo_buffers(sfx_world* w, float** send_dev, float** recv_dev);
int sfx_halo_download(sfx_world* w, float* send_host /* [n_send][8] */);   /* pack + copy, synchronises */
int sfx_halo_upload(sfx_world* w, const float* recv_host /* [n_recv][8] */); /* copy + unpack */
int sfx_halo_copy(sfx_world* dst, uint32_t dst_first, sfx_world* src, uint32_t src_first, uint32_t count);
/* Exchange / compute overlap (SURVEY.md section 5 "overlap it with interior-constraint solving"; DESIGN.md "Multi-GPU /
 * overlap"; ABI 11).  sfx_halo_overlap(w, 1): every later sfx_tick_substep runs its last kernel, the velocity sweep, as two
 * launches side by side -- the batches the halo bodies depend on, on a second stream of the library's (the halo stream),
 * and all the others on the world's stream -- and sfx_halo_pack, the transfer (sfx_halo_copy, sfx_rccl_halo_exchange, or
 * the caller's own RCCL calls on the stream sfx_halo_stream returns) and sfx_halo_download / _upload go on the halo stream
 * behind the first, so the halo travels while the second runs; sfx_halo_unpack makes the world's stream wait for it.
 * The caller's sequence stays substep, pack, transfer, unpack.  Same results, bit for bit, as with overlap off.  A tick
 * with pairs on the fallback list is not split.  Call it after sfx_halo_set, outside a tick; 0 switches it off.  Off by
 * default: the three stream hand-offs per substep cost about 35 us on an MI355X -- worth it only where the transfer takes
 * longer than that (measured: DESIGN.md).
 * sfx_halo_stream: the hipStream_t the halo's pack and transfer are enqueued on -- the world's stream with overlap off. */
int sfx_halo_overlap(sfx_world* w, int on);
int sfx_halo_stream(sfx_world* w, void** hip_stream);

int sfx_download_poses(sfx_world* w, float* pos, float* rot);
int sfx_download_velocities(sfx_world* w, float* vel, float* angvel);

/* ---- pose read-back for rendering ("double-buffered async D2H so the
 * sandbox can draw GPU-simulated scenes"; the reference draws from its own CPU state, so there is no
 * counterpart to cite) ------------------------------------------------------------------------------
 * sfx_readback_begin snapshots the poses as they are after everything enqueued so far (a device copy in
 * stream order) and starts their transfer into library-owned pinned host memory on a second stream; it
 * returns at once, and the next sfx_step overlaps the transfer.  sfx_readback_wait blocks until the OLDEST
 * read-back begun has landed and returns it: poses = [n][4] floats (x, y, cos, sin), valid until the
 * second sfx_readback_begin after the one it belongs to (the two slots alternate).  At most two read-backs are in flight (a frame loop is
 * begin(k); step(k+1); wait(k); draw).  Buffers are allocated at the first begin. */
int sfx_readback_begin(sfx_world* w);
int sfx_readback_wait(sfx_world* w, const float** poses, uint32_t* n);

/* ---- spatial queries over the device grid ("ray cast,
 * point/AABB query -- upstream's mouse-pick/raycast API"; the reference source is absent, so the
 * contract is DESIGN.md "Contract / spatial queries") -------------------------------------------
 * Queries see the colliders at their CURRENT poses.  The first query after a tick or an upload
 * sorts the colliders' tight AABBs (no margin, no velocity pad) into a grid index of its own
 * (same cell size as the tick's broadphase; allocated at the first query, like the transfer
 * buffers, which grow on demand), later queries reuse it.  A collider takes part when (category & mask) != 0.  Not allowed between sfx_tick_begin
 * and the tick's last sfx_tick_substep.  Arrays are caller-owned host memory; the calls
 * synchronise. */
/* point query: per point the lowest id of the bodies whose collider contains it (boundary
 * included; SFX_NO_BODY if none) and, if count != NULL, how many do */
int sfx_query_points(sfx_world* w, uint32_t n, const float* points /* [n][2] */, uint32_t mask,
                     uint32_t* body /* [n] */, uint32_t* count /* [n] or NULL */);
/* box query: (query << 32 | body) for every collider whose tight AABB overlaps box q =
 * (lo.x, lo.y, hi.x, hi.y) (closed intervals), ascending.  *n_hits is the full count even when
 * it exceeds cap: then SFX_ERR_CAPACITY is returned, hits holds some cap of them, and the caller
 * repeats the query with room for *n_hits */
int sfx_query_aabbs(sfx_world* w, uint32_t n, const float* boxes /* [n][4] */, uint32_t mask,
                    uint64_t* hits, size_t cap, size_t* n_hits);
/* ray cast: closest hit of origin + t * direction, 0 <= t <= max_t[q], per ray
 * (origin.x, origin.y, dir.x, dir.y); equal t: lowest body id.  direction need not be unit */
int sfx_query_rays(sfx_world* w, uint32_t n, const float* rays /* [n][4] */, const float* max_t /* [n] */,
                   uint32_t mask, sfx_ray_hit* hits /* [n] */);

/* ---- the entry point that allocates nothing ------------------------------------------------
 * sfx_query_points_static has NO counterpart in the reference (and none in SURVEY.md section 8): it is
 * sfx_query_points for callers that cannot allocate device memory or synchronise -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels and the same results as
 * sfx_query_points, but
 *  - the query index lives in a file-scope static device array of the library (10.5 MB per GPU the library
 *    runs on, held until the process ends) instead of buffers allocated at the first query;
 *  - points, body and count are DEVICE pointers ([n][2] floats, [n], [n] or NULL); nothing is copied;
 *  - everything is enqueued on the world's stream and the call returns without synchronising; it calls no
 *    hipMalloc / hipFree / allocating routine.
 * Limits: worlds of at most 131072 bodies + further colliders (one index entry per collider) on a grid of at most
 * 2^18 cells (the automatic grid of that many entries), else SFX_ERR_UNSUPPORTED; profiling must be off (event pools allocate), else SFX_ERR_INVALID.
 * The index is rebuilt by every call.  All worlds on a GPU share that GPU's array: calls that overlap on one
 * GPU (two streams, two threads) race; serialise them. */
int sfx_query_points_static(sfx_world* w, uint32_t n, const float* points_dev /* [n][2], device */, uint32_t mask,
                            uint32_t* body_dev /* [n], device */, uint32_t* count_dev /* [n] or NULL, device */);

/* ---- box and ray queries on device memory (no counterpart in the reference; SURVEY.md section 8f rank 1 for hosts
 * that keep their data on the GPU): boxes / rays / max_t / hits are DEVICE pointers, everything is enqueued on the
 * world's stream and the calls return without copying or synchronising (the query index is built on that stream too;
 * its buffers are allocated by a world's first query).
 * sfx_query_aabbs_device: hits_dev receives up to cap keys (query << 32 | BODY) in no particular order -- a body with
 * several colliders in the box appears once per collider; *n_hits_dev = the full count (compare with cap).
 * sfx_query_rays_device: as sfx_query_rays. */
int sfx_query_aabbs_device(sfx_world* w, uint32_t n, const float* boxes_dev /* [n][4] */, uint32_t mask,
                           uint64_t* hits_dev, uint32_t cap, uint32_t* n_hits_dev);
int sfx_query_rays_device(sfx_world* w, uint32_t n, const float* rays_dev /* [n][4] */, const float* max_t_dev /* [n] */,
                          uint32_t mask, sfx_ray_hit* hits_dev /* [n] */);

/* ---- parity hooks (slow paths, used by tests) ----------------------------- */
/* world AABBs [n][4] = (lo.x, lo.y, hi.x, hi.y) of the last tick's broadphase */
int sfx_debug_aabbs(sfx_world* w, float* out);
/* candidate pairs of the last tick as ascending keys (a << 32 | b), a < b */
int sfx_debug_pairs(sfx_world* w, uint64_t* keys, size_t cap, size_t* n);
/* the same pairs in device order with their owner tile (SFX_FALLBACK_TILE = fallback list)
 * and their colour inside that tile / list; tiles may be NULL */
int sfx_debug_pair_colors(sfx_world* w, uint64_t* keys, uint8_t* colors, uint32_t* tiles, size_t cap, size_t* n);
/* runs only the broadphase + colouring of a tick (no substeps), for dt = frame_dt */
int sfx_debug_broadphase(sfx_world* w, float frame_dt);
/* narrowphase of every candidate pair at the current poses, in sfx_debug_pairs order */
int sfx_debug_contacts(sfx_world* w, sfx_contact* out, size_t cap, size_t* n);
/* joint order after colouring: perm[k] = index into the joint list (distance constraints, then
 * angle constraints) */
int sfx_debug_joint_order(sfx_world* w, uint32_t* perm, uint8_t* colors, size_t cap, size_t* n);
/* the last tick's split of the velocity sweep (sfx_halo_overlap): batches on the near list (0xFFFFFFFF: that tick was
 * not split -- it had a fallback list; 0 with overlap off) and the tick's batches in all; synchronises */
int sfx_debug_near_batches(sfx_world* w, uint32_t* n_near, uint32_t* n_batches);

/* ---- reporting ------------------------------------------------------------ */
int sfx_stats(sfx_world* w, sfx_frame_stats* out); /* synchronises */
int sfx_set_profiling(sfx_world* w, int on);       /* 1: brackets every launch with events (and synchronises once per tick
                                                      to collect them); 2 * m: only the launches of the classes in the bit
                                                      mask m (bit SFX_K_x), nothing synchronises until sfx_kernel_times or
                                                      sfx_set_profiling(w, 0) collects them -- cheap enough for a timed run */
int sfx_kernel_times(sfx_world* w, sfx_kernel_time* out /* [SFX_K_CLASSES] */, int reset);
const char* sfx_kernel_class_name(int cls);
/* device time of everything enqueued between begin and end, via events on the world's stream */
int sfx_timer_begin(sfx_world* w);
int sfx_timer_end(sfx_world* w, double* ms); /* synchronises */
const char* sfx_last_error(sfx_world* w); /* w may be NULL: last create error */

#ifdef __cplusplus
}
#endif

/* Sizes of every struct that crosses the boundary: a binding in any language checks its mirrors against these
 * (bindings/rust/src/lib.rs has the same numbers as const assertions). */
#if defined(__cplusplus)
#define SFX_STATIC_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define SFX_STATIC_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
SFX_STATIC_ASSERT(sizeof(sfx_config) == 80, "sfx_config");
SFX_STATIC_ASSERT(sizeof(sfx_collider) == 56, "sfx_collider");
SFX_STATIC_ASSERT(sizeof(sfx_material) == 12, "sfx_material");
SFX_STATIC_ASSERT(sizeof(sfx_distance_constraint) == 40, "sfx_distance_constraint");
SFX_STATIC_ASSERT(sizeof(sfx_angle_constraint) == 24, "sfx_angle_constraint");
SFX_STATIC_ASSERT(sizeof(sfx_contact) == 60, "sfx_contact");
SFX_STATIC_ASSERT(sizeof(sfx_frame_stats) == 76 + 4 * SFX_MAX_COLORS, "sfx_frame_stats");
SFX_STATIC_ASSERT(sizeof(sfx_ray_hit) == 16, "sfx_ray_hit");
SFX_STATIC_ASSERT(sizeof(sfx_kernel_time) == 24, "sfx_kernel_time");
#endif /* SFX_H */
