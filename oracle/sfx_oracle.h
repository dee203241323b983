/*
 * sfx_oracle.h -- CPU oracle for the libsfx physics tick.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference implementation (m0lentum/starframe, Rust, on Codeberg) is
 * not present in /root/reference (README.md:1-3 is a redirect stub) and no Rust toolchain
 * exists in this image, so there are no reference file:line anchors, golden vectors or
 * reference runs to pin this restatement against.  It restates the published algorithm
 * named by the contract (BASELINE.json:5): Mueller et al., "Detailed Rigid Body Simulation
 * with Extended Position Based Dynamics" (SCA 2020), Algorithm 2 and sections 3.3-3.6, with
 * standard 2D SAT / closest-feature contact generation, under the [DESIGN] choices recorded
 * in SURVEY.md section 8c and DESIGN.md.  It is validated by analytic known-answer tests
 * (tests/test_oracle_kat.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  Nothing under starframe_amd/
 * may import, link or execute it.
 *
 * Plain C, single threaded, scalar f32, compiled with -ffp-contract=off so that its
 * arithmetic can be compared bit-for-bit with the HIP kernels where the contract demands
 * exactness (AABBs -> candidate pair sets, colours).
 */
#ifndef SFX_ORACLE_H
#define SFX_ORACLE_H

#include "../include/sfx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfo_world sfo_world;

/* constraint processing order of a tick */
#define SFO_ORDER_SEQUENTIAL 0 /* ascending pair key, upload order for joints: plain Gauss-Seidel,
                                  what a single-threaded CPU engine does ("reference-like")        */
#define SFO_ORDER_COLORED 1    /* tile phases, colour-major inside a tile, fallback list last: what
                                  libsfx does (DESIGN.md "Contract / tiles")                        */

sfo_world* sfo_create(const sfx_config* cfg);
void sfo_destroy(sfo_world* w);

int sfo_upload_bodies(sfo_world* w, uint32_t n, const float* pos, const float* rot,
                      const float* vel, const float* angvel, const float* inv_mass,
                      const float* inv_inertia);
int sfo_upload_colliders(sfo_world* w, uint32_t n, const sfx_collider* colliders,
                         const float* poly_verts, uint32_t n_poly_verts);
int sfo_upload_materials(sfo_world* w, uint32_t n, const sfx_material* materials);
int sfo_upload_constraints(sfo_world* w, uint32_t m, const sfx_distance_constraint* c);
int sfo_upload_angle_constraints(sfo_world* w, uint32_t m, const sfx_angle_constraint* c);
float sfo_debug_atan2(float y, float x); /* the deterministic atan2 of the angle constraint (test hook) */

/* resolved (auto-filled) configuration, for comparison with libsfx */
void sfo_resolved_config(const sfo_world* w, sfx_config* out);

int sfo_step(sfo_world* w, float frame_dt, int order);
/* the tick in pieces (sfo_step == begin + substeps x substep) and the halo state of listed bodies,
 * [n][8] = (x, y, cos, sin, vx, vy, omega, 0): what a slab driver swaps between substeps */
int sfo_tick_begin(sfo_world* w, float frame_dt, int order);
int sfo_tick_substep(sfo_world* w);
int sfo_halo_get(const sfo_world* w, const uint32_t* idx, uint32_t n, float* out);
int sfo_halo_put(sfo_world* w, const uint32_t* idx, uint32_t n, const float* in);

int sfo_download_poses(const sfo_world* w, float* pos, float* rot);
int sfo_download_velocities(const sfo_world* w, float* vel, float* angvel);

/* --- stage level access for parity tests ---------------------------------- */
int sfo_broadphase(sfo_world* w, float frame_dt);          /* AABBs + grid pairs + colouring */
int sfo_aabbs(const sfo_world* w, float* out);             /* [n][4] of the last broadphase   */
/* candidate pair set, ascending keys, three independent ways */
size_t sfo_pairs_grid(const sfo_world* w, uint64_t* keys, size_t cap);  /* last broadphase */
size_t sfo_pairs_sweep(const sfo_world* w, uint64_t* keys, size_t cap); /* sort & sweep on the same AABBs */
size_t sfo_pairs_brute(const sfo_world* w, uint64_t* keys, size_t cap); /* O(n^2) */
/* pairs in emission order with their owner tile (0xFFFFFFFF = fallback list) and their colour
 * inside that tile / list; 0 until a coloured-order broadphase or step has run */
size_t sfo_pair_colors(const sfo_world* w, uint64_t* keys, uint8_t* colors, uint32_t* tiles, size_t cap);
/* narrowphase on all candidate pairs at current poses, ascending key order */
size_t sfo_contacts(const sfo_world* w, sfx_contact* out, size_t cap);
/* narrowphase of one explicit pair at current poses (a < b not required) */
int sfo_contact_pair(const sfo_world* w, uint32_t a, uint32_t b, sfx_contact* out);
size_t sfo_joint_order(const sfo_world* w, uint32_t* perm, uint8_t* colors, size_t cap);
int sfo_stats(const sfo_world* w, sfx_frame_stats* out);

/* spatial queries at the current poses (sfx.h: sfx_query_*); brute != 0: over all colliders,
 * else through the grid index -- the two must agree exactly */
int sfo_query_points(const sfo_world* w, uint32_t n, const float* points, uint32_t mask, uint32_t* body,
                     uint32_t* count, int brute);
size_t sfo_query_aabbs(const sfo_world* w, uint32_t n, const float* boxes, uint32_t mask, uint64_t* hits, size_t cap,
                       int brute);
int sfo_query_rays(const sfo_world* w, uint32_t n, const float* rays, const float* max_t, uint32_t mask,
                   sfx_ray_hit* out, int brute);

/* threads the tick uses: 1 for libsfx_oracle.so; libsfx_oracle_omp.so (the same source built with -fopenmp, make
 * libsfx_oracle_omp.so) runs the per-body loops, the manifolds and -- in the coloured order -- the units of one phase
 * side by side and gives the same bits (tests/test_oracle_kat.py); SURVEY.md 8d "OpenMP-coloured variant" */
int sfo_threads(void);

#ifdef __cplusplus
}
#endif
#endif
