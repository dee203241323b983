/* sfx_c_tick.c -- one physics tick through the C ABI from plain C11 (no C++, no Python): what any FFI (Rust's
 * extern "C", cgo, JNI) sees of include/sfx.h.  A ball dropped on a static floor comes to rest on it; a second ball
 * hangs on a rope from a world point.  Exit status 0 when the poses are where closed-form physics puts them.
 *   gcc -std=c11 -Iinclude tools/sfx_c_tick.c -Lstarframe_amd/csrc -lsfx -lm -o sfx_c_tick */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "sfx.h"

/* every struct any binding mirrors: their sizes are part of the ABI (include/sfx.h has the same asserts) */
_Static_assert(sizeof(sfx_collider) == 56, "sfx_collider");
_Static_assert(sizeof(sfx_material) == 12, "sfx_material");
_Static_assert(sizeof(sfx_distance_constraint) == 40, "sfx_distance_constraint");

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != SFX_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sfx_last_error(w)); return 1; } \
    } while (0)

int main(void) {
    sfx_world* w = NULL;
    sfx_config cfg;
    sfx_default_config(&cfg);
    cfg.substeps = 8;
    if (sfx_abi_version() != SFX_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
    CHECK(sfx_create(&cfg, &w));
    /* bodies: 0 ball (falls), 1 bob (on a rope), 2 floor (static) */
    const float r = 0.25f, pi = 3.14159265f;
    const float m = pi * r * r, in = 0.5f * m * r * r;
    float pos[6] = {0.0f, 1.0f, 4.0f, 3.0f, 0.0f, -0.5f}, rot[6] = {1, 0, 1, 0, 1, 0}, vel[6] = {0}, angvel[3] = {0};
    float inv_mass[3] = {1.0f / m, 1.0f / m, 0.0f}, inv_inertia[3] = {1.0f / in, 1.0f / in, 0.0f};
    sfx_collider col[3];
    memset(col, 0, sizeof(col));
    for (int k = 0; k < 3; ++k) { col[k].category = 1u; col[k].mask = 0xFFFFFFFFu; }
    col[0].type = SFX_SHAPE_CIRCLE; col[0].p[2] = r;
    col[1].type = SFX_SHAPE_CIRCLE; col[1].p[2] = r;
    col[2].type = SFX_SHAPE_RECT; col[2].p[0] = 20.0f; col[2].p[1] = 0.5f;
    sfx_material mat = {0.0f, 0.5f, 0.4f};
    sfx_distance_constraint rope;
    memset(&rope, 0, sizeof(rope));
    rope.body_a = SFX_WORLD_BODY; rope.body_b = 1u;
    rope.anchor_a[0] = 4.0f; rope.anchor_a[1] = 4.0f;   /* world point */
    rope.rest_length = 1.0f; rope.limit = SFX_LIMIT_ROPE;
    CHECK(sfx_upload_bodies(w, 3, pos, rot, vel, angvel, inv_mass, inv_inertia));
    CHECK(sfx_upload_materials(w, 1, &mat));
    CHECK(sfx_upload_colliders(w, 3, col, NULL, 0));
    CHECK(sfx_upload_constraints(w, 1, &rope));
    for (int f = 0; f < 180; ++f) CHECK(sfx_step(w, 1.0f / 60.0f));
    CHECK(sfx_sync(w));
    CHECK(sfx_download_poses(w, pos, rot));
    sfx_frame_stats st;
    CHECK(sfx_stats(w, &st));
    printf("ball y %.4f (rests at %.2f), bob %.4f below its anchor (rope 1.0), %u manifolds, %u launches per tick\n", pos[1], r,
           4.0f - pos[3], st.n_manifolds, st.launches);
    const int ok = fabsf(pos[1] - r) < 5e-3f && fabsf((4.0f - pos[3]) - 1.0f) < 5e-3f && fabsf(pos[2] - 4.0f) < 1e-3f && st.overflow == 0;
    sfx_destroy(w);
    return ok ? 0 : 1;
}
