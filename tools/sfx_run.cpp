// sfx_run -- native driver over the libsfx C ABI (include/sfx.h): loads a scene dumped by
// scripts/dump_scene.py, optionally settles it with damping, runs ticks and prints timings.
// Used for rocprofv3 runs (the program itself must follow `--`) and as a C example of the ABI.
//
//   sfx_run scene.bin [ticks=100] [settle=0] [tile_cells=0] [tile_cells_y=0]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/sfx.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != SFX_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sfx_last_error(w));          \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

struct Scene {
    uint32_t n = 0, n_mat = 0, n_poly = 0, m = 0, substeps = 4;
    std::vector<float> pos, rot, vel, angvel, inv_mass, inv_inertia, poly;
    std::vector<sfx_collider> col;
    std::vector<sfx_material> mats;
    std::vector<sfx_distance_constraint> joints;
};

template <class T>
static bool rd(FILE* f, std::vector<T>& v, size_t count) {
    v.resize(count);
    return count == 0 || fread(v.data(), sizeof(T), count, f) == count;
}

static bool load(const char* path, Scene& s) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    uint32_t hdr[6];
    bool ok = fread(hdr, 4, 6, f) == 6 && hdr[0] == 0x33584653u;  // "SFX3" (56-byte colliders: local pose, body)
    if (ok) {
        s.n = hdr[1]; s.n_mat = hdr[2]; s.n_poly = hdr[3]; s.m = hdr[4]; s.substeps = hdr[5];
        ok = rd(f, s.pos, 2 * (size_t)s.n) && rd(f, s.rot, 2 * (size_t)s.n) && rd(f, s.vel, 2 * (size_t)s.n) &&
             rd(f, s.angvel, s.n) && rd(f, s.inv_mass, s.n) && rd(f, s.inv_inertia, s.n) && rd(f, s.col, s.n) &&
             rd(f, s.mats, s.n_mat) && rd(f, s.poly, 2 * (size_t)s.n_poly) && rd(f, s.joints, s.m);
    }
    fclose(f);
    return ok;
}

static int upload(sfx_world* w, const Scene& s) {
    CHECK(sfx_upload_bodies(w, s.n, s.pos.data(), s.rot.data(), s.vel.data(), s.angvel.data(), s.inv_mass.data(),
                            s.inv_inertia.data()));
    CHECK(sfx_upload_materials(w, s.n_mat, s.mats.data()));
    CHECK(sfx_upload_colliders(w, s.n, s.col.data(), s.poly.data(), s.n_poly));
    CHECK(sfx_upload_constraints(w, s.m, s.joints.data()));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s scene.bin [ticks] [settle] [tile_cells] [tile_cells_y]\n", argv[0]); return 2; }
    const int ticks = argc > 2 ? atoi(argv[2]) : 100;
    const int settle = argc > 3 ? atoi(argv[3]) : 0;
    const int tile_cells = argc > 4 ? atoi(argv[4]) : 0;
    const int tile_cells_y = argc > 5 ? atoi(argv[5]) : 0;
    Scene s;
    if (!load(argv[1], s)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const float dt = 1.0f / 60.0f;
    sfx_world* w = nullptr;
    sfx_config cfg;
    sfx_default_config(&cfg);
    cfg.substeps = s.substeps;
    cfg.tile_cells = (uint32_t)tile_cells;
    cfg.tile_cells_y = (uint32_t)tile_cells_y;
    if (settle > 0) {
        sfx_config c2 = cfg;
        c2.linear_damping = 3.3f; c2.angular_damping = 1.0f;
        CHECK(sfx_create(&c2, &w));
        if (upload(w, s)) return 1;
        for (int k = 0; k < settle; ++k) CHECK(sfx_step(w, dt));
        CHECK(sfx_download_poses(w, s.pos.data(), s.rot.data()));
        std::fill(s.vel.begin(), s.vel.end(), 0.0f);
        std::fill(s.angvel.begin(), s.angvel.end(), 0.0f);
        sfx_destroy(w);
        w = nullptr;
    }
    CHECK(sfx_create(&cfg, &w));
    if (upload(w, s)) return 1;
    for (int k = 0; k < 10; ++k) CHECK(sfx_step(w, dt));
    CHECK(sfx_sync(w));
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(sfx_timer_begin(w));
    for (int k = 0; k < ticks; ++k) CHECK(sfx_step(w, dt));
    double ms = 0.0;
    CHECK(sfx_timer_end(w, &ms));
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    sfx_frame_stats st;
    CHECK(sfx_stats(w, &st));
    printf("bodies %u pairs %u manifolds %u tiles %u fallback %u colors %u launches %u\n", st.n_bodies, st.n_pairs,
           st.n_manifolds, st.n_tiles, st.n_fallback, st.n_colors, st.launches);
    printf("ticks %d wall %.4f ms/tick device %.4f ms/tick  %.1f M body-substeps/s\n", ticks, wall / ticks, ms / ticks,
           1e-3 * s.n * s.substeps * ticks / ms);
    {   // FNV-1a over the final state: two runs of one scene must agree bit for bit (the tick is deterministic)
        std::vector<float> vel(s.vel.size()), av(s.angvel.size());
        CHECK(sfx_download_poses(w, s.pos.data(), s.rot.data()));
        CHECK(sfx_download_velocities(w, vel.data(), av.data()));
        uint64_t h = 1469598103934665603ull;
        auto eat = [&h](const std::vector<float>& v) {
            const unsigned char* p = (const unsigned char*)v.data();
            for (size_t i = 0; i < v.size() * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
        };
        eat(s.pos); eat(s.rot); eat(vel); eat(av);
        printf("state checksum %016llx\n", (unsigned long long)h);
    }
    sfx_destroy(w);
    return 0;
}
