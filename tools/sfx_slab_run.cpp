// sfx_slab_run -- slab domain decomposition over plain RCCL, no torch, no Python: the path a Rust or C host takes
// (include/sfx.h + include/sfx_rccl.h).  One process per GPU: the program forks `nranks` ranks BEFORE anything touches
// the GPU; rank 0 makes the RCCL id and hands it over through a file.
//
//   sfx_slab_run scene.bin [ticks=100] [nranks=1]      one slab per process, rank r on device r % device count
//   sfx_slab_run scene.bin [ticks=100] --self           ONE process, two logical slabs on device 0: their halos travel
//                                                       through ncclSend / ncclRecv from rank 0 to itself, and the final
//                                                       state must equal, bit for bit, the same two slabs exchanging
//                                                       through sfx_halo_copy (the single-GPU test of the RCCL path)
//
// scene.bin: scripts/dump_scene.py (circles and static rects only: the planner below knows no other shape, like
// starframe_amd/slab.py it cuts the x axis at equal body counts and ghosts the neighbours' bodies within the halo).
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../include/sfx_rccl.h"

struct Scene {
    uint32_t n = 0, n_mat = 0, substeps = 4;
    std::vector<float> pos, rot, vel, angvel, inv_mass, inv_inertia;
    std::vector<sfx_collider> col;
    std::vector<sfx_material> mats;
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
    bool ok = fread(hdr, 4, 6, f) == 6 && hdr[0] == 0x33584653u && hdr[3] == 0 && hdr[4] == 0;  // no polygons, no joints
    if (ok) {
        s.n = hdr[1]; s.n_mat = hdr[2]; s.substeps = hdr[5];
        ok = rd(f, s.pos, 2 * (size_t)s.n) && rd(f, s.rot, 2 * (size_t)s.n) && rd(f, s.vel, 2 * (size_t)s.n) &&
             rd(f, s.angvel, s.n) && rd(f, s.inv_mass, s.n) && rd(f, s.inv_inertia, s.n) && rd(f, s.col, s.n) && rd(f, s.mats, s.n_mat);
    }
    fclose(f);
    return ok;
}

// ---- the slab plan of one rank (starframe_amd/slab.py::plan_slabs restated for circles + static rects)
struct Plan {
    std::vector<uint32_t> ids;            // global id of every local body: owned, ghosts (left, right), fixed
    uint32_t n_owned = 0;
    std::vector<uint32_t> send, recv;     // local indices: send = left block then right block, recv likewise
    uint32_t n_sl = 0, n_sr = 0, n_rl = 0, n_rr = 0;
    float halo = 0.0f;
};

static bool dynamic(const Scene& s, uint32_t i) { return s.inv_mass[i] > 0.0f || s.inv_inertia[i] > 0.0f; }
static float radius(const Scene& s, uint32_t i) {   // about the body origin: a shape off it reaches further
    const sfx_collider& c = s.col[i];
    const float off = std::sqrt(c.local_pos[0] * c.local_pos[0] + c.local_pos[1] * c.local_pos[1]);
    if (c.type == SFX_SHAPE_CIRCLE) return c.p[2] + off;
    if (c.type == SFX_SHAPE_RECT) return std::sqrt(c.p[0] * c.p[0] + c.p[1] * c.p[1]) + c.p[2] + off;
    return 0.0f;
}
static float xext(const Scene& s, uint32_t i) {   // half width of a non-turning rect's box
    const sfx_collider& c = s.col[i];
    if (c.type == SFX_SHAPE_RECT && c.local_pos[0] == 0.0f && c.local_pos[1] == 0.0f && c.local_angle == 0.0f)
        return std::fabs(s.rot[2 * i]) * c.p[0] + std::fabs(s.rot[2 * i + 1]) * c.p[1] + c.p[2];
    return radius(s, i);
}

static bool plan_slab(const Scene& s, int rank, int nranks, Plan& p) {
    std::vector<double> xd;
    float rmax = 0.0f;
    for (uint32_t i = 0; i < s.n; ++i) {
        if (s.col[i].type != SFX_SHAPE_CIRCLE && s.col[i].type != SFX_SHAPE_RECT && s.col[i].type != SFX_SHAPE_NONE) return false;
        if (dynamic(s, i)) { xd.push_back(s.pos[2 * i]); rmax = std::max(rmax, radius(s, i)); }
    }
    if ((int)xd.size() < nranks) return false;
    std::sort(xd.begin(), xd.end());
    // slab.py's default halo plus 1.5 m: surface bodies of a pile roll, and the halo guard (a quarter of this) should not
    // stop a run of a few hundred ticks; the extra ghosts are a fraction of a percent of a slab
    p.halo = 4.0f * rmax + 4.0f * (0.01f + 0.03f) + 1.5f;
    std::vector<double> bounds(nranks + 1);
    bounds[0] = -INFINITY; bounds[nranks] = INFINITY;
    for (int k = 1; k < nranks; ++k) { const size_t j = (size_t)k * xd.size() / nranks; bounds[k] = 0.5 * (xd[j - 1] + xd[j]); }
    auto owner = [&](double x) { int o = (int)(std::upper_bound(bounds.begin(), bounds.end(), x) - bounds.begin()) - 1; return std::min(std::max(o, 0), nranks - 1); };
    const double lo = bounds[rank], hi = bounds[rank + 1], halo = p.halo;
    std::vector<uint32_t> owned, gl, gr, fixed;
    for (uint32_t i = 0; i < s.n; ++i) {
        const double x = s.pos[2 * i], rad = radius(s, i);
        if (dynamic(s, i)) {
            const int o = owner(x);
            if (o == rank) owned.push_back(i);
            else if (o == rank - 1 && x + rad > lo - halo) gl.push_back(i);
            else if (o == rank + 1 && x - rad < hi + halo) gr.push_back(i);
            else if (std::abs(o - rank) > 1 && x + rad > lo - halo && x - rad < hi + halo) return false;  // slabs too thin
        } else if (x + xext(s, i) > lo - 2.0 * halo && x - xext(s, i) < hi + 2.0 * halo) {
            fixed.push_back(i);
        }
    }
    p.n_owned = (uint32_t)owned.size();
    p.ids = owned;
    p.ids.insert(p.ids.end(), gl.begin(), gl.end());
    p.ids.insert(p.ids.end(), gr.begin(), gr.end());
    p.ids.insert(p.ids.end(), fixed.begin(), fixed.end());
    // what the neighbours ghost from this rank, in ascending global id (their ghost lists are in that order too)
    std::vector<uint32_t> sl, sr;
    for (uint32_t k = 0; k < p.n_owned; ++k) {
        const uint32_t i = owned[k];
        const double x = s.pos[2 * i], rad = radius(s, i);
        if (rank > 0 && x - rad < lo + halo) sl.push_back(k);
        if (rank + 1 < nranks && x + rad > hi - halo) sr.push_back(k);
    }
    p.n_sl = (uint32_t)sl.size(); p.n_sr = (uint32_t)sr.size(); p.n_rl = (uint32_t)gl.size(); p.n_rr = (uint32_t)gr.size();
    p.send = sl; p.send.insert(p.send.end(), sr.begin(), sr.end());
    p.recv.resize(p.n_rl + p.n_rr);
    for (uint32_t k = 0; k < p.n_rl + p.n_rr; ++k) p.recv[k] = p.n_owned + k;
    return true;
}

#define CHECK(w, call)                                                                                   \
    do {                                                                                                 \
        int rc_ = (call);                                                                                \
        if (rc_ != SFX_OK) { fprintf(stderr, "%s -> %d: %s | %s\n", #call, rc_, sfx_last_error(w), sfx_rccl_last_error()); return 1; } \
    } while (0)

static int make_world(const Scene& s, const Plan& p, int device, sfx_world** out) {
    const uint32_t n = (uint32_t)p.ids.size();
    std::vector<float> pos(2 * n), rot(2 * n), vel(2 * n), av(n), im(n), ii(n);
    std::vector<sfx_collider> col(n);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t i = p.ids[k];
        pos[2 * k] = s.pos[2 * i]; pos[2 * k + 1] = s.pos[2 * i + 1]; rot[2 * k] = s.rot[2 * i]; rot[2 * k + 1] = s.rot[2 * i + 1];
        vel[2 * k] = s.vel[2 * i]; vel[2 * k + 1] = s.vel[2 * i + 1]; av[k] = s.angvel[i]; im[k] = s.inv_mass[i]; ii[k] = s.inv_inertia[i];
        col[k] = s.col[i];
    }
    sfx_config cfg;
    sfx_default_config(&cfg);
    cfg.substeps = s.substeps;
    cfg.device = device;
    sfx_world* w = nullptr;
    CHECK(w, sfx_create(&cfg, &w));
    CHECK(w, sfx_upload_bodies(w, n, pos.data(), rot.data(), vel.data(), av.data(), im.data(), ii.data()));
    CHECK(w, sfx_upload_materials(w, s.n_mat, s.mats.data()));
    CHECK(w, sfx_upload_colliders(w, n, col.data(), nullptr, 0));
    CHECK(w, sfx_upload_constraints(w, 0, nullptr));
    CHECK(w, sfx_halo_set(w, p.send.data(), (uint32_t)p.send.size(), p.recv.data(), (uint32_t)p.recv.size()));
    CHECK(w, sfx_halo_guard(w, 0.25f * p.halo));
    *out = w;
    return 0;
}

static uint64_t checksum(sfx_world* w, uint32_t n_owned, uint64_t h) {
    uint32_t ns = 0, nr = 0;
    (void)ns; (void)nr;
    sfx_frame_stats st;
    if (sfx_stats(w, &st)) return 0;
    std::vector<float> pos(2 * st.n_bodies), rot(2 * st.n_bodies);
    if (sfx_download_poses(w, pos.data(), rot.data())) return 0;
    const unsigned char* p = (const unsigned char*)pos.data();
    for (size_t i = 0; i < 2 * (size_t)n_owned * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
    p = (const unsigned char*)rot.data();
    for (size_t i = 0; i < 2 * (size_t)n_owned * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

// ---- one process, two logical slabs on device 0, halos through RCCL send-to-self or through sfx_halo_copy
static int run_self(const Scene& s, int ticks) {
    const float dt = 1.0f / 60.0f;
    uint64_t sums[2] = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {   // 0: RCCL, 1: device-to-device copies
        Plan p[2];
        sfx_world* w[2] = {nullptr, nullptr};
        for (int r = 0; r < 2; ++r) {
            if (!plan_slab(s, r, 2, p[r])) { fprintf(stderr, "cannot cut the scene into two slabs\n"); return 1; }
            if (make_world(s, p[r], 0, &w[r])) return 1;
        }
        if (p[0].n_sr != p[1].n_rl || p[1].n_sl != p[0].n_rr) { fprintf(stderr, "send and ghost lists do not mirror\n"); return 1; }
        sfx_rccl_comm* comm = nullptr;
        if (mode == 0) {
            unsigned char id[SFX_RCCL_ID_BYTES];
            CHECK(w[0], sfx_rccl_unique_id(id));
            CHECK(w[0], sfx_rccl_comm_create(id, 0, 1, 0, &comm));
            // both worlds on one stream: the two exchanges are then ordered like two ranks' would be by the transfers
            void* st = nullptr;
            CHECK(w[0], sfx_get_stream(w[0], &st));
            CHECK(w[1], sfx_set_stream(w[1], st));
        }
        for (int t = 0; t < ticks; ++t) {
            for (int r = 0; r < 2; ++r) CHECK(w[r], sfx_tick_begin(w[r], dt));
            for (uint32_t sub = 0; sub < s.substeps; ++sub) {
                for (int r = 0; r < 2; ++r) CHECK(w[r], sfx_tick_substep(w[r]));
                if (mode == 0) {
                    // both worlds pack; then each direction is one grouped ncclSend + ncclRecv from rank 0 to itself,
                    // straight from one world's send buffer into the other's receive buffer, on the shared stream
                    CHECK(w[0], sfx_halo_pack(w[0]));
                    CHECK(w[1], sfx_halo_pack(w[1]));
                    float *s0, *r0, *s1, *r1;
                    void* st = nullptr;
                    CHECK(w[0], sfx_get_stream(w[0], &st));
                    CHECK(w[0], sfx_halo_buffers(w[0], &s0, &r0));
                    CHECK(w[1], sfx_halo_buffers(w[1], &s1, &r1));
                    CHECK(w[0], sfx_rccl_sendrecv(comm, st, s0, 8 * (size_t)p[0].n_sr, 0, r1, 8 * (size_t)p[1].n_rl, 0));   // slab 0's right block -> slab 1's ghosts
                    CHECK(w[1], sfx_rccl_sendrecv(comm, st, s1, 8 * (size_t)p[1].n_sl, 0, r0, 8 * (size_t)p[0].n_rr, 0));   // slab 1's left block -> slab 0's ghosts
                    CHECK(w[0], sfx_halo_unpack(w[0]));
                    CHECK(w[1], sfx_halo_unpack(w[1]));
                } else {
                    for (int r = 0; r < 2; ++r) CHECK(w[r], sfx_halo_pack(w[r]));
                    CHECK(w[1], sfx_halo_copy(w[1], 0, w[0], 0, p[0].n_sr));
                    CHECK(w[0], sfx_halo_copy(w[0], 0, w[1], 0, p[1].n_sl));
                    for (int r = 0; r < 2; ++r) CHECK(w[r], sfx_halo_unpack(w[r]));
                }
            }
        }
        for (int r = 0; r < 2; ++r) CHECK(w[r], sfx_sync(w[r]));
        uint64_t h = 1469598103934665603ull;
        for (int r = 0; r < 2; ++r) h = checksum(w[r], p[r].n_owned, h);
        sums[mode] = h;
        sfx_frame_stats st;
        CHECK(w[0], sfx_stats(w[0], &st));
        printf("%s: slabs %u + %u owned, halos %u / %u bodies, manifolds (slab 0) %u, checksum %016llx\n", mode == 0 ? "rccl send-to-self" : "sfx_halo_copy",
               p[0].n_owned, p[1].n_owned, p[0].n_sr, p[1].n_sl, st.n_manifolds, (unsigned long long)h);
        for (int r = 0; r < 2; ++r) sfx_destroy(w[r]);
        sfx_rccl_comm_destroy(comm);
    }
    if (sums[0] != sums[1] || sums[0] == 0) { printf("MISMATCH\n"); return 1; }
    printf("self test ok\n");
    return 0;
}

// ---- one rank of the real thing
static int run_rank(const Scene& s, int ticks, int rank, int nranks, const std::string& idfile) {
    const float dt = 1.0f / 60.0f;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "rank %d: no GPU\n", rank); return 1; }
    const int device = rank % ndev;
    Plan p;
    if (!plan_slab(s, rank, nranks, p)) { fprintf(stderr, "rank %d: cannot plan %d slabs (thin slabs or unsupported shapes)\n", rank, nranks); return 1; }
    sfx_world* w = nullptr;
    if (make_world(s, p, device, &w)) return 1;
    unsigned char id[SFX_RCCL_ID_BYTES];
    if (rank == 0) {
        CHECK(w, sfx_rccl_unique_id(id));
        const std::string tmp = idfile + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 1;
        fclose(f);
        rename(tmp.c_str(), idfile.c_str());
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = fopen(idfile.c_str(), "rb")); ++tries) usleep(100000);
        if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "rank %d: no RCCL id\n", rank); return 1; }
        fclose(f);
    }
    sfx_rccl_comm* comm = nullptr;
    CHECK(w, sfx_rccl_comm_create(id, rank, nranks, device, &comm));
    const int left = rank > 0 ? rank - 1 : -1, right = rank + 1 < nranks ? rank + 1 : -1;
    auto tick = [&]() -> int {
        CHECK(w, sfx_tick_begin(w, dt));
        for (uint32_t sub = 0; sub < s.substeps; ++sub) {
            CHECK(w, sfx_tick_substep(w));
            CHECK(w, sfx_rccl_halo_exchange(comm, w, left, right, p.n_sl, p.n_sr, p.n_rl, p.n_rr));
        }
        return 0;
    };
    // a rank that fails (capacity overflow, stale halo, device error) must not leave its peers waiting in an exchange:
    // it aborts its communicator and exits non-zero; the parent then ends the other ranks (main)
    auto failed = [&]() -> int {
        sfx_rccl_comm_abort(comm);
        sfx_destroy(w);
        return 1;
    };
    for (int t = 0; t < 10; ++t) if (tick()) return failed();
    CHECK(w, sfx_sync(w));
    double zero = 0.0, tmax = 0.0;
    CHECK(w, sfx_rccl_allreduce_stats(comm, &zero, 1, &tmax, 1));   // a barrier before the clock starts
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < ticks; ++t) if (tick()) return failed();
    if (sfx_sync(w)) { fprintf(stderr, "rank %d: %s\n", rank, sfx_last_error(w)); return failed(); }
    double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    sfx_frame_stats st;
    CHECK(w, sfx_stats(w, &st));
    double sums[3] = {(double)p.n_owned, (double)st.n_manifolds, (double)st.kinetic_energy}, maxs[2] = {el, (double)st.max_depth};
    CHECK(w, sfx_rccl_allreduce_stats(comm, sums, 3, maxs, 2));
    if (rank == 0)
        printf("ranks %d bodies %.0f (rank 0: %u owned, %u + %u sent, %u + %u ghosts) manifolds %.0f max_depth %.4f\n"
               "ticks %d  %.4f ms/tick (slowest rank)  %.1f M body-substeps/s\n",
               nranks, sums[0], p.n_owned, p.n_sl, p.n_sr, p.n_rl, p.n_rr, sums[1], maxs[1], ticks, 1e3 * maxs[0] / ticks,
               1e-6 * sums[0] * s.substeps * ticks / maxs[0]);
    sfx_rccl_comm_destroy(comm);
    sfx_destroy(w);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s scene.bin [ticks] [nranks | --self]\n", argv[0]); return 2; }
    const int ticks = argc > 2 ? atoi(argv[2]) : 100;
    const bool self = argc > 3 && !strcmp(argv[3], "--self");
    const int nranks = (argc > 3 && !self) ? std::max(1, atoi(argv[3])) : 1;
    Scene s;
    if (!load(argv[1], s)) { fprintf(stderr, "cannot read %s (circles and static rects only)\n", argv[1]); return 2; }
    if (self) return run_self(s, ticks);
    if (sfx_abi_version() != SFX_ABI_VERSION) { fprintf(stderr, "libsfx.so has ABI %u, this program was built for %u: rebuild\n", sfx_abi_version(), SFX_ABI_VERSION); return 2; }
    const std::string idfile = "/tmp/sfx_slab_run." + std::to_string((long)getpid()) + ".id";
    remove(idfile.c_str());   // (a killed run with the same process id may have left one: ranks > 0 would read a stale id)
    remove((idfile + ".tmp").c_str());
    if (nranks == 1) { const int rc = run_rank(s, ticks, 0, 1, idfile); remove(idfile.c_str()); return rc; }
    std::vector<pid_t> kids;
    for (int r = 0; r < nranks; ++r) {   // fork BEFORE anything touches the GPU: every rank initialises HIP itself
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0) _exit(run_rank(s, ticks, r, nranks, idfile));
        kids.push_back(pid);
    }
    // the first rank that ends badly ends the run: the others (by their process ids) are not waited for in order --
    // they may be inside a collective that will never complete
    int bad = 0;
    for (size_t left_kids = kids.size(); left_kids; --left_kids) {
        int st = 0;
        const pid_t k = waitpid(-1, &st, 0);
        if (k < 0) break;
        if (!WIFEXITED(st) || WEXITSTATUS(st)) {
            if (!bad) for (pid_t o : kids) if (o != k) kill(o, SIGKILL);
            bad = 1;
        }
    }
    remove(idfile.c_str());
    return bad;
}
