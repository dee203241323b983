// starframe_physics.hpp -- host-side mirror of starframe's physics API over the libsfx C ABI.
//
// The reference is Rust and its toolchain is absent here, so the host side above the C ABI is
// written in C++ (header-only).  It keeps the *shape* of the reference's physics API that
// BASELINE.json:5 names -- a physics config, body / collider / material components, and the
// per-frame tick entry point -- and marshals to the flat SoA calls of include/sfx.h.  Type and
// method names follow upstream as far as they could be recalled (SURVEY.md marks them
// [RECALL], unverified: the source is not in /root/reference); bindings/rust/ holds the Rust
// equivalent a maintainer would compile.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sfx.h"

namespace sf {

struct Vec2 { float x = 0, y = 0; };

struct Pose {
    Vec2 translation;
    float angle = 0;
};

struct PhysicsMaterial {
    float static_friction_coef = 0.5f;
    float dynamic_friction_coef = 0.4f;
    float restitution_coef = 0.1f;
};

struct CollisionLayers {
    uint32_t category = 1, mask = 0xFFFFFFFFu, group = 0;
};

class Collider {
public:
    static Collider new_circle(float r) { Collider c; c.raw_.type = SFX_SHAPE_CIRCLE; c.raw_.p[2] = r; return c; }
    static Collider new_rect(float width, float height) {
        Collider c; c.raw_.type = SFX_SHAPE_RECT; c.raw_.p[0] = 0.5f * width; c.raw_.p[1] = 0.5f * height; return c;
    }
    static Collider new_rounded_rect(float width, float height, float radius) {
        Collider c = new_rect(width - 2 * radius, height - 2 * radius); c.raw_.p[2] = radius; return c;
    }
    static Collider new_capsule(float length, float radius) {
        Collider c; c.raw_.type = SFX_SHAPE_CAPSULE; c.raw_.p[0] = 0.5f * length; c.raw_.p[2] = radius; return c;
    }
    // convex, counter-clockwise, 3..8 vertices around the centre of mass
    static Collider new_polygon(std::vector<Vec2> verts, float rounding = 0.0f) {
        if (verts.size() < 3 || verts.size() > SFX_MAX_POLY_VERTS) throw std::invalid_argument("3..8 vertices");
        Collider c; c.raw_.type = SFX_SHAPE_POLY; c.raw_.p[2] = rounding; c.verts_ = std::move(verts); return c;
    }
    Collider with_material(const PhysicsMaterial& m) { material_ = m; return *this; }
    Collider with_layers(const CollisionLayers& l) { raw_.category = l.category; raw_.mask = l.mask; raw_.group = l.group; return *this; }

    // area and second moment about the centre, for Body::new_dynamic(collider, density)
    float area() const {
        const float a = raw_.p[0], b = raw_.p[1], r = raw_.p[2];
        switch (raw_.type) {
        case SFX_SHAPE_CIRCLE: return 3.14159265358979f * r * r;
        case SFX_SHAPE_RECT: return 4.0f * (a + r) * (b + r);
        case SFX_SHAPE_CAPSULE: return 4.0f * a * r + 3.14159265358979f * r * r;
        case SFX_SHAPE_POLY: { float s = 0; for (size_t k = 0; k < verts_.size(); ++k) { const Vec2 &p = verts_[k], &q = verts_[(k + 1) % verts_.size()]; s += p.x * q.y - q.x * p.y; } return 0.5f * s; }
        default: return 0.0f;
        }
    }
    float unit_inertia() const {  // inertia / mass
        const float a = raw_.p[0], b = raw_.p[1], r = raw_.p[2];
        switch (raw_.type) {
        case SFX_SHAPE_CIRCLE: return 0.5f * r * r;
        case SFX_SHAPE_RECT: { float w = 2 * (a + r), h = 2 * (b + r); return (w * w + h * h) / 12.0f; }
        case SFX_SHAPE_CAPSULE: {
            float mr = 4 * a * r, mc = 3.14159265358979f * r * r;
            return (mr * (4 * a * a + 4 * r * r) / 12.0f + mc * (0.5f * r * r + a * a)) / (mr + mc);
        }
        case SFX_SHAPE_POLY: {
            float num = 0, den = 0;
            for (size_t k = 0; k < verts_.size(); ++k) {
                const Vec2 &p = verts_[k], &q = verts_[(k + 1) % verts_.size()];
                float cr = p.x * q.y - q.x * p.y;
                num += (p.x * p.x + p.x * q.x + q.x * q.x + p.y * p.y + p.y * q.y + q.y * q.y) * cr;
                den += cr;
            }
            return den != 0 ? num / (6.0f * den) : 0.0f;
        }
        default: return 0.0f;
        }
    }

    // the shape's pose in the body frame (the reference's collider "local offset"); the body's inertia then follows the
    // parallel axis theorem (Body::new_dynamic)
    Collider with_offset(Vec2 offset, float angle = 0.0f) const {
        Collider c = *this; c.raw_.local_pos[0] = offset.x; c.raw_.local_pos[1] = offset.y; c.raw_.local_angle = angle; return c;
    }
    float offset2() const { return raw_.local_pos[0] * raw_.local_pos[0] + raw_.local_pos[1] * raw_.local_pos[1]; }

    sfx_collider raw_{};
    std::vector<Vec2> verts_;
    PhysicsMaterial material_;
    Collider() { raw_.category = 1; raw_.mask = 0xFFFFFFFFu; }
};

struct Velocity { Vec2 linear; float angular = 0; };

class Body {
public:
    static Body new_dynamic(const Collider& c, float density) {
        Body b; float m = density * c.area(); b.inv_mass = m > 0 ? 1 / m : 0;
        float i = m * (c.unit_inertia() + c.offset2()); b.inv_inertia = i > 0 ? 1 / i : 0; return b;
    }
    static Body new_dynamic_const_mass(const Collider& c, float mass) {
        Body b; b.inv_mass = mass > 0 ? 1 / mass : 0; float i = mass * (c.unit_inertia() + c.offset2()); b.inv_inertia = i > 0 ? 1 / i : 0; return b;
    }
    static Body new_particle(float mass) { Body b; b.inv_mass = mass > 0 ? 1 / mass : 0; return b; }
    static Body new_kinematic() { return Body(); }
    static Body new_static() { return Body(); }
    float inv_mass = 0, inv_inertia = 0;
};

struct PhysicsConfig {  // "physics config" (field names of the reference's tuning struct are unknown)
    uint32_t substeps = 4;
    Vec2 gravity{0.0f, -9.81f};
    float aabb_margin = 0.01f;
    float max_depenetration_velocity = 0.0f;  // 0 = unlimited
    int device = 0;
};

using BodyKey = uint32_t;
constexpr BodyKey WORLD = SFX_WORLD_BODY;

// The physics world: collects components on the host, uploads once, then tick() per frame.
class Physics {
public:
    explicit Physics(const PhysicsConfig& cfg = PhysicsConfig()) : cfg_(cfg) {}
    ~Physics() { sfx_destroy(world_); }
    Physics(const Physics&) = delete;
    Physics& operator=(const Physics&) = delete;

    BodyKey add(const Body& body, const Collider& collider, const Pose& pose, const Velocity& vel = Velocity()) {
        const BodyKey k = (BodyKey)inv_mass_.size();
        pos_.push_back(pose.translation.x); pos_.push_back(pose.translation.y);
        rot_.push_back(std::cos(pose.angle)); rot_.push_back(std::sin(pose.angle));
        vel_.push_back(vel.linear.x); vel_.push_back(vel.linear.y); angvel_.push_back(vel.angular);
        inv_mass_.push_back(body.inv_mass); inv_inertia_.push_back(body.inv_inertia);
        sfx_collider c = collider.raw_;
        c.material = material_index(collider.material_);
        if (c.type == SFX_SHAPE_POLY) {
            c.poly_offset = (uint32_t)(poly_.size() / 2); c.poly_count = (uint32_t)collider.verts_.size();
            for (const Vec2& v : collider.verts_) { poly_.push_back(v.x); poly_.push_back(v.y); }
        }
        colliders_.push_back(c);
        dirty_ = true;
        return k;
    }
    // A further collider on a body that has one (a compound body; include/sfx.h sfx_collider.body).  density > 0
    // (dynamic bodies): the shape's mass and inertia -- parallel axis about the body origin, which is taken to stay
    // the centre of mass -- are added to the body's.
    void add_collider(BodyKey body, const Collider& collider, float density = 0.0f) {
        sfx_collider c = collider.raw_;
        c.material = material_index(collider.material_);
        c.body = body;
        if (c.type == SFX_SHAPE_POLY) {
            c.poly_offset = (uint32_t)(poly_.size() / 2); c.poly_count = (uint32_t)collider.verts_.size();
            for (const Vec2& v : collider.verts_) { poly_.push_back(v.x); poly_.push_back(v.y); }
        }
        further_.push_back(c);
        if (density > 0.0f && inv_mass_[body] > 0.0f) {
            const float m = density * collider.area(), i = m * (collider.unit_inertia() + collider.offset2());
            inv_mass_[body] = 1.0f / (1.0f / inv_mass_[body] + m);
            if (inv_inertia_[body] > 0.0f) inv_inertia_[body] = 1.0f / (1.0f / inv_inertia_[body] + i);
        }
        dirty_ = true;
    }
    BodyKey add_particle(const Body& body, const Pose& pose) { Collider none; none.raw_.type = SFX_SHAPE_NONE; return add(body, none, pose); }

    // distance constraint between body-local anchors (world position for WORLD)
    // (damping, 1/s: two-sided constraints only)
    void add_distance_constraint(BodyKey a, Vec2 anchor_a, BodyKey b, Vec2 anchor_b, float rest_length, float compliance = 0,
                                 uint32_t limit = SFX_LIMIT_NONE, float damping = 0) {
        sfx_distance_constraint c{a, b, {anchor_a.x, anchor_a.y}, {anchor_b.x, anchor_b.y}, rest_length, compliance, limit, damping};
        joints_.push_back(c);
        dirty_ = true;
    }
    // pin joint: the two anchors are held together
    void add_attachment(BodyKey a, Vec2 anchor_a, BodyKey b, Vec2 anchor_b, float compliance = 0, float damping = 0) {
        add_distance_constraint(a, anchor_a, b, anchor_b, 0.0f, compliance, SFX_LIMIT_NONE, damping);
    }
    // keeps angle(b) - angle(a) inside [min_angle, max_angle] (WORLD: angle 0)
    void add_angle_constraint(BodyKey a, BodyKey b, float min_angle, float max_angle, float compliance = 0, float damping = 0) {
        angles_.push_back(sfx_angle_constraint{a, b, min_angle, max_angle, compliance, damping});
        dirty_ = true;
    }
    // rope helper: `links` circle bodies from start to end tied by one-sided links, first one tied to the world at start;
    // slack makes the rope that fraction longer than the distance it spans
    std::vector<BodyKey> add_rope(Vec2 start, Vec2 end, uint32_t links, float radius = 0.04f, float density = 1.0f, bool pin_end = false,
                                  float slack = 0.0f) {
        std::vector<BodyKey> ids;
        const float dx = (end.x - start.x) / (float)links, dy = (end.y - start.y) / (float)links;
        const float seg = std::sqrt(dx * dx + dy * dy) * (1.0f + slack);
        const uint32_t group = 0x52000000u + (uint32_t)inv_mass_.size();
        for (uint32_t k = 0; k < links; ++k) {
            Collider c = Collider::new_circle(radius);
            c.raw_.group = group;
            ids.push_back(add(Body::new_dynamic(c, density), c, Pose{{start.x + (k + 0.5f) * dx, start.y + (k + 0.5f) * dy}, 0.0f}));
            if (k) add_distance_constraint(ids[k - 1], {0, 0}, ids[k], {0, 0}, seg, 0, SFX_LIMIT_ROPE);
        }
        if (links) add_distance_constraint(WORLD, start, ids[0], {0, 0}, 0.5f * seg, 0, SFX_LIMIT_ROPE);
        if (links && pin_end) add_distance_constraint(ids[links - 1], {0, 0}, WORLD, end, 0.5f * seg, 0, SFX_LIMIT_ROPE);
        return ids;
    }

    // the per-frame tick entry point
    void tick(float frame_dt) {
        if (dirty_) upload();
        check(sfx_step(world_, frame_dt), "sfx_step");
    }

    // blocks until the tick has finished; poses as (x, y) and (cos, sin)
    void read_poses(std::vector<float>& pos, std::vector<float>& rot) {
        pos.resize(pos_.size()); rot.resize(rot_.size());
        check(sfx_download_poses(world_, pos.data(), rot.data()), "sfx_download_poses");
    }
    // render loop: start the read-back of the current poses, run the next tick while they travel, return them
    // ([n][4] = x, y, cos, sin in library-owned pinned memory, valid until the call after next)
    const float* tick_and_read(float frame_dt, uint32_t* n = nullptr) {
        if (dirty_) upload();
        check(sfx_readback_begin(world_), "sfx_readback_begin");
        check(sfx_step(world_, frame_dt), "sfx_step");
        const float* poses = nullptr;
        check(sfx_readback_wait(world_, &poses, n), "sfx_readback_wait");
        return poses;
    }
    sfx_frame_stats stats() { sfx_frame_stats s; check(sfx_stats(world_, &s), "sfx_stats"); return s; }

    // ---- spatial queries at the current poses (the reference's mouse-pick / ray-cast API; include/sfx.h)
    // body under a point (lowest key if several; WORLD = none)
    BodyKey query_point(Vec2 p, uint32_t layer_mask = 0xFFFFFFFFu) {
        if (dirty_) upload();
        uint32_t body = SFX_NO_BODY;
        const float xy[2] = {p.x, p.y};
        check(sfx_query_points(world_, 1, xy, layer_mask, &body, nullptr), "sfx_query_points");
        return body;
    }
    // bodies whose collider's bounding box overlaps [lo, hi]
    std::vector<BodyKey> query_aabb(Vec2 lo, Vec2 hi, uint32_t layer_mask = 0xFFFFFFFFu) {
        if (dirty_) upload();
        const float box[4] = {lo.x, lo.y, hi.x, hi.y};
        std::vector<uint64_t> hits(64);
        size_t n = 0;
        int rc = sfx_query_aabbs(world_, 1, box, layer_mask, hits.data(), hits.size(), &n);
        if (rc == SFX_ERR_CAPACITY) { hits.resize(n); rc = sfx_query_aabbs(world_, 1, box, layer_mask, hits.data(), hits.size(), &n); }
        check(rc, "sfx_query_aabbs");
        std::vector<BodyKey> out(n);
        for (size_t k = 0; k < n; ++k) out[k] = (BodyKey)(hits[k] & 0xFFFFFFFFu);
        return out;
    }
    // closest hit of origin + t * dir, 0 <= t <= max_t; hit.body == SFX_NO_BODY when nothing is hit
    sfx_ray_hit raycast(Vec2 origin, Vec2 dir, float max_t, uint32_t layer_mask = 0xFFFFFFFFu) {
        if (dirty_) upload();
        const float ray[4] = {origin.x, origin.y, dir.x, dir.y};
        sfx_ray_hit hit;
        check(sfx_query_rays(world_, 1, ray, &max_t, layer_mask, &hit), "sfx_query_rays");
        return hit;
    }
    size_t body_count() const { return inv_mass_.size(); }

private:
    uint32_t material_index(const PhysicsMaterial& m) {
        for (size_t k = 0; k < mats_.size(); ++k)
            if (mats_[k].restitution == m.restitution_coef && mats_[k].static_friction == m.static_friction_coef &&
                mats_[k].dynamic_friction == m.dynamic_friction_coef) return (uint32_t)k;
        mats_.push_back({m.restitution_coef, m.static_friction_coef, m.dynamic_friction_coef});
        return (uint32_t)mats_.size() - 1;
    }
    void check(int rc, const char* what) {
        if (rc != SFX_OK) throw std::runtime_error(std::string(what) + ": " + sfx_last_error(world_));
    }
    void upload() {
        if (!world_) {
            sfx_config c; sfx_default_config(&c);
            c.substeps = cfg_.substeps; c.gravity[0] = cfg_.gravity.x; c.gravity[1] = cfg_.gravity.y;
            c.aabb_margin = cfg_.aabb_margin; c.max_depenetration_velocity = cfg_.max_depenetration_velocity;
            c.device = cfg_.device;
            check(sfx_create(&c, &world_), "sfx_create");
        } else if (ticked_) {
            // keep the simulated state of bodies that already exist
            std::vector<float> p(pos_.size()), r(rot_.size()), v(vel_.size()), a(angvel_.size());
            const size_t old = old_count_;
            check(sfx_download_poses(world_, p.data(), r.data()), "sfx_download_poses");
            check(sfx_download_velocities(world_, v.data(), a.data()), "sfx_download_velocities");
            for (size_t k = 0; k < old; ++k) {
                pos_[2 * k] = p[2 * k]; pos_[2 * k + 1] = p[2 * k + 1]; rot_[2 * k] = r[2 * k]; rot_[2 * k + 1] = r[2 * k + 1];
                vel_[2 * k] = v[2 * k]; vel_[2 * k + 1] = v[2 * k + 1]; angvel_[k] = a[k];
            }
        }
        const uint32_t n = (uint32_t)inv_mass_.size();
        if (mats_.empty()) material_index(PhysicsMaterial());
        check(sfx_upload_bodies(world_, n, pos_.data(), rot_.data(), vel_.data(), angvel_.data(), inv_mass_.data(), inv_inertia_.data()), "sfx_upload_bodies");
        check(sfx_upload_materials(world_, (uint32_t)mats_.size(), mats_.data()), "sfx_upload_materials");
        std::vector<sfx_collider> all(colliders_);   // the bodies' own colliders, then the further ones
        all.insert(all.end(), further_.begin(), further_.end());
        check(sfx_upload_colliders(world_, (uint32_t)all.size(), all.data(), poly_.data(), (uint32_t)(poly_.size() / 2)), "sfx_upload_colliders");
        check(sfx_upload_constraints(world_, (uint32_t)joints_.size(), joints_.data()), "sfx_upload_constraints");
        check(sfx_upload_angle_constraints(world_, (uint32_t)angles_.size(), angles_.data()), "sfx_upload_angle_constraints");
        old_count_ = n; dirty_ = false; ticked_ = true;
    }

    PhysicsConfig cfg_;
    sfx_world* world_ = nullptr;
    bool dirty_ = true, ticked_ = false;
    size_t old_count_ = 0;
    std::vector<float> pos_, rot_, vel_, angvel_, inv_mass_, inv_inertia_, poly_;
    std::vector<sfx_collider> colliders_, further_;
    std::vector<sfx_material> mats_;
    std::vector<sfx_distance_constraint> joints_;
    std::vector<sfx_angle_constraint> angles_;
};

}  // namespace sf
