// Example / link test for starframe_physics.hpp: the sandbox-style box pyramid.
//   g++ -std=c++17 example_pyramid.cpp -L../csrc -lsfx -Wl,-rpath,'$ORIGIN/../csrc' -o example_pyramid
#include <cstdio>

#include "starframe_physics.hpp"

int main() {
    sf::PhysicsConfig cfg;
    cfg.substeps = 10;
    sf::Physics physics(cfg);
    const sf::Collider box = sf::Collider::new_rect(1.0f, 1.0f);
    const int base = 12;
    for (int row = 0; row < base; ++row)
        for (int j = 0; j < base - row; ++j)
            physics.add(sf::Body::new_dynamic(box, 1.0f), box, sf::Pose{{(j - (base - row - 1) / 2.0f) * 1.05f, 0.5f + row}, 0.0f});
    const sf::Collider ground = sf::Collider::new_rect(80.0f, 1.0f);
    physics.add(sf::Body::new_static(), ground, sf::Pose{{0.0f, -0.5f}, 0.0f});
    for (int frame = 0; frame < 240; ++frame) physics.tick(1.0f / 60.0f);
    std::vector<float> pos, rot;
    physics.read_poses(pos, rot);
    float top = 0;
    for (size_t k = 0; k + 1 < physics.body_count(); ++k) top = pos[2 * k + 1] > top ? pos[2 * k + 1] : top;
    const sfx_frame_stats st = physics.stats();
    std::printf("pyramid: %zu bodies, top at %.3f (ideal %.1f), KE %.2e, %u pairs, %u colours\n", physics.body_count(), top,
                base - 0.5f, st.kinetic_energy, st.n_pairs, st.n_colors);
    // pick the apex from above with a ray, and the ground with a point query
    const sfx_ray_hit hit = physics.raycast({0.0f, base + 5.0f}, {0.0f, -1.0f}, 100.0f);
    const sf::BodyKey under = physics.query_point({30.0f, -0.5f});
    std::printf("ray from above hits body %u at t = %.3f (normal %.2f, %.2f); body under (30, -0.5): %u\n", hit.body, hit.t,
                hit.normal[0], hit.normal[1], under);
    const bool ray_ok = hit.body != SFX_NO_BODY && hit.normal[1] > 0.9f && under == (sf::BodyKey)(physics.body_count() - 1);
    // a second world: a rope between two pins and a hinged flap with stops, drawn from the overlapped read-back
    sf::Physics toys(cfg);
    const std::vector<sf::BodyKey> rope = toys.add_rope({0.0f, 3.0f}, {2.0f, 3.0f}, 16, 0.04f, 1.0f, true, 0.25f);
    const sf::Collider flap = sf::Collider::new_rect(1.0f, 0.1f);
    const sf::BodyKey bar = toys.add(sf::Body::new_dynamic(flap, 1.0f), flap, sf::Pose{{4.5f, 3.0f}, 0.0f});
    toys.add_attachment(sf::WORLD, {4.0f, 3.0f}, bar, {-0.5f, 0.0f});
    toys.add_angle_constraint(sf::WORLD, bar, -0.4f, 0.3f, 0.0f, 2.0f);
    const float* poses = nullptr;
    uint32_t n = 0;
    for (int frame = 0; frame < 240; ++frame) poses = toys.tick_and_read(1.0f / 60.0f, &n);
    float low = 3.0f;
    for (sf::BodyKey k : rope) low = poses[4 * k + 1] < low ? poses[4 * k + 1] : low;
    const float flap_angle = std::atan2(poses[4 * bar + 3], poses[4 * bar + 2]);
    std::printf("rope of %zu links sags to %.3f below its pins; flap rests at %.3f rad (stop at -0.4)\n", rope.size(), 3.0f - low, flap_angle);
    const bool toys_ok = n == toys.body_count() && low < 2.95f && low > 1.9f && std::fabs(flap_angle + 0.4f) < 0.02f;
    return top > base - 0.6f && ray_ok && toys_ok ? 0 : 1;
}
