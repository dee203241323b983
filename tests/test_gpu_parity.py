"""GPU parity tests: libsfx (HIP, through the C ABI) against the CPU oracle on identical inputs.

Oracle = this repository's CPU restatement of the published XPBD algorithm (PARITY UNPINNED
against starframe itself: its source was unavailable).  Bars (BASELINE.json:5):
candidate pair sets bit-exact; contact normals/depths and single-tick post-solve positions
within 1e-5 relative.
"""
import numpy as np
import pytest

from starframe_amd import scenes
from tests.helpers import coloring_valid, colors_by_key, rel_err, tile_phases_disjoint, ulp_diff

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0
TOL = 1e-5


def make_pair(scene, **kw):
    from oracle.oracle import OracleWorld
    from starframe_amd.world import World

    return World.from_scene(scene, **kw), OracleWorld.from_scene(scene, **kw)


SOUPS = [(300, 1, 4.0), (1000, 2, 6.0), (3000, 3, 14.0)]


@pytest.mark.parametrize("n,seed,extent", SOUPS)
def test_aabbs_and_pairs_bit_exact(n, seed, extent):
    s = scenes.random_soup(n, seed=seed, extent=extent)
    g, o = make_pair(s)
    g.broadphase(DT)
    o.broadphase(DT)
    assert np.array_equal(g.aabbs().view(np.uint32), o.aabbs().view(np.uint32)), "AABBs differ bitwise"
    gp, op = g.pairs(), o.pairs()
    assert len(gp) > n // 2
    assert np.array_equal(gp, op), "candidate pair set differs from the oracle grid"
    assert np.array_equal(gp, o.pairs_sweep()), "candidate pair set differs from sort-and-sweep"
    if n <= 1000:
        assert np.array_equal(gp, o.pairs_brute())
    gs, os_ = g.stats(), o.stats()
    assert (gs.n_small, gs.n_large, gs.n_pairs) == (os_.n_small, os_.n_large, os_.n_pairs)


@pytest.mark.parametrize("n,seed,extent", SOUPS)
def test_colouring_matches_oracle(n, seed, extent):
    s = scenes.random_soup(n, seed=seed, extent=extent)
    g, o = make_pair(s)
    g.broadphase(DT)
    o.broadphase(DT)
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    rc = o.resolved_config()
    assert coloring_valid(gk, gc, dyn, gt)
    assert tile_phases_disjoint(gk, gt, dyn, tuple(rc.grid_log2), rc.tile_cells)
    assert np.array_equal(gk, ok)
    assert np.array_equal(gt, ot), "owner tiles differ from the restated tiling"
    assert np.array_equal(gc, oc), "pair colours differ from the restated colouring"
    gs, os_ = g.stats(), o.stats()
    assert gs.n_colors == os_.n_colors and gs.n_leftover == os_.n_leftover
    assert list(gs.color_count) == list(os_.color_count)
    assert (gs.n_tiles, gs.n_fallback, gs.max_tile_pairs, gs.max_tile_bodies) == \
           (os_.n_tiles, os_.n_fallback, os_.max_tile_pairs, os_.max_tile_bodies)


@pytest.mark.parametrize("n,seed,extent", SOUPS)
def test_contacts_match_oracle(n, seed, extent):
    s = scenes.random_soup(n, seed=seed, extent=extent)
    g, o = make_pair(s)
    g.broadphase(DT)
    o.broadphase(DT)
    gc, oc = g.contacts(), o.contacts()
    assert len(gc) == len(oc)
    assert np.array_equal(gc["body_a"], oc["body_a"]) and np.array_equal(gc["body_b"], oc["body_b"])
    assert np.array_equal(gc["count"], oc["count"]), "manifold point counts differ"
    assert (gc["count"] > 0).sum() > 20
    assert rel_err(gc["normal"], oc["normal"], floor=1.0) <= TOL
    assert rel_err(gc["depth"], oc["depth"], floor=1e-2) <= TOL
    assert rel_err(gc["point_a"], oc["point_a"], floor=1.0) <= TOL
    assert rel_err(gc["point_b"], oc["point_b"], floor=1.0) <= TOL


@pytest.mark.parametrize("n,seed,extent", SOUPS)
def test_single_tick_positions(n, seed, extent):
    """One tick, colour order on both sides: post-solve poses and velocities within 1e-5."""
    from oracle.oracle import ORDER_COLORED

    s = scenes.random_soup(n, seed=seed, extent=extent)
    g, o = make_pair(s)
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    gp, gr = g.poses()
    op, orr = o.poses()
    gv, gw = g.velocities()
    ov, ow = o.velocities()
    assert rel_err(gp, op, floor=1.0) <= TOL
    assert rel_err(gr, orr, floor=1.0) <= TOL
    assert rel_err(gv, ov, floor=1.0) <= 1e-4
    assert rel_err(gw, ow, floor=10.0) <= 1e-4
    print("ulp pos", ulp_diff(gp, op), "rot", ulp_diff(gr, orr))


@pytest.mark.parametrize("kw", [dict(tile_cells=1, tile_cells_y=1), dict(tile_cells=2, tile_cells_y=2), dict(tile_cells=4, tile_cells_y=4),
                                dict(tile_cells=8, tile_cells_y=8), dict(cell_size=0.2, tile_cells_y=1), dict(color_rounds=1),
                                dict(tile_cells_y=4096), dict(tile_cells=2, tile_cells_y=4)],
                         ids=["tile1", "tile2", "tile4", "tile8", "large_dynamic_fallback", "serial_colour_finish", "strips",
                              "tile2x4"])
def test_tiling_variants_match_oracle(kw):
    """Other tile sizes, a cell size that pushes big dynamic bodies onto the fallback list, and a
    colouring that leans on the serial finish: tiles, colours and one tick still match."""
    from oracle.oracle import ORDER_COLORED

    s = scenes.random_soup(1500, seed=9, extent=7.0)
    g, o = make_pair(s, **kw)
    g.broadphase(DT)
    o.broadphase(DT)
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot) and np.array_equal(gc, oc)
    if "cell_size" in kw:
        assert g.stats().n_fallback > 0
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL
    assert rel_err(g.poses()[1], o.poses()[1], floor=1.0) <= TOL
    assert rel_err(g.velocities()[0], o.velocities()[0], floor=1.0) <= 1e-4


def test_overfull_tiles_fall_back_and_match():
    from oracle.oracle import ORDER_COLORED
    from starframe_amd.scenes import SceneBuilder

    b = SceneBuilder("dense")
    b.static_rect((0.0, -0.5), 10.0, 0.5)
    b.circle((6.0, 1.0), 1.0)
    for i in range(40):
        for j in range(40):
            b.circle((-2.0 + 0.05 * i, 0.03 + 0.05 * j), 0.026)
    s = b.build()
    g, o = make_pair(s)
    g.broadphase(DT)
    o.broadphase(DT)
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    assert g.stats().n_fallback > 1000 and g.stats().n_fallback == o.stats().n_fallback
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot) and np.array_equal(gc, oc)
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL


def test_pyramid_single_tick_and_long_run():
    """Config 1 (BASELINE.json:7): single tick within tolerance, then 300 frames compared
    statistically (stack height, kinetic energy, penetration)."""
    from oracle.oracle import ORDER_COLORED

    s = scenes.pyramid()
    g, o = make_pair(s)
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL
    for _ in range(299):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    gs, os_ = g.stats(), o.stats()
    gp, op = g.poses()[0], o.poses()[0]
    assert abs(gp[:-1, 1].max() - op[:-1, 1].max()) < 0.05, "stack height"
    assert gp[:-1, 1].max() > 30.0
    assert gs.max_depth < 0.01 and os_.max_depth < 0.01
    assert gs.kinetic_energy < 1.0 and os_.kinetic_energy < 1.0
    assert np.isfinite(gp).all()


def test_chains_joint_colouring_and_tick():
    from oracle.oracle import ORDER_COLORED

    s = scenes.chains(n_chains=20, links=30)
    g, o = make_pair(s)
    gperm, gcol = g.joint_order()
    operm, ocol = o.joint_order()
    assert np.array_equal(np.sort(gperm), np.arange(len(s.constraints)))
    cg = np.empty(len(gperm), np.uint8); cg[gperm] = gcol
    co = np.empty(len(operm), np.uint8); co[operm] = ocol
    assert np.array_equal(cg, co), "joint colours differ"
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL
    for _ in range(120):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    gp, op = g.poses()[0], o.poses()[0]
    assert np.isfinite(gp).all()
    # statistical: mean sag of the chain tips
    tips = np.arange(20) * 30 + 29
    assert abs(gp[tips, 1].mean() - op[tips, 1].mean()) < 0.1


def test_rope_links_match_oracle():
    """One-sided links (SFX_LIMIT_ROPE / _STRUT): chains of rope links that start slack, one tick within
    1e-5 and the tips' sag after 120 ticks."""
    from oracle.oracle import ORDER_COLORED

    s = scenes.chains(n_chains=12, links=25)
    s.constraints["rest_length"] = 0.14                    # bodies 0.12 apart: every rope link starts slack
    s.constraints["limit"] = 1
    s.constraints["limit"][::7] = 2                        # a few struts among them
    s.constraints["limit"][::11] = 0
    g, o = make_pair(s)
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL
    assert rel_err(g.velocities()[0], o.velocities()[0], floor=1.0) <= 1e-4
    for _ in range(120):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    gp, op = g.poses()[0], o.poses()[0]
    assert np.isfinite(gp).all()
    tips = np.arange(12) * 25 + 24
    assert abs(gp[tips, 1].mean() - op[tips, 1].mean()) < 0.1


def hinged_scene(n_chains=10, links=12):
    """Chains of bars hanging from the world: pins with offset anchors, angle stops between neighbours, hinge
    dampers, a damped soft link across each chain, falling onto a floor."""
    from starframe_amd._abi import WORLD_BODY
    from starframe_amd.scenes import SceneBuilder

    b = SceneBuilder("hinged", substeps=4)
    b.static_rect((0.0, -4.0), 40.0, 0.5)
    for c in range(n_chains):
        x0, y0 = 1.3 * c - 6.0, 0.5 + 0.07 * c
        ids = [b.rect((x0 + 0.3 * k + 0.15, y0), 0.13, 0.03, group=100 + c, angle=0.05 * ((k + c) % 3 - 1)) for k in range(links)]
        b.attach(WORLD_BODY, ids[0], anchor_a=(x0, y0), anchor_b=(-0.15, 0.0), damping=2.0 if c % 2 else 0.0)
        for k in range(links - 1):
            b.attach(ids[k], ids[k + 1], anchor_a=(0.15, 0.0), anchor_b=(-0.15, 0.0), compliance=1e-6 if k % 4 == 3 else 0.0)
            if (k + c) % 3 != 2:
                b.angle(ids[k], ids[k + 1], -0.5, 0.7 if k % 2 else 0.5, compliance=1e-3 if k % 5 == 0 else 0.0,
                        damping=3.0 if k % 2 else 0.0)
        b.distance(ids[0], ids[-1], rest=0.3 * (links - 1), compliance=5e-3, damping=4.0)
        b.angle(WORLD_BODY, ids[0], -0.3, 0.3)
    return b.build()


def test_hinged_links_match_oracle():
    """Attachments with offset anchors, angle limits / torsion springs and joint damping (include/sfx.h
    sfx_angle_constraint, sfx_distance_constraint.damping): joint order and colours equal, 1 and 5 ticks within
    1e-5 (the deterministic atan2 is shared op for op), 150 ticks statistically."""
    from oracle.oracle import ORDER_COLORED

    s = hinged_scene()
    g, o = make_pair(s)
    gp, gc = g.joint_order()
    op, oc = o.joint_order()
    assert len(gp) == len(s.constraints) + len(s.angle_constraints)
    assert np.array_equal(gp, op) and np.array_equal(gc, oc)
    for ticks in (1, 4):
        for _ in range(ticks):
            g.step(DT)
            o.step(DT, ORDER_COLORED)
        assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL
        assert rel_err(g.poses()[1], o.poses()[1], floor=1.0) <= TOL
        assert rel_err(g.velocities()[0], o.velocities()[0], floor=1.0) <= 1e-4
        assert rel_err(g.velocities()[1], o.velocities()[1], floor=1.0) <= 1e-3
    for _ in range(150):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    (gpos, grot), (opos, orot) = g.poses(), o.poses()
    assert np.isfinite(gpos).all()
    assert abs(gpos[1:, 1].mean() - opos[1:, 1].mean()) < 0.05
    # every angle stop holds on both sides
    for pos, rot in ((gpos, grot), (opos, orot)):
        th = np.arctan2(rot[:, 1], rot[:, 0])
        a = s.angle_constraints
        rigid = (a["compliance"] == 0) & (a["body_a"] != 0xFFFFFFFF)
        rel = th[a["body_b"][rigid]] - th[a["body_a"][rigid]]
        rel = (rel + np.pi) % (2 * np.pi) - np.pi
        assert (rel > a["min_angle"][rigid] - 0.05).all() and (rel < a["max_angle"][rigid] + 0.05).all()


def test_constraint_validation_through_the_abi():
    from starframe_amd._abi import WORLD_BODY
    from starframe_amd.scenes import SceneBuilder
    from starframe_amd.world import World

    for bad in (dict(limit=1, damping=1.0), dict(damping=-1.0), dict(limit=3)):
        b = SceneBuilder("bad")
        b.circle((0.0, 0.0), 0.1)
        b.circle((1.0, 0.0), 0.1)
        b.distance(0, 1, **bad)
        with pytest.raises(Exception):
            World.from_scene(b.build())
    for lo, hi in ((0.5, 0.2), (-4.0, 0.0), (0.0, 3.2)):
        b = SceneBuilder("bad")
        b.circle((0.0, 0.0), 0.1)
        b.angle(WORLD_BODY, 0, lo, hi)
        with pytest.raises(Exception):
            World.from_scene(b.build())
    # the two arrays are independent: uploading one keeps the other
    s = hinged_scene(2, 4)
    w = World.from_scene(s)
    n_all = len(w.joint_order()[0])
    assert n_all == len(s.constraints) + len(s.angle_constraints)
    w._check(w._f("upload_constraints")(w.handle, 0, None))
    w.m = n_all
    assert len(w.joint_order()[0]) == len(s.angle_constraints)
    w.step(DT)
    assert np.isfinite(w.poses()[0]).all()


def test_empty_and_tiny_worlds():
    from starframe_amd.scenes import SceneBuilder
    from starframe_amd.world import World

    b = SceneBuilder("one")
    b.circle((0, 0), 0.1)
    w = World.from_scene(b.build())
    for _ in range(3):
        w.step(DT)
    p, _ = w.poses()
    assert p[0, 1] < 0
    assert w.stats().n_pairs == 0
    b = SceneBuilder("none")
    b.particle((0, 0))
    b.particle((0, 0))
    w = World.from_scene(b.build())
    w.step(DT)
    assert w.stats().n_pairs == 0


def test_pair_capacity_overflow_reported():
    """Too few pair slots: the tick drops pairs, says so in the stats, and the next synchronising call -- or the next
    tick, once the host's counter read-back has shown it the bit -- returns SFX_ERR_CAPACITY."""
    from starframe_amd.world import SfxError, World

    s = scenes.random_soup(2000, seed=5, extent=8.0)
    w = World.from_scene(s, max_pairs=1024)
    w.step(DT)
    assert w.stats().overflow & 1
    with pytest.raises(SfxError) as e:
        w.step(DT)
        w.sync()
    assert e.value.code == -4
    with pytest.raises(SfxError):        # the condition persists: every tick overflows again
        w.sync()
        w.step(DT)
        w.sync()


@pytest.mark.parametrize("grid_log2", [(12, 12), (9, 9), (10, 10), (6, 6)], ids=["4096x4096", "512x512", "1024x1024", "64x64"])
def test_largest_grid_scans(grid_log2):
    """4096 x 4096 buckets (what a 4 M-body scene resolves to): the cell scan takes 2^24 + 1 items, 4097 scan tiles
    whose totals k_scan_sums scans.  512 x 512 (the 100 k pile's grid): 65 scan tiles, k_scan_add sums the totals
    before its tile itself; 1024 x 1024: 257 tiles, the first count over that path's limit; 64 x 64: two tiles.
    The tile lists of the small grids come from the one-workgroup k_tile_list_scan, the large ones from
    k_tile_list + the scan launches."""
    s = scenes.random_soup(400, seed=9, extent=5.0)
    g, o = make_pair(s, grid_log2=grid_log2)
    g.broadphase(DT)
    o.broadphase(DT)
    assert np.array_equal(g.pairs(), o.pairs())
    g.step(DT)
    assert g.stats().overflow == 0


def test_batches_by_ticket_equal_batches_by_workgroup_index():
    """A world alone on its device starts workgroup b of a fused tile-solve launch with batch b; with a second
    world alive on the device every batch is handed out by ticket (two partly resident launches must not wait
    for each other).  Which workgroup solves a batch is scheduling only: the states are bit-identical, also
    when the two worlds' launches overlap on the device."""
    from starframe_amd.world import World

    s = scenes.mixed_pile(n=20000, width=400)
    alone = World.from_scene(s)
    for _ in range(30):
        alone.step(DT)
    ref_pos, ref_rot = alone.poses()
    ref_vel, _ = alone.velocities()
    assert alone.stats().n_manifolds > 1000
    alone.close()
    a, b = World.from_scene(s), World.from_scene(s)      # own streams: their launches run side by side
    for _ in range(30):
        a.step(DT)
        b.step(DT)
    for w in (a, b):
        pos, rot = w.poses()
        assert w.stats().overflow == 0
        assert np.array_equal(pos.view(np.uint32), ref_pos.view(np.uint32))
        assert np.array_equal(rot.view(np.uint32), ref_rot.view(np.uint32))
        assert np.array_equal(w.velocities()[0].view(np.uint32), ref_vel.view(np.uint32))
        w.close()


def test_every_tick_starts_from_cleared_tables():
    """The tick's zero-initialised block (counters, cell counts, tile tables) has two halves; a tick works in the half
    that the previous tick's k_aabb_count cleared.  After each of several ticks the oracle takes the GPU's state over
    and both run the broadphase: pair sets, tiles and colours are equal every time (a count or a flag left over from
    two ticks earlier would show up as a missing or an extra pair, or as a tile on the wrong list)."""
    s = scenes.random_soup(1500, seed=5, extent=7.0)
    g, o = make_pair(s)
    for tick in range(5):
        g.step(DT)
        pos, rot = g.poses()
        vel, angvel = g.velocities()
        o.set_state(pos, rot, vel, angvel, s.inv_mass, s.inv_inertia)
        g.broadphase(DT)   # one more flip of the halves between two ticks
        o.broadphase(DT)
        assert np.array_equal(g.pairs(), o.pairs()), tick
        gk, gc, gt = colors_by_key(*g.pair_colors())
        ok, oc, ot = colors_by_key(*o.pair_colors())
        assert np.array_equal(gk, ok) and np.array_equal(gt, ot) and np.array_equal(gc, oc), tick
        gs, os_ = g.stats(), o.stats()
        assert gs.overflow == 0 and gs.n_pairs == os_.n_pairs and gs.n_fallback == os_.n_fallback, tick


def test_overflow_of_one_tick_among_many_is_reported_by_the_next_sync():
    """the overflow bits live in the per-tick counter block, which is cleared two ticks later, and the host
    samples it only every 8th tick -- a capacity overflow in tick 3 of 10 used to be lost.  The bits are folded into a
    sticky device word when the block is cleared: the sync after the ten ticks reports it, once."""
    from starframe_amd.world import SfxError, World

    s = scenes.random_soup(600, seed=5, extent=12.0)          # sparse: few hundred pairs
    w = World.from_scene(s, max_pairs=1024)
    crowd = s.pos * 0.12                                     # the same bodies crammed together: thousands of pairs
    reported = 0
    for tick in range(10):
        if tick == 3:
            w.set_state(crowd, s.rot, s.vel, s.angvel, s.inv_mass, s.inv_inertia)
        if tick == 4:
            w.set_state(s.pos, s.rot, s.vel, s.angvel, s.inv_mass, s.inv_inertia)
        try:
            w.step(DT)
        except SfxError as e:                                # the host may have seen the bit already (counter read-back)
            assert e.code == -4
            reported += 1
            break
        if tick == 2:
            w.sync()                                         # nothing has overflowed yet
    try:
        w.sync()
    except SfxError as e:
        assert e.code == -4 and "capacity" in str(e)
        reported += 1
    assert reported >= 1, "the overflow of tick 3 was lost"
    w.sync()                                                 # reported once: the condition is cleared
    w.step(DT)
    w.sync()
    assert w.stats().overflow == 0


def test_tiny_grid_and_ticket_path_equal_the_default(monkeypatch):
    """A fused tile-solve launch hands its batches out by ticket; the default grid holds every workgroup the device
    can run.  With a deliberately tiny grid (4 workgroups for hundreds of batches: every workgroup takes dozens of
    tickets, later phases wait on flags set by the same few workgroups) and with every batch handed out by ticket
    (SFX_SHARED_DEVICE semantics: a second world alive), the states are bit-identical to the default path."""
    from starframe_amd.world import World

    s = scenes.mixed_pile(n=20000, width=400)

    def run(extra_world=False):
        w = World.from_scene(s)
        other = World.from_scene(scenes.random_soup(100, seed=1, extent=3.0)) if extra_world else None
        for _ in range(25):
            w.step(DT)
        w.sync()
        assert w.stats().overflow == 0 and w.stats().n_manifolds > 1000
        out = (w.poses(), w.velocities())
        w.close()
        if other:
            other.close()
        return out

    ref = run()
    monkeypatch.setenv("SFX_TILE_GRID_MAX", "4")
    for extra in (False, True):
        (pos, rot), (vel, ang) = run(extra)
        assert np.array_equal(pos.view(np.uint32), ref[0][0].view(np.uint32))
        assert np.array_equal(rot.view(np.uint32), ref[0][1].view(np.uint32))
        assert np.array_equal(vel.view(np.uint32), ref[1][0].view(np.uint32))
        assert np.array_equal(ang.view(np.uint32), ref[1][1].view(np.uint32))


def test_colliders_with_a_local_pose_match_oracle():
    """sfx_collider.local_pos / local_angle (SURVEY.md section 8a row a4 "local offset"): half the shapes of a soup sit off
    their body's origin, turned in the body frame.  AABBs (with the swing pad), pair set, tiles, colours bit-exact;
    contacts and five ticks of poses within tolerance; spatial queries see the shapes where they are."""
    from oracle.oracle import ORDER_COLORED
    from tests.test_oracle_queries import query_inputs

    s = scenes.random_soup(2000, seed=31, extent=9.0, offsets=0.5)
    assert (np.abs(s.colliders["local_pos"]).sum(1) > 0).sum() > 800
    g, o = make_pair(s)
    g.broadphase(DT)
    o.broadphase(DT)
    assert np.array_equal(g.aabbs().view(np.uint32), o.aabbs().view(np.uint32)), "AABBs differ bitwise"
    assert np.array_equal(g.pairs(), o.pairs())
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot) and np.array_equal(gc, oc)
    gcon, ocon = g.contacts(), o.contacts()
    assert np.array_equal(gcon["count"], ocon["count"]) and (gcon["count"] > 0).sum() > 300
    assert rel_err(gcon["normal"], ocon["normal"], floor=1.0) <= TOL
    assert rel_err(gcon["depth"], ocon["depth"], floor=1e-2) <= TOL
    assert rel_err(gcon["point_a"], ocon["point_a"], floor=1.0) <= TOL
    pts, boxes, rays, max_t = query_inputs(31, 9.0, n=500, near=s.pos)
    gb, gn = g.query_points(pts)
    ob, on = o.query_points(pts)
    assert np.array_equal(gb, ob) and np.array_equal(gn, on) and (on > 0).sum() > 50
    assert np.array_equal(g.query_aabbs(boxes), o.query_aabbs(boxes))
    gh, oh = g.query_rays(rays, max_t), o.query_rays(rays, max_t)
    assert np.array_equal(gh["body"], oh["body"])
    for _ in range(5):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL
    assert rel_err(g.poses()[1], o.poses()[1], floor=1.0) <= TOL
    assert rel_err(g.velocities()[0], o.velocities()[0], floor=1.0) <= 1e-4
    assert rel_err(g.velocities()[1], o.velocities()[1], floor=10.0) <= 1e-4
    assert g.stats().overflow == 0


def test_several_colliders_per_body_match_oracle():
    """sfx_collider.body: a third of a soup's bodies carry one or two further colliders, the
    static floor and wall carry level geometry.  AABBs per collider, pair set, owner units (a further collider of a
    dynamic body stands in the tiles' lists under its BODY's LDS slot when it sits beside the body's first collider --
    nine in ten of their pairs here -- and on the fallback list otherwise), colours bit-exact; contacts, queries (which name
    the body of the collider hit, each body once per box) and twenty ticks of poses within tolerance."""
    from oracle.oracle import ORDER_COLORED
    from tests.test_oracle_queries import query_inputs

    s = scenes.random_soup(2000, seed=37, extent=9.0, offsets=0.3, compounds=0.35)
    assert s.n_colliders > s.n + 700
    g, o = make_pair(s)
    g.broadphase(DT)
    o.broadphase(DT)
    assert g.aabbs().shape == (s.n_colliders, 4)
    assert np.array_equal(g.aabbs().view(np.uint32), o.aabbs().view(np.uint32)), "AABBs differ bitwise"
    assert np.array_equal(g.pairs(), o.pairs())
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot) and np.array_equal(gc, oc)
    gs, os_ = g.stats(), o.stats()
    assert 0 < gs.n_fallback == os_.n_fallback < 400 and gs.n_bodies == os_.n_bodies == s.n
    body = s.colliders["body"].copy()
    body[: s.n] = np.arange(s.n)
    ka, kb = (gk >> np.uint64(32)).astype(np.int64), (gk & np.uint64(0xFFFFFFFF)).astype(np.int64)
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    cmp_ = ((ka >= s.n) & dyn[body[ka]]) | ((kb >= s.n) & dyn[body[kb]])
    assert (cmp_ & (gt != 0xFFFFFFFF)).sum() > 5 * (cmp_ & (gt == 0xFFFFFFFF)).sum() > 0   # compounds: 2286 tile pairs, 309 fallback
    gcon, ocon = g.contacts(), o.contacts()
    assert np.array_equal(gcon["count"], ocon["count"]) and (gcon["count"] > 0).sum() > 300
    assert rel_err(gcon["normal"], ocon["normal"], floor=1.0) <= TOL
    assert rel_err(gcon["depth"], ocon["depth"], floor=1e-2) <= TOL
    pts, boxes, rays, max_t = query_inputs(37, 9.0, n=500, near=s.pos)
    gb, gn = g.query_points(pts)
    ob, on = o.query_points(pts)
    assert np.array_equal(gb, ob) and np.array_equal(gn, on) and (on > 0).sum() > 50 and gb[gb != 0xFFFFFFFF].max() < s.n
    assert np.array_equal(g.query_aabbs(boxes), o.query_aabbs(boxes))
    gh, oh = g.query_rays(rays, max_t), o.query_rays(rays, max_t)
    assert np.array_equal(gh["body"], oh["body"])
    for _ in range(20):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    assert g.poses()[0].shape == (s.n, 2)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= 1e-4
    assert rel_err(g.poses()[1], o.poses()[1], floor=1.0) <= 1e-4
    assert rel_err(g.velocities()[0], o.velocities()[0], floor=1.0) <= 1e-3
    assert g.stats().overflow == 0
    # the same bodies, new state: the further colliders stay and follow
    pos, rot = o.poses()
    vel, ang = o.velocities()
    for w in (g, o):
        w.set_state(pos + np.float32(0.25), rot, vel, ang, s.inv_mass, s.inv_inertia)
        w.broadphase(DT)
    assert np.array_equal(g.aabbs().view(np.uint32), o.aabbs().view(np.uint32))
    assert np.array_equal(g.pairs(), o.pairs())


def test_dumbbell_and_static_level_geometry_on_the_gpu():
    """The oracle's known answers for compound bodies (tests/test_oracle_compound.py) on the GPU: a dumbbell (two discs
    and a bar, one body) comes to rest on both discs; a disc rolls down a ramp into the corner of a static body made of
    three colliders, whose pairs stay in the tiles; 300 ticks equal the oracle's within tolerance."""
    from oracle.oracle import ORDER_COLORED
    from tests.test_oracle_compound import dumbbell_scene
    from starframe_amd.scenes import SceneBuilder

    s, floor, k = dumbbell_scene()
    g, o = make_pair(s)
    for _ in range(240):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    gp, gr = g.poses()
    assert abs(gp[k, 1] - 0.2) < 3e-3 and abs(gr[k, 1]) < 2e-3
    assert rel_err(gp, o.poses()[0], floor=1.0) <= 1e-4 and rel_err(gr, o.poses()[1], floor=1.0) <= 1e-4
    assert g.stats().n_fallback == 1          # floor x second disc (the bar hovers above the floor)
    b = SceneBuilder("level", materials=[(0.0, 0.5, 0.4)])
    level = b.compound((0.0, 0.0), [("rect", (3.0, 0.5), (0.0, -0.5), 0.0), ("rect", (0.5, 2.0), (-3.5, 1.0), 0.0)], density=0.0)
    b.add_collider(level, "rect", (1.5, 0.05), (1.0, 0.6), 0.35)
    d = b.circle((1.5, 1.6), 0.15)
    g2, o2 = make_pair(b.build())
    for _ in range(600):
        g2.step(DT)
        o2.step(DT, ORDER_COLORED)
    p = g2.poses()[0]
    assert abs(p[d, 0] - (-3.0 + 0.15)) < 5e-3 and abs(p[d, 1] - 0.15) < 5e-3
    assert rel_err(p, o2.poses()[0], floor=1.0) <= 1e-4
    st = g2.stats()
    assert st.n_fallback == 0 and st.n_tiles >= 1 and st.overflow == 0


def test_second_grid_level_matches_oracle_and_the_one_level_grid():
    """The broadphase grid's second level (DESIGN.md "Contract / grid"; SURVEY.md 8a row a7: upstream's HGrid): a soup with
    400 mid-size static platforms and 40 mid-size dynamic boxes.  AABBs, pairs, owner units, colours bit-exact against the
    oracle; the same world with SFX_CFG_ONE_LEVEL (every collider too large for a base cell on the flat list) has the
    same pairs, owners and colours and, after ten ticks, the same poses bit for bit."""
    from oracle.oracle import ORDER_COLORED

    s = scenes.random_soup(4000, seed=41, extent=16.0, platforms=400)
    # square tiles: with strips a batch of sixteen 1.2 m columns spans the whole soup and its pairs would touch more than
    # the 64 outside colliders a batch has LDS slots for (the next test is about that)
    g, o = make_pair(s, cell_size=0.6, tile_cells_y=2)
    from starframe_amd.world import World
    g1 = World.from_scene(s, cell_size=0.6, tile_cells_y=2, flags=4)
    for w in (g, o, g1):
        w.broadphase(DT)
    assert np.array_equal(g.aabbs().view(np.uint32), o.aabbs().view(np.uint32))
    gp = g.pairs()
    assert len(gp) > 8_000 and np.array_equal(gp, o.pairs()) and np.array_equal(gp, g1.pairs())
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    k1, c1, t1 = colors_by_key(*g1.pair_colors())
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot) and np.array_equal(gc, oc)
    assert np.array_equal(gk, k1) and np.array_equal(gt, t1) and np.array_equal(gc, c1)
    gs, os_ = g.stats(), o.stats()
    assert (gs.n_small, gs.n_large, gs.n_pairs, gs.n_fallback, gs.n_tiles) == (os_.n_small, os_.n_large, os_.n_pairs, os_.n_fallback, os_.n_tiles)
    assert gs.n_large >= 430 and gs.overflow == 0
    for _ in range(10):
        g.step(DT)
        g1.step(DT)
        o.step(DT, ORDER_COLORED)
    assert np.array_equal(g.poses()[0].view(np.uint32), g1.poses()[0].view(np.uint32))
    assert np.array_equal(g.velocities()[0].view(np.uint32), g1.velocities()[0].view(np.uint32))
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= 1e-4
    assert g.stats().overflow == 0 and g1.stats().overflow == 0


def test_batches_shrink_when_their_pairs_touch_too_many_outside_colliders():
    """A batch of tiles keeps the colliders outside its cells that its pairs touch (container walls, platforms) in 64 LDS
    slots.  The soup of 400 platforms under full-height strips overruns that: the ticks that did drop those pairs and say
    so (overflow bit 64 -> SFX_ERR_CAPACITY at the next sync, once); the library then halves the tiles per batch until the
    batches fit, and the run goes on without errors."""
    from starframe_amd.world import SfxError, World

    s = scenes.random_soup(4000, seed=41, extent=16.0, platforms=400)
    w = World.from_scene(s, cell_size=0.6)
    reported = clean = 0
    for _ in range(12):
        try:
            for _ in range(9):       # the counters come back every eighth tick
                w.step(DT)
            w.sync()
            clean += 1
        except SfxError as e:
            assert e.code == -4 and "64" in str(e)
            reported += 1
            clean = 0
            w.sync()                 # reported once: the next sync is clean
    assert reported >= 1 and clean >= 4, (reported, clean)
    assert w.stats().overflow == 0
    w.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_small_world_path_equals_launch_per_stage(seed):
    """Worlds of at most 1024 bodies run their substeps in one workgroup with the bodies in LDS (k_small_substeps; one
    launch per sfx_step).  Scheduling only: the states are bit-identical to the launch-per-stage path -- random soups with
    every shape, local poses, several colliders per body (static and dynamic: the fallback list), joints of every kind
    with damping, 1..8 substeps, strips and square tiles -- tick by tick with sfx_step and substep by substep; and the
    one-workgroup world agrees with the oracle like any other."""
    from oracle.oracle import ORDER_COLORED, OracleWorld
    from starframe_amd._abi import CFG_MULTI_LAUNCH, CFG_ONE_WORKGROUP
    from starframe_amd.world import World

    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(30, 900))
    s = scenes.random_soup(n, seed=500 + seed, extent=max(2.0, 0.15 * n ** 0.5) * float(rng.uniform(1.0, 2.0)), speed=float(rng.uniform(0.0, 1.5)),
                           offsets=float(rng.choice([0.0, 0.3])), compounds=float(rng.choice([0.0, 0.2])),
                           platforms=int(rng.choice([0, 12])), joints=float(rng.choice([0.0, 0.3, 0.8])))
    cfg = dict(substeps=int(rng.integers(1, 9)), tile_cells=int(rng.choice([0, 2, 4])), tile_cells_y=int(rng.choice([0, 2, 4096])))
    one = World.from_scene(s, flags=CFG_ONE_WORKGROUP, **cfg)
    many = World.from_scene(s, flags=CFG_MULTI_LAUNCH, **cfg)
    auto = World.from_scene(s, **cfg)      # changes path when the first counters arrive
    o = OracleWorld.from_scene(s, **cfg)
    for t in range(6):
        if t % 2:
            for w in (one, many, auto):
                w.step(DT)
        else:       # substep by substep (what a slab world does)
            for w in (one, many, auto):
                w.tick_begin(DT)
                for _ in range(cfg["substeps"]):
                    w.tick_substep()
        o.step(DT, ORDER_COLORED)
        if t == 2:
            auto.sync()     # the counters are in: the next tick_begin adopts them
        p1, r1 = one.poses()
        v1, a1 = one.velocities()
        for w in (many, auto):
            p2, r2 = w.poses()
            v2, a2 = w.velocities()
            assert np.array_equal(p1.view(np.uint32), p2.view(np.uint32)) and np.array_equal(r1.view(np.uint32), r2.view(np.uint32)), (t, cfg)
            assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)) and np.array_equal(a1.view(np.uint32), a2.view(np.uint32)), (t, cfg)
        assert np.array_equal(one.aabbs().view(np.uint32), many.aabbs().view(np.uint32))    # the mirror entries too
    assert one.stats().launches < many.stats().launches
    assert one.stats().overflow == 0 and many.stats().overflow == 0
    assert rel_err(p1, o.poses()[0], floor=1.0) <= 1e-5 and rel_err(v1, o.velocities()[0], floor=1.0) <= 1e-4
    for w in (one, many, auto, o):
        w.close()


@pytest.mark.gpu
def test_small_world_batches_one_after_the_other(monkeypatch):
    """The one-workgroup kernel tabulates the colour runs of up to 64 batches in LDS; a world with more takes its batches
    one after the other.  SFX_SMALL_TABLED=0 (read at every launch) forces that path: same bits again."""
    from starframe_amd._abi import CFG_MULTI_LAUNCH, CFG_ONE_WORKGROUP
    from starframe_amd.world import World

    s = scenes.random_soup(700, seed=77, extent=6.0, compounds=0.2, joints=0.3, platforms=12)
    monkeypatch.setenv("SFX_SMALL_TABLED", "0")
    one = World.from_scene(s, flags=CFG_ONE_WORKGROUP, substeps=3)
    many = World.from_scene(s, flags=CFG_MULTI_LAUNCH, substeps=3)
    for _ in range(4):
        one.step(DT)
        many.step(DT)
    assert one.stats().launches < many.stats().launches and one.stats().overflow == 0
    for a, b in zip(one.poses() + one.velocities(), many.poses() + many.velocities()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    one.close()
    many.close()


@pytest.mark.gpu
def test_velocities_derived_where_first_read_equal_the_separate_pass():
    """Inside sfx_step the velocity solve derives a body's velocity when it first loads it (stamp in the velocity record's
    fourth word), bodies no pair touches are derived by the next integration or by one k_derive at the end of the tick.
    Same bits as sfx_tick_begin + sfx_tick_substep, which keep the separate k_derive pass after every position solve:
    a soup with free-flying bodies (no pair at all), statics, compounds on the fallback list and undamped joints."""
    from starframe_amd._abi import CFG_MULTI_LAUNCH
    from starframe_amd.world import World

    s = scenes.random_soup(2500, seed=21, extent=16.0, compounds=0.15, joints=0.3, platforms=12, speed=1.5)
    s.constraints["damping"] = 0.0
    s.angle_constraints["damping"] = 0.0
    a = World.from_scene(s, flags=CFG_MULTI_LAUNCH, substeps=5)   # sfx_step: derive-on-load
    b = World.from_scene(s, flags=CFG_MULTI_LAUNCH, substeps=5)   # substep by substep: k_derive every substep
    for t in range(8):
        a.step(DT)
        b.tick_begin(DT)
        for _ in range(5):
            b.tick_substep()
        for x, y in zip(a.poses() + a.velocities(), b.poses() + b.velocities()):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), t
    assert a.stats().n_fallback > 0 and a.stats().overflow == 0
    assert a.stats().launches < b.stats().launches
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,extent,kw,one", [
    (300, 4.0, dict(), True),                                   # automatic: at most SFX_ONE_TILE_COLLIDERS colliders
    (300, 4.0, dict(tile_cells=2), False),                      # tiles asked for
    (800, 9.0, dict(tile_cells=0xFFFF), True),                  # one tile asked for, within a tile's capacity
    (450, 5.0, dict(tile_cells=0xFFFF, tile_cells_y=2), True),  # (the height is then the grid's too)
    (1000, 4.5, dict(tile_cells=0xFFFF), True),                 # more pairs than a tile holds: all on the fallback list
], ids=["auto", "tiles_asked", "whole_grid", "whole_grid_y", "whole_grid_overfull"])
def test_one_tile_worlds_match_oracle(n, extent, kw, one):
    """The whole grid as ONE tile (DESIGN.md "Contract / tiles"): every pair -- also those across the torus' wrap lines --
    is an interior pair of the one unit: one colouring, no seam phase.  Pairs, units, colours and three ticks against
    the oracle; such a world runs its substeps in one workgroup unless told otherwise, with the same bits."""
    from oracle.oracle import ORDER_COLORED
    from starframe_amd._abi import CFG_MULTI_LAUNCH

    s = scenes.random_soup(n, seed=31, extent=extent, offsets=0.3, compounds=0.1 if n < 1000 else 0.0)
    g, o = make_pair(s, **kw)
    m, _ = make_pair(s, flags=CFG_MULTI_LAUNCH, **kw)
    g.broadphase(DT)
    o.broadphase(DT)
    assert np.array_equal(g.pairs(), o.pairs())
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot) and np.array_equal(gc, oc)
    gs, os_ = g.stats(), o.stats()
    assert (gs.n_pairs, gs.n_solver_pairs, gs.n_fallback, gs.n_colors) == (os_.n_pairs, os_.n_solver_pairs, os_.n_fallback, os_.n_colors)
    tile_units = gt[gt != 0xFFFFFFFF] if len(gt) else gt
    if one:
        assert np.all(tile_units == 0)            # unit = class << 28 | tile: the interior class of tile 0
    else:
        assert len(np.unique(tile_units)) > 1
    if n == 1000:
        assert gs.n_fallback == gs.n_pairs > 2048
    for _ in range(3):
        g.step(DT)
        m.step(DT)
        o.step(DT, ORDER_COLORED)
    assert g.stats().overflow == 0 and o.stats().overflow == 0
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL and rel_err(g.poses()[1], o.poses()[1], floor=1.0) <= TOL
    assert rel_err(g.velocities()[0], o.velocities()[0], floor=1.0) <= 1e-4
    for a, b in zip(g.poses() + g.velocities(), m.poses() + m.velocities()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if one and n <= 2048:
        assert g.stats().launches < m.stats().launches      # one launch for the substeps


@pytest.mark.gpu
def test_small_level_with_many_platforms_stays_tiled():
    """A small world whose large colliders (wider than a grid cell: platforms) outnumber the 64 outside-body slots of a
    batch is NOT made one tile by itself: tiles spread them.  Sixty bodies falling over a hundred platforms: no overflow,
    the oracle's pairs, colours and poses; asked for as one tile, the same world reports the overflow on both sides."""
    from oracle.oracle import ORDER_COLORED
    from starframe_amd.world import SfxError

    s = scenes.random_soup(60, seed=3, extent=12.0, platforms=100, speed=0.5)
    g, o = make_pair(s, cell_size=0.6)
    assert o.resolved_config().tile_cells == 2
    for _ in range(30):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    assert g.stats().overflow == 0 and o.stats().overflow == 0
    assert np.array_equal(g.pairs(), o.pairs())
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= 1e-4   # (thirty ticks: roundoff grows)
    g.close()
    o.close()
