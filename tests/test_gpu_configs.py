"""GPU parity on the BASELINE.json configs at their real sizes (VERDICT r1 "configs_untested").

libsfx (HIP, through the C ABI) against the CPU oracle on identical inputs: the 100 k mixed pile (configs[2], the
headline) one tick from the lattice and one tick from the settled state, 10 k circles (configs[1]), 1000 x 100 chains
(configs[3]); then the north star's long-horizon statistics (stack height, kinetic energy, penetration depth) on 600
frames of configs[1] and a 20 k mixed pile.  Bars: candidate pairs, owner tiles, colours bit-exact; contacts and
single-tick poses within 1e-5 relative, velocities 1e-4.

The oracle is this repository's CPU restatement (PARITY UNPINNED against starframe itself); it shares the
contract's helper arithmetic with the device code by construction, so equality here shows that the parallel
implementation is faithful to the contract -- the contract itself is pinned by the analytic KATs
(tests/test_oracle_kat.py, test_oracle_broadphase.py)."""
import numpy as np
import pytest

from starframe_amd import scenes
from tests.helpers import coloring_valid, colors_by_key, rel_err, tile_phases_disjoint

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0
TOL = 1e-5


def make_pair(scene, **kw):
    from oracle.oracle import OracleWorld
    from starframe_amd.world import World

    return World.from_scene(scene, **kw), OracleWorld.from_scene(scene, **kw)


def settle_on_gpu(scene, frames, **kw):
    from starframe_amd.world import World

    w = World.from_scene(scene, linear_damping=3.3, angular_damping=1.0, **kw)
    for _ in range(frames):
        w.step(DT)
    w.sync()
    pos, rot = w.poses()
    assert w.stats().overflow == 0
    w.close()
    return scene.with_state(pos, rot, name=scene.name + "_settled")


def depths(contacts):
    d = np.concatenate([contacts["depth"][:, 0][contacts["count"] > 0], contacts["depth"][:, 1][contacts["count"] > 1]])
    return d[d > 0]


def check_one_tick(s, min_pairs, min_manifolds, contacts=True):
    """Broadphase objects bit-exact, narrowphase and one colour-ordered tick within tolerance."""
    from oracle.oracle import ORDER_COLORED

    g, o = make_pair(s)
    g.broadphase(DT)
    o.broadphase(DT)
    assert np.array_equal(g.aabbs().view(np.uint32), o.aabbs().view(np.uint32)), "AABBs differ bitwise"
    gp = g.pairs()
    assert len(gp) >= min_pairs
    assert np.array_equal(gp, o.pairs()), "candidate pair set differs"
    gk, gc, gt = colors_by_key(*g.pair_colors())
    ok, oc, ot = colors_by_key(*o.pair_colors())
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    rc = o.resolved_config()
    assert coloring_valid(gk, gc, dyn, gt)
    if rc.tile_cells_y == rc.tile_cells:   # square tiles: the four-phase disjointness check of tests/helpers.py applies
        assert tile_phases_disjoint(gk, gt, dyn, tuple(rc.grid_log2), rc.tile_cells)
    assert np.array_equal(gk, ok) and np.array_equal(gt, ot), "owner tiles differ"
    assert np.array_equal(gc, oc), "colours differ"
    gs, os_ = g.stats(), o.stats()
    assert gs.overflow == 0
    assert (gs.n_small, gs.n_large, gs.n_pairs, gs.n_solver_pairs, gs.n_colors, gs.n_tiles, gs.n_fallback) == \
           (os_.n_small, os_.n_large, os_.n_pairs, os_.n_solver_pairs, os_.n_colors, os_.n_tiles, os_.n_fallback)
    if contacts:
        gcon, ocon = g.contacts(), o.contacts()
        assert np.array_equal(gcon["body_a"], ocon["body_a"]) and np.array_equal(gcon["body_b"], ocon["body_b"])
        assert np.array_equal(gcon["count"], ocon["count"]), "manifold point counts differ"
        assert (gcon["count"] > 0).sum() >= min_manifolds
        assert rel_err(gcon["normal"], ocon["normal"], floor=1.0) <= TOL
        assert rel_err(gcon["depth"], ocon["depth"], floor=1e-2) <= TOL
        assert rel_err(gcon["point_a"], ocon["point_a"], floor=1.0) <= TOL
        assert rel_err(gcon["point_b"], ocon["point_b"], floor=1.0) <= TOL
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    g.sync()
    (gpos, grot), (opos, orot) = g.poses(), o.poses()
    (gv, gw), (ov, ow) = g.velocities(), o.velocities()
    assert rel_err(gpos, opos, floor=1.0) <= TOL
    assert rel_err(grot, orot, floor=1.0) <= TOL
    assert rel_err(gv, ov, floor=1.0) <= 1e-4
    assert rel_err(gw, ow, floor=10.0) <= 1e-4
    gs, os_ = g.stats(), o.stats()
    assert gs.overflow == 0 and gs.n_manifolds == os_.n_manifolds and gs.n_contacts == os_.n_contacts
    assert gs.n_manifolds >= min_manifolds
    g.close()
    o.close()


def test_mixed_pile_100k_one_tick_from_the_lattice():
    """configs[2] as generated: 100 k bodies on the lattice (pitch 0.30 m, bodies up to 0.28 m across: a few dozen
    candidate pairs), and the same scene 150 undamped ticks later, when the lower rows have landed on each other."""
    from starframe_amd.world import World

    s = scenes.mixed_pile()
    check_one_tick(s, min_pairs=50, min_manifolds=0)
    w = World.from_scene(s)
    for _ in range(150):
        w.step(DT)
    w.sync()
    (pos, rot), (vel, ang) = w.poses(), w.velocities()
    w.close()
    check_one_tick(s.with_state(pos, rot, vel, ang, name="mixed_pile_falling"), min_pairs=100_000, min_manifolds=30_000)


def test_mixed_pile_100k_one_tick_from_the_settled_state():
    """configs[2] as benchmarked: the pile settled on the GPU (as bench.py does), handed to both sides."""
    s = settle_on_gpu(scenes.mixed_pile(), 600)
    check_one_tick(s, min_pairs=300_000, min_manifolds=200_000)


def test_mixed_pile_100k_w400_one_tick_from_the_settled_state():
    """SURVEY.md 8d's own lattice for configs[2] (400 wide, 250 rows: the pile bench.py reports as config.survey_scene,
    about 55 layers deep, 2-D tiles with all four pair classes): settled on the GPU, one tick on both sides."""
    s = settle_on_gpu(scenes.mixed_pile(width=400), 1500)
    check_one_tick(s, min_pairs=300_000, min_manifolds=200_000)


def test_circles_10k_one_tick_lattice_and_settled():
    """configs[1]: 10 k circles in the container."""
    s = scenes.circles_container()
    check_one_tick(s, min_pairs=100, min_manifolds=0)
    check_one_tick(settle_on_gpu(s, 500), min_pairs=20_000, min_manifolds=15_000)


def test_chains_1000x100_one_tick():
    """configs[3]: 1000 chains x 100 links: joint colours equal, one tick within tolerance."""
    from oracle.oracle import ORDER_COLORED

    s = scenes.chains()
    g, o = make_pair(s)
    gperm, gcol = g.joint_order()
    operm, ocol = o.joint_order()
    assert len(gperm) == 100_000
    assert np.array_equal(gperm, operm) and np.array_equal(gcol, ocol), "joint order / colours differ"
    g.broadphase(DT)
    o.broadphase(DT)
    assert np.array_equal(g.pairs(), o.pairs())
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    assert rel_err(g.poses()[0], o.poses()[0], floor=1.0) <= TOL
    assert rel_err(g.poses()[1], o.poses()[1], floor=1.0) <= TOL
    assert rel_err(g.velocities()[0], o.velocities()[0], floor=1.0) <= 1e-4
    assert g.stats().overflow == 0


def long_run_statistics(s, frames, n_dyn):
    """Both sides tick `frames` frames (colour order); returns per side: pile height, kinetic energy, depths."""
    from oracle.oracle import ORDER_COLORED

    g, o = make_pair(s)
    for _ in range(frames):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    g.sync()
    out = []
    for w in (g, o):
        pos, _ = w.poses()
        w.broadphase(DT)
        d = depths(w.contacts())
        st = w.stats()
        assert np.isfinite(pos).all()
        out.append(dict(height=float(np.percentile(pos[:n_dyn, 1], 99.5)), floor=float(pos[:n_dyn, 1].min()),
                        ke=float(st.kinetic_energy), p50=float(np.percentile(d, 50)), p99=float(np.percentile(d, 99)),
                        dmax=float(d.max()), n=len(d)))
    assert g.stats().overflow == 0
    g.close()
    o.close()
    return out


def compare_statistics(gs, os_, height_tol, tag):
    print(tag, "gpu", gs, "oracle", os_)
    assert abs(gs["height"] - os_["height"]) < height_tol, "pile height"
    assert gs["floor"] > -0.02 and os_["floor"] > -0.02, "nothing below the floor"
    assert abs(gs["n"] - os_["n"]) < 0.03 * os_["n"], "contact point count"
    assert abs(gs["p50"] - os_["p50"]) < 0.25 * os_["p50"] + 2e-4, "median penetration"
    assert abs(gs["p99"] - os_["p99"]) < 0.25 * os_["p99"] + 1e-3, "p99 penetration"
    assert gs["dmax"] < 2.0 * os_["dmax"] + 5e-3 and os_["dmax"] < 2.0 * gs["dmax"] + 5e-3, "deepest penetration"
    # kinetic energy of a pile that has come to rest: both small, same order
    assert gs["ke"] < 10.0 * os_["ke"] + 0.05 and os_["ke"] < 10.0 * gs["ke"] + 0.05, "kinetic energy"


def test_circles_10k_600_frames_statistics():
    """configs[1], 600 frames from the lattice, undamped: the circles fall into the container and come to rest.  Colour
    order on both sides; trajectories diverge chaotically, the statistics the north star names agree."""
    s = scenes.circles_container()
    gs, os_ = long_run_statistics(s, 600, 10_000)
    compare_statistics(gs, os_, 0.15, "circles10k")
    assert gs["p99"] < 0.02 and gs["ke"] < 1.0


def test_mixed_pile_20k_600_frames_statistics():
    """A 20 k mixed pile (200 wide, 100 rows like the headline scene), 600 damped frames on both sides."""
    from oracle.oracle import ORDER_COLORED

    s = scenes.mixed_pile(n=20_000, width=200)
    g, o = make_pair(s, linear_damping=3.3, angular_damping=1.0)
    for _ in range(600):
        g.step(DT)
        o.step(DT, ORDER_COLORED)
    g.sync()
    stats = []
    for w in (g, o):
        pos, _ = w.poses()
        w.broadphase(DT)
        d = depths(w.contacts())
        st = w.stats()
        assert np.isfinite(pos).all()
        stats.append(dict(height=float(np.percentile(pos[:20_000, 1], 99.5)), floor=float(pos[:20_000, 1].min()),
                          ke=float(st.kinetic_energy), p50=float(np.percentile(d, 50)), p99=float(np.percentile(d, 99)),
                          dmax=float(d.max()), n=len(d)))
    assert g.stats().overflow == 0
    compare_statistics(stats[0], stats[1], 0.25, "pile20k")
    assert stats[0]["p99"] < 0.04 and stats[0]["p50"] < 0.006


def test_granular_slab_one_gpus_share_500k_one_tick():
    """configs[4] (4 M circles over 8 GPUs) at ONE GPU's share, 500 k circles, settled on the GPU: broadphase objects
    bit-exact against the oracle, one tick within tolerance (the contacts' hook is skipped: 1.4 M manifolds on the host)."""
    s = settle_on_gpu(scenes.granular_slab(n=500_000, width=5000), 300)
    check_one_tick(s, min_pairs=1_200_000, min_manifolds=900_000, contacts=False)


def test_granular_slab_in_eight_logical_slabs_equals_the_oracles_slabs():
    """configs[4]'s decomposition in small: a 160 k-circle slab cut into EIGHT slabs on x (eight worlds on the one GPU,
    halos copied device to device per substep, the halo guard armed), against the same eight slabs on the oracle: one tick
    bit-comparable, then 60 ticks stay a pile at rest."""
    from oracle.oracle import OracleWorld
    from starframe_amd.slab import LocalSlabs
    from starframe_amd.world import World

    s = settle_on_gpu(scenes.granular_slab(n=160_000, width=1600), 400)
    G = LocalSlabs(s, 8, World, halo=2.0)
    O = LocalSlabs(s, 8, OracleWorld, halo=2.0)
    assert sum(len(p.recv_idx) for p in G.plans) > 8000          # 14 cut faces, some hundred ghosts each
    G.step(DT)
    O.step(DT)
    gp, gr, gv, ga = G.gather()
    op, orr, ov, oa = O.gather()
    assert rel_err(gp, op, floor=1.0) <= TOL and rel_err(gr, orr, floor=1.0) <= TOL
    assert rel_err(gv, ov, floor=1.0) <= 1e-4
    O.close()
    W = World.from_scene(s)          # the same pile undecomposed: released from the damped settling, both sag alike
    W.step(DT)
    for _ in range(60):
        G.step(DT)
        W.step(DT)
    pos, ref = G.gather()[0], W.poses()[0]
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    assert np.isfinite(pos).all() and pos[dyn, 1].min() > -0.05
    assert abs(pos[dyn, 1].mean() - ref[dyn, 1].mean()) < 0.003
    assert abs(np.percentile(pos[dyn, 1], 99.5) - np.percentile(ref[dyn, 1], 99.5)) < 0.02
    W.close()
    assert all(w.stats().overflow == 0 for w in G.worlds)
    G.close()
