"""CPU tests of the oracle's broadphase and colouring restatement: the candidate pair set is
a pure function of the AABBs (three independent enumerations agree and the grid parameters do
not matter), and the colouring is a valid edge colouring."""
import numpy as np
import pytest

from oracle.oracle import OracleWorld
from starframe_amd import scenes
from starframe_amd.scenes import SceneBuilder
from tests.helpers import PRUNED_COLOR, coloring_valid, colors_by_key, tile_phases_disjoint

DT = 1.0 / 60.0


@pytest.mark.parametrize("n,seed,extent", [(200, 1, 3.0), (800, 2, 6.0), (2500, 3, 12.0)])
def test_three_enumerations_agree(n, seed, extent):
    w = OracleWorld.from_scene(scenes.random_soup(n, seed=seed, extent=extent))
    w.broadphase(DT)
    g, s = w.pairs(), w.pairs_sweep()
    assert len(g) > n // 2
    assert np.array_equal(g, s)
    if n <= 800:
        assert np.array_equal(g, w.pairs_brute())
    assert np.all(np.diff(g.astype(np.int64)) > 0), "keys must be strictly ascending (no duplicates)"


@pytest.mark.parametrize("kw", [dict(grid_log2=(2, 2)), dict(grid_log2=(3, 7)), dict(cell_size=0.9),
                                dict(cell_size=3.0), dict(cell_size=0.2)])
def test_pair_set_independent_of_grid(kw):
    s = scenes.random_soup(600, seed=7, extent=5.0)
    ref = OracleWorld.from_scene(s)
    ref.broadphase(DT)
    w = OracleWorld.from_scene(s, **kw)
    w.broadphase(DT)
    assert np.array_equal(w.pairs(), ref.pairs())
    assert np.array_equal(w.pairs(), w.pairs_sweep())


def test_negative_and_far_coordinates_wrap():
    s = scenes.random_soup(400, seed=9, extent=4.0)
    s.pos += np.float32((-1234.5, 987.25))
    w = OracleWorld.from_scene(s, grid_log2=(3, 3))
    w.broadphase(DT)
    assert len(w.pairs()) > 200
    assert np.array_equal(w.pairs(), w.pairs_sweep())


def test_filters_group_mask_static():
    b = SceneBuilder("filters")
    b.circle((0, 0), 0.5)
    b.circle((0.2, 0), 0.5, group=3)
    b.circle((0.4, 0), 0.5, group=3)                  # same group as body 1: filtered
    b.circle((0.1, 0.1), 0.5, category=2, mask=0xFFFFFFFE)  # does not collide with category 1
    b.circle((0.3, 0.1), 0.5, category=2, mask=0xFFFFFFFF)
    b.static_rect((0, -1), 5, 0.6)
    b.static_rect((0, -1.2), 5, 0.6)                 # static-static: filtered
    b.particle((0, 0))                               # no collider
    w = OracleWorld.from_scene(b.build())
    w.broadphase(DT)
    keys = set(int(k) for k in w.pairs())
    key = lambda a, c: (a << 32) | c
    assert key(1, 2) not in keys
    assert key(0, 3) not in keys and key(1, 3) not in keys and key(2, 3) not in keys
    assert key(3, 4) in keys        # both category 2, 4's mask has bit 2, 3's mask has bit 2
    assert key(0, 4) in keys        # 4 is category 2 with a full mask, 0 accepts everything
    assert key(5, 6) not in keys
    assert all((k >> 32) != 7 and (k & 0xFFFFFFFF) != 7 for k in keys)
    assert key(0, 1) in keys and key(0, 5) in keys


@pytest.mark.parametrize("rounds", [1, 3, 12])
def test_colouring_is_valid_and_deterministic(rounds):
    s = scenes.random_soup(1500, seed=4, extent=7.0)
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    w = OracleWorld.from_scene(s, color_rounds=rounds)
    w.broadphase(DT)
    k1, c1, t1 = w.pair_colors()
    assert coloring_valid(k1, c1, dyn, t1)
    rc = w.resolved_config()
    assert tile_phases_disjoint(k1, t1, dyn, tuple(rc.grid_log2), rc.tile_cells)
    st = w.stats()
    assert st.n_colors <= 40
    assert sum(st.color_count) == st.n_solver_pairs == int((c1 != PRUNED_COLOR).sum())
    assert 0 < st.n_solver_pairs < st.n_pairs   # the reach test dropped some AABB-only candidates
    assert st.n_tiles > 4 and st.n_fallback == int((t1 == 0xFFFFFFFF).sum())
    if rounds == 1:
        assert st.n_leftover > 0  # the serial finish did real work
    w2 = OracleWorld.from_scene(s, color_rounds=rounds)
    w2.broadphase(DT)
    k2, c2, t2 = w2.pair_colors()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2) and np.array_equal(t1, t2)


@pytest.mark.parametrize("tile_cells,cell", [(2, 0.0), (4, 0.0), (8, 0.0), (2, 0.2)])
def test_tile_ownership_keeps_phases_disjoint(tile_cells, cell):
    """Every tile size, and a cell size that makes big dynamic bodies 'large' (fallback list):
    the units (tile, class) of one class -- one phase -- never share a dynamic body, colours are valid per unit;
    all four classes occur, the interior class holds most pairs."""
    s = scenes.random_soup(1500, seed=9, extent=7.0)
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    w = OracleWorld.from_scene(s, tile_cells=tile_cells, tile_cells_y=tile_cells, cell_size=cell)
    w.broadphase(DT)
    k, c, t = w.pair_colors()
    rc = w.resolved_config()
    assert rc.tile_cells == tile_cells
    assert coloring_valid(k, c, dyn, t)
    assert tile_phases_disjoint(k, t, dyn, tuple(rc.grid_log2), rc.tile_cells)
    cls = (t[t != 0xFFFFFFFF] >> np.uint32(28)).astype(int)
    hist = np.bincount(cls, minlength=4)
    assert cell or (hist > 0).all(), hist
    assert cell or tile_cells <= 2 or hist[0] == hist.max()
    st = w.stats()
    assert st.max_tile_pairs <= 2048 and st.max_tile_bodies <= 1024
    if cell:
        assert st.n_fallback > 0 and st.n_large > 3


def test_a_tile_is_at_least_two_cells_wide_and_high():
    """tile_cells = 1 is raised to 2: with one-cell tiles a body would sit in the last and the first cell column of
    its tile at once and two x seams could share it."""
    s = scenes.random_soup(300, seed=1)
    rc = OracleWorld.from_scene(s, tile_cells=1, tile_cells_y=1).resolved_config()
    assert rc.tile_cells == 2 and rc.tile_cells_y == 2


def test_strip_tiles_have_two_phases():
    """tile_cells_y >= grid height: full-height strips, only the classes 0 (interior) and 1 (x seam)."""
    s = scenes.random_soup(1500, seed=9, extent=7.0)
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    w = OracleWorld.from_scene(s, tile_cells_y=4096)
    w.broadphase(DT)
    rc = w.resolved_config()
    assert rc.tile_cells == 2 and rc.tile_cells_y == 1 << rc.grid_log2[1]
    k, c, t = w.pair_colors()
    ok = t != 0xFFFFFFFF
    assert ok.sum() > 1000
    assert (t[ok] & 0x0FFFFFFF).max() < (1 << (rc.grid_log2[0] - 1))   # one tile row
    assert set(np.unique(t[ok] >> np.uint32(28)).tolist()) == {0, 1}
    assert coloring_valid(k, c, dyn, t)
    assert tile_phases_disjoint(k, t, dyn)
    # a seam joins a tile with its right neighbour: every seam pair has one body homed in each
    w2 = OracleWorld.from_scene(s, tile_cells=2, tile_cells_y=4)
    w2.broadphase(DT)
    rc2 = w2.resolved_config()
    assert (rc2.tile_cells, rc2.tile_cells_y) == (2, 4)
    k2, c2, t2 = w2.pair_colors()
    assert coloring_valid(k2, c2, dyn, t2) and tile_phases_disjoint(k2, t2, dyn)
    for _ in range(3):
        w.step(DT)
        w2.step(DT)
    assert np.isfinite(w.poses()[0]).all() and np.isfinite(w2.poses()[0]).all()


def test_overfull_tiles_fall_back():
    """Many tiny bodies next to one big dynamic one: the cell is sized by the big body, tiles
    overflow SFX_TILE_MAX_PAIRS / _BODIES and hand their pairs to the fallback list."""
    from starframe_amd.scenes import SceneBuilder

    b = SceneBuilder("dense")
    b.static_rect((0.0, -0.5), 10.0, 0.5)
    b.circle((6.0, 1.0), 1.0)
    for i in range(40):
        for j in range(40):
            b.circle((-2.0 + 0.05 * i, 0.03 + 0.05 * j), 0.026)
    s = b.build()
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    w = OracleWorld.from_scene(s)
    rc = w.resolved_config()
    assert rc.tile_cells_y == rc.tile_cells == 2   # 1600 colliders in two columns: no automatic strips
    w.broadphase(DT)
    k, c, t = w.pair_colors()
    st = w.stats()
    assert st.n_fallback > 1000
    assert coloring_valid(k, c, dyn, t)
    for _ in range(5):
        w.step(DT)
    assert np.isfinite(w.poses()[0]).all()


def test_chain_joint_colouring_two_colours_possible():
    s = scenes.chains(n_chains=5, links=40)
    w = OracleWorld.from_scene(s)
    perm, col = w.joint_order()
    assert sorted(perm.tolist()) == list(range(200))
    assert np.all(np.diff(col.astype(int)) >= 0), "colour-major order"
    # validity: within a colour no body twice
    for c in np.unique(col):
        idx = perm[col == c]
        ends = np.concatenate([s.constraints["body_a"][idx], s.constraints["body_b"][idx]])
        ends = ends[ends != 0xFFFFFFFF]
        assert len(np.unique(ends)) == len(ends)
    assert w.stats().n_joint_colors <= 8


def test_auto_grid_and_cell_resolution():
    s = scenes.random_soup(300, seed=1)
    w = OracleWorld.from_scene(s)
    rc = w.resolved_config()
    assert rc.cell_size > 0.4
    assert rc.grid_log2[0] == rc.grid_log2[1] == 5
    assert rc.max_pairs == 12 * s.n
    assert rc.tile_cells == 0xFFFF and rc.tile_cells_y == 32   # at most SFX_ONE_TILE_COLLIDERS colliders: the whole grid is ONE tile
    w.broadphase(1.0 / 60.0)
    k, c, t = w.pair_colors()
    assert len(k) and np.all(t == 0)                            # ... every pair an interior pair of the one unit
    rc = OracleWorld.from_scene(s, tile_cells=2).resolved_config()
    assert rc.tile_cells == 2 and rc.tile_cells_y == 32         # asked for tiles: sparse scene, automatic full-height strips
    big = scenes.random_soup(1000, seed=2, extent=6.0)
    rc = OracleWorld.from_scene(big).resolved_config()
    assert rc.tile_cells == 2                                   # larger worlds are tiled
    rc = OracleWorld.from_scene(big, tile_cells=0xFFFF).resolved_config()
    assert rc.tile_cells == 0xFFFF                              # unless one tile is asked for
    level = scenes.random_soup(60, seed=3, extent=12.0, platforms=100)   # a level: 100 platforms wider than a cell
    rc = OracleWorld.from_scene(level, cell_size=0.6).resolved_config()
    assert rc.tile_cells == 2                                   # more large colliders than one batch keeps: tiled
    assert OracleWorld.from_scene(level).resolved_config().tile_cells == 0xFFFF   # (automatic cells are as large as its big boxes)


@pytest.mark.parametrize("speed", [0.0, 1.5])
def test_reach_test_only_drops_pairs_that_cannot_touch(speed):
    """DESIGN.md "Contract / solver pairs": a dropped candidate pair has no contact point, not even
    a speculative one, at the tick's start, and stays apart for the whole tick when the bodies move
    as far as their AABB pads allow for."""
    s = scenes.random_soup(1500, seed=11, extent=7.0)
    rng = np.random.default_rng(5)
    s.vel[:] = speed * rng.uniform(-1, 1, s.vel.shape).astype(np.float32) * (s.inv_mass > 0)[:, None]
    w = OracleWorld.from_scene(s, gravity=(0.0, 0.0))
    w.broadphase(DT)
    k, c, t = w.pair_colors()
    dropped = set(int(x) for x in k[c == PRUNED_COLOR])
    assert len(dropped) > 50
    assert set(int(x) for x in k) == set(int(x) for x in w.pairs())  # the candidate set itself is untouched

    def touching():
        con = w.contacts()
        keys = (con["body_a"].astype(np.uint64) << np.uint64(32)) | con["body_b"].astype(np.uint64)
        return set(int(x) for x in keys[con["count"] > 0])

    assert not (touching() & dropped)
    if speed:
        # move everything by its velocity over the tick (no forces, no solve): still nothing dropped touches
        pos, rot = w.poses()
        w2 = OracleWorld.from_scene(s.with_state(pos + s.vel * DT, rot), gravity=(0.0, 0.0))
        w2.broadphase(DT)
        con = w2.contacts()
        keys = (con["body_a"].astype(np.uint64) << np.uint64(32)) | con["body_b"].astype(np.uint64)
        assert not (set(int(x) for x in keys[con["count"] > 0]) & dropped)


@pytest.mark.parametrize("seed", [5, 6])
def test_second_grid_level_changes_nothing_but_the_cost(seed):
    """Two-level grid (DESIGN.md "Contract / grid"; upstream's broadphase is a hierarchical grid): a soup with 120 mid-size
    static platforms and a dozen mid-size dynamic boxes.  The medium colliders are sorted into level-1 cells instead of the
    flat list: grid = sweep = brute force; SFX_CFG_ONE_LEVEL gives the same pairs, owners and colours; a small collider
    with a static medium one stays in its tile, a dynamic medium one goes to the fallback list; ten ticks agree bit for bit."""
    s = scenes.random_soup(1200, seed=seed, extent=10.0, platforms=120)
    # (the cell size is set: derived from the largest dynamic body it would be 3 m here and the platforms small)
    two, one = OracleWorld.from_scene(s, cell_size=0.6), OracleWorld.from_scene(s, cell_size=0.6, flags=4)
    for w in (two, one):
        w.broadphase(DT)
    g, sw, br = two.pairs(), two.pairs_sweep(), two.pairs_brute()
    assert len(g) > 2000 and np.array_equal(g, sw) and np.array_equal(g, br)
    assert np.array_equal(g, one.pairs())
    k2, c2, t2 = colors_by_key(*two.pair_colors())
    k1, c1, t1 = colors_by_key(*one.pair_colors())
    assert np.array_equal(k2, k1) and np.array_equal(c2, c1) and np.array_equal(t2, t1)
    st2, st1 = two.stats(), one.stats()
    assert st2.n_small == st1.n_small and st2.n_large == st1.n_large >= 120     # medium + listed colliders
    # which colliders are medium: bounding boxes too large for a base cell, small enough for a level-1 cell
    rc = two.resolved_config()
    A = two.aabbs()
    ext = np.maximum(A[:, 2] - A[:, 0], A[:, 3] - A[:, 1])
    medium = (ext > np.float32(0.9) * np.float32(rc.cell_size)) & (ext <= np.float32(0.9) * (np.float32(8.0) * np.float32(rc.cell_size)))
    assert medium.sum() >= 110
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    a, b = (k2 >> np.uint64(32)).astype(np.int64), (k2 & np.uint64(0xFFFFFFFF)).astype(np.int64)
    with_static_medium = (medium[b] & ~dyn[b] & ~medium[a]) | (medium[a] & ~dyn[a] & ~medium[b])
    with_dynamic_medium = (medium[a] & dyn[a]) | (medium[b] & dyn[b])
    assert with_static_medium.sum() > 300 and np.all(t2[with_static_medium & ~with_dynamic_medium] != 0xFFFFFFFF)
    assert with_dynamic_medium.sum() > 20 and np.all(t2[with_dynamic_medium] == 0xFFFFFFFF)
    for _ in range(10):
        two.step(DT)
        one.step(DT)
    assert np.array_equal(two.poses()[0].view(np.uint32), one.poses()[0].view(np.uint32))
    assert np.array_equal(two.velocities()[0].view(np.uint32), one.velocities()[0].view(np.uint32))
