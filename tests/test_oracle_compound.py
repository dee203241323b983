"""Several colliders per body (include/sfx.h sfx_collider.body; DESIGN.md "Several colliders per body") on the CPU
oracle: known answers.  SURVEY.md section 8a row a4."""
import numpy as np
import pytest

from oracle.oracle import ORDER_COLORED, ORDER_SEQUENTIAL, OracleWorld
from starframe_amd._abi import SHAPE_NONE, WORLD_BODY
from starframe_amd.scenes import SceneBuilder, random_soup

DT = 1.0 / 60.0
FALLBACK = 0xFFFFFFFF


def dumbbell_scene(angle=0.0, y=0.6):
    b = SceneBuilder("dumbbell", materials=[(0.0, 0.6, 0.5)])
    floor = b.static_rect((0.0, -0.5), 5.0, 0.5)
    k = b.compound((0.0, y), [("circle", (0.2,), (-0.5, 0.0), 0.0), ("circle", (0.2,), (0.5, 0.0), 0.0),
                              ("rect", (0.5, 0.04), (0.0, 0.0), 0.0)], angle=angle)
    return b.build(), floor, k


@pytest.mark.parametrize("order", [ORDER_SEQUENTIAL, ORDER_COLORED])
def test_dumbbell_comes_to_rest_on_both_wheels(order):
    """Two discs and a bar as ONE body, dropped level: it lands on both discs and stays level, its origin one disc
    radius above the floor (less the contact compression of XPBD at 4 substeps)."""
    s, floor, k = dumbbell_scene()
    assert s.n == 2 and s.n_colliders == 4 and list(s.colliders["body"][2:]) == [k, k]
    w = OracleWorld.from_scene(s)
    for _ in range(240):
        w.step(DT, order)
    pos, rot = w.poses()
    vel, ang = w.velocities()
    assert abs(pos[k, 1] - 0.2) < 3e-3, pos[k]
    assert abs(pos[k, 0]) < 1e-3 and abs(rot[k, 1]) < 2e-3
    assert np.abs(vel[k]).max() < 2e-2 and abs(ang[k]) < 2e-2
    w.broadphase(DT)
    c = w.contacts()
    touching = {(int(a), int(b)) for a, b, n in zip(c["body_a"], c["body_b"], c["count"]) if n}
    # colliders 1 (the first disc, the body's own) and 2 (the second disc, a further collider) touch the floor (0)
    assert (0, 1) in touching and (0, 2) in touching
    w.close()


def test_tilted_dumbbell_rights_itself_and_no_collider_pair_within_a_body():
    s, floor, k = dumbbell_scene(angle=0.5, y=0.9)
    w = OracleWorld.from_scene(s)
    w.broadphase(DT)
    keys = w.pairs()
    body = s.colliders["body"].copy()
    body[: s.n] = np.arange(s.n)
    for key in keys:   # the three shapes overlap each other: never a pair
        a, b = int(key >> np.uint64(32)), int(key & np.uint64(0xFFFFFFFF))
        assert body[a] != body[b]
    for _ in range(400):
        w.step(DT)
    pos, rot = w.poses()
    assert abs(pos[k, 1] - 0.2) < 5e-3 and abs(rot[k, 1]) < 1e-2, (pos[k], rot[k])
    w.close()


def test_a_further_collider_acts_like_the_bodys_own():
    """The same disc once as the body's own collider, once as a further collider of a body whose own slot is empty:
    the same fall, bounce and rest (one pair: the order of the solve cannot differ), bit for bit."""
    def run(as_further):
        b = SceneBuilder("one", materials=[(0.3, 0.5, 0.4)])
        b.static_rect((0.0, -0.5), 5.0, 0.5)
        if as_further:
            k = b.circle((0.1, 0.8), 0.2, vel=(0.5, 0.0), local_pos=(0.1, 0.0))
            b.col[k] = b._col(SHAPE_NONE, (0, 0, 0))
            b.add_collider(k, "circle", (0.2,), (0.1, 0.0), 0.0)
        else:
            k = b.circle((0.1, 0.8), 0.2, vel=(0.5, 0.0), local_pos=(0.1, 0.0))
        w = OracleWorld.from_scene(b.build())
        for _ in range(120):
            w.step(DT)
        out = w.poses(), w.velocities(), w.stats().n_fallback
        w.close()
        return out
    (p0, r0), (v0, a0), f0 = run(False)
    (p1, r1), (v1, a1), f1 = run(True)
    assert f0 == 0 and f1 == 1      # the further collider of a dynamic body is solved on the fallback list
    assert np.array_equal(p0, p1) and np.array_equal(r0, r1) and np.array_equal(v0, v1) and np.array_equal(a0, a1)


def test_static_compound_is_level_geometry_in_the_tiles():
    """Floor and wall as two colliders of ONE static body: a disc comes to rest in the corner, touching both; its
    pairs are ordinary tile pairs (further colliders of a non-dynamic body cost what any static collider costs)."""
    b = SceneBuilder("level", materials=[(0.0, 0.5, 0.4)])
    level = b.compound((0.0, 0.0), [("rect", (3.0, 0.5), (0.0, -0.5), 0.0), ("rect", (0.5, 2.0), (-3.5, 1.0), 0.0)], density=0.0)
    ramp = b.add_collider(level, "rect", (1.5, 0.05), (1.0, 0.6), 0.35)
    d = b.circle((1.5, 1.6), 0.15)
    s = b.build()
    assert s.inv_mass[level] == 0 and s.n_colliders == 4
    w = OracleWorld.from_scene(s)
    for _ in range(600):
        w.step(DT)
    pos, _ = w.poses()
    vel, _ = w.velocities()
    assert abs(pos[d, 0] - (-3.0 + 0.15)) < 5e-3 and abs(pos[d, 1] - 0.15) < 5e-3, pos[d]   # rolled down the ramp into the corner
    assert np.abs(vel[d]).max() < 1e-2
    w.broadphase(DT)
    keys, colors, tiles = w.pair_colors()
    assert len(keys) >= 2 and np.all(tiles != FALLBACK) and w.stats().n_fallback == 0
    w.close()


def test_arm_of_a_spinning_body_is_stopped_by_a_peg():
    """A hub pinned to the world with an arm (a small disc at 0.6 m, a further collider): the arm's tip meets a static
    peg and the spin stops there -- the contact acts on the body through the collider's lever arm."""
    b = SceneBuilder("arm", materials=[(0.0, 0.3, 0.2)])
    hub = b.circle((0.0, 0.0), 0.2, angvel=3.0)
    b.add_collider(hub, "circle", (0.08,), (0.6, 0.0), 0.0)
    b.attach(WORLD_BODY, hub, anchor_a=(0.0, 0.0))
    b.static_rect((0.0, 0.68), 0.05, 0.05)          # a peg above the hub, in the path of the tip
    s = b.build()
    w = OracleWorld.from_scene(s, gravity=(0.0, 0.0))
    seen = []
    for _ in range(120):
        w.step(DT)
        _, rot = w.poses()
        _, ang = w.velocities()
        seen.append((np.arctan2(rot[hub, 1], rot[hub, 0]), ang[hub]))
    seen = np.array(seen)
    assert seen[:, 0].max() < np.pi / 2            # the tip never passes the peg (it would at 90 degrees)
    assert seen[10, 1] > 2.9 and abs(seen[-1, 1]) < 0.6, seen[-1]   # free spin before, stopped (or slowly bouncing back) after
    assert np.hypot(*w.poses()[0][hub]) < 1e-3     # the pin holds
    w.close()


def test_queries_name_the_body_of_the_collider_hit():
    s, floor, k = dumbbell_scene(y=1.0)
    w = OracleWorld.from_scene(s)
    pts = np.array([[0.5, 1.0], [-0.5, 1.1], [0.0, 1.0], [0.0, 1.3]], np.float32)   # second disc, first disc, bar, nothing
    for brute in (True, False):
        body, count = w.query_points(pts, brute=brute)
        assert list(body) == [k, k, k, 0xFFFFFFFF] and list(count) == [2, 1, 1, 0]   # the bar passes through the discs (y = 1.1 is above it)
        hits = w.query_rays(np.array([[0.5, 3.0, 0.0, -1.0], [2.0, 1.0, -1.0, 0.0]], np.float32), 10.0, brute=brute)
        assert list(hits["body"]) == [k, k]
        assert abs(hits["t"][0] - 1.8) < 1e-5 and abs(hits["t"][1] - 1.3) < 1e-5   # top of the second disc; its right side
        boxes = w.query_aabbs(np.array([[-1.0, 0.7, 1.0, 1.3]], np.float32), brute=brute)
        assert list(boxes) == [k]                    # three colliders of one body: named once
    w.close()


@pytest.mark.parametrize("seed,n,extent", [(3, 250, None), (4, 250, None), (37, 2000, 9.0)])
def test_soup_of_compounds_three_pair_enumerations_agree_and_it_runs(seed, n, extent):
    """(250 bodies: the whole grid is one tile; 2000: many tiles, seams and corners)"""
    s = random_soup(n, seed=seed, offsets=0.3, compounds=0.3, **({} if extent is None else {"extent": extent}))
    assert s.n_colliders > s.n + 40
    w = OracleWorld.from_scene(s)
    w.broadphase(DT)
    g, sw, br = w.pairs(), w.pairs_sweep(), w.pairs_brute()
    assert len(g) > 100 and np.array_equal(g, sw) and np.array_equal(g, br)
    body = s.colliders["body"].copy()
    body[: s.n] = np.arange(s.n)
    a, b = (g >> np.uint64(32)).astype(np.int64), (g & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.all(body[a] != body[b])
    keys, colors, tiles = w.pair_colors()
    ka, kb = (keys >> np.uint64(32)).astype(np.int64), (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    further_dyn = lambda e: (e >= s.n) & dyn[body[e]]
    # pairs of a further collider of a dynamic body: in a tile's list when the collider sits beside its body's first one
    # (oracle/sfx_oracle.c tile_end_ok), else on the fallback list -- both happen in a soup
    cmp_ = further_dyn(ka) | further_dyn(kb)
    in_tiles = cmp_ & (tiles != FALLBACK)
    assert cmp_.sum() > 20 and in_tiles.sum() > 10
    if n >= 2000:
        assert (cmp_ & (tiles == FALLBACK)).sum() > 50 and in_tiles.sum() > 8 * (cmp_ & (tiles == FALLBACK)).sum()
        assert len(np.unique(tiles[tiles != FALLBACK] >> 28)) >= 2   # (strips here: interior and x seam)
    # what the rule is for: the units (tile, class) of ONE class share no dynamic body -- they are solved side by side
    for cls in range(4):
        owner = {}
        sel = (tiles != FALLBACK) & ((tiles >> 28) == cls) & (colors < 0xFE)
        for x, y, u in zip(body[ka[sel]], body[kb[sel]], tiles[sel]):
            for v in (x, y):
                if dyn[v]:
                    assert owner.setdefault(int(v), int(u)) == int(u), (cls, v, u, owner[int(v)])
    # colours: two pairs of one unit that share a dynamic BODY never share a colour
    PRUNED = 0xFE
    for unit in np.unique(tiles):
        sel = (tiles == unit) & (colors < PRUNED)
        seen = set()
        for x, y, c in zip(body[ka[sel]], body[kb[sel]], colors[sel]):
            for v in (x, y):
                if dyn[v]:
                    assert (v, c) not in seen, (unit, v, c)
                    seen.add((v, c))
    for _ in range(60 if n < 2000 else 10):
        w.step(DT)
    pos, rot = w.poses()
    assert np.isfinite(pos).all() and np.isfinite(rot).all() and pos.shape[0] == s.n
    assert w.stats().overflow == 0
    w.close()
