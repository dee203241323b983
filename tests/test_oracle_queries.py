"""CPU tests of the oracle's spatial queries (DESIGN.md "Contract / spatial queries"):
closed-form known answers per shape, and the grid enumeration against brute force, exactly."""
import numpy as np
import pytest

from oracle.oracle import OracleWorld
from starframe_amd import scenes
from starframe_amd._abi import NO_BODY
from starframe_amd.scenes import SceneBuilder


def one_shape_world(add):
    b = SceneBuilder("q", substeps=1)
    add(b)
    return OracleWorld.from_scene(b.build())


def query_inputs(seed, extent, n=400, near=None):
    """Random points, boxes and rays over the scene; half of the points lie close to body centres."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-extent - 1.5, extent + 1.5, (n, 2)).astype(np.float32)
    if near is not None:
        k = rng.integers(0, len(near), n // 2)
        pts[: n // 2] = near[k] + rng.normal(0, 0.08, (n // 2, 2))
    lo = rng.uniform(-extent - 1.5, extent + 1.5, (n, 2))
    size = rng.choice([0.05, 0.5, 3.0], (n, 1)) * rng.uniform(0.2, 1.0, (n, 2))
    boxes = np.concatenate([lo, lo + size], 1).astype(np.float32)
    boxes[0] = (-1e9, -1e9, 1e9, 1e9)      # everything
    boxes[1] = (1.0, 1.0, 0.5, 0.5)        # inverted: empty
    org = rng.uniform(-extent - 2.0, extent + 2.0, (n, 2))
    ang = rng.uniform(0, 2 * np.pi, n)
    d = np.stack([np.cos(ang), np.sin(ang)], 1) * rng.choice([0.3, 1.0, 7.0], (n, 1))
    d[2] = (1.0, 0.0); d[3] = (0.0, -2.0); d[4] = (0.0, 0.0); d[5] = (-1.0, 0.0)   # axis-aligned and degenerate
    rays = np.concatenate([org, d], 1).astype(np.float32)
    max_t = rng.choice([0.5, 3.0, 50.0, np.inf], n).astype(np.float32)
    max_t[6] = 0.0
    max_t[7] = -1.0
    return pts, boxes, rays, max_t


# ---------------------------------------------------------------- known answers
def test_ray_circle_known_answer():
    w = one_shape_world(lambda b: b.circle((2.0, 0.0), 0.5))
    h = w.query_rays([[0, 0, 1, 0], [0, 0, 2, 0], [0, 1, 1, 0], [0, 0, -1, 0], [2.1, 0.1, 1, 0]], np.inf)
    assert h["body"].tolist() == [0, 0, NO_BODY, NO_BODY, 0]
    assert h["t"][0] == pytest.approx(1.5, rel=1e-6)
    assert h["t"][1] == pytest.approx(0.75, rel=1e-6)       # t is in units of the direction
    np.testing.assert_allclose(h["normal"][0], (-1.0, 0.0), atol=1e-6)
    assert h["t"][4] == 0.0 and h["normal"][4].tolist() == [0.0, 0.0]   # origin inside
    # max_t cuts the ray off
    assert w.query_rays([[0, 0, 1, 0]], 1.4)["body"][0] == NO_BODY
    assert w.query_rays([[0, 0, 1, 0]], 1.6)["body"][0] == 0


def test_ray_box_faces_and_rounded_corner():
    ang = 0.0
    w = one_shape_world(lambda b: b.rect((0.0, 0.0), 1.0, 0.5, rounding=0.25, angle=ang))
    h = w.query_rays([[-5, 0, 1, 0], [0, 5, 0, -1], [-5, 0.75 - 1e-3, 1, 0]], np.inf)
    assert h["body"].tolist() == [0, 0, 0]
    assert h["t"][0] == pytest.approx(5 - 1.25, rel=1e-6)
    np.testing.assert_allclose(h["normal"][0], (-1, 0), atol=1e-6)
    assert h["t"][1] == pytest.approx(5 - 0.75, rel=1e-6)
    np.testing.assert_allclose(h["normal"][1], (0, 1), atol=1e-6)
    # grazing the top: first met by the rounded corner at (-1, 0.5), radius 0.25
    y = 0.75 - 1e-3 - 0.5  # height above the corner centre
    x_hit = -1.0 - np.sqrt(0.25 ** 2 - y ** 2)
    assert h["t"][2] == pytest.approx(x_hit + 5.0, rel=1e-4)
    n = h["normal"][2]
    assert n[0] < 0 and n[1] > 0 and np.hypot(*n) == pytest.approx(1.0, rel=1e-5)


def test_ray_capsule_and_rotated_poly():
    w = one_shape_world(lambda b: b.capsule((0.0, 0.0), 1.0, 0.25, angle=np.pi / 2))  # vertical capsule
    h = w.query_rays([[-3, 0.5, 1, 0], [0, 3, 0, -1]], np.inf)
    assert h["t"][0] == pytest.approx(3 - 0.25, rel=1e-6)
    assert h["t"][1] == pytest.approx(3 - 1.25, rel=1e-6)
    np.testing.assert_allclose(h["normal"][1], (0, 1), atol=1e-6)
    tri = [(-1.0, -1.0), (1.0, -1.0), (0.0, 2.0)]  # centroid at the origin
    w = one_shape_world(lambda b: b.poly((0.0, 0.0), tri))
    h = w.query_rays([[0, -5, 0, 1], [0, 5, 0, -1], [3, -2, 0, 1]], np.inf)
    assert h["t"][0] == pytest.approx(4.0, rel=1e-6)
    assert h["t"][1] == pytest.approx(3.0, rel=1e-5)
    assert h["body"][2] == NO_BODY


def test_point_containment_known_answers():
    b = SceneBuilder("q", substeps=1)
    b.circle((0.0, 0.0), 0.5)                                    # 0
    b.rect((3.0, 0.0), 1.0, 0.5, rounding=0.1, angle=np.pi / 2)   # 1: 1.2 high... turned: x half 0.6, y half 1.1
    b.capsule((6.0, 0.0), 1.0, 0.25)                              # 2
    b.circle((0.2, 0.0), 0.5, category=2)                         # 3 overlaps 0, other layer
    w = OracleWorld.from_scene(b.build())
    pts = [[0.0, 0.49], [0.0, 0.51], [3.55, 1.05], [3.59, 1.09], [3.0, 1.09], [7.2, 0.0], [7.26, 0.0], [0.3, 0.0]]
    body, count = w.query_points(pts)
    assert body.tolist() == [0, NO_BODY, 1, NO_BODY, 1, 2, NO_BODY, 0]
    assert count.tolist() == [1, 0, 1, 0, 1, 1, 0, 2]
    body, count = w.query_points(pts, mask=2)
    assert body[7] == 3 and count[7] == 1 and body[0] == NO_BODY and body[2] == NO_BODY


def test_aabb_query_known_answers():
    b = SceneBuilder("q", substeps=1)
    b.circle((0.0, 0.0), 0.5)
    b.circle((2.0, 0.0), 0.5)
    b.static_rect((0.0, -3.0), 10.0, 0.5)
    w = OracleWorld.from_scene(b.build())
    hits = w.query_aabbs([[-0.1, -0.1, 0.1, 0.1], [0.5, 0.0, 1.5, 0.1], [0.6, 0.0, 1.4, 0.1], [-20, -20, 20, 20]])
    assert hits.tolist() == [(0 << 32) | 0, (1 << 32) | 0, (1 << 32) | 1, (3 << 32) | 0, (3 << 32) | 1, (3 << 32) | 2]


# ---------------------------------------------------------------- grid == brute force
@pytest.mark.parametrize("seed,n,extent,grid", [(1, 300, 3.0, (0, 0)), (2, 500, 6.0, (0, 0)), (3, 400, 4.0, (3, 2)),
                                                (4, 200, 40.0, (4, 4))])
def test_grid_queries_equal_brute_force(seed, n, extent, grid):
    s = scenes.random_soup(n, seed=seed, extent=extent)
    w = OracleWorld.from_scene(s, grid_log2=grid)
    pts, boxes, rays, max_t = query_inputs(seed, extent, near=s.pos)
    for mask in (2, 0xFFFFFFFF):
        bg, cg = w.query_points(pts, mask)
        bb, cb = w.query_points(pts, mask, brute=True)
        assert np.array_equal(bg, bb) and np.array_equal(cg, cb)
        assert np.array_equal(w.query_aabbs(boxes, mask), w.query_aabbs(boxes, mask, brute=True))
        hg = w.query_rays(rays, max_t, mask)
        hb = w.query_rays(rays, max_t, mask, brute=True)
        assert np.array_equal(hg["body"], hb["body"])
        assert np.array_equal(hg["t"], hb["t"]) and np.array_equal(hg["normal"], hb["normal"])
    assert (bg != NO_BODY).sum() > 50 and (hg["body"] != NO_BODY).sum() > 50


def test_queries_on_settled_pile_hit_what_a_dense_sampling_sees():
    """Ray hits are consistent with point containment: just before the hit the ray is outside every
    shape, just after it is inside the body hit."""
    s = scenes.random_soup(300, seed=7, extent=3.0, with_statics=False)
    w = OracleWorld.from_scene(s)
    _, _, rays, _ = query_inputs(7, 3.0, n=200)
    rays = rays[8:]
    nrm = np.hypot(rays[:, 2], rays[:, 3])[:, None]
    rays[:, 2:] /= nrm
    h = w.query_rays(rays, np.inf)
    hit = (h["body"] != NO_BODY) & (h["t"] > 1e-3)
    assert hit.sum() > 30
    o, d, t = rays[hit, :2], rays[hit, 2:], h["t"][hit, None]
    before, _ = w.query_points(o + (t - 1e-3) * d)
    assert (before == NO_BODY).all()
    _, cnt_after = w.query_points(o + (t + 1e-3) * d)
    assert (cnt_after >= 1).mean() > 0.9   # grazing hits may leave again within 1 mm
    # the hit body contains the point just inside along the inward normal
    inside, _ = w.query_points(o + t * d - 1e-3 * h["normal"][hit])
    _, cnt = w.query_points(o + t * d - 1e-3 * h["normal"][hit])
    assert (cnt >= 1).all()
