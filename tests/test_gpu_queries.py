"""GPU parity of the spatial queries (include/sfx.h "spatial queries"):
libsfx through the C ABI against the CPU oracle on identical inputs.  Body ids, containment counts
and box hits bit-exact; ray parameters and normals within 1e-5 relative (same operation order on
both sides, so 0 ulp is what is observed).  PARITY UNPINNED against starframe itself."""
import numpy as np
import pytest

from starframe_amd import scenes
from starframe_amd._abi import NO_BODY
from tests.test_oracle_queries import query_inputs

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0


def make_pair(scene, **kw):
    from oracle.oracle import OracleWorld
    from starframe_amd.world import World

    return World.from_scene(scene, **kw), OracleWorld.from_scene(scene, **kw)


def check_queries(g, o, pts, boxes, rays, max_t, masks=(0xFFFFFFFF, 2)):
    n_hit = 0
    for mask in masks:
        gb, gc = g.query_points(pts, mask)
        ob, oc = o.query_points(pts, mask)
        assert np.array_equal(gb, ob) and np.array_equal(gc, oc)
        assert np.array_equal(g.query_aabbs(boxes, mask), o.query_aabbs(boxes, mask))
        gh = g.query_rays(rays, max_t, mask)
        oh = o.query_rays(rays, max_t, mask)
        assert np.array_equal(gh["body"], oh["body"])
        hit = oh["body"] != NO_BODY
        assert np.allclose(gh["t"][hit], oh["t"][hit], rtol=1e-5, atol=1e-7)
        assert np.allclose(gh["normal"][hit], oh["normal"][hit], rtol=0, atol=1e-5)
        n_hit += int(hit.sum())
    return n_hit


@pytest.mark.parametrize("n,seed,extent,grid", [(300, 1, 3.0, (0, 0)), (2000, 2, 10.0, (0, 0)), (400, 3, 4.0, (3, 2))])
def test_queries_match_oracle_on_soups(n, seed, extent, grid):
    s = scenes.random_soup(n, seed=seed, extent=extent)
    g, o = make_pair(s, grid_log2=grid)
    pts, boxes, rays, max_t = query_inputs(seed, extent, n=600, near=s.pos)
    assert check_queries(g, o, pts, boxes, rays, max_t) > 100


def test_queries_follow_the_poses_after_ticks():
    """The index is rebuilt at the current poses: same answers as the oracle after both have ticked
    (colour order on both sides), and a query inside an open tick is refused."""
    from oracle.oracle import ORDER_COLORED
    from starframe_amd.world import SfxError

    s = scenes.random_soup(500, seed=5, extent=4.0)
    g, o = make_pair(s)
    pts, boxes, rays, max_t = query_inputs(5, 4.0, n=300, near=s.pos)
    check_queries(g, o, pts, boxes, rays, max_t, masks=(0xFFFFFFFF,))
    g.step(DT)
    o.step(DT, ORDER_COLORED)
    gp, gr = g.poses()
    op, orr = o.poses()
    if np.array_equal(gp, op) and np.array_equal(gr, orr):   # 0 ulp is the observed state; then the answers are equal too
        check_queries(g, o, pts, boxes, rays, max_t, masks=(0xFFFFFFFF,))
    else:                                                    # else put the oracle on the GPU's poses
        v, av = g.velocities()
        o.set_state(gp, gr, v, av, s.inv_mass, s.inv_inertia)
        check_queries(g, o, pts, boxes, rays, max_t, masks=(0xFFFFFFFF,))
    g.tick_begin(DT)
    with pytest.raises(SfxError):
        g.query_points(pts)
    for _ in range(s.substeps):
        g.tick_substep()
    g.query_points(pts)


def test_queries_on_a_pile_with_large_static_colliders():
    s = scenes.circles_container(n=3000, width=60, floor_hw=9.0, wall_hh=8.0)
    g, o = make_pair(s)
    ext = 9.0
    pts, boxes, rays, max_t = query_inputs(11, ext, n=500, near=s.pos)
    rays[:250, 1] = np.abs(rays[:250, 1]) + 12.0   # from above the lattice, downwards
    rays[:250, 2:] = (0.05, -1.0)
    assert check_queries(g, o, pts, boxes, rays, max_t) > 200
    st = g.stats()
    assert st.n_bodies == s.n


def test_box_query_capacity_is_reported():
    from starframe_amd._abi import SFX_ERR_CAPACITY, fptr, vptr
    import ctypes as C

    s = scenes.random_soup(300, seed=1, extent=3.0)
    g, o = make_pair(s)
    box = np.array([[-100, -100, 100, 100]], np.float32)
    full = o.query_aabbs(box)
    hits = np.empty(8, np.uint64)
    n = C.c_size_t()
    rc = g.lib.sfx_query_aabbs(g.handle, 1, fptr(box), 0xFFFFFFFF, vptr(hits), 8, C.byref(n))
    assert rc == SFX_ERR_CAPACITY and n.value == len(full)
    assert np.array_equal(g.query_aabbs(box), full)


def test_empty_and_degenerate_queries():
    from starframe_amd.scenes import SceneBuilder

    s = scenes.random_soup(50, seed=3, extent=2.0)
    g, o = make_pair(s)
    b, c = g.query_points(np.zeros((0, 2), np.float32))
    assert len(b) == 0 and len(c) == 0
    assert len(g.query_aabbs(np.zeros((0, 4), np.float32))) == 0
    assert len(g.query_rays(np.zeros((0, 4), np.float32), np.zeros(0, np.float32))) == 0
    # a world whose colliders are all large (one static box) and bodies without colliders
    bld = SceneBuilder("big", substeps=1)
    bld.static_rect((0.0, 0.0), 50.0, 1.0)
    bld.particle((0.0, 5.0))
    g2, o2 = make_pair(bld.build())
    rays = np.array([[0, 10, 0, -1], [0, 10, 0, 1], [100, 0.5, -1, 0], [0, 0.5, 1, 0]], np.float32)
    gh, oh = g2.query_rays(rays, np.inf), o2.query_rays(rays, np.inf)
    assert np.array_equal(gh["body"], oh["body"]) and np.array_equal(gh["t"], oh["t"])
    assert gh["body"].tolist() == [0, NO_BODY, 0, 0] and gh["t"][0] == 9.0 and gh["t"][3] == 0.0
    assert np.array_equal(g2.query_points([[0, 0], [0, 5]])[0], o2.query_points([[0, 0], [0, 5]])[0])


def test_device_pointer_box_and_ray_queries_equal_the_host_ones():
    """sfx_query_aabbs_device / sfx_query_rays_device: inputs and outputs in device memory (plain
    hipMalloc buffers here), enqueued on the world's stream without a synchronisation; the answers are those of the
    host-array calls -- box hits as a set (the device call neither sorts nor merges a body's colliders), ray hits record
    for record.  Includes a soup with compound bodies (hits name bodies, not colliders)."""
    from starframe_amd._abi import ray_hit_dtype
    from starframe_amd.world import World
    from tests.hipmem import Hip

    hip = Hip()
    for kw in (dict(), dict(compounds=0.3)):
        s = scenes.random_soup(1500, seed=21, extent=8.0, **kw)
        w = World.from_scene(s)
        w.step(DT)
        pts, boxes, rays, max_t = query_inputs(21, 8.0, n=700, near=w.poses()[0])
        d_boxes = hip.to_device(np.ascontiguousarray(boxes, np.float32))
        d_rays = hip.to_device(np.ascontiguousarray(rays, np.float32))
        d_mt = hip.to_device(np.ascontiguousarray(np.broadcast_to(np.asarray(max_t, np.float32), (len(rays),))))
        d_n = hip.malloc(4)
        for mask in (0xFFFFFFFF, 2):
            ref_hits = w.query_aabbs(boxes, mask)
            ref_rays = w.query_rays(rays, max_t, mask)
            cap = max(2 * len(ref_hits), 64)
            d_hits = hip.malloc(8 * cap)
            d_out = hip.malloc(len(rays) * ray_hit_dtype.itemsize)
            hip.fill(d_n, 0x5A, 4)
            w.query_aabbs_device(d_boxes, len(boxes), d_hits, cap, d_n, mask)
            w.query_rays_device(d_rays, d_mt, len(rays), d_out, mask)
            w.sync()   # the calls only enqueue
            n = int(hip.to_host(d_n, np.uint32, 1)[0])
            assert n <= cap and n >= len(ref_hits)
            got = np.unique(hip.to_host(d_hits, np.uint64, cap)[:n])
            assert np.array_equal(got, ref_hits)
            out = hip.to_host(d_out, np.uint8, len(rays) * ray_hit_dtype.itemsize).view(ray_hit_dtype)
            assert np.array_equal(out["body"], ref_rays["body"])
            assert np.array_equal(out["t"], ref_rays["t"]) and np.array_equal(out["normal"], ref_rays["normal"])
        # capacity: the full count comes back, the buffer holds `cap` of the hits
        small = hip.malloc(8 * 4)
        w.query_aabbs_device(d_boxes, len(boxes), small, 4, d_n)
        w.sync()
        assert int(hip.to_host(d_n, np.uint32, 1)[0]) > 4
        w.close()
        hip.free_all()
