"""The entry point that allocates nothing (include/sfx.h sfx_query_points_static) against the existing
sfx_query_points: body ids and containment counts bit-exact, on two calls in a row -- the second on a smaller
world with fewer points, so that anything the first call left in the static scratch would show.

Device memory for the inputs and outputs comes from the HIP runtime directly (ctypes on libamdhip64): the
entry point takes device pointers and must not see an allocation of its own."""
import numpy as np
import pytest

from starframe_amd import scenes
from starframe_amd._abi import SFX_ERR_INVALID, SFX_ERR_UNSUPPORTED
from tests.hipmem import Hip
from tests.test_oracle_queries import query_inputs

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0


def static_query(hip, w, pts, mask=0xFFFFFFFF, with_count=True):
    n = len(pts)
    d_pts = hip.to_device(np.ascontiguousarray(pts, np.float32))
    d_body, d_count = hip.malloc(4 * n), hip.malloc(4 * n)
    hip.fill(d_body, 0xEE, 4 * n)
    hip.fill(d_count, 0xEE, 4 * n)
    # the property the entry point exists for: the call allocates nothing -- the device's free memory is the
    # same before and after it (a lazily allocated buffer on this path would show here)
    before = hip.free_bytes()
    w.query_points_static(d_pts, n, d_body, d_count if with_count else 0, mask)
    assert hip.free_bytes() == before, "sfx_query_points_static changed the device's free memory: it allocated"
    w.sync()   # the call only enqueues on the world's stream
    return hip.to_host(d_body, np.uint32, n), hip.to_host(d_count, np.uint32, n)


def test_static_point_query_equals_the_allocating_one_twice_in_a_row():
    from starframe_amd.world import World

    hip = Hip()
    big = scenes.random_soup(3000, seed=3, extent=14.0)
    small = scenes.random_soup(400, seed=7, extent=4.0)
    wb, ws = World.from_scene(big), World.from_scene(small)
    for _ in range(3):
        wb.step(DT)
        ws.step(DT)
    pts_big, _, _, _ = query_inputs(3, 14.0, n=5000, near=wb.poses()[0])
    pts_small, _, _, _ = query_inputs(7, 4.0, n=700, near=ws.poses()[0])
    # first call: the larger world, the larger input
    body, count = static_query(hip, wb, pts_big)
    ref_body, ref_count = wb.query_points(pts_big)
    assert (ref_count > 0).sum() > 200
    assert np.array_equal(body, ref_body) and np.array_equal(count, ref_count)
    # second call straight after it: a smaller world (fewer bodies, fewer grid cells) and fewer points
    body, count = static_query(hip, ws, pts_small)
    ref_body, ref_count = ws.query_points(pts_small)
    assert (ref_count > 0).sum() > 50
    assert np.array_equal(body, ref_body) and np.array_equal(count, ref_count)
    # and the large one again, with a layer mask and without the count output
    body, count = static_query(hip, wb, pts_big[:1234], mask=2, with_count=False)
    ref_body, _ = wb.query_points(pts_big[:1234], 2)
    assert np.array_equal(body, ref_body)
    assert (count == 0xEEEEEEEE).all(), "count_dev == NULL must leave nothing written"
    # the poses move on: the index follows (it is rebuilt by every call)
    wb.step(DT)
    body, count = static_query(hip, wb, pts_big)
    ref_body, ref_count = wb.query_points(pts_big)
    assert np.array_equal(body, ref_body) and np.array_equal(count, ref_count)
    hip.free_all()
    wb.close()
    ws.close()


def test_static_point_query_rejects_what_does_not_fit():
    from starframe_amd.world import SfxError, World

    hip = Hip()
    d = hip.malloc(64)
    s = scenes.random_soup(300, seed=1, extent=3.0)
    w = World.from_scene(s, grid_log2=(10, 10))       # 2^20 cells: more than the scratch holds
    with pytest.raises(SfxError) as e:
        w.query_points_static(d, 4, d, 0)
    assert e.value.code == SFX_ERR_UNSUPPORTED and "static scratch" in str(e.value)
    w.close()
    w = World.from_scene(s)
    w.set_profiling(True)
    with pytest.raises(SfxError) as e:
        w.query_points_static(d, 4, d, 0)
    assert e.value.code == SFX_ERR_INVALID
    w.set_profiling(False)
    w.tick_begin(DT)
    with pytest.raises(SfxError):
        w.query_points_static(d, 4, d, 0)             # not inside an open tick
    for _ in range(s.substeps):
        w.tick_substep()
    w.query_points_static(d, 0, 0, 0)                 # n = 0: nothing to do, no pointers needed
    w.sync()
    hip.free_all()
    w.close()
