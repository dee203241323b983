"""GPU tests of the host-side mirrors of the reference's physics API (config / body / collider / material /
tick / queries): the Python one (starframe_amd/physics.py) and the C++ one (starframe_amd/cpp), both over the
C ABI.  Checked against closed-form answers (the oracle parity of the same calls is in the other GPU tests)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_physics_api_tick_and_queries():
    from starframe_amd.physics import Body, Collider, Material, Physics, PhysicsConfig

    ph = Physics(PhysicsConfig(substeps=8))
    ground = ph.add(Body.static(), Collider.rect(40.0, 1.0), (0.0, -0.5))
    box = ph.add(Body.dynamic(), Collider.rect(1.0, 1.0, material=Material(0.0, 0.6, 0.5)), (0.0, 0.52))
    ball = ph.add(Body.dynamic(), Collider.circle(0.25), (3.0, 0.27))
    bob = ph.add(Body.dynamic(), Collider.circle(0.1), (6.0, 3.0))
    ph.add_distance_constraint(None, bob, anchor_a=(6.0, 4.0), rest_length=1.0, limit=1)   # a rope from a world point
    for _ in range(120):
        ph.tick(1.0 / 60.0)
    pos, ang = ph.poses()
    assert abs(pos[box, 1] - 0.5) < 5e-3 and abs(pos[ball, 1] - 0.25) < 5e-3       # resting on the ground
    assert abs(np.hypot(pos[bob, 0] - 6.0, pos[bob, 1] - 4.0) - 1.0) < 5e-3        # hanging on the taut rope
    assert ph.query_point((0.0, 0.5)) == box and ph.query_point((3.0, 0.3)) == ball
    assert ph.query_point((1.5, 2.0)) is None and ph.query_point((10.0, -0.5)) == ground
    assert sorted(ph.query_aabb((-1.0, 0.1), (3.5, 0.2))) == [box, ball]
    hit = ph.raycast((0.0, 5.0), (0.0, -1.0))
    assert hit[0] == box and abs(hit[1] - (5.0 - pos[box, 1] - 0.5)) < 1e-3 and hit[2][1] > 0.99
    assert ph.raycast((-10.0, 10.0), (1.0, 0.0), max_t=100.0) is None
    assert ph.raycast((-10.0, 0.25), (1.0, 0.0))[0] == box                          # the box is in front of the ball


def test_cpp_host_api_example_builds_and_runs(tmp_path):
    """starframe_amd/cpp/example_pyramid.cpp over starframe_physics.hpp: a 20-row pyramid stands, a ray from above
    finds its apex, a point query finds the ground (the program's exit status)."""
    exe = str(tmp_path / "example_pyramid")
    csrc = os.path.join(ROOT, "starframe_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "starframe_amd", "cpp", "example_pyramid.cpp"),
                           "-L", csrc, "-lsfx", "-Wl,-rpath," + csrc, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ray from above hits body" in out.stdout


def test_plain_c_driver_of_one_tick(tmp_path):
    """tools/sfx_c_tick.c, compiled as C11 (no C++): upload, 180 ticks, download -- a ball rests on the floor, a bob
    hangs on its rope.  What a Rust / Go / Java FFI sees of include/sfx.h, struct sizes asserted at compile time."""
    exe = str(tmp_path / "sfx_c_tick")
    csrc = os.path.join(ROOT, "starframe_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "sfx_c_tick.c"), "-L", csrc, "-lsfx", "-lm", "-Wl,-rpath," + csrc, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr


def test_native_rccl_slab_driver_send_to_self(tmp_path):
    """tools/sfx_slab_run --self (C++, no torch, include/sfx_rccl.h): two logical slabs of a settled circle pile on this
    GPU exchange their halos through ncclSend / ncclRecv from rank 0 to itself every substep; the final owned states equal,
    bit for bit, the same slabs exchanging through sfx_halo_copy.  Then the one-process-per-GPU path with one rank."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bench
    from dump_scene import dump
    from starframe_amd import scenes

    s = bench.settle(scenes.circles_container(n=3000, width=150, floor_hw=20.0, wall_hh=8.0), 0, frames=300)
    scene_file = str(tmp_path / "circles_settled.bin")
    dump(s, scene_file)
    exe = os.path.join(ROOT, "tools", "sfx_slab_run")
    # always through make (a no-op when the binary is newer than its sources, the headers and the two libraries): a binary
    # left over from before an ABI change must not run against the new library (it also checks sfx_abi_version itself)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "sfx_slab_run"])
    out = subprocess.run([exe, scene_file, "30", "--self"], capture_output=True, text=True, timeout=200)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "self test ok" in out.stdout and "rccl send-to-self" in out.stdout
    out = subprocess.run([exe, scene_file, "30", "1"], capture_output=True, text=True, timeout=200)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ranks 1 bodies 3000" in out.stdout


def test_pose_readback_overlaps_the_next_tick():
    """sfx_readback_begin / _wait (include/sfx.h): the snapshot is the state at begin, whatever is enqueued after
    it; two may be in flight, a third is refused; a view stays valid until the second begin after its own."""
    from starframe_amd import scenes
    from starframe_amd.world import World

    s = scenes.random_soup(3000, seed=11, extent=8.0)
    w = World.from_scene(s)
    for _ in range(3):
        w.step(1.0 / 60.0)
    pos0, rot0 = w.poses()
    w.readback_begin()
    for _ in range(2):
        w.step(1.0 / 60.0)                       # overwrites the poses while the copy travels
    pos2, rot2 = w.poses()
    w.readback_begin()
    with pytest.raises(Exception):
        w.readback_begin()                       # two in flight already
    a = w.readback_wait(copy=False)
    assert a.shape == (s.n, 4)
    assert np.array_equal(a[:, :2], pos0) and np.array_equal(a[:, 2:], rot0)
    b = w.readback_wait(copy=False)
    assert np.array_equal(b[:, :2], pos2) and np.array_equal(b[:, 2:], rot2)
    assert not np.array_equal(pos0, pos2)
    assert np.array_equal(a[:, :2], pos0)        # both views are live: the first one's slot is taken by the third begin
    with pytest.raises(Exception):
        w.readback_wait()                        # nothing begun
    w.readback_begin()                           # the third begin reuses the first slot, not the second
    w.readback_wait()
    assert np.array_equal(b[:, :2], pos2)
    # a frame loop: every frame's read-back equals the synchronous download taken at the same point
    for _ in range(5):
        want = np.concatenate(w.poses(), axis=1)
        w.readback_begin()
        w.step(1.0 / 60.0)
        assert np.array_equal(w.readback_wait(), want)
    w.close()


def test_python_physics_api_joints_rope_and_frame_loop():
    from starframe_amd.physics import Body, Collider, Physics, PhysicsConfig

    ph = Physics(PhysicsConfig(substeps=8))
    ph.add(Body.static(), Collider.rect(40.0, 1.0), (0.0, -3.0))
    bar = ph.add(Body.dynamic(), Collider.rect(1.0, 0.1), (0.5, 2.0))
    ph.add_attachment(None, bar, anchor_a=(0.0, 2.0), anchor_b=(-0.5, 0.0))
    ph.add_angle_constraint(None, bar, -0.4, 0.3, damping=2.0)
    rope = ph.add_rope((3.0, 2.0), (5.0, 2.0), links=16, pin_end=True, slack=0.25)
    last = None
    for _ in range(240):
        last = ph.tick_and_read(1.0 / 60.0)
    pos, ang = ph.poses()
    assert abs(ang[bar] + 0.4) < 1e-2                                              # the bar rests on its lower stop
    assert np.hypot(pos[bar, 0] - 0.5 * np.cos(ang[bar]), pos[bar, 1] - 2.0 - 0.5 * np.sin(ang[bar])) < 5e-3   # pinned
    seg = np.linalg.norm(np.diff(pos[rope], axis=0), axis=1)
    assert seg.max() < 0.125 * 1.25 * 1.01 and pos[rope, 1].min() < 1.6                    # taut links, hanging below the pins
    assert last.shape == (len(pos), 4) and np.abs(last[:, :2] - pos).max() < 0.05   # one frame behind


def test_kernel_times_of_all_launches_and_of_chosen_classes():
    """sfx_set_profiling: 1 brackets every launch (collected per tick), 2 * mask only the chosen classes (collected
    on demand, nothing synchronises in between); either way the tick's results are the same."""
    from starframe_amd import scenes
    from starframe_amd.world import World

    s = scenes.mixed_pile(n=3000, width=100)
    plain, prof = World.from_scene(s), World.from_scene(s)
    for _ in range(5):
        plain.step(1.0 / 60.0)
    prof.set_profiling(True)
    for _ in range(2):
        prof.step(1.0 / 60.0)
    kt = prof.kernel_times(reset=True)
    # (the narrowphase runs inside the position sweep: no k_narrow launch unless SFX_FUSE_NARROW=0 asks for it)
    assert kt["solve_vel"]["launches"] == 2 * s.substeps and kt["solve_pos"]["launches"] == 2 * s.substeps
    assert kt["narrow"]["launches"] == (2 * s.substeps if os.environ.get("SFX_FUSE_NARROW") == "0" else 0)
    assert kt["pair_count"]["launches"] == 2 and kt["solve_vel"]["total_ms"] > 0.0
    prof.set_profiling_classes(["solve_vel", "integrate"])
    for _ in range(3):
        prof.step(1.0 / 60.0)
    kt = prof.kernel_times(reset=True)
    # (integration on load, libsfx's default for such a world: no k_integrate -- the class holds one k_tick_free per tick)
    on_load = os.environ.get("SFX_FUSE_TICK", "3") == "3"
    assert kt["solve_vel"]["launches"] == 3 * s.substeps and kt["integrate"]["launches"] == (3 if on_load else 3 * s.substeps)
    assert kt["solve_pos"]["launches"] == 0 and kt["pair_count"]["launches"] == 0
    assert 0.0 < kt["solve_vel"]["total_ms"] < 50.0
    prof.set_profiling(False)
    assert np.array_equal(plain.poses()[0].view(np.uint32), prof.poses()[0].view(np.uint32))
    plain.close()
    prof.close()
