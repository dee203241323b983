"""Known answers for the joint flavours beyond plain links: attachments with
offset anchors, angle limits / torsion springs, joint damping, the rope helper.  CPU oracle only; the GPU
parity of the same flavours is tests/test_gpu_parity.py::test_hinged_links_match_oracle."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as oracle_mod
from oracle.oracle import ORDER_COLORED, ORDER_SEQUENTIAL, OracleWorld
from starframe_amd._abi import WORLD_BODY
from starframe_amd.scenes import SceneBuilder

DT = 1.0 / 60.0


def world(builder, **kw):
    return OracleWorld.from_scene(builder.build(), **kw)


def angles(w):
    _, rot = w.poses()
    return np.arctan2(rot[:, 1], rot[:, 0])


def test_deterministic_atan2_is_accurate():
    lib = oracle_mod.load()
    f = lib.sfo_debug_atan2
    f.restype = C.c_float
    f.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(5)
    th = np.concatenate([rng.uniform(-np.pi, np.pi, 4000), np.linspace(-np.pi, np.pi, 721),
                         [0.0, np.pi / 8, np.pi / 4, 3 * np.pi / 8, np.pi / 2, 1e-6, -1e-6, np.pi - 1e-6]])
    r = np.concatenate([rng.uniform(0.1, 3.0, 4000), np.ones(721 + 8)])
    y, x = (r * np.sin(th)).astype(np.float32), (r * np.cos(th)).astype(np.float32)
    got = np.array([f(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    d = np.abs(got - want)
    d = np.minimum(d, 2 * np.pi - d)          # +pi and -pi are the same direction
    assert d.max() < 5e-7
    assert f(0.0, 0.0) == 0.0 and f(0.0, 1.0) == 0.0 and f(1.0, 0.0) == pytest.approx(np.pi / 2, rel=1e-7)
    assert f(0.0, -1.0) == pytest.approx(np.pi, rel=1e-7) and f(-1.0, 0.0) == pytest.approx(-np.pi / 2, rel=1e-7)


def test_angle_limit_first_step_closed_form():
    """One substep, no gravity: a body turned 0.5 rad, limit [-0.2, 0.2] against the world, compliance
    alpha: d_theta = -C iw / (iw + alpha / h^2), applied as rotate_by's first-order turn atan(d_theta)."""
    for alpha in (0.0, 1e-3):
        b = SceneBuilder("ang", substeps=1)
        b.rect((0.0, 0.0), 0.5, 0.1, angle=0.5)
        b.angle(WORLD_BODY, 0, -0.2, 0.2, compliance=alpha)
        s = b.build()
        w = OracleWorld.from_scene(s, gravity=(0.0, 0.0))
        w.step(DT, ORDER_SEQUENTIAL)
        iw = float(s.inv_inertia[0])
        dth = -(0.5 - 0.2) * iw / (iw + alpha / DT ** 2)
        assert angles(w)[0] == pytest.approx(0.5 + math.atan(dth), abs=2e-6)
        # inside the range nothing happens
        b = SceneBuilder("ang2", substeps=1)
        b.rect((0.0, 0.0), 0.5, 0.1, angle=0.1)
        b.angle(WORLD_BODY, 0, -0.2, 0.2, compliance=alpha)
        w = OracleWorld.from_scene(b.build(), gravity=(0.0, 0.0))
        w.step(DT, ORDER_SEQUENTIAL)
        _, rot = w.poses()
        assert rot[0, 0] == np.float32(math.cos(0.1)) and rot[0, 1] == np.float32(math.sin(0.1))


def test_angle_constraint_between_two_bodies_shares_the_turn_by_inverse_inertia():
    b = SceneBuilder("ang", substeps=1)
    b.rect((0.0, 0.0), 0.5, 0.1, angle=0.0)
    b.rect((3.0, 0.0), 1.0, 0.2, angle=0.1)       # heavier: turns less (small angle: rotate_by is first order)
    b.angle(0, 1, 0.0, 0.0)
    s = b.build()
    w = OracleWorld.from_scene(s, gravity=(0.0, 0.0))
    for _ in range(30):
        w.step(DT, ORDER_SEQUENTIAL)
    a = angles(w)
    assert a[1] - a[0] == pytest.approx(0.0, abs=1e-5)
    ia, ib = 1.0 / s.inv_inertia[0], 1.0 / s.inv_inertia[1]
    assert a[0] == pytest.approx(0.1 * ib / (ia + ib), rel=2e-2)   # angular momentum about the centres is kept
    pos, _ = w.poses()
    np.testing.assert_array_equal(pos, s.pos)                      # a pure torque pair: nothing translates


def hinged_bar(limit=None, damping=0.0, attach_damping=0.0, substeps=8):
    """A bar pinned to the world at its left end (offset anchor), starting horizontal."""
    b = SceneBuilder("hinge", substeps=substeps)
    b.rect((0.5, 0.0), 0.5, 0.05)
    b.attach(WORLD_BODY, 0, anchor_a=(0.0, 0.0), anchor_b=(-0.5, 0.0), damping=attach_damping)
    if limit is not None:
        b.angle(WORLD_BODY, 0, limit[0], limit[1], damping=damping)
    return b


def pin_error(w):
    pos, rot = w.poses()
    ax = pos[0, 0] + rot[0, 0] * -0.5
    ay = pos[0, 1] + rot[0, 1] * -0.5
    return float(np.hypot(ax, ay))


def test_attachment_with_offset_anchor_is_a_pendulum():
    w = world(hinged_bar())
    lo = 0.0
    for _ in range(150):
        w.step(DT)
        assert pin_error(w) < 2e-3
        lo = min(lo, float(angles(w)[0]))
    assert lo < -2.8           # swung through the bottom and up the other side: (almost) no energy lost


def test_hinge_stops_at_its_angle_limit():
    w = world(hinged_bar(limit=(-0.4, 0.3)))
    for _ in range(240):
        w.step(DT)
        assert angles(w)[0] > -0.4 - 5e-3
    assert angles(w)[0] == pytest.approx(-0.4, abs=5e-3)   # rests on the stop
    assert pin_error(w) < 2e-3
    assert abs(w.velocities()[1][0]) < 0.05


def test_hinge_damper_brings_the_pendulum_to_rest_at_the_bottom():
    free = world(hinged_bar(limit=(-3.0, 3.0)))
    damped = world(hinged_bar(limit=(-3.0, 3.0), damping=6.0))
    for _ in range(600):
        free.step(DT)
        damped.step(DT)
    assert abs(free.velocities()[1][0]) + abs(angles(free)[0] + np.pi / 2) > 0.2     # still swinging
    assert angles(damped)[0] == pytest.approx(-np.pi / 2, abs=2e-2)
    # (what is left is the one-iteration scheme's period-two flutter about the pin: 1e-4 rad per substep)
    assert abs(damped.velocities()[1][0]) < 0.1
    # damping at the pin (an attachment damps the whole relative velocity of the anchors) does nothing to a
    # pin that holds: the anchors do not move against each other
    pin = world(hinged_bar(attach_damping=50.0))
    ref = world(hinged_bar())
    for _ in range(60):
        pin.step(DT)
        ref.step(DT)
    assert angles(pin)[0] == pytest.approx(angles(ref)[0], abs=2e-2)


def test_link_damping_takes_the_stated_fraction_of_the_length_rate():
    """Two free particles on a very soft link flying apart: after one substep the rate of change of the link's
    length is (1 - min(c h, 1)) of what it was, the sideways relative velocity and the momentum are kept."""
    for c, frac in ((6.0, 6.0 * DT), (1000.0, 1.0)):
        b = SceneBuilder("damp", substeps=1)
        b.particle((0.0, 0.0), mass=1.0, vel=(-1.0, 0.5))
        b.particle((1.0, 0.0), mass=3.0, vel=(2.0, -0.25))
        b.distance(0, 1, rest=1.0, compliance=1e9, damping=c)
        w = OracleWorld.from_scene(b.build(), gravity=(0.0, 0.0))
        w.step(DT, ORDER_SEQUENTIAL)
        pos, _ = w.poses()
        vel, _ = w.velocities()
        u = (pos[1] - pos[0]) / np.linalg.norm(pos[1] - pos[0])
        rate = float((vel[1] - vel[0]) @ u)
        rate0 = float(np.array([3.0, -0.75]) @ u)
        assert rate == pytest.approx((1.0 - frac) * rate0, abs=2e-5)
        t = np.array([-u[1], u[0]])
        assert float((vel[1] - vel[0]) @ t) == pytest.approx(float(np.array([3.0, -0.75]) @ t), abs=2e-5)
        np.testing.assert_allclose(1.0 * vel[0] + 3.0 * vel[1], (-1.0 + 6.0, 0.5 - 0.75), atol=1e-5)


def test_attachment_damping_removes_all_relative_anchor_velocity():
    b = SceneBuilder("damp", substeps=1)
    b.particle((0.0, 0.0), mass=1.0, vel=(-1.0, 0.5))
    b.particle((0.0, 0.0), mass=1.0, vel=(2.0, -0.25))
    b.attach(0, 1, compliance=1e9, damping=1000.0)
    w = OracleWorld.from_scene(b.build(), gravity=(0.0, 0.0))
    w.step(DT, ORDER_SEQUENTIAL)
    vel, _ = w.velocities()
    np.testing.assert_allclose(vel[0], vel[1], atol=1e-6)
    np.testing.assert_allclose(vel[0], (0.5, 0.125), atol=1e-6)


def test_rope_helper_hangs_between_its_pins():
    b = SceneBuilder("rope", substeps=8)
    ids = b.rope((0.0, 0.0), (1.6, 0.0), links=20, pin_end=True, slack=0.25)
    b.static_rect((1.0, -3.0), 5.0, 0.2)
    s = b.build()
    assert len(ids) == 20 and len(s.constraints) == 21 and (s.constraints["limit"] == 1).all()
    assert len(set(s.colliders["group"][ids])) == 1 and s.colliders["group"][ids[0]] != 0
    w = OracleWorld.from_scene(s)
    for _ in range(300):
        w.step(DT)
    pos, _ = w.poses()
    p = pos[ids]
    seg = np.linalg.norm(np.diff(p, axis=0), axis=1)
    assert seg.max() < 0.1 * 1.01 and np.isclose(s.constraints["rest_length"][0], 0.1)                    # no link longer than its rest length (+1 %)
    assert p[:, 1].min() < -0.3 and p[:, 1].argmin() in range(6, 14)   # sags in the middle
    assert np.hypot(*(p[0] - (0.0, 0.0))) < 0.0505 * 1.01 and np.hypot(*(p[-1] - (1.6, 0.0))) < 0.0505 * 1.01


def test_constraint_validation():
    b = SceneBuilder("bad", substeps=1)
    b.circle((0.0, 0.0), 0.1)
    b.circle((1.0, 0.0), 0.1)
    for bad in (dict(limit=1, damping=1.0), dict(damping=-1.0), dict(limit=3)):
        bb = SceneBuilder("bad", substeps=1)
        bb.circle((0.0, 0.0), 0.1)
        bb.circle((1.0, 0.0), 0.1)
        bb.distance(0, 1, **bad)
        with pytest.raises(Exception):
            OracleWorld.from_scene(bb.build())
    for lo, hi in ((0.5, 0.2), (-4.0, 0.0), (0.0, 3.2)):
        bb = SceneBuilder("bad", substeps=1)
        bb.circle((0.0, 0.0), 0.1)
        bb.angle(WORLD_BODY, 0, lo, hi)
        with pytest.raises(Exception):
            OracleWorld.from_scene(bb.build())


def test_joint_list_is_distance_then_angle_and_coloured_together():
    b = SceneBuilder("mix", substeps=1)
    for k in range(6):
        b.rect((float(k), 0.0), 0.4, 0.1)
    for k in range(5):
        b.attach(k, k + 1, anchor_a=(0.5, 0.0), anchor_b=(-0.5, 0.0))
    for k in range(5):
        b.angle(k, k + 1, -0.3, 0.3)
    s = b.build()
    w = OracleWorld.from_scene(s)
    perm, col = w.joint_order()
    assert sorted(perm.tolist()) == list(range(10))
    ends = [(int(s.constraints["body_a"][k]), int(s.constraints["body_b"][k])) if k < 5 else
            (int(s.angle_constraints["body_a"][k - 5]), int(s.angle_constraints["body_b"][k - 5])) for k in range(10)]
    for c in set(col.tolist()):            # no two joints of one colour share a body
        used = [x for k, cc in zip(perm, col) if cc == c for x in ends[k]]
        assert len(used) == len(set(used))
    assert (np.diff(col.astype(int)) >= 0).all()
