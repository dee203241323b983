"""Committed golden vectors (tests/golden/*.npz, SELF-ORACLE: produced by this repository's
CPU oracle with tests/golden/make_golden.py; the reference itself was unavailable).

CPU: the oracle still reproduces them bit-for-bit.  GPU: libsfx matches them at the contract's
bars (pairs, colours bit-exact; contacts and one-tick poses within 1e-5)."""
import glob
import os

import numpy as np
import pytest

from starframe_amd import scenes
from starframe_amd.scenes import Scene
from tests.helpers import colors_by_key, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
DT = 1.0 / 60.0


def load(path):
    z = np.load(path, allow_pickle=False)
    s = Scene(os.path.basename(path), z["pos"], z["rot"], z["vel"], z["angvel"], z["inv_mass"], z["inv_inertia"],
              scenes.as_colliders(z["colliders"]), z["materials"], z["poly_verts"], z["constraints"], substeps=int(z["substeps"]))
    return s, z


def test_fixtures_present():
    assert len(FIXTURES) >= 6


@pytest.mark.parametrize("path", FIXTURES, ids=os.path.basename)
def test_oracle_reproduces_golden(path):
    from oracle.oracle import ORDER_COLORED, ORDER_SEQUENTIAL, OracleWorld

    s, z = load(path)
    w = OracleWorld.from_scene(s)
    w.broadphase(DT)
    assert np.array_equal(w.aabbs().view(np.uint32), z["aabbs"].view(np.uint32))
    assert np.array_equal(w.pairs(), z["pairs"])
    k, c, t = colors_by_key(*w.pair_colors())
    assert np.array_equal(c, z["pair_colors"]) and np.array_equal(t, z["pair_tiles"])
    assert w.contacts().tobytes() == z["contacts"].tobytes()
    w.step(DT, ORDER_COLORED)
    p, r = w.poses()
    assert np.array_equal(p.view(np.uint32), z["pos_colored"].view(np.uint32))
    assert np.array_equal(r.view(np.uint32), z["rot_colored"].view(np.uint32))
    w2 = OracleWorld.from_scene(s)
    w2.step(DT, ORDER_SEQUENTIAL)
    assert np.array_equal(w2.poses()[0].view(np.uint32), z["pos_sequential"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=os.path.basename)
def test_gpu_matches_golden(path):
    from starframe_amd.world import World

    s, z = load(path)
    g = World.from_scene(s)
    g.broadphase(DT)
    assert np.array_equal(g.aabbs().view(np.uint32), z["aabbs"].view(np.uint32))
    assert np.array_equal(g.pairs(), z["pairs"])
    k, c, t = colors_by_key(*g.pair_colors())
    assert np.array_equal(c, z["pair_colors"]) and np.array_equal(t, z["pair_tiles"])
    gc, oc = g.contacts(), z["contacts"]
    assert np.array_equal(gc["count"], oc["count"])
    assert rel_err(gc["normal"], oc["normal"], floor=1.0) <= 1e-5
    assert rel_err(gc["depth"], oc["depth"], floor=1e-2) <= 1e-5
    if len(s.constraints):
        perm, col = g.joint_order()
        assert np.array_equal(perm, z["joint_perm"]) and np.array_equal(col, z["joint_colors"])
    g.step(DT)
    p, r = g.poses()
    v, a = g.velocities()
    assert rel_err(p, z["pos_colored"], floor=1.0) <= 1e-5
    assert rel_err(r, z["rot_colored"], floor=1.0) <= 1e-5
    assert rel_err(v, z["vel_colored"], floor=1.0) <= 1e-4
    for _ in range(9):
        g.step(DT)
    p10, _ = g.poses()
    # ten ticks: chaotic divergence is allowed to grow, but not beyond a millimetre here
    assert rel_err(p10, z["pos_colored10"], floor=1.0) <= 1e-3
