#!/usr/bin/env python3
"""Generates tests/golden/*.npz: SELF-ORACLE fixtures (inputs + the CPU oracle's outputs).

These are NOT starframe outputs -- the reference was unavailable (Oracle).  They
pin this repository's oracle against regressions and give the GPU tests committed vectors.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.oracle import ORDER_COLORED, ORDER_SEQUENTIAL, OracleWorld  # noqa: E402
from starframe_amd import scenes  # noqa: E402

DT = 1.0 / 60.0
HERE = os.path.dirname(os.path.abspath(__file__))


def scene_arrays(s):
    return dict(pos=s.pos, rot=s.rot, vel=s.vel, angvel=s.angvel, inv_mass=s.inv_mass, inv_inertia=s.inv_inertia,
                colliders=s.colliders, materials=s.materials, poly_verts=s.poly_verts, constraints=s.constraints,
                substeps=np.int32(s.substeps))


def make(name, s):
    w = OracleWorld.from_scene(s)
    w.broadphase(DT)
    out = scene_arrays(s)
    out["aabbs"] = w.aabbs()
    out["pairs"] = w.pairs()
    k, c, t = w.pair_colors()
    o = np.argsort(k, kind="stable")
    out["pair_colors"] = c[o]
    out["pair_tiles"] = t[o]
    out["contacts"] = w.contacts()
    perm, col = w.joint_order()
    out["joint_perm"], out["joint_colors"] = perm, col
    w.step(DT, ORDER_COLORED)
    out["pos_colored"], out["rot_colored"] = w.poses()
    out["vel_colored"], out["angvel_colored"] = w.velocities()
    for _ in range(9):
        w.step(DT, ORDER_COLORED)
    out["pos_colored10"], out["rot_colored10"] = w.poses()
    w2 = OracleWorld.from_scene(s)
    w2.step(DT, ORDER_SEQUENTIAL)
    out["pos_sequential"], out["rot_sequential"] = w2.poses()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, s.n, "bodies", len(out["pairs"]), "pairs")


def settled_pile(n=3000, width=50, frames=600):
    """The headline workload in small: a mixed circle / rect / capsule pile settled with damping (on the oracle)."""
    s = scenes.mixed_pile(n=n, width=width)
    w = OracleWorld.from_scene(s, linear_damping=3.3, angular_damping=1.0)
    for _ in range(frames):
        w.step(DT, ORDER_COLORED)
    pos, rot = w.poses()
    return s.with_state(pos, rot, name="pile%d_settled" % n)


if __name__ == "__main__":
    make("pile3000_settled", settled_pile())
    make("soup200", scenes.random_soup(200, seed=21, extent=3.0))
    make("soup300_offsets", scenes.random_soup(300, seed=23, extent=3.5, offsets=0.5))   # half the shapes off their body's origin
    make("soup300_compounds", scenes.random_soup(300, seed=29, extent=3.5, offsets=0.3, compounds=0.4))   # several colliders per body
    make("pyramid6", scenes.pyramid(base=6))
    make("chains4x20", scenes.chains(n_chains=4, links=20))
    make("circles500", scenes.circles_container(n=500, width=25, floor_hw=4.0, wall_hh=6.0))
