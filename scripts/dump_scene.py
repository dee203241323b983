"""Writes a scenes.Scene to the flat binary format tools/sfx_run.cpp reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from starframe_amd import scenes


def dump(s, path):
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pv = f32(s.poly_verts).reshape(-1, 2)
    with open(path, "wb") as f:
        np.array([0x31584653, s.n, len(s.materials), pv.shape[0], len(s.constraints), s.substeps], np.uint32).tofile(f)
        for a in (s.pos, s.rot, s.vel, s.angvel, s.inv_mass, s.inv_inertia):
            f32(a).tofile(f)
        np.ascontiguousarray(s.colliders).tofile(f)
        np.ascontiguousarray(s.materials).tofile(f)
        pv.tofile(f)
        np.ascontiguousarray(s.constraints).tofile(f)


if __name__ == "__main__":
    kind, n, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    s = {"pile": lambda: scenes.mixed_pile(n=n, width=max(50, n // 100)),
         "circles": lambda: scenes.circles_container(n=n),
         "chains": lambda: scenes.chains(n_chains=n // 100, links=100)}[kind]()
    dump(s, path)
    print(path, s.n, "bodies")
