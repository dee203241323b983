"""Writes a scenes.Scene to the flat binary format tools/sfx_run.cpp reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from starframe_amd import scenes


def dump(s, path):
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    pv = f32(s.poly_verts).reshape(-1, 2)
    with open(path, "wb") as f:
        assert len(s.colliders) == s.n, "scene files hold one collider per body"
        np.array([0x33584653, s.n, len(s.materials), pv.shape[0], len(s.constraints), s.substeps], np.uint32).tofile(f)
        for a in (s.pos, s.rot, s.vel, s.angvel, s.inv_mass, s.inv_inertia):
            f32(a).tofile(f)
        np.ascontiguousarray(s.colliders).tofile(f)
        np.ascontiguousarray(s.materials).tofile(f)
        pv.tofile(f)
        np.ascontiguousarray(s.constraints).tofile(f)


if __name__ == "__main__":
    # python scripts/dump_scene.py KIND N OUT.bin [SETTLE_TICKS]   (settling runs on the GPU, damped, like bench.py)
    kind, n, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    settle = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    relax = int(sys.argv[5]) if len(sys.argv) > 5 else 0   # further undamped ticks (bench.py's warm-up + timed region)
    s = {"pile": lambda: scenes.mixed_pile(n=n, width=max(50, n // 100)),
         "circles": lambda: scenes.circles_container(n=n),
         "slab": lambda: scenes.granular_slab(n=n, width=max(50, n // 100)),
         "chains": lambda: scenes.chains(n_chains=n // 100, links=100)}[kind]()
    if settle:
        import bench
        s = bench.settle(s, 0, frames=settle)
        if relax:   # undamped ticks; the velocities are kept: the file is the steady (jittering) state bench.py times
            from starframe_amd.world import World
            w = World.from_scene(s)
            for _ in range(relax):
                w.step(bench.DT)
            pos, rot = w.poses()
            vel, ang = w.velocities()
            w.close()
            s = s.with_state(pos, rot, vel, ang, name=s.name + "_relaxed")
    dump(s, path)
    print(path, s.n, "bodies", "settled" if settle else "")
