"""Worker of tests/test_gpu_fused_tick.py: ticks a few worlds through sfx_step under the SFX_FUSE_TICK setting of its
environment (read once per process by libsfx) and prints a hash of every world's final state, the launches of the
one-launch tick seen (kernel class "tick") and what the worlds held (solver pairs, fallback pairs)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DT = 1.0 / 60.0
MULTI_LAUNCH = 1


def main():
    from starframe_amd import scenes
    from starframe_amd.world import World

    h = hashlib.sha256()
    out = {"tick_launches": 0, "integrate_launches": 0, "ticks": 0, "fallback": 0, "solver_pairs": 0, "per_world": []}

    def run(scene, n_ticks, **kw):
        w = World.from_scene(scene, flags=MULTI_LAUNCH, **kw)   # the tile kernels, never the one-workgroup path
        w.set_profiling_classes(["tick", "integrate"])
        w.kernel_times(reset=True)
        for _ in range(n_ticks):
            w.step(DT)
        kt = w.kernel_times()
        out["tick_launches"] += kt["tick"]["launches"]
        out["integrate_launches"] += kt["integrate"]["launches"]
        out["ticks"] += n_ticks
        w.set_profiling(False)
        w.tick_begin(DT)                 # one tick substep by substep (the launch-per-stage path in both settings) ...
        for _ in range(scene.substeps):
            w.tick_substep()
        for _ in range(3):               # ... and back
            w.step(DT)
        st = w.stats()
        assert st.overflow == 0, (scene.name, st.overflow)
        out["fallback"] += int(st.n_fallback)
        out["solver_pairs"] += int(st.n_solver_pairs)
        hw = hashlib.sha256()
        pos, rot = w.poses()
        vel, om = w.velocities()
        for a in (pos, rot, vel, om):
            b = np.ascontiguousarray(a, dtype=np.float32).tobytes()
            h.update(b)
            hw.update(b)
        out["per_world"].append([scene.name, int(st.n_solver_pairs), int(st.n_fallback), hw.hexdigest()[:12]])
        w.close()

    rng = np.random.default_rng(7)
    for seed in range(6):
        n = int(rng.integers(600, 3000))
        s = scenes.random_soup(n, seed=500 + seed, extent=float(rng.uniform(6.0, 14.0)), offsets=0.3, platforms=int(rng.integers(0, 4)))
        if seed % 2 == 0:   # bodies that are not dynamic but move: nobody writes them inside the launch
            st = np.flatnonzero(np.asarray(s.inv_mass) == 0.0)
            s.vel[st[-1]] = (0.4, -0.1)
            s.angvel[st[0]] = 0.3
        run(s, 6, tile_cells=int(rng.choice([0, 2, 4])), color_rounds=int(rng.choice([4, 12])))
    run(scenes.mixed_pile(n=20_000, width=80), 120, tile_cells=2, tile_cells_y=2)   # square tiles: all four pair classes
    run(scenes.mixed_pile(n=30_000, width=300), 120)                                 # strips
    run(scenes.mixed_pile(n=12_000, width=40, pitch=0.16), 60, tile_cells=8, tile_cells_y=8)   # over-full tiles: a fallback list
    run(scenes.circles_container(n=10_000), 60)
    run(scenes.random_soup(300, seed=77, extent=80.0, with_statics=False), 12)   # sparse: hardly a pair -- nearly every body is stepped by k_tick_free
    print(json.dumps(out))


if __name__ == "__main__":
    main()
