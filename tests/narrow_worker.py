"""Worker of tests/test_gpu_fused_narrow.py: ticks a few worlds under the SFX_FUSE_NARROW setting of its environment (read
once per process by libsfx) and prints a hash of every world's final state and the number of k_narrow launches seen."""
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
    narrow_launches, fallback, pairs = 0, 0, 0

    def run(scene, n_ticks, **kw):
        nonlocal narrow_launches, fallback, pairs
        w = World.from_scene(scene, flags=MULTI_LAUNCH, **kw)   # the tile kernels, never the one-workgroup path
        for _ in range(n_ticks):
            w.step(DT)
        w.set_profiling(True)
        w.kernel_times(reset=True)
        w.step(DT)
        narrow_launches += w.kernel_times()["narrow"]["launches"]
        w.set_profiling(False)
        for _ in range(2):           # substep by substep (the slab drivers' path)
            w.tick_begin(DT)
            for _ in range(scene.substeps):
                w.tick_substep()
        st = w.stats()
        assert st.overflow == 0, st.overflow
        fallback += int(st.n_fallback)
        pairs += int(st.n_solver_pairs)
        pos, rot = w.poses()
        vel, om = w.velocities()
        for a in (pos, rot, vel, om):
            h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        w.close()

    rng = np.random.default_rng(5)
    for seed in range(8):
        n = int(rng.integers(300, 2500))
        s = scenes.random_soup(n, seed=300 + seed, extent=float(rng.uniform(6.0, 14.0)), offsets=0.3, compounds=0.3, platforms=int(rng.integers(0, 4)))
        run(s, 5, tile_cells=int(rng.choice([0, 2, 4])), color_rounds=int(rng.choice([4, 12])))
    run(scenes.mixed_pile(n=20_000, width=80), 150, tile_cells=2, tile_cells_y=2)   # square tiles: all four pair classes
    run(scenes.mixed_pile(n=30_000, width=300), 150)                                 # strips
    run(scenes.chains(n_chains=40, links=30), 20)                                   # joints between the launches
    print(json.dumps({"state": h.hexdigest(), "narrow_launches": narrow_launches, "fallback": fallback, "solver_pairs": pairs}))


if __name__ == "__main__":
    main()
