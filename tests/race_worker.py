"""Worker of tests/test_gpu_races.py: runs under SFX_LIBRARY=libsfx_races.so (the -DSFX_DEBUG_RACES build, whose tile
solves check on the device that no two pairs of one colour write the same body) and prints the overflow bits seen."""
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
    from starframe_amd.world import SfxError, World

    which = sys.argv[1]
    bits, ticks, pairs = 0, 0, 0

    def run(scene, n_ticks, **kw):
        nonlocal bits, ticks, pairs
        w = World.from_scene(scene, flags=MULTI_LAUNCH, **kw)   # the tile kernels, never the one-workgroup path
        for _ in range(n_ticks):
            w.step(DT)
            ticks += 1
        st = w.stats()
        bits |= int(st.overflow)
        pairs += int(st.n_solver_pairs)
        try:
            w.sync()
        except SfxError:
            pass
        w.close()

    if which == "pile":
        s = scenes.mixed_pile(n=100_000)
        w = World.from_scene(s, linear_damping=3.3, angular_damping=1.0)
        for _ in range(400):
            w.step(DT)
        bits |= int(w.stats().overflow)
        pos, rot = w.poses()
        w.close()
        run(s.with_state(pos, rot), 30)
        run(scenes.mixed_pile(n=20_000, width=80), 200, tile_cells=2, tile_cells_y=2)   # square tiles: all four pair classes
    else:
        rng = np.random.default_rng(77)
        for seed in range(12):
            n = int(rng.integers(300, 3000))
            s = scenes.random_soup(n, seed=100 + seed, extent=float(rng.uniform(3.0, 9.0)), offsets=0.3, compounds=0.2, platforms=int(rng.integers(0, 4)))
            run(s, 6, tile_cells=int(rng.choice([0, 2, 4])), color_rounds=int(rng.integers(1, 13)))
    print(json.dumps({"overflow": bits, "ticks": ticks, "solver_pairs": pairs}))


if __name__ == "__main__":
    main()
