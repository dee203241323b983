"""How the tile solves' launch time follows their instruction count: the settled pile with and without friction
(SFX_CFG_NO_FRICTION drops about 40 % of a velocity step's arithmetic, nothing of its memory traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from starframe_amd import scenes
from starframe_amd.world import World

s = bench.settle(scenes.mixed_pile(), 0, frames=600)
for flags in (0, 2, 0, 2):
    w = World.from_scene(s, flags=flags)
    for _ in range(20):
        w.step(bench.DT)
    w.sync()
    w.set_profiling(True)
    w.kernel_times(reset=True)
    for _ in range(30):
        w.step(bench.DT)
    kt = w.kernel_times()
    st = w.stats()
    print("flags", flags, {k: round(1e3 * v["total_ms"] / v["launches"], 2) for k, v in kt.items() if k in ("solve_pos", "solve_vel", "narrow")},
          "manifolds", st.n_manifolds, "colors", st.n_colors, flush=True)
    w.close()
