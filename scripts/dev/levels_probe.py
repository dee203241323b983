"""What the grid's second level buys: a soup of 60 k small bodies among 3000 mid-size static platforms, per-frame kernel
classes with two levels (default) and with SFX_CFG_ONE_LEVEL (the platforms on the flat list, tested against everything)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from starframe_amd import scenes
from starframe_amd.world import World, SfxError

s = scenes.random_soup(60_000, seed=43, extent=70.0, platforms=3000, speed=0.5)
for flags in (0, 4, 0, 4):
    w = World.from_scene(s, cell_size=0.6, tile_cells_y=2, flags=flags)
    try:
        for _ in range(20):
            w.step(bench.DT)
        w.sync()
    except SfxError as e:
        print("overflow while warming up:", e)
    w.set_profiling(True)
    w.kernel_times(reset=True)
    for _ in range(20):
        w.step(bench.DT)
    kt = w.kernel_times()
    st = w.stats()
    per = {k: round(1e3 * v["total_ms"] / 20, 1) for k, v in kt.items() if v["launches"]}
    print("flags", flags, "tick", round(sum(per.values()), 1), "us;", {k: per[k] for k in ("aabb_count", "scan", "pair_count", "pair_fill", "color")},
          "small", st.n_small, "large", st.n_large, "pairs", st.n_pairs, "overflow", st.overflow, flush=True)
    w.close()
