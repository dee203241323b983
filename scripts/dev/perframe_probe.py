"""Per-frame kernel classes of the settled pile, us per tick (events around every launch), for the library in use."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from starframe_amd import scenes
from starframe_amd.world import World

s = bench.settle(scenes.mixed_pile(), 0, frames=600)
w = World.from_scene(s)
for _ in range(40):
    w.step(bench.DT)
w.sync()
w.timer_begin()
for _ in range(100):
    w.step(bench.DT)
ms = w.timer_end()
w.set_profiling(True)
w.kernel_times(reset=True)
for _ in range(40):
    w.step(bench.DT)
kt = w.kernel_times()
st = w.stats()
per = {k: round(1e3 * v["total_ms"] / 40, 1) for k, v in kt.items() if v["launches"]}
frame = sum(per[k] for k in ("aabb_count", "scan", "scatter", "pair_count", "pair_fill", "tile_list", "color"))
print(os.environ.get("SFX_LIBRARY", "default"), f"tick {ms / 100:.4f} ms  per-frame {frame:.1f} us", per, "pairs", st.n_pairs, flush=True)
w.close()
