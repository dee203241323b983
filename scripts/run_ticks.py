"""Runs N un-profiled ticks of the settled bench scene (for rocprofv3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starframe_amd import scenes
from starframe_amd.world import World
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v else int(v)
s = scenes.mixed_pile(n=n)
s = bench.settle(s, 0, frames=300)
w = World.from_scene(s, **kw)
for _ in range(10):
    w.step(bench.DT)
w.sync()
t0 = time.perf_counter()
w.timer_begin()
for _ in range(ticks):
    w.step(bench.DT)
ms = w.timer_end()
print("wall ms/tick", (time.perf_counter() - t0) * 1e3 / ticks, "device ms/tick", ms / ticks)
