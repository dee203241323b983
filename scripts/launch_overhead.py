"""Fixed cost of the tick's launches: a world of far-apart bodies (no candidate pairs at all), per-kernel-class
device time per launch from HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from starframe_amd.scenes import SceneBuilder
from starframe_amd.world import World

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
b = SceneBuilder("apart", substeps=4)
for i in range(n):
    b.circle((3.0 * (i % 100), 3.0 * (i // 100)), 0.1)
w = World.from_scene(b.build(), gravity=(0.0, 0.0))
for _ in range(20):
    w.step(1 / 60)
w.sync()
w.timer_begin()
for _ in range(100):
    w.step(1 / 60)
ms = w.timer_end()
print("pairs", w.stats().n_pairs, "device ms/tick", ms / 100)
w.set_profiling(True); w.kernel_times(reset=True)
for _ in range(50):
    w.step(1 / 60)
kt = w.kernel_times()
print({k: round(1e3 * v["total_ms"] / v["launches"], 2) for k, v in kt.items() if v["launches"]}, "us per launch")
