"""Development: wall time of the per-frame stage (broadphase + colouring) run back to back on the settled bench pile,
beside the sum of its kernels' durations in the rocprofv3 trace (launch floors of consecutive small kernels overlap)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from starframe_amd import scenes
from starframe_amd.world import World

s = bench.settle(scenes.mixed_pile(), 0, frames=600)
w = World.from_scene(s)
for _ in range(60):
    w.step(bench.DT)
w.sync(); w.timer_begin()
for _ in range(200):
    w.step(bench.DT)
tick = w.timer_end() / 200
w.sync(); w.timer_begin()
for _ in range(200):
    w.broadphase(bench.DT)
bp = w.timer_end() / 200
print("tick %.4f ms; per-frame stage alone, back to back: %.4f ms (%.1f %% of the tick)" % (tick, bp, 100 * bp / tick))
