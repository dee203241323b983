import sys, time
sys.path.insert(0, '.')
import bench
from starframe_amd import scenes
from starframe_amd.world import World
s = bench.settle(scenes.mixed_pile(n=100000), 0, frames=400)
w = World.from_scene(s)
for _ in range(30): w.step(bench.DT)
w.sync()
t0 = time.perf_counter()
for _ in range(200): w.step(bench.DT)
t1 = time.perf_counter()
w.sync()
t2 = time.perf_counter()
print("enqueue ms/tick", 1e3*(t1-t0)/200, "total ms/tick", 1e3*(t2-t0)/200)
