import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starframe_amd import scenes
from starframe_amd.world import World
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
s = scenes.mixed_pile(n=n)
s = bench.settle(s, 0, frames=int(sys.argv[2]) if len(sys.argv) > 2 else 300)
w = World.from_scene(s)
for _ in range(10):
    w.step(bench.DT)
w.sync()
for rep in range(3):
    t0 = time.perf_counter(); w.timer_begin()
    for _ in range(100):
        w.step(bench.DT)
    t1 = time.perf_counter()
    ms = w.timer_end()
    print(f"async 100: enqueue {1e3*(t1-t0)/100:.3f} ms/tick, wall {(time.perf_counter()-t0)*10:.3f} ms/tick, device {ms/100:.3f}")
t0 = time.perf_counter()
for _ in range(100):
    w.step(bench.DT); w.sync()
print(f"sync each: wall {(time.perf_counter()-t0)*10:.3f} ms/tick")
st = w.stats()
print("pairs", st.n_pairs, "manifolds", st.n_manifolds, "tiles", st.n_tiles, "fallback", st.n_fallback, "colors", st.n_colors)
