"""ms/tick and pair counts in windows of 20 ticks after the settled pile is restarted without damping."""
import os, sys, time
sys.path.insert(0, ".")
import bench
from starframe_amd import scenes
from starframe_amd.world import World
s = bench.settle(scenes.mixed_pile(n=100000), 0, frames=int(os.environ.get("FRAMES", "600")))
w = World.from_scene(s)
w.sync()
for k in range(int(os.environ.get("WINDOWS", "24"))):
    t0 = time.perf_counter()
    for _ in range(20):
        w.step(bench.DT)
    w.sync()
    t = time.perf_counter() - t0
    st = w.stats()
    print(f"ticks {20*k:4d}-{20*k+20:4d}: {1e3*t/20:.4f} ms/tick pairs {st.n_pairs} solver {st.n_solver_pairs} manifolds {st.n_manifolds} colors {st.n_colors} fallback {st.n_fallback} ke {st.kinetic_energy:.3g}", flush=True)
w.close()
