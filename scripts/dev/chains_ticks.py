"""Per-tick wall time of the chains config (sync after every tick), with the stats that change."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from starframe_amd import scenes
from starframe_amd.world import World

n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
w = World.from_scene(scenes.chains())
for i in range(n):
    t0 = time.perf_counter()
    w.step(bench.DT)
    w.sync()
    ms = 1e3 * (time.perf_counter() - t0)
    if i % 10 == 0 or ms > 2.0:
        st = w.stats()
        print(i, f"{ms:.3f} ms pairs {st.n_pairs} solver {st.n_solver_pairs} manifolds {st.n_manifolds} tiles {st.n_tiles} fallback {st.n_fallback} "
              f"colors {st.n_colors} launches {st.launches} overflow {st.overflow} ke {st.kinetic_energy:.3g}", flush=True)
w.close()
