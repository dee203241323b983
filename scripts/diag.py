"""Diagnostic: tile / fallback statistics and per-kernel times of the bench scene."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from starframe_amd import scenes
from starframe_amd.world import World
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v else int(v)
s = scenes.mixed_pile(n=n)
s = bench.settle(s, 0, frames=400)
w = World.from_scene(s, **kw)
for _ in range(20):
    w.step(bench.DT)
st = w.stats()
print("pairs", st.n_pairs, "tiles", st.n_tiles, "fallback", st.n_fallback, "max_tile_pairs", st.max_tile_pairs,
      "max_tile_bodies", st.max_tile_bodies, "colors", st.n_colors, "manifolds", st.n_manifolds, "overflow", st.overflow,
      "launches", st.launches, "large", st.n_large)
print("color_count", list(st.color_count)[:st.n_colors])
w.set_profiling(True); w.kernel_times(reset=True)
for _ in range(20):
    w.step(bench.DT)
kt = w.kernel_times()
w.set_profiling(False)
for k, v in kt.items():
    if v["launches"]:
        print(f"{k:14s} {v['total_ms']/20:8.4f} ms/tick  {v['launches']//20:4d} launches  {1e3*v['total_ms']/v['launches']:8.2f} us each")
import time
w.sync(); t0 = time.perf_counter()
for _ in range(100):
    w.step(bench.DT)
w.sync(); print("ms/tick", (time.perf_counter() - t0) * 10)
