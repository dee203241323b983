import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from starframe_amd import scenes
from starframe_amd.world import World
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 600
s = scenes.mixed_pile(n=n)
w = World.from_scene(s, linear_damping=3.3, angular_damping=1.0)
def report(tag, w):
    st = w.stats()
    p, _ = w.poses()
    print(f"{tag}: pairs {st.n_pairs} manif {st.n_manifolds} contacts {st.n_contacts} tiles {st.n_tiles} fb {st.n_fallback} "
          f"maxtp {st.max_tile_pairs} maxtb {st.max_tile_bodies} col {st.n_colors} depth {st.max_depth:.4f} KE {st.kinetic_energy:.3g} "
          f"ovf {st.overflow} ymax {p[:-3,1].max():.2f} ymin {p[:-3,1].min():.2f}", flush=True)
for k in range(frames):
    w.step(bench.DT)
    if k % 100 == 99:
        report(f"settle {k+1}", w)
pos, rot = w.poses()
w.close()
s2 = s.with_state(pos, rot)
w = World.from_scene(s2)
for k in range(300):
    w.step(bench.DT)
    if k % 25 == 24:
        report(f"run {k+1}", w)
