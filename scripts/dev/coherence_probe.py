"""How long a broadphase of the settled pile stays valid: frames until some body has left the padded box it had
when the pair list was built (margin + min(|v| dt, 0.03) per axis), and what fraction of the bodies has by then."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from starframe_amd import scenes
from starframe_amd.world import World

s = bench.settle(scenes.mixed_pile(), 0, frames=600)
w = World.from_scene(s)
for _ in range(60):
    w.step(bench.DT)
dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
for margin in (0.01, 0.02):
    p0 = w.poses()[0].copy()
    v0 = w.velocities()[0].copy()
    pad = margin + np.minimum(np.abs(v0) * bench.DT, 0.03)
    first = None
    for f in range(1, 61):
        w.step(bench.DT)
        d = np.abs(w.poses()[0] - p0)
        out = (d > pad).any(axis=1) & dyn
        if f in (1, 2, 3, 5, 10, 20, 40, 60):
            print(f"margin {margin}: frame {f}: {out.sum()} bodies outside their box, max move {d[dyn].max() * 1e3:.1f} mm, p99.9 {np.percentile(d[dyn].max(axis=1), 99.9) * 1e3:.2f} mm", flush=True)
st = w.stats()
print("ke", st.kinetic_energy, "v rms", float(np.sqrt((w.velocities()[0][dyn] ** 2).sum(1).mean())))
