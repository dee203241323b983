"""How often would a persistent pair list have to be rebuilt on the benchmarked pile?  ("measure
rebuild frequency on the bench pile; keep only if it wins".)  Settles the pile as bench.py does, then runs `frames`
undamped ticks and simulates the rule "rebuild when any body has moved more than s since the last rebuild" for several
slacks s, from the real per-frame poses; also the speed distribution that decides it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import DT, settle  # noqa: E402
from starframe_amd import scenes  # noqa: E402
from starframe_amd.world import World  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
s = settle(scenes.mixed_pile(n=100_000, width=width), 0, frames=600 if width == 1000 else 1500)
w = World.from_scene(s)
for _ in range(60):
    w.step(DT)
dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
slacks = [0.005, 0.01, 0.02, 0.03, 0.05]
ref = {k: w.poses()[0].copy() for k in slacks}
rebuilds = {k: 0 for k in slacks}
speeds = []
for f in range(frames):
    w.step(DT)
    pos = w.poses()[0]
    v = np.linalg.norm(w.velocities()[0][dyn], axis=1)
    speeds.append([v.max(), np.percentile(v, 99.9), np.percentile(v, 99), np.sqrt((v * v).mean())])
    for k in slacks:
        if np.abs(pos - ref[k]).max() > k:
            rebuilds[k] += 1
            ref[k] = pos.copy()
sp = np.array(speeds)
print(f"width {width}: {frames} frames; speed m/s max/p99.9/p99/rms: mean over frames {sp.mean(0).round(4).tolist()}, worst frame {sp.max(0).round(4).tolist()}")
print(f"per-frame travel of the fastest body: mean {1e3 * DT * sp[:, 0].mean():.2f} mm, worst {1e3 * DT * sp[:, 0].max():.2f} mm")
for k in slacks:
    print(f"slack {1e3 * k:5.1f} mm: {rebuilds[k]:4d} rebuilds in {frames} frames (every {frames / max(rebuilds[k], 1):.1f} frames)")
