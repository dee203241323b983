"""Times the BASELINE.json configs on the GPU: python scripts/configs.py NAME [ticks] [settle]
NAME in pyramid, circles10k, pile100k, chains, slab500k, slab4m.  Prints stats, ms/tick,
body-substeps/s and the per-kernel-class device time per tick (HIP events around every launch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from starframe_amd import scenes
from starframe_amd.world import World

name = sys.argv[1] if len(sys.argv) > 1 else "pile100k"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 400
make = {
    "pyramid": lambda: scenes.pyramid(),
    "circles10k": lambda: scenes.circles_container(),
    "pile100k": lambda: scenes.mixed_pile(),
    "pile100k_w400": lambda: scenes.mixed_pile(width=400),      # SURVEY.md 8d's lattice: 400 wide, 250 rows
    "chains": lambda: scenes.chains(),
    "slab500k": lambda: scenes.granular_slab(n=500_000, width=5000),
    "slab4m": lambda: scenes.granular_slab(),
}[name]
t0 = time.perf_counter()
s = make()
print(f"{name}: {s.n} bodies, {s.n_dynamic()} dynamic, {s.substeps} substeps, built in {time.perf_counter() - t0:.1f} s", flush=True)
if settle and name not in ("pyramid", "chains"):
    t0 = time.perf_counter()
    s = bench.settle(s, 0, frames=settle)
    print(f"settled {settle} ticks in {time.perf_counter() - t0:.1f} s", flush=True)
w = World.from_scene(s)
for _ in range(10):
    w.step(bench.DT)
w.sync()
w.timer_begin()
t0 = time.perf_counter()
for _ in range(ticks):
    w.step(bench.DT)
ms = w.timer_end()
wall = time.perf_counter() - t0
st = w.stats()
print(f"pairs {st.n_pairs} solver_pairs {st.n_solver_pairs} manifolds {st.n_manifolds} contacts {st.n_contacts} tiles {st.n_tiles} "
      f"fallback {st.n_fallback} colors {st.n_colors} joint_colors {st.n_joint_colors} launches {st.launches} "
      f"max_depth {st.max_depth:.4f} ke {st.kinetic_energy:.3g} overflow {st.overflow}")
print(f"device {ms / ticks:.4f} ms/tick  wall {1e3 * wall / ticks:.4f} ms/tick  "
      f"{1e-6 * s.n_dynamic() * s.substeps * ticks / (1e-3 * ms):.1f} M body-substeps/s", flush=True)
w.set_profiling(True)
w.kernel_times(reset=True)
k = min(ticks, 30)
for _ in range(k):
    w.step(bench.DT)
kt = w.kernel_times()
print("per tick, us:", {a: round(1e3 * v["total_ms"] / k, 1) for a, v in kt.items() if v["launches"]})
st = w.stats()
kb = bench.kernel_bytes(st, fused_narrow=kt["narrow"]["launches"] == 0)
for cls, nbytes in kb.items():
    v = kt[cls]
    if v["launches"]:
        dur = 1e-3 * v["total_ms"] / v["launches"]
        per = nbytes * k * s.substeps / v["launches"]
        print(f"  {cls}: {1e6 * dur:.1f} us/launch, {per / 1e6:.1f} MB algorithmic, {per / dur / 1e9:.0f} GB/s = {100 * per / dur / 8e12:.1f} % of 8 TB/s")
print("launches per tick:", {a: v["launches"] // k for a, v in kt.items() if v["launches"]})
print("penetration:", bench.penetration_mm(w))
pos, rot = w.poses()
dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
print("pile height (99.5 % of the dynamic bodies below)", float(np.percentile(pos[dyn, 1], 99.5)))
print("finite", bool(np.isfinite(pos).all() and np.isfinite(rot).all()), "y range", float(pos[:, 1].min()), float(pos[:, 1].max()))
w.close()
