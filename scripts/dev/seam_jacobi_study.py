"""What would ONE-PHASE sweeps cost in solver quality?  ("solve seam pairs on both neighbouring tiles against
a read-only ghost of the other body ... judge it on the oracle by the pile statistics before writing a kernel".)

The slab decomposition already IS that scheme at its cuts: a pair across a cut is solved on both sides, each side keeping its own
body, the other body a ghost refreshed once per substep.  Cutting a pile into slabs as narrow as the bench's tile strips
(about 1 m) therefore shows what a sweep without seam phases would do to the released pile: same oracle, same contract, the
seams Jacobi instead of Gauss-Seidel.  Compared with the undecomposed pile: v_rms, kinetic energy, contact depths, height.

  python scripts/dev/seam_jacobi_study.py [n] [width] [slabs] [frames]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import ORDER_COLORED, OracleWorld  # noqa: E402
from starframe_amd import scenes  # noqa: E402
from starframe_amd.slab import LocalSlabs  # noqa: E402

DT = 1 / 60.0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
width = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nslabs = int(sys.argv[3]) if len(sys.argv) > 3 else 48
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 300
halo = float(sys.argv[5]) if len(sys.argv) > 5 else 0.45


def depths_of(w):
    w.broadphase(DT)
    c = w.contacts()
    d = np.concatenate([c["depth"][:, 0][c["count"] > 0], c["depth"][:, 1][c["count"] > 1]])
    return d[d > 0]


def report(tag, f, vel, pos, dyn, d):
    sp = np.sqrt((vel[dyn] ** 2).sum(1))
    p = 1e3 * np.percentile(d, [50, 90, 99, 100]) if len(d) else [0, 0, 0, 0]
    print(f"  {tag:14s} frame {f:4d}: v_rms {np.sqrt((sp ** 2).mean()):.4f} p99 {np.percentile(sp, 99):.3f} max {sp.max():.2f} m/s  "
          f"height p99.5 {np.percentile(pos[dyn, 1], 99.5):.3f} m  depth n={len(d)} p50 {p[0]:.2f} p90 {p[1]:.2f} p99 {p[2]:.2f} max {p[3]:.2f} mm",
          flush=True)


s = scenes.mixed_pile(n=n, width=width)
dyn = s.inv_mass > 0
t0 = time.time()
w = OracleWorld.from_scene(s, openmp=True, linear_damping=3.3, angular_damping=1.0)
for _ in range(600):
    w.step(DT, ORDER_COLORED)
pos, rot = w.poses()
w.close()
s2 = s.with_state(pos, rot, name="settled")
print(f"{n} bodies, {width} wide: settled in {time.time() - t0:.0f} s; released for {frames} frames", flush=True)

W = OracleWorld.from_scene(s2, openmp=True)
L = LocalSlabs(s2, nslabs, OracleWorld, halo=halo, rebalance_every=2)
ghosts = sum(len(p.recv_idx) for p in L.plans)
print(f"{nslabs} slabs of {np.diff(L.bounds[1:-1]).mean():.2f} m (halo {halo} m): {ghosts} ghosts for {int(dyn.sum())} bodies", flush=True)
for f in range(1, frames + 1):
    W.step(DT, ORDER_COLORED)
    try:
        L.step(DT)
    except Exception as e:   # the host-side halo guard of the CPU engines
        print("  slabs stopped:", e)
        break
    if f in (1, 50, 100, 200, 300, 450, 600) or f == frames:
        report("undecomposed", f, W.velocities()[0], W.poses()[0], dyn, depths_of(W))
        gp, gr, gv, ga = L.gather()
        X = OracleWorld.from_scene(s2.with_state(gp, gr, gv, ga))   # (a throw-away world: the contact hook of the gathered state)
        report(f"{nslabs} slabs", f, gv, gp, dyn, depths_of(X))
        X.close()
