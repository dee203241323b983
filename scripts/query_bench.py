"""Times the spatial queries on the settled 100k pile: python scripts/query_bench.py [n_queries]
Prints wall time per call (host arrays in and out, index rebuild included in the first call) and the device
time of the query kernels (HIP events, class "query"), with the oracle's answers checked on a sample."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from starframe_amd import scenes
from starframe_amd._abi import NO_BODY
from starframe_amd.world import World

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
s = bench.settle(scenes.mixed_pile(), 0, frames=400)
w = World.from_scene(s)
w.step(bench.DT)
rng = np.random.default_rng(1)
lo, hi = s.pos[:-3].min(0), s.pos[:-3].max(0)
pts = rng.uniform(lo, hi, (nq, 2)).astype(np.float32)
ang = rng.uniform(0, 2 * np.pi, nq)
rays = np.concatenate([rng.uniform(lo, hi + (0, 5.0), (nq, 2)), np.stack([np.cos(ang), np.sin(ang)], 1)], 1).astype(np.float32)
max_t = np.full(nq, 5.0, np.float32)
b0 = rng.uniform(lo, hi, (nq // 10, 2))
boxes = np.concatenate([b0, b0 + rng.uniform(0.1, 1.0, (nq // 10, 2))], 1).astype(np.float32)


def timed(name, fn, n):
    fn()  # first call: index build + allocation effects
    w.set_profiling(True)
    w.kernel_times(reset=True)
    t0 = time.perf_counter()
    out = fn()
    wall = time.perf_counter() - t0
    kt = w.kernel_times()["query"]
    w.set_profiling(False)
    print(f"{name}: {n} queries, wall {1e3 * wall:.2f} ms ({n / wall / 1e6:.1f} M/s incl. PCIe copies), "
          f"kernel {kt['total_ms']:.3f} ms ({n / (1e-3 * kt['total_ms']) / 1e6:.1f} M/s device)", flush=True)
    return out


body, cnt = timed("points", lambda: w.query_points(pts), nq)
hits = timed("rays(max_t=5)", lambda: w.query_rays(rays, max_t), nq)
bh = timed("boxes", lambda: w.query_aabbs(boxes, cap=40 * len(boxes)), len(boxes))
print("points inside a body:", float((body != NO_BODY).mean()), " rays that hit:", float((hits["body"] != NO_BODY).mean()),
      " box hits per box:", len(bh) / len(boxes))
# parity on a sample against the oracle (same scene state: poses copied over)
from oracle.oracle import OracleWorld

o = OracleWorld.from_scene(s)
p, r = w.poses()
v, av = w.velocities()
o.set_state(p, r, v, av, s.inv_mass, s.inv_inertia)
k = 2000
ob, oc = o.query_points(pts[:k])
oh = o.query_rays(rays[:k], max_t[:k])
print("sample parity: points", bool(np.array_equal(ob, body[:k]) and np.array_equal(oc, cnt[:k])),
      "rays", bool(np.array_equal(oh["body"], hits["body"][:k]) and np.array_equal(oh["t"], hits["t"][:k])))
t0 = time.perf_counter()
for k in range(200):
    w.query_points(pts[k:k + 1])
print(f"one point per call: {1e6 * (time.perf_counter() - t0) / 200:.1f} us per call (index already built)")
t0 = time.perf_counter()
for k in range(200):
    w.step(bench.DT)
    w.query_rays(rays[k:k + 1], max_t[k:k + 1])
w.sync()
t1 = time.perf_counter()
for k in range(200):
    w.step(bench.DT)
    w.sync()
t2 = time.perf_counter()
print(f"one ray per tick (index rebuilt every time): {1e6 * ((t1 - t0) - (t2 - t1)) / 200:.1f} us on top of a synchronised tick")
