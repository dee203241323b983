#!/usr/bin/env python3
"""Where do the deep overlaps of settled piles come from?    Runs on the CPU oracle, which the GPU
matches bit for bit.  Columns and piles of growing depth, substep counts and solver knobs; prints the distribution of
positive contact depths and which pairs carry the deep ones."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import ORDER_COLORED, ORDER_SEQUENTIAL, OracleWorld
from starframe_amd import scenes
from starframe_amd.scenes import SceneBuilder

DT = 1 / 60.0

def depth_stats(w, dyn):
    c = w.contacts()
    d = c["depth"][:, 0][c["count"] > 0]
    d2 = c["depth"][:, 1][c["count"] > 1]
    d = np.concatenate([d, d2])
    d = d[d > 0]
    if not len(d):
        return "no contacts"
    return "n=%d p50=%.1f p90=%.1f p99=%.1f max=%.1f mm" % (len(d), *(1e3 * np.percentile(d, [50, 90, 99, 100])))

def column(layers, r=0.075, substeps=4, frames=400, order=ORDER_COLORED, **kw):
    b = SceneBuilder("col", substeps=substeps)
    b.static_rect((0, -0.5), 5, 0.5)
    for k in range(layers):
        b.circle((0.0, r + 2 * r * k), r)
    s = b.build()
    w = OracleWorld.from_scene(s, **kw)
    for _ in range(frames):
        w.step(DT, order)
    p, _ = w.poses()
    y = p[1:, 1]
    gaps = np.diff(np.concatenate([[0.0 - r], y]))  # centre distance between consecutive
    pen = 2 * r - gaps
    pen[0] = r - y[0]
    return pen

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "column"
    if which == "column":
        for layers in (10, 25, 50, 100):
            for sub in (4, 8, 16):
                for order, on in ((ORDER_COLORED, "col"), (ORDER_SEQUENTIAL, "seq")):
                    pen = column(layers, substeps=sub, order=order)
                    print(f"layers {layers:3d} substeps {sub:2d} {on}: bottom pen {1e3*pen[0]:7.2f} mm, max {1e3*pen.max():7.2f}, mean {1e3*pen.mean():7.2f}, height loss {1e3*pen.sum():8.1f} mm", flush=True)
    if which == "pile":
        n, width = int(sys.argv[2]), int(sys.argv[3])
        for sub, kw in ((4, {}), (4, dict(max_depenetration_velocity=-1.0)), (8, {}), (16, {})):
            s = scenes.mixed_pile(n=n, width=width, substeps=sub)
            w = OracleWorld.from_scene(s, linear_damping=3.3, angular_damping=1.0, **kw)
            t0 = time.time()
            for _ in range(700):
                w.step(DT, ORDER_COLORED)
            p, _ = w.poses()
            dyn = s.inv_mass > 0
            print(f"pile n={n} width={width} substeps={sub} {kw}: height {p[dyn,1].max():.2f} m, {depth_stats(w, dyn)}  ({time.time()-t0:.0f}s)", flush=True)
