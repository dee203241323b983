#!/usr/bin/env python3
"""Why does the benchmarked pile never come to rest, and which contacts carry the deep tail?  

Runs on the CPU oracle (bit-identical to the GPU on these scenes).  A mixed pile is settled with damping, released
without, and for `frames` undamped frames the script logs v_rms, kinetic energy and the contact-depth distribution;
at the end it splits the contact depths by shape-pair class, by height in the pile and by wall / floor contact and lists
the fastest bodies.  Variants switch one solver feature at a time so that the cause can be told apart.

  python scripts/jitter_study.py <n> <width> <settle> <frames> <variant> [<variant> ...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import ORDER_COLORED, OracleWorld  # noqa: E402
from starframe_amd import scenes  # noqa: E402

DT = 1 / 60.0
NAMES = {1: "circle", 2: "rect", 3: "capsule", 4: "poly"}

VARIANTS = {
    "base": {},
    "nofric": dict(flags=2),
    "nospec": dict(speculative_margin=-1.0),
    "sub8": dict(substeps=8),
    "rest0": dict(_restitution=0.0),
    "thr8": dict(restitution_threshold=8.0),
    "margin0": dict(aabb_margin=0.002),
    "clamp1": dict(max_depenetration_velocity=1.0),
    "clamp05": dict(max_depenetration_velocity=0.5),
    "pad10": dict(vel_pad_max=0.10),
    "spec5": dict(speculative_margin=0.005),
    "spec10": dict(speculative_margin=0.010),
    "spec20": dict(speculative_margin=0.020),
    "damp01": dict(_release_damping=0.1),
}


def depths(c):
    d = np.concatenate([c["depth"][:, 0][c["count"] > 0], c["depth"][:, 1][c["count"] > 1]])
    return d[d > 0]


def pct(d):
    if not len(d):
        return "none"
    return "n=%6d p50 %5.2f p90 %5.2f p99 %5.2f max %6.2f mm" % (len(d), *(1e3 * np.percentile(d, [50, 90, 99, 100])))


def run(n, width, settle, frames, name, openmp=True):
    kw = dict(VARIANTS[name])
    rest = kw.pop("_restitution", None)
    sub = kw.pop("substeps", 4)
    rdamp = kw.pop("_release_damping", 0.0)
    s = scenes.mixed_pile(n=n, width=width, substeps=sub)
    if rest is not None:
        s.materials["restitution"] = rest
    w = OracleWorld.from_scene(s, openmp=openmp, linear_damping=3.3, angular_damping=1.0, **kw)
    t0 = time.time()
    for _ in range(settle):
        w.step(DT, ORDER_COLORED)
    pos, rot = w.poses()
    w.close()
    s2 = s.with_state(pos, rot, name="settled")
    w = OracleWorld.from_scene(s2, openmp=openmp, linear_damping=rdamp, angular_damping=rdamp, **kw)
    dyn = (s.inv_mass > 0)
    mass = np.where(dyn, 1.0 / np.maximum(s.inv_mass, 1e-30), 0.0)
    print(f"== {name} {kw} n={n} width={width} substeps={sub}: settled {settle} ticks in {time.time() - t0:.0f}s, "
          f"height {pos[dyn, 1].max():.2f} m", flush=True)
    types = np.asarray(s.colliders["type"][: s.n])
    for f in range(frames + 1):
        if f % 50 == 0:
            vel, om = w.velocities()
            sp = np.sqrt((vel[dyn] ** 2).sum(1))
            w.broadphase(DT)
            c = w.contacts()
            st = w.stats()
            print(f"  frame {f:4d}: v_rms {np.sqrt((sp ** 2).mean()):.4f} p50 {np.median(sp):.4f} p99 {np.percentile(sp, 99):.3f} "
                  f"max {sp.max():.2f} m/s  w_rms {np.sqrt((om[dyn] ** 2).mean()):.3f} rad/s  KE {0.5 * (mass[dyn] * sp ** 2).sum():.4f} J  "
                  f"pairs {st.n_pairs} solver {st.n_solver_pairs}  depth {pct(depths(c))}", flush=True)
        if f < frames:
            w.step(DT, ORDER_COLORED)
    # ---- where the deep contacts are
    vel, om = w.velocities()
    pos, rot = w.poses()
    w.broadphase(DT)
    c = w.contacts()
    c = c[c["count"] > 0]
    a, b = c["body_a"], c["body_b"]
    dmax = np.maximum(c["depth"][:, 0], np.where(c["count"] > 1, c["depth"][:, 1], -1))
    keep = dmax > 0
    c, a, b, dmax = c[keep], a[keep], b[keep], dmax[keep]
    ta, tb = types[a], types[b]
    stat = ~dyn[a] | ~dyn[b]
    print("  by class (deepest point of each touching pair):")
    for x in (1, 2, 3):
        for y in (x, *[z for z in (1, 2, 3) if z > x]):
            m = (((ta == x) & (tb == y)) | ((ta == y) & (tb == x))) & ~stat
            print(f"    {NAMES[x]:8s}-{NAMES[y]:8s}: {pct(dmax[m])}")
    print(f"    with a static (floor/walls)  : {pct(dmax[stat])}")
    for cnt in (1, 2):
        print(f"    manifolds with {cnt} point(s)    : {pct(dmax[(c['count'] == cnt) & ~stat])}")
    ymid = 0.5 * (pos[a, 1] + pos[b, 1])
    top = pos[dyn, 1].max()
    print("  by height (fraction of pile height):")
    for lo in np.arange(0.0, 1.0, 0.2):
        m = (ymid >= lo * top) & (ymid < (lo + 0.2) * top) & ~stat
        print(f"    {lo:.1f}-{lo + 0.2:.1f}: {pct(dmax[m])}")
    # force chains: within one height band the contact depths (= loads: a contact carries its load by its depth) of a random
    # packing are spread like the contact forces of a granular pile -- roughly exponentially about their mean
    band = (ymid < 0.2 * top) & ~stat
    if band.sum() > 100:
        x = dmax[band] / dmax[band].mean()
        print("  bottom fifth, depth / mean depth: fraction above 1, 2, 3, 4, 6 x mean: "
              + ", ".join(f"{(x > k).mean():.4f}" for k in (1, 2, 3, 4, 6))
              + "   (exponential: " + ", ".join(f"{np.exp(-k):.4f}" for k in (1, 2, 3, 4, 6)) + f"); mean {1e3 * dmax[band].mean():.2f} mm")
    sp = np.sqrt((vel ** 2).sum(1))
    print("  speed by body type: " + ", ".join(f"{NAMES[t]} v_rms {np.sqrt((sp[dyn & (types == t)] ** 2).mean()):.4f} "
                                              f"w_rms {np.sqrt((om[dyn & (types == t)] ** 2).mean()):.3f}" for t in (1, 2, 3)))
    print("  speed by height: " + ", ".join(
        f"{lo:.1f}: {np.sqrt((sp[dyn & (pos[:, 1] >= lo * top) & (pos[:, 1] < (lo + 0.2) * top)] ** 2).mean()):.4f}" for lo in np.arange(0.0, 1.0, 0.2)))
    order = np.argsort(-dmax)[:12]
    print("  deepest pairs:")
    for k in order:
        print(f"    {NAMES[int(ta[k])]}-{NAMES[int(tb[k])]} count {c['count'][k]} depth {1e3 * c['depth'][k][0]:.1f}/{1e3 * c['depth'][k][1]:.1f} mm "
              f"y {ymid[k]:.2f} static {bool(stat[k])} normal ({c['normal'][k][0]:.2f},{c['normal'][k][1]:.2f})")
    w.close()


if __name__ == "__main__":
    n, width, settle, frames = (int(x) for x in sys.argv[1:5])
    for v in sys.argv[5:]:
        run(n, width, settle, frames, v)
