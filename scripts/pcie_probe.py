"""PCIe-inclusive frame rates on the settled bench scene (Measurement): what a host loop sees when it
takes the poses back every frame, or hands the whole state over and back every frame.  `bench.py`'s `value` has the
state resident in HBM; these figures are never `value`.

  python scripts/pcie_probe.py [frames=240] [settle=600]
"""
import os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from starframe_amd import scenes
from starframe_amd.world import World
import bench

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 240
settle = int(sys.argv[2]) if len(sys.argv) > 2 else 600
s = bench.settle(scenes.mixed_pile(), 0, frames=settle)
nd, sub = s.n_dynamic(), s.substeps
w = World.from_scene(s)
for _ in range(60):
    w.step(bench.DT)
w.sync()


def report(name, seconds):
    print(f"{name}: {1e3 * seconds / frames:.4f} ms per frame, {1e-6 * nd * sub * frames / seconds:.1f} M body-substeps/s", flush=True)


# (a) state resident, no transfers (bench.py's figure)
t0 = time.perf_counter()
for _ in range(frames):
    w.step(bench.DT)
w.sync()
report("resident, no transfer", time.perf_counter() - t0)

# (b) overlapped pose read-back: snapshot + transfer on a second stream while the next tick runs
t0 = time.perf_counter()
w.readback_begin()
for _ in range(frames):
    w.step(bench.DT)
    a = w.readback_wait(copy=False)
    w.readback_begin()
w.readback_wait(copy=False)
w.sync()
report("tick + overlapped pose read-back (1.6 MB per frame)", time.perf_counter() - t0)

# (c) blocking pose download into caller buffers after every tick
t0 = time.perf_counter()
for _ in range(frames):
    w.step(bench.DT)
    pos, rot = w.poses()
report("tick + blocking sfx_download_poses", time.perf_counter() - t0)

# (d) the whole body state handed over and taken back every frame (host-owned state: upload 28 B, download 28 B per body)
pos, rot = w.poses()
vel, angvel = w.velocities()
im, ii = np.asarray(s.inv_mass, np.float32), np.asarray(s.inv_inertia, np.float32)
t0 = time.perf_counter()
for _ in range(frames):
    w.set_state(pos, rot, vel, angvel, im, ii)
    w.step(bench.DT)
    pos, rot = w.poses()
    vel, angvel = w.velocities()
report("upload state + tick + download state, every frame", time.perf_counter() - t0)
