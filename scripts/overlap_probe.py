"""Is the per-substep halo exchange off the substep's critical path with sfx_halo_overlap?  

One GPU, one rank: a native RCCL communicator of one sends the boundary bodies of a settled granular slab to ITSELF
(send list = receive list = the bodies within `halo` of the slab's two ends, so the unpack writes back what it read and
the scene stays physical), per substep, exactly as DistSlab(transport="native") does between ranks.  Timed: the tick
in pieces without any exchange, with the serial exchange, with the overlapped one; the states of the last two must be
the same bits.

  python scripts/overlap_probe.py [bodies] [ticks]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from starframe_amd import _abi, scenes  # noqa: E402
from starframe_amd.world import World  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 100
s = scenes.granular_slab(n=n, width=max(n // 100, 100))
s = bench.settle(s, 0, frames=300)
dyn = np.flatnonzero(s.inv_mass > 0)
x = s.pos[dyn, 0]
halo = 2.0
left = dyn[x < x.min() + halo].astype(np.uint32)
right = dyn[x > x.max() - halo].astype(np.uint32)
idx = np.concatenate([left, right])
print(f"{n} bodies, halo lists {len(left)} + {len(right)} bodies", flush=True)

rccl = _abi.load_libsfx_rccl()
buf = (C.c_ubyte * _abi.RCCL_ID_BYTES)()
assert rccl.sfx_rccl_unique_id(buf) == 0, rccl.sfx_rccl_last_error()
comm = C.c_void_p()
assert rccl.sfx_rccl_comm_create(buf, 0, 1, 0, C.byref(comm)) == 0, rccl.sfx_rccl_last_error()


def run(mode):
    w = World.from_scene(s)
    if mode != "none":
        w.halo_set(idx, idx)
    if mode == "overlap":
        w.halo_overlap(True)

    def tick():
        w.tick_begin(bench.DT)
        for _ in range(s.substeps):
            w.tick_substep()
            if mode != "none":
                rc = rccl.sfx_rccl_halo_exchange(comm, w.handle, 0, 0, len(left), len(right), len(left), len(right))
                assert rc == 0, rccl.sfx_rccl_last_error()

    for _ in range(30):
        tick()
    w.sync()
    t0 = time.perf_counter()
    for _ in range(ticks):
        tick()
    w.sync()
    dt = (time.perf_counter() - t0) / ticks
    st = w.stats()
    near = w.debug_near_batches()
    out = (w.poses(), w.velocities())
    w.close()
    print(f"{mode:8s} {1e3 * dt:.4f} ms/tick  {n * s.substeps / dt / 1e6:.1f} M body-substeps/s  pairs {st.n_solver_pairs} "
          f"colours {st.n_colors} near batches {near[0]} of {near[1]}", flush=True)
    return dt, out


t_none, _ = run("none")
t_ser, a = run("serial")
t_ovl, b = run("overlap")
same = all(np.array_equal(p.view(np.uint32), q.view(np.uint32)) for p, q in zip(a[0] + a[1], b[0] + b[1]))
print(f"exchange on the critical path: serial {1e6 * (t_ser - t_none) / s.substeps:.1f} us per substep, "
      f"overlapped {1e6 * (t_ovl - t_none) / s.substeps:.1f} us per substep; same bits: {same}")
rccl.sfx_rccl_comm_destroy(comm)
