"""Worker of tests/test_slab.py::test_two_ranks_over_gloo: one slab per process, CPU oracle engine,
halo exchange over torch.distributed (gloo).  Rank 0 saves the gathered global state."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, ticks = sys.argv[1], int(sys.argv[2])
    drifting = len(sys.argv) > 3 and sys.argv[3] == "drift"
    import torch.distributed as dist

    from oracle.oracle import OracleWorld
    from starframe_amd import scenes
    from starframe_amd.slab import DistSlab

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    s = scenes.circles_container(n=600, width=60, floor_hw=8.0, wall_hh=8.0)
    if drifting:   # the pile slides to the right wall: bodies change slab, the cut is renewed every 3 frames
        dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
        vel = np.zeros_like(s.pos)
        vel[dyn, 0] = 2.0
        s = s.with_state(s.pos, s.rot, vel, s.angvel)
    slab = DistSlab(s, OracleWorld, dist, halo=1.5, rebalance_every=2)
    for _ in range(ticks):
        slab.step(1.0 / 60.0)
    pos, rot, vel, ang = slab.gather()
    gs = slab.global_stats()
    if dist.get_rank() == 0:
        np.savez(out, pos=pos, rot=rot, vel=vel, ang=ang, ke=gs["kinetic_energy"], bodies=gs["bodies"],
                 manifolds=gs["manifolds"], max_depth=gs["max_depth"], replans=slab.replans, bounds=slab.bounds)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
