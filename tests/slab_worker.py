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
    if len(sys.argv) > 3 and sys.argv[3] == "trip":
        # one body of the LEFT slab shoots off at 30 m/s: rank 0's halo guard trips in the first frames, rank 1 sees nothing.
        # Both ranks must stop, together, at the next collective check -- and neither may hang in an exchange
        from starframe_amd.slab import SlabError

        dyn = np.nonzero((s.inv_mass > 0) | (s.inv_inertia > 0))[0]
        left = dyn[np.argmin(s.pos[dyn, 0])]
        vel = np.zeros_like(s.pos)
        vel[left] = (0.0, 30.0)
        vel[left, 0] = 30.0
        s = s.with_state(s.pos, s.rot, vel, s.angvel)
        slab = DistSlab(s, OracleWorld, dist, halo=1.5, check_every=4)
        msg = "no error"
        try:
            for _ in range(ticks):
                slab.step(1.0 / 60.0)
            slab.sync()
        except SlabError as e:
            msg = str(e)
        with open(f"{out}.rank{dist.get_rank()}", "w") as f:
            f.write(f"{slab.frames}\n{msg}\n")
        dist.barrier()
        dist.destroy_process_group()
        return
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
