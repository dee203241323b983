"""Worker of tests/test_slab.py (gpu): torch views of libsfx's halo buffers + the one-rank "nccl"
driver.  torch is imported first so that libsfx binds to the HIP runtime torch already loaded."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: see the module docstring)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from starframe_amd import scenes  # noqa: E402
from starframe_amd.slab import DistSlab, _DeviceArray  # noqa: E402
from starframe_amd.world import World  # noqa: E402

DT = 1.0 / 60.0


def main():
    torch.cuda.init()
    s = scenes.circles_container(n=500, width=50, floor_hw=7.0, wall_hh=6.0)
    w = World.from_scene(s)
    stream = torch.cuda.Stream()
    w.set_stream(stream.cuda_stream)
    idx = np.arange(10, 40, dtype=np.uint32)
    w.halo_set(idx, idx[::-1].copy())
    for _ in range(5):
        w.step(DT)
    sp, rp = w.halo_buffer_pointers()
    send = torch.as_tensor(_DeviceArray(sp, 8 * len(idx)), device="cuda")
    recv = torch.as_tensor(_DeviceArray(rp, 8 * len(idx)), device="cuda")
    w.halo_pack()
    w.sync()
    host = w.halo_get()
    assert np.array_equal(send.cpu().numpy().reshape(-1, 8), host), "torch alias does not see the packed halo"
    pos, rot = w.poses()
    assert np.array_equal(host[:, 0:2], pos[idx]) and np.array_equal(host[:, 2:4], rot[idx])
    with torch.cuda.stream(stream):
        recv.copy_(send)              # a "transfer" on the world's stream
    w.halo_unpack()                   # body idx[::-1][k] <- state of body idx[k]
    pos2, _ = w.poses()
    assert np.array_equal(pos2[idx[::-1]], pos[idx]), "unpack did not read what torch wrote"

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # the exchange DistSlab runs between ranks, here from rank 0 to itself: RCCL send/recv straight
        # from / into the library's buffers on the world's stream, ordered between pack and unpack
        for _ in range(3):
            w.step(DT)
        w.halo_pack()
        w.sync()
        sent = send.cpu().numpy().reshape(-1, 8).copy()
        with torch.cuda.stream(stream):
            for work in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]):
                work.wait()
        w.halo_unpack()
        pos3, _ = w.poses()
        assert np.array_equal(pos3[idx[::-1]], sent[:, 0:2]), "RCCL self send/recv did not carry the packed halo"
        w.close()

        # the same exchange through libsfx_rccl (plain rccl.h, a communicator of the library's own): rank 0 names itself
        # as its left neighbour, so its "left" send block lands in its own "from the left" ghosts
        import ctypes as C

        from starframe_amd import _abi
        rl = _abi.load_libsfx_rccl()
        uid = (C.c_ubyte * _abi.RCCL_ID_BYTES)()
        assert rl.sfx_rccl_unique_id(uid) == 0, rl.sfx_rccl_last_error()
        comm = C.c_void_p()
        assert rl.sfx_rccl_comm_create(uid, 0, 1, 0, C.byref(comm)) == 0, rl.sfx_rccl_last_error()
        w2 = World.from_scene(s)
        w2.halo_set(idx, idx[::-1].copy())
        for _ in range(4):
            w2.step(DT)
        w2.sync()
        before, _ = w2.poses()
        assert rl.sfx_rccl_halo_exchange(comm, w2.handle, 0, -1, len(idx), 0, len(idx), 0) == 0, rl.sfx_rccl_last_error()
        w2.sync()
        after, _ = w2.poses()
        assert np.array_equal(after[idx[::-1]], before[idx]), "native RCCL self exchange did not carry the packed halo"
        assert rl.sfx_rccl_halo_exchange(comm, w2.handle, 0, -1, 3, 0, len(idx), 0) != 0      # counts must match sfx_halo_set
        sums = (C.c_double * 2)(1.5, 2.0)
        maxs = (C.c_double * 1)(7.0)
        assert rl.sfx_rccl_allreduce_stats(comm, sums, 2, maxs, 1) == 0 and (sums[0], sums[1], maxs[0]) == (1.5, 2.0, 7.0)
        rl.sfx_rccl_comm_destroy(comm)
        w2.close()

        slab = DistSlab(s, World, dist, device_index=0, halo=3.0)   # falling circles: the halo guard (halo / 4) must outlast the run
        ref = World.from_scene(s)
        assert slab.on_device and slab.native
        for _ in range(20):
            slab.step(DT)
            ref.step(DT)
        slab.sync()
        assert np.array_equal(slab.gather()[0].view(np.uint32), ref.poses()[0].view(np.uint32))

        # count exchange + migration on the device path: the forced re-cut carries every body's state over
        # exactly (new world, new halo buffers on the same stream) and the run goes on
        counts, moved = slab.drift()
        dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
        assert counts.sum() == dyn.sum() and moved > 0.0
        before = slab.gather()
        assert slab.rebalance(force=True) and slab.replans == 1 and slab.on_device
        for a, b in zip(before, slab.gather()):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "state changed in the migration"
        assert slab.drift()[1] == 0.0
        for _ in range(20):
            slab.step(DT)
            ref.step(DT)
        slab.sync()
        pos4 = slab.gather()[0]
        assert np.isfinite(pos4).all() and abs(pos4[dyn, 1].mean() - ref.poses()[0][dyn, 1].mean()) < 0.02
        slab.close()
        ref.close()
    finally:
        dist.destroy_process_group()
    print("slab gpu worker ok")


if __name__ == "__main__":
    main()
