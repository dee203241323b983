"""Where the one-rank slab path loses time against World.step: the same settled pile ticked
(a) by sfx_step, (b) piecewise from Python, (c) piecewise + empty halo calls, (d) on a torch stream,
(e) after torch.distributed (nccl) is up, (f) through DistSlab.  One process, one GPU."""
import os
import sys
import time

sys.path.insert(0, ".")
if os.environ.get("PROBE_TORCH_FIRST", "1") == "1":
    import torch  # torch's bundled HIP runtime is then the one libsfx binds to (as in bench.py --gpus N)

    torch.cuda.set_device(0)
import bench
from starframe_amd import scenes
from starframe_amd.world import World

N = int(os.environ.get("PROBE_TICKS", "200"))
s = bench.settle(scenes.mixed_pile(n=100000), 0, frames=600)


def timed(name, w, fn, n=N):
    for _ in range(60):
        fn()
    w.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    w.sync()
    t2 = time.perf_counter()
    print(f"{name:34s} enqueue {1e3 * (t1 - t0) / n:.4f}  total {1e3 * (t2 - t0) / n:.4f} ms/tick", flush=True)


def piecewise(w, halo=False):
    def f():
        w.tick_begin(bench.DT)
        for _ in range(s.substeps):
            w.tick_substep()
            if halo:
                w.halo_pack()
                w.halo_unpack()
    return f


w = World.from_scene(s)
timed("a sfx_step", w, lambda: w.step(bench.DT))
timed("b piecewise", w, piecewise(w))
timed("c piecewise + empty halo", w, piecewise(w, True))
w.close()

import torch

torch.cuda.set_device(0)
w = World.from_scene(s)
timed("a2 sfx_step, torch imported", w, lambda: w.step(bench.DT))
st = torch.cuda.Stream()
w.set_stream(st.cuda_stream)
timed("d sfx_step on a torch stream", w, lambda: w.step(bench.DT))
w.set_stream(0)
timed("d2 back on the own stream", w, lambda: w.step(bench.DT))
w.close()

import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = World.from_scene(s)
timed("e sfx_step, nccl group up", w, lambda: w.step(bench.DT))
dist.barrier()
timed("e2 sfx_step, after a barrier", w, lambda: w.step(bench.DT))
w.close()

from starframe_amd.slab import DistSlab

sl = DistSlab(s, World, dist, device_index=0)
timed("f DistSlab.step", sl.world, lambda: sl.step(bench.DT))
print("stats", sl.world.stats().n_pairs, sl.world.stats().n_manifolds, sl.world.stats().n_colors)
sl.close()

# (g) the exchange mechanism itself: 2 x 1500 bodies sent from rank 0 to itself over RCCL every substep
# (received into the same bodies, so the physics does not change): pack, send/recv, unpack on the world's stream
import numpy as np
from starframe_amd.slab import _DeviceArray

w = World.from_scene(s)
stream = torch.cuda.Stream()
w.set_stream(stream.cuda_stream)
idx = np.arange(2000, 5000, dtype=np.uint32)
w.halo_set(idx, idx)
sp, rp = w.halo_buffer_pointers()
send = torch.as_tensor(_DeviceArray(sp, 8 * len(idx)), device="cuda")
recv = torch.as_tensor(_DeviceArray(rp, 8 * len(idx)), device="cuda")
h = 8 * 1500


def exch_step(pack_only=False):
    w.tick_begin(bench.DT)
    for _ in range(s.substeps):
        w.tick_substep()
        w.halo_pack()
        if not pack_only:
            ops = [dist.P2POp(dist.isend, send[:h], 0), dist.P2POp(dist.irecv, recv[:h], 0),
                   dist.P2POp(dist.isend, send[h:], 0), dist.P2POp(dist.irecv, recv[h:], 0)]
            with torch.cuda.stream(stream):
                for wk in dist.batch_isend_irecv(ops):
                    wk.wait()
        else:
            with torch.cuda.stream(stream):
                recv.copy_(send)
        w.halo_unpack()


timed("g0 pack + copy + unpack per substep", w, lambda: exch_step(True))
timed("g self send/recv 2 x 1500 bodies", w, exch_step)
w.close()
dist.destroy_process_group()
