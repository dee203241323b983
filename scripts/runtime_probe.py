"""Host enqueue cost per tick under the HIP runtime libsfx ends up bound to: MODE=sys (ROCm's, libsfx alone),
MODE=torchlib (torch's bundled libamdhip64 loaded first, torch not imported), MODE=torch (import torch first)."""
import ctypes, os, sys, time
sys.path.insert(0, ".")
mode = os.environ.get("MODE", "sys")
if mode == "torchlib":
    d = "/usr/local/lib/python3.10/dist-packages/torch/lib"
    ctypes.CDLL(os.path.join(d, "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
elif mode == "torch":
    import torch
    if os.environ.get("INIT", "1") == "1":
        torch.cuda.set_device(0)
import bench
from starframe_amd import scenes
from starframe_amd.world import World
s = bench.settle(scenes.mixed_pile(n=100000), 0, frames=int(os.environ.get("FRAMES", "100")))
w = World.from_scene(s)
for _ in range(30): w.step(bench.DT)
w.sync()
t0 = time.perf_counter()
for _ in range(200): w.step(bench.DT)
t1 = time.perf_counter()
w.sync()
t2 = time.perf_counter()
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})
print(mode, os.environ.get("INIT", ""), "enqueue ms/tick %.4f total %.4f" % (1e3*(t1-t0)/200, 1e3*(t2-t0)/200), libs, flush=True)
