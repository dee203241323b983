"""Development: where a colour step of k_tile_pos / k_tile_vel spends its cycles (needs a -DSFX_STEP_PROBE build of libsfx:
python -m starframe_amd.build ... or hipcc with -DSFX_STEP_PROBE into devlib/probe/libsfx.so, SFX_LIB=devlib/probe/libsfx.so).
Shares only: the stamps forbid overlaps the real kernel has."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from starframe_amd import _abi, scenes  # noqa: E402
from starframe_amd.world import World  # noqa: E402

lib = _abi.load_libsfx(os.environ.get("SFX_LIB"))
_abi._lib = lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
s = bench.settle(scenes.mixed_pile(n=n, width=max(50, n // 100)), 0, frames=600)
w = World.from_scene(s)
out = (C.c_ulonglong * 128)()
lib.sfx_debug_tile_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for _ in range(30):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 1)
for _ in range(50):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 0)
o = list(out)
names = ["step table read", "record wait (vmcnt 0)", "decode + next fetch issue", "solve (LDS read, math, LDS write)", "barrier"]
for base, kern in ((64, "k_tile_pos"), (72, "k_tile_vel")):
    ns = max(o[base + 5], 1)
    seg = [o[base + 0], o[base + 1], o[base + 2], o[base + 3]]
    # stamps: pq0 top, pq1 after the record wait, pq2 after decode + fetch, pq3 after the solve, pq4 after the barrier
    print(f"{kern}: {ns} steps stamped in workgroup 3; cycles per step (stamps included, ~40 each):")
    tot = sum(seg)
    for nm, v in zip(["step table read + record wait", "decode + next fetch issue", "solve (LDS read, math, LDS write)", "barrier"], seg):
        print(f"   {nm:40s} {v / ns:8.1f}  {100.0 * v / max(tot, 1):5.1f} %")
    print(f"   {'sum':40s} {tot / ns:8.1f}")
w.sync()
w.timer_begin()
for _ in range(50):
    w.step(bench.DT)
print("device ms/tick (this build): %.4f" % (w.timer_end() / 50))
nb = max(o[89], 1)
print(f"k_batch_color, workgroup (5, 0, 0): {nb} batches stamped; cycles per batch (every stamp drains the loads in flight):")
names = ["tile info + greedy cut", "batch ranges + tables", "filter gather + table init", "pair load loop", "body mark scan",
         "body_cursor atomic + outside bodies", "body list + masses", "colouring rounds", "finish + output"]
tot = sum(o[80:89])
for nm, v in zip(names, o[80:89]):
    print(f"   {nm:40s} {v / nb:9.0f}  {100.0 * v / max(tot, 1):5.1f} %")
print(f"   {'sum':40s} {tot / nb:9.0f}")
