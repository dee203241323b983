"""Development: stage timing of k_tile_pos (needs a -DSFX_TILE_TIMING build of libsfx)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from starframe_amd import scenes, _abi
from starframe_amd.world import World
import bench
lib = _abi.load_libsfx(os.environ.get("SFX_LIB"))
_abi._lib = lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
s = scenes.mixed_pile(n=n, width=max(50, n // 100))
s = bench.settle(s, 0, frames=400)
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = int(v)
w = World.from_scene(s, **kw)
out = (C.c_ulonglong * 128)()
lib.sfx_debug_tile_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for _ in range(10):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 1)
for _ in range(10):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 1)
o = list(out)
w.step(bench.DT)   # spans of the last launches (slots cleared above): one tick = 8 fused launches
lib.sfx_debug_tile_timing(w.handle, out, 0)
sp = list(out)
spans = [(sp[17 + 2 * k] - ((~sp[16 + 2 * k]) & 0xFFFFFFFFFFFFFFFF)) / 100 for k in range(16) if sp[17 + 2 * k]]
print("k_tile_vel: first batch start to last batch end per launch (us):", [round(x, 1) for x in spans])
M = 0xFFFFFFFFFFFFFFFF
ks = [(((~sp[48 + 2 * k]) & M), sp[49 + 2 * k], ((~sp[16 + 2 * k]) & M), sp[17 + 2 * k]) for k in range(16) if sp[49 + 2 * k]]
print("k_tile_vel: entry->exit (us):", [round((b - a) / 100, 1) for a, b, _, _ in ks], " entry->first batch:", [round((c - a) / 100, 1) for a, _, c, _ in ks],
      " last batch->exit:", [round((b - d) / 100, 1) for _, b, _, d in ks])
ks.sort()
print("k_tile_vel: exit of one launch -> entry of the next vel launch (us):", [round((ks[i + 1][0] - ks[i][1]) / 100, 1) for i in range(len(ks) - 1)])
tiles = max(o[4], 1)
nv = max(o[13], 1)
print("k_tile_pos per batch (us): wait %.2f load %.2f narrow %.2f colours %.2f writeback %.2f total %.2f | batches %d mean colours %.2f" % (
    o[6] / tiles / 100, o[0] / tiles / 100, o[1] / tiles / 100, o[2] / tiles / 100, o[7] / tiles / 100, o[3] / tiles / 100, tiles, o[5] / tiles))
print("k_tile_vel per batch (us): wait %.2f load %.2f colours %.2f writeback %.2f total %.2f | batches %d" % (
    o[8] / nv / 100, o[9] / nv / 100, o[10] / nv / 100, o[11] / nv / 100, o[12] / nv / 100, nv))
nc_ = max(o[101], 1)
print("k_batch_color per batch (us): setup %.2f rounds %.2f (%.1f rounds) finish %.2f output %.2f total %.2f longest %.2f | batches %d" % (
    o[96] / nc_ / 100, o[97] / nc_ / 100, o[98] / nc_, o[99] / nc_ / 100, o[100] / nc_ / 100, o[102] / nc_ / 100, o[103] / 100, nc_))
print("longest batch without its wait (us): pos %.2f vel %.2f" % (o[15] / 100, o[14] / 100))
print("per tile (us): load %.2f narrow %.2f colours %.2f total %.2f  tiles/launch %.0f  mean colours %.2f" % (
    o[0] / tiles / 100, o[1] / tiles / 100, o[2] / tiles / 100, o[3] / tiles / 100, tiles / (40.0 * (2 if kw.get("tile_cells_y", 0) >= 4096 else 4)), o[5] / tiles))

w.sync()
w.timer_begin()
for _ in range(50):
    w.step(bench.DT)
print("device ms/tick (this build): %.4f" % (w.timer_end() / 50))
w.set_profiling(True); w.kernel_times(reset=True)
for _ in range(20):
    w.step(bench.DT)
kt = w.kernel_times()
print("hipEvent us per launch:", {k: round(1e3 * v["total_ms"] / v["launches"], 1) for k, v in kt.items() if v["launches"] and k in ("solve_pos", "solve_vel", "narrow")})
