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
out = (C.c_ulonglong * 16)()
lib.sfx_debug_tile_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for _ in range(10):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 1)
for _ in range(10):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 1)
o = list(out)
tiles = max(o[4], 1)
nv = max(o[13], 1)
print("k_tile_pos per batch (us): wait %.2f load %.2f narrow %.2f colours %.2f writeback %.2f total %.2f | batches %d mean colours %.2f" % (
    o[6] / tiles / 100, o[0] / tiles / 100, o[1] / tiles / 100, o[2] / tiles / 100, o[7] / tiles / 100, o[3] / tiles / 100, tiles, o[5] / tiles))
print("k_tile_vel per batch (us): wait %.2f load %.2f colours %.2f writeback %.2f total %.2f | batches %d" % (
    o[8] / nv / 100, o[9] / nv / 100, o[10] / nv / 100, o[11] / nv / 100, o[12] / nv / 100, nv))
print("per tile (us): load %.2f narrow %.2f colours %.2f total %.2f  tiles/launch %.0f  mean colours %.2f" % (
    o[0] / tiles / 100, o[1] / tiles / 100, o[2] / tiles / 100, o[3] / tiles / 100, tiles / (40.0 * (2 if kw.get("tile_cells_y", 0) >= 4096 else 4)), o[5] / tiles))
