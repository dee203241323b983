"""Development: where a substep of the one-workgroup path (k_small_substeps) spends its time.  Needs a -DSFX_TILE_TIMING
build: python -m starframe_amd.build -DSFX_TILE_TIMING -o ... (see scripts/variants.sh), SFX_LIBRARY=<that build>."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from starframe_amd import scenes, _abi
from starframe_amd.world import World
import bench

lib = _abi.load_libsfx(os.environ.get("SFX_LIBRARY"))
lib.sfx_debug_tile_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
s = scenes.pyramid() if len(sys.argv) < 2 else scenes.random_soup(int(sys.argv[1]), seed=3, extent=float(sys.argv[2]))
w = World.from_scene(s, flags=_abi.CFG_ONE_WORKGROUP)
out = (C.c_ulonglong * 128)()
for _ in range(100):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 1)
for _ in range(20):
    w.step(bench.DT)
lib.sfx_debug_tile_timing(w.handle, out, 0)
o = list(out)
n = max(o[72], 1)
names = ["setup (per launch)", "integrate+joints", "narrow", "pos phases", "fallback pos + derive", "damp + vel phases", "fallback vel", "substep"]
print("substeps", n, "stats", w.stats().n_pairs, "pairs", w.stats().n_colors, "colours")
for k, nm in enumerate(names):
    d = 20 if k == 0 else n
    print("%-24s %.2f us" % (nm, o[64 + k] / d / 100))
w.sync(); w.timer_begin()
for _ in range(50):
    w.step(bench.DT)
print("device ms/tick (this build): %.4f" % (w.timer_end() / 50))
