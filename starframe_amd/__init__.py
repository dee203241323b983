"""starframe_amd: MI355X-native 2D XPBD physics tick behind the libsfx C ABI.

Only the hot path of starframe's physics tick lives here:
  csrc/        HIP kernels (gfx950) + the C-ABI host runtime -> libsfx.so
  world.py     ctypes handle over the C ABI
  physics.py   host-side mirror of the reference's physics API (config, body, collider,
               material, tick)
  scenes.py    synthetic scenes of the five BASELINE.json configs
  slab.py      slab domain decomposition across GPUs (RCCL halo exchange)
"""
from ._abi import make_config  # noqa: F401
