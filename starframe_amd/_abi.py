"""ctypes view of include/sfx.h: struct layouts, numpy dtypes and the libsfx loader.

The product path has no CPU fallback: `load_libsfx()` raises if the HIP shared library has
not been built (run `python -c "import __graft_entry__ as g; g.build()"`), and
`sfx_create` itself fails with SFX_ERR_DEVICE when no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBSFX_PATH = os.path.join(HERE, "csrc", "libsfx.so")

SFX_OK = 0
SFX_ERR_INVALID = -1
SFX_ERR_UNSUPPORTED = -2
SFX_ERR_DEVICE = -3
SFX_ERR_CAPACITY = -4
SFX_ERR_NOMEM = -5

SHAPE_NONE, SHAPE_CIRCLE, SHAPE_RECT, SHAPE_CAPSULE, SHAPE_POLY = 0, 1, 2, 3, 4
MAX_POLY_VERTS = 8
WORLD_BODY = 0xFFFFFFFF
MAX_COLORS = 64
TILE_MAX_PAIRS = 2048
TILE_MAX_BODIES = 1024
FALLBACK_TILE = 0xFFFFFFFF

LIMIT_NONE, LIMIT_ROPE, LIMIT_STRUT = 0, 1, 2
CFG_MULTI_LAUNCH = 1     # small worlds too take the launch-per-stage path
CFG_NO_FRICTION = 2
CFG_ONE_LEVEL = 4
CFG_ONE_WORKGROUP = 8    # worlds of <= 1024 bodies: substeps in one workgroup from the first tick on


class SfxConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("substeps", C.c_uint32),
        ("gravity", C.c_float * 2),
        ("aabb_margin", C.c_float),
        ("vel_pad_max", C.c_float),
        ("cell_size", C.c_float),
        ("grid_log2", C.c_uint32 * 2),
        ("max_pairs", C.c_uint32),
        ("color_rounds", C.c_uint32),
        ("tile_cells", C.c_uint32),
        ("tile_cells_y", C.c_uint32),
        ("restitution_threshold", C.c_float),
        ("max_depenetration_velocity", C.c_float),
        ("speculative_margin", C.c_float),
        ("linear_damping", C.c_float),
        ("angular_damping", C.c_float),
        ("flags", C.c_uint32),
    ]


class SfxFrameStats(C.Structure):
    _fields_ = [
        ("n_bodies", C.c_uint32),
        ("n_small", C.c_uint32),
        ("n_large", C.c_uint32),
        ("n_pairs", C.c_uint32),
        ("n_colors", C.c_uint32),
        ("n_leftover", C.c_uint32),
        ("n_manifolds", C.c_uint32),
        ("n_manifolds2", C.c_uint32),
        ("n_contacts", C.c_uint32),
        ("n_joint_colors", C.c_uint32),
        ("overflow", C.c_uint32),
        ("launches", C.c_uint32),
        ("n_tiles", C.c_uint32),
        ("n_fallback", C.c_uint32),
        ("max_tile_pairs", C.c_uint32),
        ("max_tile_bodies", C.c_uint32),
        ("n_solver_pairs", C.c_uint32),
        ("max_depth", C.c_float),
        ("kinetic_energy", C.c_float),
        ("color_count", C.c_uint32 * MAX_COLORS),
    ]


class SfxKernelTime(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("launches", C.c_uint64), ("items", C.c_uint64)]


K_CLASSES = 17
NO_BODY = 0xFFFFFFFF

collider_dtype = np.dtype(
    [
        ("type", "<u4"),
        ("material", "<u4"),
        ("category", "<u4"),
        ("mask", "<u4"),
        ("group", "<u4"),
        ("p", "<f4", (3,)),
        ("poly_offset", "<u4"),
        ("poly_count", "<u4"),
        ("local_pos", "<f4", (2,)),
        ("local_angle", "<f4"),
        ("body", "<u4"),
    ]
)
material_dtype = np.dtype([("restitution", "<f4"), ("static_friction", "<f4"), ("dynamic_friction", "<f4")])
constraint_dtype = np.dtype(
    [
        ("body_a", "<u4"),
        ("body_b", "<u4"),
        ("anchor_a", "<f4", (2,)),
        ("anchor_b", "<f4", (2,)),
        ("rest_length", "<f4"),
        ("compliance", "<f4"),
        ("limit", "<u4"),
        ("damping", "<f4"),
    ]
)
angle_constraint_dtype = np.dtype(
    [
        ("body_a", "<u4"),
        ("body_b", "<u4"),
        ("min_angle", "<f4"),
        ("max_angle", "<f4"),
        ("compliance", "<f4"),
        ("damping", "<f4"),
    ]
)
contact_dtype = np.dtype(
    [
        ("body_a", "<u4"),
        ("body_b", "<u4"),
        ("count", "<u4"),
        ("normal", "<f4", (2,)),
        ("depth", "<f4", (2,)),
        ("point_a", "<f4", (2, 2)),
        ("point_b", "<f4", (2, 2)),
    ]
)
assert collider_dtype.itemsize == 56
assert material_dtype.itemsize == 12
assert constraint_dtype.itemsize == 40
assert angle_constraint_dtype.itemsize == 24
ray_hit_dtype = np.dtype([("body", "<u4"), ("t", "<f4"), ("normal", "<f4", (2,))])
assert contact_dtype.itemsize == 60
assert ray_hit_dtype.itemsize == 16


def make_config(
    substeps=4,
    gravity=(0.0, -9.81),
    aabb_margin=0.01,
    vel_pad_max=0.0,
    cell_size=0.0,
    grid_log2=(0, 0),
    max_pairs=0,
    color_rounds=0,
    tile_cells=0,
    tile_cells_y=0,
    restitution_threshold=0.0,
    max_depenetration_velocity=0.0,
    speculative_margin=0.0,
    linear_damping=0.0,
    angular_damping=0.0,
    flags=0,
    device=0,
):
    cfg = SfxConfig()
    cfg.struct_size = C.sizeof(SfxConfig)
    cfg.device = device
    cfg.substeps = substeps
    cfg.gravity[0], cfg.gravity[1] = gravity
    cfg.aabb_margin = aabb_margin
    cfg.vel_pad_max = vel_pad_max
    cfg.cell_size = cell_size
    cfg.grid_log2[0], cfg.grid_log2[1] = grid_log2
    cfg.max_pairs = max_pairs
    cfg.color_rounds = color_rounds
    cfg.tile_cells = tile_cells
    cfg.tile_cells_y = tile_cells_y
    cfg.restitution_threshold = restitution_threshold
    cfg.max_depenetration_velocity = max_depenetration_velocity
    cfg.speculative_margin = speculative_margin
    cfg.linear_damping = linear_damping
    cfg.angular_damping = angular_damping
    cfg.flags = flags
    return cfg


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def vptr(a):
    return a.ctypes.data_as(C.c_void_p)


# every symbol include/sfx.h declares; tests check the built library exports all of them
SFX_SYMBOLS = [
    "sfx_create", "sfx_destroy", "sfx_abi_version", "sfx_build_id", "sfx_default_config",
    "sfx_query_aabbs_device", "sfx_query_rays_device",
    "sfx_upload_bodies", "sfx_upload_colliders", "sfx_upload_materials", "sfx_upload_constraints",
    "sfx_upload_angle_constraints",
    "sfx_step", "sfx_sync", "sfx_download_poses", "sfx_download_velocities",
    "sfx_readback_begin", "sfx_readback_wait",
    "sfx_tick_begin", "sfx_tick_substep", "sfx_set_stream", "sfx_halo_set", "sfx_halo_pack", "sfx_halo_unpack",
    "sfx_halo_buffers", "sfx_halo_download", "sfx_halo_upload", "sfx_halo_copy", "sfx_halo_guard", "sfx_get_stream", "sfx_halo_counts",
    "sfx_halo_overlap", "sfx_halo_stream",
    "sfx_query_points", "sfx_query_aabbs", "sfx_query_rays", "sfx_query_points_static",
    "sfx_debug_aabbs", "sfx_debug_pairs", "sfx_debug_pair_colors", "sfx_debug_broadphase",
    "sfx_debug_contacts", "sfx_debug_joint_order", "sfx_debug_near_batches",
    "sfx_stats", "sfx_set_profiling", "sfx_kernel_times", "sfx_kernel_class_name",
    "sfx_timer_begin", "sfx_timer_end", "sfx_last_error",
]

_lib = None


def bind_world_api(lib, prefix):
    """Declare argument types of the world API shared by libsfx (`sfx_`) and the oracle (`sfo_`)."""
    P = C.POINTER
    vp, u32, f32p = C.c_void_p, C.c_uint32, P(C.c_float)
    g = lambda name: getattr(lib, prefix + name)
    g("upload_bodies").argtypes = [vp, u32, f32p, f32p, f32p, f32p, f32p, f32p]
    g("upload_colliders").argtypes = [vp, u32, vp, f32p, u32]
    g("upload_materials").argtypes = [vp, u32, vp]
    g("upload_constraints").argtypes = [vp, u32, vp]
    g("upload_angle_constraints").argtypes = [vp, u32, vp]
    g("download_poses").argtypes = [vp, f32p, f32p]
    g("download_velocities").argtypes = [vp, f32p, f32p]
    g("stats").argtypes = [vp, P(SfxFrameStats)]
    for n in ("upload_bodies", "upload_colliders", "upload_materials", "upload_constraints",
              "upload_angle_constraints", "download_poses", "download_velocities", "stats"):
        g(n).restype = C.c_int


def load_libsfx(path=None):
    """Load the HIP library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SFX_LIBRARY") or LIBSFX_PATH  # SFX_LIBRARY: another build of libsfx (tuning runs)
    if not os.path.exists(p):
        raise RuntimeError(
            f"libsfx.so not found at {p}: build it with __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback"
        )
    lib = C.CDLL(p)
    P = C.POINTER
    vp = C.c_void_p
    lib.sfx_create.argtypes = [P(SfxConfig), P(vp)]
    lib.sfx_create.restype = C.c_int
    lib.sfx_destroy.argtypes = [vp]
    lib.sfx_destroy.restype = None
    lib.sfx_abi_version.restype = C.c_uint32
    lib.sfx_build_id.restype = C.c_char_p
    lib.sfx_default_config.argtypes = [P(SfxConfig)]
    lib.sfx_default_config.restype = None
    bind_world_api(lib, "sfx_")
    lib.sfx_step.argtypes = [vp, C.c_float]
    lib.sfx_step.restype = C.c_int
    lib.sfx_sync.argtypes = [vp]
    lib.sfx_sync.restype = C.c_int
    lib.sfx_tick_begin.argtypes = [vp, C.c_float]
    lib.sfx_tick_substep.argtypes = [vp]
    lib.sfx_set_stream.argtypes = [vp, vp]
    lib.sfx_halo_set.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32]
    lib.sfx_halo_pack.argtypes = [vp]
    lib.sfx_halo_guard.argtypes = [vp, C.c_float]
    lib.sfx_get_stream.argtypes = [vp, P(vp)]
    lib.sfx_get_stream.restype = C.c_int
    lib.sfx_halo_overlap.argtypes = [vp, C.c_int]
    lib.sfx_halo_overlap.restype = C.c_int
    lib.sfx_halo_stream.argtypes = [vp, P(vp)]
    lib.sfx_halo_stream.restype = C.c_int
    lib.sfx_halo_counts.argtypes = [vp, P(C.c_uint32), P(C.c_uint32)]
    lib.sfx_halo_counts.restype = C.c_int
    lib.sfx_halo_guard.restype = C.c_int
    lib.sfx_halo_unpack.argtypes = [vp]
    lib.sfx_halo_buffers.argtypes = [vp, P(vp), P(vp)]
    lib.sfx_halo_download.argtypes = [vp, P(C.c_float)]
    lib.sfx_halo_upload.argtypes = [vp, P(C.c_float)]
    lib.sfx_halo_copy.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32]
    for n in ("sfx_tick_begin", "sfx_tick_substep", "sfx_set_stream", "sfx_halo_set", "sfx_halo_pack",
              "sfx_halo_unpack", "sfx_halo_buffers", "sfx_halo_download", "sfx_halo_upload", "sfx_halo_copy"):
        getattr(lib, n).restype = C.c_int
    lib.sfx_readback_begin.argtypes = [vp]
    lib.sfx_readback_begin.restype = C.c_int
    lib.sfx_readback_wait.argtypes = [vp, P(P(C.c_float)), P(C.c_uint32)]
    lib.sfx_readback_wait.restype = C.c_int
    lib.sfx_query_points.argtypes = [vp, C.c_uint32, P(C.c_float), C.c_uint32, vp, vp]
    lib.sfx_query_aabbs.argtypes = [vp, C.c_uint32, P(C.c_float), C.c_uint32, vp, C.c_size_t, P(C.c_size_t)]
    lib.sfx_query_rays.argtypes = [vp, C.c_uint32, P(C.c_float), P(C.c_float), C.c_uint32, vp]
    lib.sfx_query_points_static.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, vp]
    lib.sfx_query_aabbs_device.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp]
    lib.sfx_query_rays_device.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint32, vp]
    for n in ("sfx_query_points", "sfx_query_aabbs", "sfx_query_rays", "sfx_query_points_static", "sfx_query_aabbs_device",
              "sfx_query_rays_device"):
        getattr(lib, n).restype = C.c_int
    lib.sfx_debug_aabbs.argtypes = [vp, P(C.c_float)]
    lib.sfx_debug_pairs.argtypes = [vp, vp, C.c_size_t, P(C.c_size_t)]
    lib.sfx_debug_pair_colors.argtypes = [vp, vp, vp, vp, C.c_size_t, P(C.c_size_t)]
    lib.sfx_debug_broadphase.argtypes = [vp, C.c_float]
    lib.sfx_debug_contacts.argtypes = [vp, vp, C.c_size_t, P(C.c_size_t)]
    lib.sfx_debug_joint_order.argtypes = [vp, vp, vp, C.c_size_t, P(C.c_size_t)]
    lib.sfx_debug_near_batches.argtypes = [vp, P(C.c_uint32), P(C.c_uint32)]
    lib.sfx_debug_near_batches.restype = C.c_int
    lib.sfx_set_profiling.argtypes = [vp, C.c_int]
    lib.sfx_kernel_times.argtypes = [vp, P(SfxKernelTime), C.c_int]
    lib.sfx_kernel_class_name.argtypes = [C.c_int]
    lib.sfx_kernel_class_name.restype = C.c_char_p
    lib.sfx_timer_begin.argtypes = [vp]
    lib.sfx_timer_end.argtypes = [vp, P(C.c_double)]
    lib.sfx_last_error.argtypes = [vp]
    lib.sfx_last_error.restype = C.c_char_p
    for n in ("sfx_debug_aabbs", "sfx_debug_pairs", "sfx_debug_pair_colors", "sfx_debug_broadphase",
              "sfx_debug_contacts", "sfx_debug_joint_order", "sfx_set_profiling", "sfx_kernel_times",
              "sfx_timer_begin", "sfx_timer_end"):
        getattr(lib, n).restype = C.c_int
    if path is None:
        _lib = lib
    return lib


# ---- libsfx_rccl.so (include/sfx_rccl.h): the slab halo exchange over plain RCCL
LIBSFX_RCCL_PATH = os.path.join(HERE, "csrc", "libsfx_rccl.so")
SFX_RCCL_SYMBOLS = ["sfx_rccl_unique_id", "sfx_rccl_comm_create", "sfx_rccl_comm_destroy", "sfx_rccl_comm_abort", "sfx_rccl_halo_exchange", "sfx_rccl_sendrecv",
                    "sfx_rccl_allreduce_stats", "sfx_rccl_last_error"]
RCCL_ID_BYTES = 128
_rccl = None


def load_libsfx_rccl():
    """Loads libsfx_rccl.so (after libsfx.so, which it links).  Raises when it has not been built."""
    global _rccl
    if _rccl is not None:
        return _rccl
    load_libsfx()
    if not os.path.exists(LIBSFX_RCCL_PATH):
        raise RuntimeError(f"libsfx_rccl.so not found at {LIBSFX_RCCL_PATH}: build it with __graft_entry__.build()")
    lib = C.CDLL(LIBSFX_RCCL_PATH)
    vp, u32 = C.c_void_p, C.c_uint32
    lib.sfx_rccl_unique_id.argtypes = [vp]
    lib.sfx_rccl_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.sfx_rccl_comm_destroy.argtypes = [vp]
    lib.sfx_rccl_comm_destroy.restype = None
    lib.sfx_rccl_comm_abort.argtypes = [vp]
    lib.sfx_rccl_comm_abort.restype = None
    lib.sfx_rccl_halo_exchange.argtypes = [vp, vp, C.c_int, C.c_int, u32, u32, u32, u32]
    lib.sfx_rccl_sendrecv.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_int]
    lib.sfx_rccl_sendrecv.restype = C.c_int
    lib.sfx_rccl_allreduce_stats.argtypes = [vp, C.POINTER(C.c_double), u32, C.POINTER(C.c_double), u32]
    lib.sfx_rccl_last_error.restype = C.c_char_p
    for n in ("sfx_rccl_unique_id", "sfx_rccl_comm_create", "sfx_rccl_halo_exchange", "sfx_rccl_allreduce_stats"):
        getattr(lib, n).restype = C.c_int
    _rccl = lib
    return lib
