"""Thin Python handle over the libsfx C ABI (include/sfx.h).

`World` owns one `sfx_world*`.  `WorldBase` holds the marshalling shared with the CPU
oracle's handle (oracle/oracle.py), so parity tests drive both through identical calls.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (
    K_CLASSES, SFX_ERR_CAPACITY, SFX_OK, SfxFrameStats, SfxKernelTime, contact_dtype, fptr, make_config,
    ray_hit_dtype, vptr,
)


class SfxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sfx error {code}: {msg}")
        self.code = code


class WorldBase:
    """Marshalling over a `<prefix>_*` world API; subclasses provide lib, prefix, handle."""

    lib = None
    prefix = ""
    handle = None
    n = 0      # bodies
    nc = 0     # colliders (>= n: the further colliders of compound bodies follow the bodies' own)
    m = 0
    n_halo_send = 0
    n_halo_recv = 0

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def _check(self, rc):
        if rc != SFX_OK:
            raise SfxError(rc, self.last_error())
        return rc

    def last_error(self):
        return ""

    def upload(self, scene):
        """Upload a scenes.Scene: bodies, colliders, materials, constraints."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self.n = scene.n
        pos, rot, vel = f32(scene.pos), f32(scene.rot), f32(scene.vel)
        angvel, im, ii = f32(scene.angvel), f32(scene.inv_mass), f32(scene.inv_inertia)
        self._check(self._f("upload_bodies")(self.handle, self.n, fptr(pos), fptr(rot), fptr(vel),
                                              fptr(angvel), fptr(im), fptr(ii)))
        col = np.ascontiguousarray(scene.colliders)
        self.nc = len(col)
        pv = f32(scene.poly_verts).reshape(-1, 2)
        self._check(self._f("upload_materials")(self.handle, len(scene.materials),
                                                 vptr(np.ascontiguousarray(scene.materials))))
        self._check(self._f("upload_colliders")(self.handle, len(col), vptr(col), fptr(pv), pv.shape[0]))
        con = np.ascontiguousarray(scene.constraints)
        self.m = len(con)
        self._check(self._f("upload_constraints")(self.handle, len(con), vptr(con)))
        ang = np.ascontiguousarray(getattr(scene, "angle_constraints", ()))
        if len(ang) or getattr(self, "m_angle", 0):
            self._check(self._f("upload_angle_constraints")(self.handle, len(ang), vptr(ang)))
        self.m_angle = len(ang)
        self.m += len(ang)
        return self

    def set_state(self, pos, rot, vel, angvel, inv_mass, inv_inertia):
        """Replace the body state (same body count; colliders and constraints are kept)."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        a = [f32(x) for x in (pos, rot, vel, angvel, inv_mass, inv_inertia)]
        assert a[0].shape[0] == self.n
        self._check(self._f("upload_bodies")(self.handle, self.n, *[fptr(x) for x in a]))

    def poses(self):
        pos = np.empty((self.n, 2), np.float32)
        rot = np.empty((self.n, 2), np.float32)
        self._check(self._f("download_poses")(self.handle, fptr(pos), fptr(rot)))
        return pos, rot

    def velocities(self):
        vel = np.empty((self.n, 2), np.float32)
        angvel = np.empty(self.n, np.float32)
        self._check(self._f("download_velocities")(self.handle, fptr(vel), fptr(angvel)))
        return vel, angvel

    def stats(self):
        st = SfxFrameStats()
        self._check(self._f("stats")(self.handle, C.byref(st)))
        return st


class World(WorldBase):
    """A libsfx world on one MI355X.  Raises if the HIP library or the device is missing."""

    prefix = "sfx_"

    def __init__(self, cfg=None, **cfg_kw):
        self.lib = _abi.load_libsfx()
        self.cfg = cfg or make_config(**cfg_kw)
        h = C.c_void_p()
        rc = self.lib.sfx_create(C.byref(self.cfg), C.byref(h))
        if rc != SFX_OK:
            raise SfxError(rc, (self.lib.sfx_last_error(None) or b"").decode())
        self.handle = h

    @classmethod
    def from_scene(cls, scene, **cfg_kw):
        kw = dict(substeps=scene.substeps, gravity=scene.gravity, aabb_margin=scene.aabb_margin)
        kw.update(cfg_kw)
        return cls(**kw).upload(scene)

    def last_error(self):
        return (self.lib.sfx_last_error(self.handle) or b"").decode()

    def close(self):
        if self.handle:
            self.lib.sfx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, dt):
        self._check(self.lib.sfx_step(self.handle, dt))

    def sync(self):
        self._check(self.lib.sfx_sync(self.handle))

    # ---- the tick in pieces + slab halos (starframe_amd/slab.py drives these)
    def tick_begin(self, dt):
        self._check(self.lib.sfx_tick_begin(self.handle, dt))

    def tick_substep(self):
        self._check(self.lib.sfx_tick_substep(self.handle))

    def set_stream(self, hip_stream):
        """Run on an existing HIP stream (an int handle, e.g. torch.cuda.current_stream().cuda_stream)."""
        self._check(self.lib.sfx_set_stream(self.handle, C.c_void_p(hip_stream or None)))

    def halo_set(self, send_idx, recv_idx):
        s = np.ascontiguousarray(send_idx, np.uint32)
        r = np.ascontiguousarray(recv_idx, np.uint32)
        self.n_halo_send, self.n_halo_recv = len(s), len(r)
        self._check(self.lib.sfx_halo_set(self.handle, vptr(s), len(s), vptr(r), len(r)))

    def halo_guard(self, max_move):
        """Ticks flag (overflow bit 128 -> SFX_ERR_CAPACITY) a body that moved further than max_move on x since halo_set."""
        self._check(self.lib.sfx_halo_guard(self.handle, float(max_move)))

    def halo_overlap(self, on=True):
        """The velocity sweep as two launches with the halo's pack and transfer on a stream of their own behind the first
        (include/sfx.h sfx_halo_overlap): same results, the exchange off the substep's critical path."""
        self._check(self.lib.sfx_halo_overlap(self.handle, 1 if on else 0))

    def halo_stream(self):
        """The hipStream_t (as an integer) the halo's pack and transfer are enqueued on."""
        s = C.c_void_p()
        self._check(self.lib.sfx_halo_stream(self.handle, C.byref(s)))
        return s.value or 0

    def halo_get(self):
        """Host copy of the send bodies' (pose, velocity): [n_send, 8]."""
        out = np.empty((self.n_halo_send, 8), np.float32)
        self._check(self.lib.sfx_halo_download(self.handle, fptr(out)))
        return out

    def halo_put(self, arr):
        a = np.ascontiguousarray(arr, np.float32)
        assert a.shape == (self.n_halo_recv, 8)
        self._check(self.lib.sfx_halo_upload(self.handle, fptr(a)))

    def halo_pack(self):
        self._check(self.lib.sfx_halo_pack(self.handle))

    def halo_unpack(self):
        self._check(self.lib.sfx_halo_unpack(self.handle))

    def halo_buffer_pointers(self):
        """Device addresses of the library-owned send / receive buffers."""
        s, r = C.c_void_p(), C.c_void_p()
        self._check(self.lib.sfx_halo_buffers(self.handle, C.byref(s), C.byref(r)))
        return s.value or 0, r.value or 0

    def halo_copy_from(self, src, dst_first, src_first, count):
        """Device-to-device: src's send buffer [src_first, +count) -> this world's receive buffer."""
        self._check(self.lib.sfx_halo_copy(self.handle, dst_first, src.handle, src_first, count))

    # ---- spatial queries at the current poses (include/sfx.h "spatial queries")
    def query_points(self, points, mask=0xFFFFFFFF):
        """Per point: lowest id of the bodies whose collider contains it (NO_BODY if none), and how many do."""
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        body = np.empty(len(p), np.uint32)
        count = np.empty(len(p), np.uint32)
        self._check(self.lib.sfx_query_points(self.handle, len(p), fptr(p), mask, vptr(body), vptr(count)))
        return body, count

    def query_aabbs(self, boxes, mask=0xFFFFFFFF, cap=None):
        """Ascending keys (query << 32 | body) of the colliders whose tight AABB overlaps each box."""
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
        cap = cap if cap is not None else max(16 * len(b), 1024)
        while True:
            hits = np.empty(cap, np.uint64)
            n = C.c_size_t()
            rc = self.lib.sfx_query_aabbs(self.handle, len(b), fptr(b), mask, vptr(hits), cap, C.byref(n))
            if rc == SFX_ERR_CAPACITY and n.value > cap:
                cap = n.value
                continue
            self._check(rc)
            return hits[: n.value]

    def query_rays(self, rays, max_t, mask=0xFFFFFFFF):
        """Closest hit per ray (origin.x, origin.y, dir.x, dir.y): structured array (body, t, normal)."""
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 4)
        mt = np.ascontiguousarray(np.broadcast_to(np.asarray(max_t, np.float32), (len(r),)))
        out = np.zeros(len(r), ray_hit_dtype)
        self._check(self.lib.sfx_query_rays(self.handle, len(r), fptr(r), fptr(mt), mask, vptr(out)))
        return out

    def query_points_static(self, points_dev, n, body_dev, count_dev=0, mask=0xFFFFFFFF):
        """sfx_query_points_static: device addresses (ints) of [n][2] float points and [n] uint32 outputs; enqueues on
        the world's stream, allocates nothing, does not synchronise."""
        self._check(self.lib.sfx_query_points_static(self.handle, n, C.c_void_p(points_dev), mask, C.c_void_p(body_dev),
                                                     C.c_void_p(count_dev or None)))

    def query_aabbs_device(self, boxes_dev, n, hits_dev, cap, n_hits_dev, mask=0xFFFFFFFF):
        """sfx_query_aabbs_device: device addresses (ints) of [n][4] float boxes, `cap` uint64 hit keys (query << 32 | body,
        unordered) and one uint32 for the full hit count; enqueues on the world's stream, does not synchronise."""
        self._check(self.lib.sfx_query_aabbs_device(self.handle, n, C.c_void_p(boxes_dev), mask, C.c_void_p(hits_dev or None), cap,
                                                    C.c_void_p(n_hits_dev)))

    def query_rays_device(self, rays_dev, max_t_dev, n, hits_dev, mask=0xFFFFFFFF):
        """sfx_query_rays_device: device addresses of [n][4] float rays, [n] float max_t and [n] sfx_ray_hit records."""
        self._check(self.lib.sfx_query_rays_device(self.handle, n, C.c_void_p(rays_dev), C.c_void_p(max_t_dev), mask,
                                                   C.c_void_p(hits_dev)))

    # ---- parity hooks
    def broadphase(self, dt):
        self._check(self.lib.sfx_debug_broadphase(self.handle, dt))

    def aabbs(self):
        out = np.empty((self.nc, 4), np.float32)   # one box per collider
        self._check(self.lib.sfx_debug_aabbs(self.handle, fptr(out)))
        return out

    def pairs(self):
        cap = max(int(self.stats().n_pairs), 1)
        keys = np.empty(cap, np.uint64)
        n = C.c_size_t()
        self._check(self.lib.sfx_debug_pairs(self.handle, vptr(keys), cap, C.byref(n)))
        return keys[: min(n.value, cap)]

    def pair_colors(self):
        cap = max(int(self.stats().n_pairs), 1)
        keys = np.empty(cap, np.uint64)
        col = np.empty(cap, np.uint8)
        tiles = np.empty(cap, np.uint32)
        n = C.c_size_t()
        self._check(self.lib.sfx_debug_pair_colors(self.handle, vptr(keys), vptr(col), vptr(tiles), cap, C.byref(n)))
        k = min(n.value, cap)
        return keys[:k], col[:k], tiles[:k]

    def contacts(self):
        cap = max(int(self.stats().n_pairs), 1)
        out = np.zeros(cap, contact_dtype)
        n = C.c_size_t()
        self._check(self.lib.sfx_debug_contacts(self.handle, vptr(out), cap, C.byref(n)))
        return out[: min(n.value, cap)]

    # ---- pose read-back that overlaps the next tick (rendering)
    def readback_begin(self):
        self._check(self.lib.sfx_readback_begin(self.handle))

    def readback_wait(self, copy=True):
        """Oldest read-back begun: [n][4] = (x, y, cos, sin).  copy=False returns a view of the library's pinned
        buffer, valid until the second readback_begin after the one it belongs to."""
        ptr = C.POINTER(C.c_float)()
        n = C.c_uint32()
        self._check(self.lib.sfx_readback_wait(self.handle, C.byref(ptr), C.byref(n)))
        a = np.ctypeslib.as_array(ptr, shape=(n.value, 4)) if n.value else np.zeros((0, 4), np.float32)
        return a.copy() if copy else a

    def debug_near_batches(self):
        """(batches on the near list of the last tick's velocity sweep, batches in all); halo_overlap worlds."""
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self.lib.sfx_debug_near_batches(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def joint_order(self):
        cap = max(self.m, 1)
        perm = np.empty(cap, np.uint32)
        col = np.empty(cap, np.uint8)
        n = C.c_size_t()
        self._check(self.lib.sfx_debug_joint_order(self.handle, vptr(perm), vptr(col), cap, C.byref(n)))
        return perm[: n.value], col[: n.value]

    # ---- reporting
    def set_profiling(self, on):
        self._check(self.lib.sfx_set_profiling(self.handle, int(on)))

    def set_profiling_classes(self, names):
        """Events around the launches of the named kernel classes only ("solve_vel", "narrow", ...); nothing
        synchronises until kernel_times() collects them, so a timed run can carry it."""
        known = {self.lib.sfx_kernel_class_name(k).decode(): k for k in range(K_CLASSES)}
        mask = 0
        for n in names:
            mask |= 1 << known[n]
        self._check(self.lib.sfx_set_profiling(self.handle, 2 * mask))

    def kernel_times(self, reset=False):
        arr = (SfxKernelTime * K_CLASSES)()
        self._check(self.lib.sfx_kernel_times(self.handle, arr, int(reset)))
        out = {}
        for k in range(K_CLASSES):
            name = self.lib.sfx_kernel_class_name(k).decode()
            out[name] = dict(total_ms=arr[k].total_ms, launches=arr[k].launches, items=arr[k].items)
        return out

    def timer_begin(self):
        self._check(self.lib.sfx_timer_begin(self.handle))

    def timer_end(self):
        ms = C.c_double()
        self._check(self.lib.sfx_timer_end(self.handle, C.byref(ms)))
        return ms.value
