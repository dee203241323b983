"""Python handle over the CPU oracle (oracle/libsfx_oracle.so).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/sfx_oracle.h): this checks libsfx against this repository's own
CPU restatement of the published XPBD algorithm, not against starframe's Rust code, which
was unavailable.  Import this only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from starframe_amd import _abi
from starframe_amd._abi import SfxConfig, contact_dtype, fptr, make_config, vptr
from starframe_amd.world import WorldBase

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsfx_oracle.so")
OMP_LIB_PATH = os.path.join(HERE, "libsfx_oracle_omp.so")   # the same source with -fopenmp: same bits, several threads

ORDER_SEQUENTIAL = 0
ORDER_COLORED = 1

_libs = {}


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def load(openmp=False):
    if openmp in _libs:
        return _libs[openmp]
    path = OMP_LIB_PATH if openmp else LIB_PATH
    src_newer = (not os.path.exists(path)) or any(
        os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(path)
        for f in ("sfx_oracle.c", "sfx_oracle.h")
    )
    if src_newer:
        build()
    lib = C.CDLL(path)
    P, vp = C.POINTER, C.c_void_p
    lib.sfo_create.argtypes = [P(SfxConfig)]
    lib.sfo_create.restype = vp
    lib.sfo_destroy.argtypes = [vp]
    lib.sfo_destroy.restype = None
    _abi.bind_world_api(lib, "sfo_")
    lib.sfo_resolved_config.argtypes = [vp, P(SfxConfig)]
    lib.sfo_resolved_config.restype = None
    lib.sfo_step.argtypes = [vp, C.c_float, C.c_int]
    lib.sfo_step.restype = C.c_int
    lib.sfo_tick_begin.argtypes = [vp, C.c_float, C.c_int]
    lib.sfo_tick_begin.restype = C.c_int
    lib.sfo_tick_substep.argtypes = [vp]
    lib.sfo_tick_substep.restype = C.c_int
    lib.sfo_halo_get.argtypes = [vp, vp, C.c_uint32, P(C.c_float)]
    lib.sfo_halo_get.restype = C.c_int
    lib.sfo_halo_put.argtypes = [vp, vp, C.c_uint32, P(C.c_float)]
    lib.sfo_halo_put.restype = C.c_int
    lib.sfo_broadphase.argtypes = [vp, C.c_float]
    lib.sfo_broadphase.restype = C.c_int
    lib.sfo_aabbs.argtypes = [vp, P(C.c_float)]
    for n in ("sfo_pairs_grid", "sfo_pairs_sweep", "sfo_pairs_brute"):
        getattr(lib, n).argtypes = [vp, vp, C.c_size_t]
        getattr(lib, n).restype = C.c_size_t
    lib.sfo_pair_colors.argtypes = [vp, vp, vp, vp, C.c_size_t]
    lib.sfo_pair_colors.restype = C.c_size_t
    lib.sfo_contacts.argtypes = [vp, vp, C.c_size_t]
    lib.sfo_contacts.restype = C.c_size_t
    lib.sfo_contact_pair.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    lib.sfo_contact_pair.restype = C.c_int
    lib.sfo_joint_order.argtypes = [vp, vp, vp, C.c_size_t]
    lib.sfo_joint_order.restype = C.c_size_t
    lib.sfo_query_points.argtypes = [vp, C.c_uint32, P(C.c_float), C.c_uint32, vp, vp, C.c_int]
    lib.sfo_query_points.restype = C.c_int
    lib.sfo_query_aabbs.argtypes = [vp, C.c_uint32, P(C.c_float), C.c_uint32, vp, C.c_size_t, C.c_int]
    lib.sfo_query_aabbs.restype = C.c_size_t
    lib.sfo_query_rays.argtypes = [vp, C.c_uint32, P(C.c_float), P(C.c_float), C.c_uint32, vp, C.c_int]
    lib.sfo_query_rays.restype = C.c_int
    lib.sfo_threads.restype = C.c_int
    _libs[openmp] = lib
    return lib


class OracleWorld(WorldBase):
    prefix = "sfo_"

    def __init__(self, cfg=None, openmp=False, **cfg_kw):
        self.lib = load(openmp)
        self.cfg = cfg or make_config(**cfg_kw)
        self.handle = C.c_void_p(self.lib.sfo_create(C.byref(self.cfg)))
        if not self.handle:
            raise MemoryError("sfo_create")

    @classmethod
    def from_scene(cls, scene, **cfg_kw):
        kw = dict(substeps=scene.substeps, gravity=scene.gravity, aabb_margin=scene.aabb_margin)
        kw.update(cfg_kw)
        return cls(**kw).upload(scene)

    def close(self):
        if self.handle:
            self.lib.sfo_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resolved_config(self):
        out = SfxConfig()
        self.lib.sfo_resolved_config(self.handle, C.byref(out))
        return out

    def step(self, dt, order=ORDER_COLORED):
        self._check(self.lib.sfo_step(self.handle, dt, order))

    def sync(self):
        pass

    # ---- the tick in pieces + slab halos, same interface as starframe_amd.world.World
    def tick_begin(self, dt, order=ORDER_COLORED):
        self._check(self.lib.sfo_tick_begin(self.handle, dt, order))

    def tick_substep(self):
        self._check(self.lib.sfo_tick_substep(self.handle))

    def halo_set(self, send_idx, recv_idx):
        self._send_idx = np.ascontiguousarray(send_idx, np.uint32)
        self._recv_idx = np.ascontiguousarray(recv_idx, np.uint32)
        self.n_halo_send, self.n_halo_recv = len(self._send_idx), len(self._recv_idx)

    def halo_get(self):
        out = np.empty((self.n_halo_send, 8), np.float32)
        self._check(self.lib.sfo_halo_get(self.handle, vptr(self._send_idx), self.n_halo_send, fptr(out)))
        return out

    def halo_put(self, arr):
        a = np.ascontiguousarray(arr, np.float32)
        assert a.shape == (self.n_halo_recv, 8)
        self._check(self.lib.sfo_halo_put(self.handle, vptr(self._recv_idx), self.n_halo_recv, fptr(a)))

    # ---- spatial queries: brute=True walks all colliders, else the grid index (must agree exactly)
    def query_points(self, points, mask=0xFFFFFFFF, brute=False):
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        body = np.empty(len(p), np.uint32)
        count = np.empty(len(p), np.uint32)
        self._check(self.lib.sfo_query_points(self.handle, len(p), fptr(p), mask, vptr(body), vptr(count), int(brute)))
        return body, count

    def query_aabbs(self, boxes, mask=0xFFFFFFFF, brute=False):
        b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
        cap = max(16 * len(b), 1024)
        while True:
            hits = np.empty(cap, np.uint64)
            n = self.lib.sfo_query_aabbs(self.handle, len(b), fptr(b), mask, vptr(hits), cap, int(brute))
            if n > cap:
                cap = n
                continue
            return hits[:n]

    def query_rays(self, rays, max_t, mask=0xFFFFFFFF, brute=False):
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 4)
        mt = np.ascontiguousarray(np.broadcast_to(np.asarray(max_t, np.float32), (len(r),)))
        out = np.zeros(len(r), _abi.ray_hit_dtype)
        self._check(self.lib.sfo_query_rays(self.handle, len(r), fptr(r), fptr(mt), mask, vptr(out), int(brute)))
        return out

    def broadphase(self, dt):
        self._check(self.lib.sfo_broadphase(self.handle, dt))

    def aabbs(self):
        out = np.empty((self.nc, 4), np.float32)   # one box per collider
        self.lib.sfo_aabbs(self.handle, fptr(out))
        return out

    def _pairs(self, fn):
        cap = max(int(self.stats().n_pairs), 1024)
        while True:
            keys = np.empty(cap, np.uint64)
            n = getattr(self.lib, fn)(self.handle, vptr(keys), cap)
            if n <= cap:
                return keys[:n]
            cap = n

    def pairs(self):
        return self._pairs("sfo_pairs_grid")

    def pairs_sweep(self):
        return self._pairs("sfo_pairs_sweep")

    def pairs_brute(self):
        return self._pairs("sfo_pairs_brute")

    def pair_colors(self):
        cap = max(int(self.stats().n_pairs), 1)
        keys = np.empty(cap, np.uint64)
        col = np.empty(cap, np.uint8)
        tiles = np.empty(cap, np.uint32)
        n = self.lib.sfo_pair_colors(self.handle, vptr(keys), vptr(col), vptr(tiles), cap)
        return keys[:n], col[:n], tiles[:n]

    def contacts(self):
        cap = max(int(self.stats().n_pairs), 1)
        out = np.zeros(cap, contact_dtype)
        n = self.lib.sfo_contacts(self.handle, vptr(out), cap)
        return out[:n]

    def contact_pair(self, a, b):
        out = np.zeros(1, contact_dtype)
        self._check(self.lib.sfo_contact_pair(self.handle, a, b, vptr(out)))
        return out[0]

    def joint_order(self):
        cap = max(self.m, 1)
        perm = np.empty(cap, np.uint32)
        col = np.empty(cap, np.uint8)
        n = self.lib.sfo_joint_order(self.handle, vptr(perm), vptr(col), cap)
        return perm[:n], col[:n]
