"""Slab domain decomposition of one scene over several worlds (BASELINE.json:5: "large scenes are
split across the 8 GPUs of one node by slab domain decomposition, exchanging boundary bodies each
substep with RCCL over xGMI").

The x axis is cut into `nranks` slabs holding equal numbers of dynamic bodies.  A rank's world
holds the bodies it owns, ghost copies of the neighbours' bodies within `halo` of its two
boundaries, and the non-dynamic bodies in reach.  Every substep the owners' (pose, velocity) of
the boundary bodies replace the neighbours' ghosts: 32 bytes per body to at most two peers, so
the exchange is latency-bound and there is no collective on the data path.

    plan_slabs      pure numpy: who owns what, who ghosts what, in which buffer order
    LocalSlabs      all slabs in one process (logical slabs on one GPU or on the CPU oracle)
    DistSlab        one slab per process over torch.distributed: "nccl" (= RCCL) moves the
                    library's device buffers directly, "gloo" goes through host memory

Both drivers work with any engine that has the World interface (starframe_amd.world.World on an
MI355X; the CPU oracle's OracleWorld in the CPU tests).
"""
from dataclasses import dataclass, field

import os

import numpy as np

from ._abi import SHAPE_CAPSULE, SHAPE_CIRCLE, SHAPE_POLY, SHAPE_RECT


def bounding_radius(scene):
    """Radius of a circle around the body origin that contains the collider."""
    c = scene.colliders
    p = c["p"]
    r = np.zeros(scene.n, np.float32)
    t = c["type"]
    r = np.where(t == SHAPE_CIRCLE, p[:, 2], r)
    r = np.where(t == SHAPE_RECT, np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2) + p[:, 2], r)
    r = np.where(t == SHAPE_CAPSULE, p[:, 0] + p[:, 2], r)
    if np.any(t == SHAPE_POLY):
        pv = np.asarray(scene.poly_verts, np.float32).reshape(-1, 2)
        vr = np.sqrt((pv ** 2).sum(1))
        for i in np.nonzero(t == SHAPE_POLY)[0]:
            o, k = int(c["poly_offset"][i]), int(c["poly_count"][i])
            r[i] = vr[o:o + k].max() + p[i, 2]
    if "local_pos" in (c.dtype.names or ()):   # a shape off the body origin reaches further from the body's centre
        r = r + np.sqrt((c["local_pos"].astype(np.float32) ** 2).sum(1))
    return r.astype(np.float32)


def x_half_extent(scene):
    """Half width of the collider's AABB at the current rotation (tight for boxes that never turn)."""
    c = scene.colliders
    p, t = c["p"], c["type"]
    co, si = np.abs(scene.rot[:, 0]), np.abs(scene.rot[:, 1])
    e = bounding_radius(scene)
    e = np.where(t == SHAPE_RECT, co * p[:, 0] + si * p[:, 1] + p[:, 2], e)
    e = np.where(t == SHAPE_CAPSULE, co * p[:, 0] + p[:, 2], e)
    if "local_pos" in (c.dtype.names or ()):
        off = np.sqrt((c["local_pos"].astype(np.float32) ** 2).sum(1))
        e = np.where(off > 0, bounding_radius(scene), e)   # turned and shifted: the bounding circle
    return e.astype(np.float32)


@dataclass
class SlabPlan:
    rank: int
    nranks: int
    lo: float                      # slab = [lo, hi) on x (owner of a dynamic body = slab of its centre)
    hi: float
    ids: np.ndarray                # global id of every local body: owned, ghosts (left, right), fixed
    n_owned: int
    send_left: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))   # local indices
    send_right: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    recv_left: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    recv_right: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))

    @property
    def send_idx(self):
        return np.concatenate([self.send_left, self.send_right]).astype(np.uint32)

    @property
    def recv_idx(self):
        return np.concatenate([self.recv_left, self.recv_right]).astype(np.uint32)


def default_halo(scene):
    """Two bodies touch only if their centres are closer than the sum of their bounding radii;
    twice the largest dynamic diameter plus the AABB padding also covers the second ring."""
    dyn = (scene.inv_mass > 0) | (scene.inv_inertia > 0)
    r = bounding_radius(scene)
    rmax = float(r[dyn].max()) if dyn.any() else 0.0
    return 4.0 * rmax + 4.0 * (scene.aabb_margin + 0.03)


def plan_slabs(scene, nranks, halo=None, only_rank=None):
    """Cuts `scene` into `nranks` slabs.  Returns (plans, boundaries).  only_rank: build that rank's plan alone (the
    others are None): what a rank of a distributed run needs -- the cuts are global, the lists are its own (at 4 M
    bodies and 8 ranks all plans take 4 s of numpy per rank, one takes 0.6 s)."""
    if (scene.constraints is not None and len(scene.constraints)) or len(getattr(scene, "angle_constraints", ())):
        raise ValueError("scenes with constraints are not decomposed (chains stay on one GPU)")
    halo = default_halo(scene) if halo is None else float(halo)
    x = scene.pos[:, 0].astype(np.float64)
    dyn = (scene.inv_mass > 0) | (scene.inv_inertia > 0)
    rad = bounding_radius(scene).astype(np.float64)
    xext = x_half_extent(scene).astype(np.float64)   # non-dynamic bodies do not turn
    xd = np.sort(x[dyn])
    if len(xd) < nranks:
        raise ValueError("fewer dynamic bodies than slabs")
    # equal-count cuts, each halfway between two neighbouring bodies
    cuts = []
    for k in range(1, nranks):
        j = (k * len(xd)) // nranks
        cuts.append(0.5 * (xd[j - 1] + xd[j]))
    bounds = np.array([-np.inf] + cuts + [np.inf])
    if nranks > 2 and np.min(np.diff(bounds[1:-1])) <= 2.0 * halo:
        raise ValueError("slabs thinner than two halo widths: use fewer ranks")
    owner = np.searchsorted(bounds, x, side="right") - 1
    owner = np.clip(owner, 0, nranks - 1)
    gid = np.arange(scene.n)
    plans = []
    wanted = range(nranks) if only_rank is None else (only_rank,)
    for r in range(nranks):
        if r not in wanted:
            plans.append(None)
            continue
        lo, hi = bounds[r], bounds[r + 1]
        owned = gid[dyn & (owner == r)]
        ghost_l = gid[dyn & (owner == r - 1) & (x + rad > lo - halo)] if r > 0 else gid[:0]
        ghost_r = gid[dyn & (owner == r + 1) & (x - rad < hi + halo)] if r + 1 < nranks else gid[:0]
        # anything further out must not reach this slab
        far = dyn & (np.abs(owner - r) > 1) & (x + rad > lo - halo) & (x - rad < hi + halo)
        if far.any():
            raise ValueError("a body two slabs away reaches into this slab's halo: use fewer ranks")
        fixed = gid[~dyn & (x + xext > lo - 2.0 * halo) & (x - xext < hi + 2.0 * halo)]
        ids = np.concatenate([owned, ghost_l, ghost_r, fixed])
        plans.append(SlabPlan(r, nranks, float(lo), float(hi), ids, len(owned)))
    # send lists mirror the neighbours' ghost lists (ascending global id on both sides)
    for r in wanted:
        p = plans[r]
        owned = p.ids[:p.n_owned]                       # ascending global ids: local index = position in this list
        no, nl = p.n_owned, 0
        if r > 0:
            gl = dyn & (owner == r - 1) & (x + rad > p.lo - halo)
            nl = int(gl.sum())
            p.recv_left = np.arange(no, no + nl, dtype=np.uint32)
            want = gid[dyn & (owner == r) & (x - rad < bounds[r] + halo)]     # what r-1 ghosts from r (its hi = this lo)
            p.send_left = np.searchsorted(owned, want).astype(np.uint32)
        if r + 1 < nranks:
            gr = dyn & (owner == r + 1) & (x - rad < p.hi + halo)
            p.recv_right = np.arange(no + nl, no + nl + int(gr.sum()), dtype=np.uint32)
            want = gid[dyn & (owner == r) & (x + rad > bounds[r + 1] - halo)] # what r+1 ghosts from r (its lo = this hi)
            p.send_right = np.searchsorted(owned, want).astype(np.uint32)
    return plans, bounds


def slab_counts(x, bounds, nranks):
    """How many of the bodies at `x` lie in each slab of `bounds` now (the per-frame count exchange)."""
    o = np.clip(np.searchsorted(bounds, np.asarray(x, np.float64), side="right") - 1, 0, nranks - 1)
    return np.bincount(o, minlength=nranks).astype(np.int64)


def needs_replan(counts, moved, halo, tol):
    """Re-cut when the slabs' populations differ by more than `tol` of the mean (10 %) or
    when a body has moved so far on x since the cut that the ghost lists may miss a contact: a body that
    was not ghosted lay more than `halo` = 4 r_max + pads beyond the boundary and touches nothing further
    than 2 r_max away, so the lists are good while no body has moved more than a quarter of the halo (where the
    halo guard stops the run); the re-cut is taken at an eighth, so that a check every few frames comes in time."""
    mean = counts.sum() / max(len(counts), 1)
    imbalance = float(counts.max() / mean - 1.0) if mean > 0 else 0.0
    return imbalance > tol or moved > 0.125 * halo, imbalance


class StaleHalo(RuntimeError):
    """A body has moved more than a quarter of the halo since the slabs were cut: ghost lists may miss contacts."""


class SlabError(RuntimeError):
    """A rank of a distributed slab run reported a capacity overflow or a stale halo: raised on EVERY rank at the
    same frame (the ranks agree by all-reduce), so that no rank is left waiting in a halo exchange for a peer that
    has stopped."""


def arm_guard(world, halo):
    """Device-side guard where the engine has one (libsfx: checked by every tick's first kernel, reported as
    SFX_ERR_CAPACITY bit 128 by the next sync or tick); returns False for engines that need the host check."""
    if hasattr(world, "halo_guard"):
        world.halo_guard(0.25 * halo)
        return True
    return False


def host_guard(world, plan, scene, halo):
    """The same check on the host, for engines without the device guard (the CPU oracle in the tests)."""
    k = plan.n_owned
    if k and float(np.abs(world.poses()[0][:k, 0] - scene.pos[plan.ids[:k], 0]).max()) > 0.25 * halo:
        raise StaleHalo("a body moved more than halo / 4 since the slabs were cut: rebalance (rebalance_every > 0)")


def local_scene(scene, plan):
    s = scene.subset(plan.ids, name=f"{scene.name}_slab{plan.rank}")
    return s


class LocalSlabs:
    """All slabs of a scene in one process: logical slabs on one GPU (device-to-device halo
    copies) or on any engine with halo_get / halo_put (host copies)."""

    def __init__(self, scene, nslabs, engine, halo=None, rebalance_every=0, imbalance_tol=0.10, overlap=False, **cfg_kw):
        """overlap: the halo travels while the velocity sweep's far batches run (sfx_halo_overlap; engines that have it)."""
        self.engine, self.nslabs, self.cfg_kw = engine, nslabs, cfg_kw
        self.overlap = bool(overlap)
        self.halo = default_halo(scene) if halo is None else float(halo)
        self.rebalance_every, self.imbalance_tol = int(rebalance_every), float(imbalance_tol)
        self.frames = self.replans = 0
        self.worlds = []
        self._build(scene)

    def _build(self, scene):
        self.scene = scene
        self.plans, self.bounds = plan_slabs(scene, self.nslabs, self.halo)
        self.worlds = [self.engine.from_scene(local_scene(scene, p), **self.cfg_kw) for p in self.plans]
        self.device_guard = True
        for w, p in zip(self.worlds, self.plans):
            w.halo_set(p.send_idx, p.recv_idx)
            self.device_guard = arm_guard(w, self.halo) and self.device_guard
            if self.overlap and hasattr(w, "halo_overlap"):
                w.halo_overlap(True)
        self.device_copies = all(hasattr(w, "halo_copy_from") for w in self.worlds)

    def rebalance(self, force=False):
        """Count exchange + migration, all slabs in one process: see DistSlab.rebalance."""
        counts, moved = np.zeros(self.nslabs, np.int64), 0.0
        for w, p in zip(self.worlds, self.plans):
            x = w.poses()[0][:p.n_owned, 0]
            counts += slab_counts(x, self.bounds, self.nslabs)
            if p.n_owned:
                moved = max(moved, float(np.abs(x - self.scene.pos[p.ids[:p.n_owned], 0]).max()))
        need, self.imbalance = needs_replan(counts, moved, self.halo, self.imbalance_tol)
        if not (need or force):
            return False
        state = self.gather()
        self.close()
        self._build(self.scene.with_state(*state))
        self.replans += 1
        return True

    def exchange(self):
        ws, ps = self.worlds, self.plans
        if self.device_copies:
            for w in ws:
                w.halo_pack()
            for r, (w, p) in enumerate(zip(ws, ps)):
                if r > 0:       # left neighbour's "to the right" segment -> my "from the left" segment
                    w.halo_copy_from(ws[r - 1], 0, len(ps[r - 1].send_left), len(p.recv_left))
                if r + 1 < len(ws):
                    w.halo_copy_from(ws[r + 1], len(p.recv_left), 0, len(p.recv_right))
            for w in ws:
                w.halo_unpack()
            return
        out = [w.halo_get() for w in ws]
        for r, (w, p) in enumerate(zip(ws, ps)):
            parts = []
            if r > 0:
                parts.append(out[r - 1][len(ps[r - 1].send_left):])
            if r + 1 < len(ws):
                parts.append(out[r + 1][:len(ps[r + 1].send_left)])
            w.halo_put(np.concatenate(parts) if parts else np.zeros((0, 8), np.float32))

    def step(self, dt):
        for w in self.worlds:
            w.tick_begin(dt)
        for _ in range(self.scene.substeps):
            for w in self.worlds:
                w.tick_substep()
            self.exchange()
        self.frames += 1
        if self.rebalance_every and self.frames % self.rebalance_every == 0:
            self.rebalance()
        elif not self.device_guard:   # never silently: without the device guard the host looks every frame
            for w, p in zip(self.worlds, self.plans):
                host_guard(w, p, self.scene, self.halo)

    def gather(self):
        """Global (pos, rot, vel, angvel): every body from its owner (fixed bodies from any copy)."""
        n = self.scene.n
        pos, rot = np.array(self.scene.pos, np.float32), np.array(self.scene.rot, np.float32)
        vel, ang = np.array(self.scene.vel, np.float32), np.array(self.scene.angvel, np.float32)
        for w, p in zip(self.worlds, self.plans):
            lp, lr = w.poses()
            lv, la = w.velocities()
            own = p.ids[:p.n_owned]
            pos[own], rot[own], vel[own], ang[own] = lp[:p.n_owned], lr[:p.n_owned], lv[:p.n_owned], la[:p.n_owned]
        assert pos.shape[0] == n
        return pos, rot, vel, ang

    def close(self):
        for w in self.worlds:
            w.close()


class _DeviceArray:
    """Lets torch view a device buffer owned by libsfx (no copy): __cuda_array_interface__."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {
            "shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None,
        }


class DistSlab:
    """One slab per process.  `dist` is an initialised torch.distributed module / group.

    backend "nccl" (RCCL): the library's send / receive buffers are wrapped as torch tensors and
    moved peer to peer on the stream the world runs on -- no host copy, no synchronisation.
    Any other backend ("gloo"): the halo goes through host memory (tests, CPU engines)."""

    def __init__(self, scene, engine, dist, halo=None, device_index=None, rebalance_every=0, imbalance_tol=0.10,
                 transport="auto", check_every=16, overlap=False, **cfg_kw):
        """transport (device path only): "native" = libsfx_rccl.so, plain rccl.h ncclSend / ncclRecv on the world's own
        stream through a communicator of its own (the path a host without torch uses; include/sfx_rccl.h);
        "torch" = torch.distributed batch_isend_irecv on tensors that alias the library's buffers; "auto" = native when it comes up on every rank, else torch (all ranks together, with a warning).
        The environment variable SFX_SLAB_TRANSPORT overrides the argument.
        check_every: every that many frames (and in sync / rebalance) the ranks all-reduce an error flag; a capacity
        overflow or a stale halo on ANY rank then raises SlabError on ALL of them at that frame.  Between two checks
        the rank that saw the condition keeps ticking (its results are void), so that its peers' exchanges complete."""
        import torch

        self.torch, self.dist = torch, dist
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()
        # the device every tensor of this driver lives on and the native communicator is made for: resolved ONCE, and made
        # torch's current device, so that no all-reduce lands on cuda:0 of every rank
        self.on_cuda = dist.get_backend() == "nccl"
        if self.on_cuda:
            if device_index is None:
                device_index = torch.cuda.current_device()
            torch.cuda.set_device(device_index)
        self.engine, self.device_index = engine, device_index
        self.check_every = int(check_every)
        # overlap: the exchange off the substep's critical path (sfx_halo_overlap): the velocity sweep in two launches, the
        # halo's pack and transfer on the library's halo stream behind the first.  Device transports only.
        self.overlap = bool(overlap)
        self.halo_stream = None
        self._local_error = ""
        self.halo = default_halo(scene) if halo is None else float(halo)
        self.rebalance_every, self.imbalance_tol = int(rebalance_every), float(imbalance_tol)
        self.frames = self.replans = 0
        self.imbalance = 0.0
        if device_index is not None:
            cfg_kw = dict(cfg_kw, device=device_index)
        self.cfg_kw = cfg_kw
        self.stream = None
        self.transport = transport
        self.rccl = self.comm = None
        self._build(scene)

    def _build(self, scene):
        torch, dist, device_index = self.torch, self.dist, self.device_index
        self.scene = scene
        plans, self.bounds = plan_slabs(scene, self.nranks, self.halo, only_rank=self.rank)
        self.plan = plans[self.rank]
        self.world = self.engine.from_scene(local_scene(scene, self.plan), **self.cfg_kw)
        self.world.halo_set(self.plan.send_idx, self.plan.recv_idx)
        self.device_guard = arm_guard(self.world, self.halo)
        self.on_device = dist.get_backend() == "nccl" and hasattr(self.world, "halo_buffer_pointers")
        self.halo_stream = None
        p = self.plan
        self.n_sl, self.n_sr, self.n_rl, self.n_rr = len(p.send_left), len(p.send_right), len(p.recv_left), len(p.recv_right)
        self.send = self.recv = None
        self.native = False
        transport = os.environ.get("SFX_SLAB_TRANSPORT", self.transport)
        if self.on_device and transport in ("auto", "native"):
            self.native = self._native_comm(required=transport == "native")
            if self.native and self.overlap and hasattr(self.world, "halo_overlap"):
                self.world.halo_overlap(True)   # sfx_rccl_halo_exchange enqueues on the halo stream by itself
        if self.on_device and not self.native:
            try:
                dev = torch.device("cuda", device_index)
                # one stream carries the world's kernels and orders the RCCL transfers against them
                if self.stream is None:
                    self.stream = torch.cuda.Stream(dev)
                self.world.set_stream(self.stream.cuda_stream)
                sp, rp = self.world.halo_buffer_pointers()
                ns, nr = 8 * (self.n_sl + self.n_sr), 8 * (self.n_rl + self.n_rr)
                self.send = torch.as_tensor(_DeviceArray(sp, ns), device=dev) if ns else torch.empty(0, device=dev)
                self.recv = torch.as_tensor(_DeviceArray(rp, nr), device=dev) if nr else torch.empty(0, device=dev)
                if self.overlap and hasattr(self.world, "halo_overlap"):
                    self.world.halo_overlap(True)   # torch's transfers then go on the library's halo stream
                    self.halo_stream = torch.cuda.ExternalStream(self.world.halo_stream(), device=dev)
            except Exception as e:  # the transfer then goes through host memory: slower, same results
                import warnings

                warnings.warn(f"device halo buffers unavailable ({e!r}); staging the halo through host memory")
                self.on_device = False
                self.world.set_stream(0)

    def _dev(self):
        """Where this driver's collective tensors live: its GPU under nccl, the host under gloo."""
        t = self.torch
        return t.device("cuda", self.device_index) if self.on_cuda else t.device("cpu")

    def _agree(self, ok):
        """True when every rank says so (one all-reduce): the ranks switch transport together or not at all."""
        t = self.torch
        flag = t.tensor([1 if ok else 0], dtype=t.int32, device=self._dev())
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item())

    def check(self):
        """The collective error decision: one all-reduce (MAX) of "this rank has seen a capacity overflow
        or a stale halo"; when any rank has, every rank raises SlabError here, together."""
        t = self.torch
        flag = t.tensor([1 if self._local_error else 0], dtype=t.int32, device=self._dev())
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        if int(flag.item()):
            raise SlabError(f"rank {self.rank} of {self.nranks}, frame {self.frames}: "
                            + (self._local_error or "another rank reported an overflow or a stale halo; this rank stops with it"))

    def _note(self, what):
        if not self._local_error:
            self._local_error = what

    def _native_comm(self, required):
        """A RCCL communicator of the library's own (once per driver; it outlives re-cuts): rank 0 makes the id, the
        existing process group carries its 128 bytes to the others.  Returns whether the native transport is up on
        every rank; when it is not and it was not `required`, all ranks use torch.distributed instead (with a warning):
        the ranks agree after each step that can fail on one of them alone, before the next collective one."""
        import ctypes as C
        import warnings

        from . import _abi

        if self.comm is not None:
            return True

        def settle(ok, what):
            if self._agree(ok):
                return True
            if required:
                raise RuntimeError(f"native slab transport: {what} failed on a rank: " + (err[0] or "see the other ranks"))
            warnings.warn(f"native slab transport unavailable ({what}: {err[0] or 'failed on another rank'}); using torch.distributed")
            return False

        err = [""]
        buf = (C.c_ubyte * _abi.RCCL_ID_BYTES)()
        try:
            self.rccl = _abi.load_libsfx_rccl()
            if self.rank == 0 and self.rccl.sfx_rccl_unique_id(buf) != 0:
                err[0] = "sfx_rccl_unique_id: " + self.rccl.sfx_rccl_last_error().decode()
        except OSError as e:
            err[0] = repr(e)
        if not settle(not err[0], "loading libsfx_rccl.so"):
            return False
        box = [bytes(buf)]
        self.dist.broadcast_object_list(box, src=0)
        comm = C.c_void_p()
        dev = self.device_index
        rc = self.rccl.sfx_rccl_comm_create((C.c_ubyte * _abi.RCCL_ID_BYTES).from_buffer_copy(box[0]), self.rank, self.nranks, dev,
                                            C.byref(comm))
        if rc != 0:
            err[0] = "sfx_rccl_comm_create: " + self.rccl.sfx_rccl_last_error().decode()
        if not settle(rc == 0, "creating the communicator"):
            if rc == 0:
                self.rccl.sfx_rccl_comm_destroy(comm)
            return False
        self.comm = comm
        return True

    def drift(self):
        """The per-frame count exchange: every rank bins its owned bodies' current x into the
        slabs, one all-reduce (SUM) of `nranks` counts and one (MAX) of the largest x displacement since the
        cut.  Returns (counts per slab, largest displacement), the same on every rank.  Reads the owned
        poses back, so it synchronises: call it every few frames, not every substep."""
        t, p = self.torch, self.plan
        x = self.world.poses()[0][:p.n_owned, 0]
        moved = float(np.abs(x - self.scene.pos[p.ids[:p.n_owned], 0]).max()) if p.n_owned else 0.0
        dev = self._dev()
        c = t.tensor(slab_counts(x, self.bounds, self.nranks), dtype=t.int64, device=dev)
        m = t.tensor([moved], dtype=t.float64, device=dev)
        self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(m, op=self.dist.ReduceOp.MAX)
        return c.cpu().numpy(), float(m.item())

    def rebalance(self, force=False):
        """Migration: when the populations differ by more than `imbalance_tol` or a body has
        moved a quarter of the halo since the cut, the ranks all-gather the owned states, cut the scene again
        at equal counts and rebuild their worlds -- bodies change owner, ghost and send lists are new.  Every
        rank decides on the same all-reduced figures.  The migration goes through host memory (a re-upload):
        it is a per-many-frames event, not on the substep path.  Returns True when the slabs were re-cut."""
        self.check()
        counts, moved = self.drift()
        need, self.imbalance = needs_replan(counts, moved, self.halo, self.imbalance_tol)
        if not (need or force):
            return False
        state = self.gather()
        self.send = self.recv = None   # views of the old world's buffers
        self.world.close()
        self._build(self.scene.with_state(*state))
        self.replans += 1
        return True

    def _ops(self, send, recv):
        d, r = self.dist, self.rank
        ops = []
        if r > 0:
            if self.n_sl:
                ops.append(d.P2POp(d.isend, send[: 8 * self.n_sl], r - 1))
            if self.n_rl:
                ops.append(d.P2POp(d.irecv, recv[: 8 * self.n_rl], r - 1))
        if r + 1 < self.nranks:
            if self.n_sr:
                ops.append(d.P2POp(d.isend, send[8 * self.n_sl:], r + 1))
            if self.n_rr:
                ops.append(d.P2POp(d.irecv, recv[8 * self.n_rl:], r + 1))
        return ops

    def exchange(self):
        if self.native:   # pack, grouped ncclSend / ncclRecv, unpack: all enqueued on the world's stream by libsfx_rccl
            left = self.rank - 1 if self.rank > 0 else -1
            right = self.rank + 1 if self.rank + 1 < self.nranks else -1
            rc = self.rccl.sfx_rccl_halo_exchange(self.comm, self.world.handle, left, right, self.n_sl, self.n_sr, self.n_rl, self.n_rr)
            if rc != 0:
                raise RuntimeError("sfx_rccl_halo_exchange: " + self.rccl.sfx_rccl_last_error().decode())
            return
        if self.on_device:
            self.world.halo_pack()
            ops = self._ops(self.send, self.recv)
            if ops:
                with self.torch.cuda.stream(self.halo_stream or self.stream):
                    for w in self.dist.batch_isend_irecv(ops):
                        w.wait()       # orders the world's stream after the transfer; no host wait with NCCL
            self.world.halo_unpack()
            return
        t = self.torch
        send = t.from_numpy(self.world.halo_get().reshape(-1))
        recv = t.empty(8 * (self.n_rl + self.n_rr), dtype=t.float32)
        ops = self._ops(send, recv)
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        self.world.halo_put(recv.numpy().reshape(-1, 8))

    def _local(self, call, *a):
        """An engine call whose capacity / halo errors are rank-local: noted, not raised -- this rank goes on ticking
        (the library refuses a tick only until sync has reported the condition once), so that the peers' exchanges
        complete; the next collective check stops all ranks together."""
        from .world import SfxError
        from ._abi import SFX_ERR_CAPACITY
        try:
            return call(*a)
        except SfxError as e:
            if e.code != SFX_ERR_CAPACITY:
                raise
            self._note(str(e))
        try:
            self.world.sync()      # reports the sticky condition (again) and clears it
        except SfxError as e:
            if e.code != SFX_ERR_CAPACITY:
                raise
        return call(*a)

    def step(self, dt):
        if self.check_every and self.frames % self.check_every == 0:
            self.check()
        self._local(self.world.tick_begin, dt)
        for _ in range(self.scene.substeps):
            self._local(self.world.tick_substep)
            self.exchange()
        self.frames += 1
        if self.rebalance_every and self.frames % self.rebalance_every == 0:
            self.rebalance()
        elif not self.device_guard:
            try:
                host_guard(self.world, self.plan, self.scene, self.halo)
            except StaleHalo as e:
                self._note(str(e))

    def sync(self):
        """Waits for this rank's work; then the ranks agree on whether any of them failed (SlabError on all)."""
        from .world import SfxError
        from ._abi import SFX_ERR_CAPACITY
        try:
            self.world.sync()
        except SfxError as e:
            if e.code != SFX_ERR_CAPACITY:
                raise
            self._note(str(e))
        self.check()

    def owned_state(self):
        p = self.plan
        lp, lr = self.world.poses()
        lv, la = self.world.velocities()
        k = p.n_owned
        return p.ids[:k], lp[:k], lr[:k], lv[:k], la[:k]

    def gather(self):
        """Global state on every rank (host all-gather; for tests and read-back, not per tick)."""
        ids, lp, lr, lv, la = self.owned_state()
        parts = [None] * self.nranks
        self.dist.all_gather_object(parts, (ids, lp, lr, lv, la))
        pos, rot = np.array(self.scene.pos, np.float32), np.array(self.scene.rot, np.float32)
        vel, ang = np.array(self.scene.vel, np.float32), np.array(self.scene.angvel, np.float32)
        for i, a, b, c, d in parts:
            pos[i], rot[i], vel[i], ang[i] = a, b, c, d
        return pos, rot, vel, ang

    def global_stats(self):
        """Whole-scene reporting figures on every rank (SURVEY.md 8e "all-reduce (stats)"): kinetic energy of
        the owned bodies and pair / manifold counts summed, deepest penetration maximised -- two all-reduces
        of a few bytes.  Pairs between an owned and a ghost body are counted on both sides."""
        t = self.torch
        ids, _, _, lv, la = self.owned_state()
        im, ii = np.asarray(self.scene.inv_mass)[ids], np.asarray(self.scene.inv_inertia)[ids]
        m = np.where(im > 0, 1.0 / np.where(im > 0, im, 1.0), 0.0)
        inertia = np.where(ii > 0, 1.0 / np.where(ii > 0, ii, 1.0), 0.0)
        ke = float((0.5 * m * (lv.astype(np.float64) ** 2).sum(1)).sum() + (0.5 * inertia * la.astype(np.float64) ** 2).sum())
        st = self.world.stats()
        dev = self._dev()
        sums = t.tensor([ke, float(st.n_pairs), float(st.n_manifolds), float(st.n_contacts), float(len(ids))],
                        dtype=t.float64, device=dev)
        mx = t.tensor([float(st.max_depth), float(st.n_colors)], dtype=t.float64, device=dev)
        self.dist.all_reduce(sums, op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(mx, op=self.dist.ReduceOp.MAX)
        sums, mx = sums.cpu().numpy(), mx.cpu().numpy()
        return {"kinetic_energy": float(sums[0]), "pairs": int(sums[1]), "manifolds": int(sums[2]), "contacts": int(sums[3]),
                "bodies": int(sums[4]), "max_depth": float(mx[0]), "colors": int(mx[1])}

    def close(self):
        self.world.close()
        if self.comm is not None:
            self.rccl.sfx_rccl_comm_destroy(self.comm)
            self.comm = None
