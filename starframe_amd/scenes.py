"""Synthetic scenes for the five BASELINE.json configs plus small
test scenes.  Pure numpy: a Scene is the set of flat arrays the C ABI takes, so the same
Scene feeds libsfx and the CPU oracle.

PRNG (SURVEY.md 8d "Common"): splitmix64(seed) -> uniform f32 from the top 24 bits.
"""
from dataclasses import dataclass, field

import numpy as np

from ._abi import (
    SHAPE_CAPSULE, SHAPE_CIRCLE, SHAPE_NONE, SHAPE_POLY, SHAPE_RECT, WORLD_BODY,
    angle_constraint_dtype, collider_dtype, constraint_dtype, material_dtype,
)

_M64 = (1 << 64) - 1


def splitmix64_uniform(seed, count):
    """count uniform f32 in [0, 1) from the splitmix64 stream started at `seed`."""
    with np.errstate(over="ignore"):
        i = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & _M64) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def as_colliders(arr):
    """A collider array of an older layout (fields missing from the end of the struct: local pose, body) in today's."""
    arr = np.asarray(arr)
    if arr.dtype == collider_dtype:
        return arr
    out = np.zeros(len(arr), collider_dtype)
    for f in arr.dtype.names:
        out[f] = arr[f]
    return out


@dataclass
class Scene:
    name: str
    pos: np.ndarray
    rot: np.ndarray
    vel: np.ndarray
    angvel: np.ndarray
    inv_mass: np.ndarray
    inv_inertia: np.ndarray
    colliders: np.ndarray
    materials: np.ndarray
    poly_verts: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))
    constraints: np.ndarray = field(default_factory=lambda: np.zeros(0, constraint_dtype))
    substeps: int = 4
    dt: float = 1.0 / 60.0
    gravity: tuple = (0.0, -9.81)
    aabb_margin: float = 0.01
    angle_constraints: np.ndarray = field(default_factory=lambda: np.zeros(0, angle_constraint_dtype))

    @property
    def n(self):
        return self.pos.shape[0]

    @property
    def n_colliders(self):
        """Bodies + further colliders (colliders[n:] name their body in the `body` field)."""
        return len(self.colliders)

    def n_dynamic(self):
        return int(np.count_nonzero((self.inv_mass > 0) | (self.inv_inertia > 0)))

    def with_state(self, pos, rot, vel=None, angvel=None, name=None):
        """Copy of the scene with another body state (e.g. the settled pile)."""
        import copy

        s = copy.copy(self)
        s.name = name or self.name
        s.pos = np.ascontiguousarray(pos, np.float32)
        s.rot = np.ascontiguousarray(rot, np.float32)
        s.vel = np.zeros_like(s.pos) if vel is None else np.ascontiguousarray(vel, np.float32)
        s.angvel = np.zeros(s.n, np.float32) if angvel is None else np.ascontiguousarray(angvel, np.float32)
        return s

    def subset(self, idx, name=None):
        """Scene restricted to bodies idx (constraints are dropped)."""
        idx = np.asarray(idx)
        if self.n_colliders != self.n:
            raise ValueError("subset of a scene with several colliders per body is not supported")
        return Scene(
            name or self.name, self.pos[idx].copy(), self.rot[idx].copy(), self.vel[idx].copy(),
            self.angvel[idx].copy(), self.inv_mass[idx].copy(), self.inv_inertia[idx].copy(),
            self.colliders[idx].copy(), self.materials.copy(), self.poly_verts.copy(),
            np.zeros(0, constraint_dtype), self.substeps, self.dt, self.gravity, self.aabb_margin,
        )


class SceneBuilder:
    """Accumulates bodies; mass properties follow from shape and density."""

    def __init__(self, name, substeps=4, materials=None):
        self.name = name
        self.substeps = substeps
        self.pos, self.rot, self.vel, self.angvel = [], [], [], []
        self.inv_mass, self.inv_inertia, self.col = [], [], []
        self.extra = []     # further colliders of bodies that already have one: (collider tuple with its body id last)
        self.poly_pool = []
        self.constraints = []
        self.angle_constraints = []
        self.materials = materials or [(0.1, 0.5, 0.4)]

    def _add(self, pos, angle, vel, angvel, mass, inertia, col):
        lp = col[8]
        if mass > 0 and (lp[0] or lp[1]):   # the shape sits off the body origin: parallel axis theorem
            inertia = inertia + mass * (lp[0] ** 2 + lp[1] ** 2)
        self.pos.append(pos)
        self.rot.append((np.cos(angle), np.sin(angle)))
        self.vel.append(vel)
        self.angvel.append(angvel)
        self.inv_mass.append(0.0 if mass <= 0 else 1.0 / mass)
        self.inv_inertia.append(0.0 if inertia <= 0 else 1.0 / inertia)
        self.col.append(col)
        return len(self.pos) - 1

    @staticmethod
    def _col(type_, p, material=0, category=1, mask=0xFFFFFFFF, group=0, poly_offset=0, poly_count=0, local_pos=(0.0, 0.0),
             local_angle=0.0):
        return (type_, material, category, mask, group, tuple(p), poly_offset, poly_count, tuple(local_pos), local_angle, 0)

    # ---- several colliders per body (include/sfx.h, sfx_collider.body)
    @staticmethod
    def _shape_mass(kind, dims, density):
        """(mass, inertia about the shape's own centre) of kind in ("circle", "rect", "capsule") with dims as the
        builder methods take them: (r,), (hw, hh[, rounding]), (hl, r)."""
        if kind == "circle":
            (r,) = dims
            m = density * np.pi * r * r
            return m, 0.5 * m * r * r
        if kind == "rect":
            hw, hh, rounding = (tuple(dims) + (0.0,))[:3]
            w, h = 2 * (hw + rounding), 2 * (hh + rounding)
            m = density * w * h
            return m, m * (w * w + h * h) / 12.0
        if kind == "capsule":
            hl, r = dims
            mr, mc = density * (2 * hl) * (2 * r), density * np.pi * r * r
            return mr + mc, mr * ((2 * hl) ** 2 + (2 * r) ** 2) / 12.0 + mc * (0.5 * r * r + hl * hl)
        raise ValueError(kind)

    def _shape_col(self, kind, dims, **kw):
        if kind == "circle":
            return self._col(SHAPE_CIRCLE, (0, 0, dims[0]), **kw)
        if kind == "rect":
            hw, hh, rounding = (tuple(dims) + (0.0,))[:3]
            return self._col(SHAPE_RECT, (hw, hh, rounding), **kw)
        if kind == "capsule":
            return self._col(SHAPE_CAPSULE, (dims[0], 0, dims[1]), **kw)
        raise ValueError(kind)

    def add_collider(self, body, kind, dims, local_pos=(0.0, 0.0), local_angle=0.0, density=None, **kw):
        """A further collider on an existing body.  density (dynamic bodies): the shape's mass and inertia (parallel
        axis about the body origin) are added to the body's; the body origin is taken to stay the centre of mass --
        lay the shapes out accordingly, or use compound(), which recentres."""
        col = self._shape_col(kind, dims, local_pos=local_pos, local_angle=local_angle, **kw)
        self.extra.append(col[:-1] + (int(body),))
        if density and self.inv_mass[body] > 0:
            m, i = self._shape_mass(kind, dims, density)
            i += m * (local_pos[0] ** 2 + local_pos[1] ** 2)
            self.inv_mass[body] = 1.0 / (1.0 / self.inv_mass[body] + m)
            self.inv_inertia[body] = 1.0 / (1.0 / self.inv_inertia[body] + i)
        return len(self.extra) - 1

    def compound(self, pos, parts, density=1.0, angle=0.0, vel=(0, 0), angvel=0.0, **kw):
        """One body of several shapes.  parts: [(kind, dims, local_pos, local_angle), ...] in any frame of the
        compound; `pos`, `angle` place that frame.  The body origin is moved to the centre of mass (every local_pos
        shifts by -com, the body sits at pos + R(angle) com); density 0 = a static compound (level geometry).
        Returns the body id."""
        ms = [self._shape_mass(k, d, density if density > 0 else 1.0) for k, d, _, _ in parts]
        mt = sum(m for m, _ in ms)
        com = np.sum([m * np.asarray(lp, float) for (m, _), (_, _, lp, _) in zip(ms, parts)], axis=0) / mt
        if density <= 0:
            com = np.zeros(2)
        inertia = sum(i + m * float(np.sum((np.asarray(lp, float) - com) ** 2)) for (m, i), (_, _, lp, _) in zip(ms, parts))
        c, s = np.cos(angle), np.sin(angle)
        origin = (pos[0] + c * com[0] - s * com[1], pos[1] + s * com[0] + c * com[1])
        k0, d0, lp0, la0 = parts[0]
        first = self._shape_col(k0, d0, local_pos=tuple(np.asarray(lp0, float) - com), local_angle=la0, **kw)
        self.pos.append(origin)
        self.rot.append((c, s))
        self.vel.append(vel)
        self.angvel.append(angvel)
        self.inv_mass.append(0.0 if density <= 0 else 1.0 / mt)
        self.inv_inertia.append(0.0 if density <= 0 else 1.0 / inertia)
        self.col.append(first)
        body = len(self.pos) - 1
        for k, d, lp, la in parts[1:]:
            self.add_collider(body, k, d, local_pos=tuple(np.asarray(lp, float) - com), local_angle=la, **kw)
        return body

    def circle(self, pos, r, density=1.0, angle=0.0, vel=(0, 0), angvel=0.0, **kw):
        m = density * np.pi * r * r
        return self._add(pos, angle, vel, angvel, m, 0.5 * m * r * r, self._col(SHAPE_CIRCLE, (0, 0, r), **kw))

    def rect(self, pos, hw, hh, density=1.0, angle=0.0, vel=(0, 0), angvel=0.0, rounding=0.0, **kw):
        w, h = 2 * (hw + rounding), 2 * (hh + rounding)
        m = density * w * h
        return self._add(pos, angle, vel, angvel, m, m * (w * w + h * h) / 12.0,
                         self._col(SHAPE_RECT, (hw, hh, rounding), **kw))

    def capsule(self, pos, hl, r, density=1.0, angle=0.0, vel=(0, 0), angvel=0.0, **kw):
        mr = density * (2 * hl) * (2 * r)
        mc = density * np.pi * r * r
        ir = mr * ((2 * hl) ** 2 + (2 * r) ** 2) / 12.0
        ic = mc * (0.5 * r * r + hl * hl)
        return self._add(pos, angle, vel, angvel, mr + mc, ir + ic, self._col(SHAPE_CAPSULE, (hl, 0, r), **kw))

    def poly(self, pos, verts, density=1.0, angle=0.0, vel=(0, 0), angvel=0.0, rounding=0.0, **kw):
        v = np.asarray(verts, np.float64)
        x, y = v[:, 0], v[:, 1]
        x1, y1 = np.roll(x, -1), np.roll(y, -1)
        cr = x * y1 - x1 * y
        area = 0.5 * cr.sum()
        cx = ((x + x1) * cr).sum() / (6 * area)
        cy = ((y + y1) * cr).sum() / (6 * area)
        v = v - (cx, cy)  # centre of mass at the body origin
        x, y = v[:, 0], v[:, 1]
        x1, y1 = np.roll(x, -1), np.roll(y, -1)
        cr = x * y1 - x1 * y
        inertia = density * ((x * x + x * x1 + x1 * x1 + y * y + y * y1 + y1 * y1) * cr).sum() / 12.0
        off = len(self.poly_pool)
        self.poly_pool.extend(v.tolist())
        return self._add(pos, angle, vel, angvel, density * area, inertia,
                          self._col(SHAPE_POLY, (0, 0, rounding), poly_offset=off, poly_count=len(v), **kw))

    def particle(self, pos, mass=1.0, vel=(0, 0)):
        """Body without collider."""
        return self._add(pos, 0.0, vel, 0.0, mass, 0.0, self._col(SHAPE_NONE, (0, 0, 0)))

    def static_rect(self, pos, hw, hh, angle=0.0, **kw):
        return self._add(pos, angle, (0, 0), 0.0, 0.0, 0.0, self._col(SHAPE_RECT, (hw, hh, 0.0), **kw))

    def distance(self, a, b, anchor_a=(0, 0), anchor_b=(0, 0), rest=None, compliance=0.0, limit=0, damping=0.0):
        if rest is None:
            pa = np.asarray(anchor_a, float) if a == WORLD_BODY else np.asarray(self.pos[a], float)
            pb = np.asarray(anchor_b, float) if b == WORLD_BODY else np.asarray(self.pos[b], float)
            rest = float(np.linalg.norm(pb - pa))
        self.constraints.append((a, b, tuple(anchor_a), tuple(anchor_b), rest, compliance, limit, damping))

    def attach(self, a, b, anchor_a=(0, 0), anchor_b=(0, 0), compliance=0.0, damping=0.0):
        """Pin: holds the two anchors together (a distance constraint of rest length 0)."""
        self.distance(a, b, anchor_a, anchor_b, rest=0.0, compliance=compliance, damping=damping)

    def angle(self, a, b, min_angle, max_angle, compliance=0.0, damping=0.0):
        """Keeps angle(b) - angle(a) inside [min_angle, max_angle]."""
        self.angle_constraints.append((a, b, min_angle, max_angle, compliance, damping))

    def rope(self, start, end, links, radius=0.04, density=1.0, pin_start=True, pin_end=False, compliance=0.0, slack=0.0, **kw):
        """Rope helper: `links` circle bodies evenly spaced from start to end, neighbours tied by
        one-sided (rope) distance links of the spacing's length (times 1 + slack); the ends optionally
        pinned to the world.  Links of one rope do not collide with each other.  Returns the body ids."""
        start, end = np.asarray(start, float), np.asarray(end, float)
        group = kw.pop("group", 0x52000000 + len(self.pos))
        ids = []
        for k in range(links):
            t = (k + 0.5) / links
            ids.append(self.circle(tuple(start + t * (end - start)), radius, density=density, group=group, **kw))
        seg = float(np.linalg.norm(end - start)) / links * (1.0 + slack)
        for k in range(links - 1):
            self.distance(ids[k], ids[k + 1], rest=seg, compliance=compliance, limit=1)
        if pin_start:
            self.distance(WORLD_BODY, ids[0], anchor_a=tuple(start), rest=0.5 * seg, compliance=compliance, limit=1)
        if pin_end:
            self.distance(ids[-1], WORLD_BODY, anchor_b=tuple(end), rest=0.5 * seg, compliance=compliance, limit=1)
        return ids

    def build(self):
        n = len(self.pos)
        f = lambda a, shape: np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape))
        return Scene(
            self.name, f(self.pos, (n, 2)), f(self.rot, (n, 2)), f(self.vel, (n, 2)), f(self.angvel, (n,)),
            f(self.inv_mass, (n,)), f(self.inv_inertia, (n,)),
            np.array(self.col + self.extra, dtype=collider_dtype), np.array(self.materials, dtype=material_dtype),
            f(self.poly_pool, (len(self.poly_pool), 2)), np.array(self.constraints, dtype=constraint_dtype),
            substeps=self.substeps,
            angle_constraints=np.array(self.angle_constraints, dtype=angle_constraint_dtype),
        )


def from_arrays(name, pos, angle, types, params, density=1.0, substeps=4, materials=None, statics=()):
    """Vectorised builder for big scenes: dynamic bodies from arrays, plus static rects
    given as (x, y, hw, hh) tuples appended at the end."""
    pos = np.asarray(pos, np.float32)
    n = pos.shape[0]
    ns = len(statics)
    params = np.asarray(params, np.float64)
    a, b, r = params[:, 0], params[:, 1], params[:, 2]
    mass = np.zeros(n)
    inertia = np.zeros(n)
    c = types == SHAPE_CIRCLE
    mass[c] = density * np.pi * r[c] ** 2
    inertia[c] = 0.5 * mass[c] * r[c] ** 2
    q = types == SHAPE_RECT
    w, h = 2 * (a[q] + r[q]), 2 * (b[q] + r[q])
    mass[q] = density * w * h
    inertia[q] = mass[q] * (w * w + h * h) / 12.0
    k = types == SHAPE_CAPSULE
    mr = density * (2 * a[k]) * (2 * r[k])
    mc = density * np.pi * r[k] ** 2
    mass[k] = mr + mc
    inertia[k] = mr * ((2 * a[k]) ** 2 + (2 * r[k]) ** 2) / 12.0 + mc * (0.5 * r[k] ** 2 + a[k] ** 2)
    col = np.zeros(n + ns, collider_dtype)
    col["type"][:n] = types
    col["p"][:n] = params.astype(np.float32)
    col["category"] = 1
    col["mask"] = 0xFFFFFFFF
    P = np.zeros((n + ns, 2), np.float32)
    P[:n] = pos
    R = np.zeros((n + ns, 2), np.float32)
    R[:n, 0] = np.cos(angle)
    R[:n, 1] = np.sin(angle)
    R[n:, 0] = 1.0
    im = np.zeros(n + ns, np.float32)
    ii = np.zeros(n + ns, np.float32)
    im[:n] = 1.0 / mass
    ii[:n] = 1.0 / inertia
    for s, (x, y, hw, hh) in enumerate(statics):
        P[n + s] = (x, y)
        col["type"][n + s] = SHAPE_RECT
        col["p"][n + s] = (hw, hh, 0.0)
    mats = np.array(materials or [(0.1, 0.5, 0.4)], dtype=material_dtype)
    return Scene(name, P, R, np.zeros((n + ns, 2), np.float32), np.zeros(n + ns, np.float32), im, ii,
                 col, mats, substeps=substeps)


# ------------------------------------------------------------------ the five configs

def pyramid(base=31, substeps=10):
    """Config 1 (BASELINE.json:7): box pyramid, rows base..1 of unit boxes on a static ground."""
    b = SceneBuilder("pyramid", substeps=substeps)
    for row in range(base):
        cnt = base - row
        for j in range(cnt):
            b.rect(((j - (cnt - 1) / 2.0) * 1.05, 0.5 + row * 1.0), 0.5, 0.5)
    b.static_rect((0.0, -0.5), 40.0, 0.5)
    return b.build()


def _container(floor_hw, wall_hh, t=5.0):
    """Floor (top at y = 0) and two walls (inner faces at x = +-floor_hw), each 2 t thick so that
    fast bodies cannot tunnel through them between two broadphase passes."""
    return [(0.0, -t, floor_hw + 2 * t, t), (-floor_hw - t, wall_hh, t, wall_hh + t), (floor_hw + t, wall_hh, t, wall_hh + t)]


def _lattice(n, width, pitch, jitter, u, y0=0.3):
    k = np.arange(n)
    x = ((k % width) - (width - 1) / 2.0) * pitch + (u[0::2][:n] * 2 - 1) * jitter
    y = y0 + (k // width) * pitch + (u[1::2][:n] * 2 - 1) * jitter
    return np.stack([x, y], 1)


def circles_container(n=10_000, seed=0x5EED0002, width=100, pitch=0.22, floor_hw=24.0, wall_hh=30.0, substeps=4):
    """Config 2 (BASELINE.json:8): n circles r~U[0.05,0.10] on a jittered lattice above a
    static container of three rects."""
    u = splitmix64_uniform(seed, 3 * n)
    r = 0.05 + 0.05 * u[:n].astype(np.float64)
    pos = _lattice(n, width, pitch, 0.01, u[n:])
    types = np.full(n, SHAPE_CIRCLE, np.uint32)
    params = np.stack([np.zeros(n), np.zeros(n), r], 1)
    s = from_arrays("circles_container", pos, np.zeros(n), types, params, substeps=substeps,
                    statics=_container(floor_hw, wall_hh))
    return s


def mixed_pile(n=100_000, seed=0x5EED0003, width=1000, pitch=0.30, floor_hw=None, wall_hh=40.0, substeps=4):
    """Config 3, the headline (BASELINE.json:9): 40 % circles, 30 % rects, 30 % capsules,
    random initial rotation, on a lattice above the container.  The lattice is 1000 bodies
    wide (SURVEY.md 8d guessed 400) so that the settled pile is ~50 layers deep: XPBD with
    4 substeps cannot carry a deeper 2D pile without compressing it (Scenes)."""
    if floor_hw is None:
        floor_hw = width * pitch / 2.0 + 2.0
    u = splitmix64_uniform(seed, 7 * n)
    kind, u1, u2, u3, ang = u[:n], u[n:2 * n], u[2 * n:3 * n], u[3 * n:4 * n], u[4 * n:5 * n]
    types = np.where(kind < 0.4, SHAPE_CIRCLE, np.where(kind < 0.7, SHAPE_RECT, SHAPE_CAPSULE)).astype(np.uint32)
    params = np.zeros((n, 3))
    c, q, k = types == SHAPE_CIRCLE, types == SHAPE_RECT, types == SHAPE_CAPSULE
    params[c, 2] = 0.05 + 0.05 * u1[c]
    params[q, 0] = 0.05 + 0.05 * u1[q]
    params[q, 1] = 0.05 + 0.05 * u2[q]
    params[k, 0] = 0.05 + 0.05 * u1[k]
    params[k, 2] = 0.03 + 0.03 * u3[k]
    pos = _lattice(n, width, pitch, 0.01, u[5 * n:])
    return from_arrays("mixed_pile", pos, ang.astype(np.float64) * 2 * np.pi, types, params,
                       substeps=substeps, statics=_container(floor_hw, wall_hh))


def chains(n_chains=1000, links=100, r=0.05, rest=0.12, substeps=4):
    """Config 4 (BASELINE.json:10): chains of circle links joined by rigid distance
    constraints, link 0 tied to a world anchor; chain k starts horizontal at (0.5 k, 0.3 k).
    Links of one chain share a collision group (no self collision)."""
    n = n_chains * links
    ch = np.repeat(np.arange(n_chains), links)
    li = np.tile(np.arange(links), n_chains)
    pos = np.stack([0.5 * ch + rest * (li + 1), 0.3 * ch], 1)
    types = np.full(n, SHAPE_CIRCLE, np.uint32)
    params = np.stack([np.zeros(n), np.zeros(n), np.full(n, r)], 1)
    s = from_arrays("chains", pos, np.zeros(n), types, params, substeps=substeps)
    s.colliders["group"] = ch + 1
    con = np.zeros(n, constraint_dtype)
    idx = np.arange(n)
    con["body_b"] = idx
    con["body_a"] = np.where(li == 0, WORLD_BODY, idx - 1)
    con["anchor_a"][li == 0] = np.stack([0.5 * np.arange(n_chains), 0.3 * np.arange(n_chains)], 1)
    con["rest_length"] = rest
    s.constraints = con
    return s


def granular_slab(n=4_000_000, seed=0x5EED0005, width=40000, pitch=0.22, wall_hh=40.0, substeps=4):
    """Config 5 (BASELINE.json:11): n circles on a `width`-wide lattice over a long floor
    (100 rows at the default sizes, for the same reason as mixed_pile)."""
    u = splitmix64_uniform(seed, 3 * n)
    r = 0.05 + 0.05 * u[:n].astype(np.float64)
    pos = _lattice(n, width, pitch, 0.01, u[n:])
    floor_hw = width * pitch / 2.0 + 2.0
    types = np.full(n, SHAPE_CIRCLE, np.uint32)
    params = np.stack([np.zeros(n), np.zeros(n), r], 1)
    return from_arrays("granular_slab", pos, np.zeros(n), types, params, substeps=substeps,
                       statics=_container(floor_hw, wall_hh))


# ------------------------------------------------------------------ small test scenes

def random_soup(n=300, seed=1, extent=4.0, with_poly=True, with_statics=True, speed=1.0, offsets=0.0, compounds=0.0, platforms=0, joints=0.0):
    """Overlapping random shapes of every type with random velocities: a broadphase and
    narrowphase stress scene for parity tests.  offsets: fraction of bodies whose shape has a local pose;
    compounds: fraction of bodies that get one or two further colliders (and one more on every static);
    platforms: that many mid-size static rects (0.4 .. 1.6 m half extents) strewn over the scene and a tenth as many
    mid-size DYNAMIC boxes -- the colliders the second level of the broadphase grid is for; joints: fraction of bodies that
    get a random constraint (distance link, spring, rope, world pin, hinge with an angle range, world angle lock)."""
    rng = np.random.default_rng(seed)
    b = SceneBuilder(f"soup{n}_{seed}", materials=[(0.1, 0.5, 0.4), (0.6, 0.8, 0.6), (0.0, 0.2, 0.1)])
    for i in range(n):
        p = rng.uniform(-extent, extent, 2)
        ang = rng.uniform(0, 2 * np.pi)
        v = rng.normal(0, speed, 2)
        w = rng.normal(0, 2.0)
        t = rng.integers(0, 4 if with_poly else 3)
        kw = dict(angle=ang, vel=v, angvel=w, material=int(rng.integers(0, 3)))
        if rng.random() < 0.1:
            kw["group"] = int(rng.integers(1, 4))
        if rng.random() < 0.1:
            kw["category"], kw["mask"] = 2, 0xFFFFFFFD
        if offsets and rng.random() < offsets:   # the shape off the body origin, turned in the body frame
            kw["local_pos"] = tuple(rng.uniform(-0.15, 0.15, 2))
            kw["local_angle"] = float(rng.uniform(-np.pi, np.pi)) if rng.random() < 0.7 else 0.0
        if t == 0:
            b.circle(p, rng.uniform(0.05, 0.2), **kw)
        elif t == 1:
            b.rect(p, rng.uniform(0.05, 0.2), rng.uniform(0.05, 0.2), rounding=rng.choice([0.0, 0.02]), **kw)
        elif t == 2:
            b.capsule(p, rng.uniform(0.05, 0.2), rng.uniform(0.03, 0.08), **kw)
        else:
            k = int(rng.integers(3, 9))
            a = np.sort(rng.uniform(0, 2 * np.pi, k))
            a = a + np.linspace(0, 1e-3, k)
            rad = rng.uniform(0.08, 0.2)
            b.poly(p, np.stack([rad * np.cos(a), rad * np.sin(a)], 1), rounding=rng.choice([0.0, 0.015]), **kw)
    if with_statics:
        b.static_rect((0.0, -extent - 0.5), extent + 2.0, 0.5)
        b.static_rect((-extent - 1.0, 0.0), 0.5, extent + 1.0)
        b.static_rect((extent + 1.0, 0.0), 0.5, extent + 1.0, angle=0.1)
    if platforms:
        prng = np.random.default_rng(seed + 104729)
        for _ in range(platforms):
            b.static_rect(tuple(prng.uniform(-extent, extent, 2)), prng.uniform(0.4, 1.6), prng.uniform(0.1, 0.5),
                          angle=float(prng.uniform(-0.5, 0.5)))
        for _ in range(max(platforms // 10, 1)):
            b.rect(tuple(prng.uniform(-extent, extent, 2)), prng.uniform(0.5, 0.9), prng.uniform(0.4, 0.8),
                   vel=tuple(prng.normal(0, speed, 2)), angvel=float(prng.normal(0, 1.0)))
    if compounds:
        crng = np.random.default_rng(seed + 7919)
        for i in range(n):
            if b.col[i][0] == SHAPE_NONE or crng.random() >= compounds:
                continue
            for _ in range(int(crng.integers(1, 3))):
                lp = tuple(crng.uniform(-0.25, 0.25, 2))
                la = float(crng.uniform(-np.pi, np.pi))
                kw = dict(material=int(crng.integers(0, 3)), group=b.col[i][4], category=b.col[i][2], mask=b.col[i][3])
                if crng.random() < 0.5:
                    b.add_collider(i, "circle", (crng.uniform(0.04, 0.15),), lp, 0.0, density=1.0, **kw)
                elif crng.random() < 0.5:
                    b.add_collider(i, "rect", (crng.uniform(0.04, 0.15), crng.uniform(0.04, 0.15)), lp, la, density=1.0, **kw)
                else:
                    b.add_collider(i, "capsule", (crng.uniform(0.05, 0.15), crng.uniform(0.03, 0.06)), lp, la, density=1.0, **kw)
        if with_statics:   # level geometry as compounds: a ledge on the floor body, a step on a wall body
            b.add_collider(n, "rect", (0.8, 0.15), (1.0, 0.65), 0.0)
            b.add_collider(n + 1, "rect", (0.3, 0.3), (0.8, -1.0), 0.3)
            b.add_collider(n, "circle", (1.2,), (-2.0, 0.2), 0.0)    # a large further collider
    if joints:
        jrng = np.random.default_rng(seed + 15485863)
        P = np.asarray(b.pos[:n], float)
        for i in range(n):
            if jrng.random() >= joints:
                continue
            d = np.linalg.norm(P - P[i], axis=1)
            d[i] = np.inf
            j = int(np.argsort(d)[int(jrng.integers(0, 3))])   # one of the three nearest bodies
            anchors = dict(anchor_a=tuple(jrng.uniform(-0.1, 0.1, 2)), anchor_b=tuple(jrng.uniform(-0.1, 0.1, 2)))
            kind = int(jrng.integers(0, 6))
            if kind == 0:      # rigid distance link at the current distance
                b.distance(i, j)
            elif kind == 1:    # soft spring with damping, anchors off the origins
                b.distance(i, j, rest=float(d[j]) * 0.8, compliance=float(jrng.uniform(1e-4, 1e-2)),
                           damping=float(jrng.uniform(0.0, 2.0)), **anchors)
            elif kind == 2:    # rope (one-sided), slack or taut
                b.distance(i, j, rest=float(d[j]) * float(jrng.uniform(0.7, 1.3)), limit=1, **anchors)
            elif kind == 3:    # pin to the world
                b.distance(WORLD_BODY, i, anchor_a=tuple(P[i] + jrng.uniform(-0.2, 0.2, 2)), anchor_b=anchors["anchor_b"],
                           rest=float(jrng.uniform(0.0, 0.3)), compliance=float(jrng.choice([0.0, 1e-3])))
            elif kind == 4:    # hinge: pin plus an angle range
                b.attach(i, j, anchor_a=tuple(0.5 * (P[j] - P[i])), anchor_b=tuple(0.5 * (P[i] - P[j])))
                b.angle(i, j, -0.4, 0.6, compliance=float(jrng.choice([0.0, 1e-3])), damping=float(jrng.choice([0.0, 0.5])))
            else:              # angle lock against the world
                b.angle(WORLD_BODY, i, -0.2, 0.2)
    return b.build()
