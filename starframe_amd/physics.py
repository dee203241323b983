"""Host-side mirror of starframe's physics API shape over the libsfx C ABI (Python harness).

BASELINE.json:5 names the API surface that stays on the host: a physics config, body / collider /
material components and the per-frame tick entry point.  Upstream's exact names are not
verifiable (its source is absent), so these follow SURVEY.md's [RECALL] hints.  The C++
equivalent is starframe_amd/cpp/starframe_physics.hpp; the Rust one bindings/rust/.
"""
from dataclasses import dataclass

import numpy as np

from ._abi import WORLD_BODY
from .scenes import SceneBuilder
from .world import World


@dataclass
class PhysicsConfig:
    substeps: int = 4
    gravity: tuple = (0.0, -9.81)
    aabb_margin: float = 0.01
    max_depenetration_velocity: float = 0.0   # 0 = unlimited
    device: int = 0


@dataclass(frozen=True)
class Material:
    restitution: float = 0.1
    static_friction: float = 0.5
    dynamic_friction: float = 0.4


@dataclass
class Collider:
    kind: str
    a: float = 0.0
    b: float = 0.0
    r: float = 0.0
    verts: tuple = ()
    material: Material = Material()
    category: int = 1
    mask: int = 0xFFFFFFFF
    group: int = 0
    offset: tuple = (0.0, 0.0)   # the shape's origin in the body frame (the reference's collider "local offset")
    local_angle: float = 0.0     # and its rotation there

    def with_offset(self, offset, angle=0.0):
        from dataclasses import replace
        return replace(self, offset=tuple(offset), local_angle=float(angle))

    @staticmethod
    def circle(r, **kw):
        return Collider("circle", r=r, **kw)

    @staticmethod
    def rect(width, height, rounding=0.0, **kw):
        return Collider("rect", a=0.5 * width - rounding, b=0.5 * height - rounding, r=rounding, **kw)

    @staticmethod
    def capsule(length, radius, **kw):
        return Collider("capsule", a=0.5 * length, r=radius, **kw)

    @staticmethod
    def polygon(verts, rounding=0.0, **kw):
        return Collider("poly", r=rounding, verts=tuple(map(tuple, verts)), **kw)


@dataclass
class Body:
    kind: str
    density: float = 1.0

    @staticmethod
    def dynamic(collider=None, density=1.0):
        return Body("dynamic", density)

    @staticmethod
    def static():
        return Body("static", 0.0)


class Physics:
    """Collects components, uploads once, then `tick(dt)` per frame."""

    def __init__(self, config=None):
        self.config = config or PhysicsConfig()
        self._b = SceneBuilder("physics", substeps=self.config.substeps, materials=[])
        self._mats = {}
        self._world = None

    def _mat(self, m):
        if m not in self._mats:
            self._mats[m] = len(self._b.materials)
            self._b.materials.append((m.restitution, m.static_friction, m.dynamic_friction))
        return self._mats[m]

    def add(self, body, collider, position, angle=0.0, velocity=(0.0, 0.0), angular_velocity=0.0):
        if self._world is not None:
            raise RuntimeError("add bodies before the first tick")
        kw = dict(angle=angle, material=self._mat(collider.material), category=collider.category,
                  mask=collider.mask, group=collider.group, local_pos=collider.offset, local_angle=collider.local_angle)
        dens = body.density if body.kind == "dynamic" else 0.0
        dyn = dict(density=dens, vel=velocity, angvel=angular_velocity)
        if collider.kind == "circle":
            return self._b.circle(position, collider.r, **dyn, **kw)
        if collider.kind == "rect":
            return self._b.rect(position, collider.a, collider.b, rounding=collider.r, **dyn, **kw)
        if collider.kind == "capsule":
            return self._b.capsule(position, collider.a, collider.r, **dyn, **kw)
        if collider.kind == "poly":
            return self._b.poly(position, collider.verts, rounding=collider.r, **dyn, **kw)
        raise ValueError(collider.kind)

    def add_collider(self, body_id, collider, density=None):
        """A further collider on a body that has one (a compound body; include/sfx.h sfx_collider.body).  With a
        density (dynamic bodies) the shape's mass and inertia are added to the body's; the body origin is taken to
        stay the centre of mass.  Polygons are not offered here (use one polygon per body)."""
        if self._world is not None:
            raise RuntimeError("add colliders before the first tick")
        kw = dict(material=self._mat(collider.material), category=collider.category, mask=collider.mask, group=collider.group)
        dims = {"circle": (collider.r,), "rect": (collider.a, collider.b, collider.r), "capsule": (collider.a, collider.r)}
        if collider.kind not in dims:
            raise ValueError(collider.kind)
        self._b.add_collider(body_id, collider.kind, dims[collider.kind], collider.offset, collider.local_angle, density=density, **kw)

    def add_distance_constraint(self, a, b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), rest_length=None, compliance=0.0,
                                limit=0, damping=0.0):
        """limit: 0 holds the length both ways, 1 (rope) only pulls, 2 (strut) only pushes.
        damping (1/s, two-sided constraints only) takes min(damping h, 1) of the rate of change of the length
        away every substep."""
        self._b.distance(WORLD_BODY if a is None else a, WORLD_BODY if b is None else b, anchor_a, anchor_b,
                         rest_length, compliance, limit, damping)

    def add_attachment(self, a, b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), compliance=0.0, damping=0.0):
        """Pin joint: the two body-local anchors (world position for None) are held together."""
        self._b.attach(WORLD_BODY if a is None else a, WORLD_BODY if b is None else b, anchor_a, anchor_b, compliance, damping)

    def add_angle_constraint(self, a, b, min_angle, max_angle, compliance=0.0, damping=0.0):
        """Keeps angle(b) - angle(a) (None: the world, angle 0) inside [min_angle, max_angle];
        min == max with a compliance is a torsion spring; damping (1/s) is a hinge damper."""
        self._b.angle(WORLD_BODY if a is None else a, WORLD_BODY if b is None else b, min_angle, max_angle, compliance, damping)

    def add_rope(self, start, end, links, radius=0.04, density=1.0, pin_start=True, pin_end=False, compliance=0.0,
                 slack=0.0, material=None):
        """Rope helper (the reference's rope builder): `links` small circle bodies from start to end tied by
        one-sided links; returns their ids."""
        if self._world is not None:
            raise RuntimeError("add bodies before the first tick")
        return self._b.rope(start, end, links, radius, density, pin_start, pin_end, compliance, slack,
                            material=self._mat(material or Material()))

    def _ensure_world(self):
        if self._world is None:
            if not self._b.materials:
                self._mat(Material())
            c = self.config
            self._world = World.from_scene(self._b.build(), substeps=c.substeps, gravity=c.gravity,
                                           aabb_margin=c.aabb_margin, device=c.device,
                                           max_depenetration_velocity=c.max_depenetration_velocity)
        return self._world

    def tick(self, frame_dt):
        """The per-frame tick entry point: asynchronous, one sfx_step."""
        self._ensure_world().step(frame_dt)

    # ---- spatial queries at the current poses (the reference's mouse-pick / ray-cast API)
    def query_point(self, point, layer_mask=0xFFFFFFFF):
        """Body under a point (lowest id if several), or None."""
        body, _ = self._ensure_world().query_points([point], layer_mask)
        return None if body[0] == 0xFFFFFFFF else int(body[0])

    def query_aabb(self, lo, hi, layer_mask=0xFFFFFFFF):
        """Bodies whose collider's bounding box overlaps [lo, hi]."""
        hits = self._ensure_world().query_aabbs([[lo[0], lo[1], hi[0], hi[1]]], layer_mask)
        return [int(h & 0xFFFFFFFF) for h in hits]

    def raycast(self, origin, direction, max_t=float("inf"), layer_mask=0xFFFFFFFF):
        """Closest hit of origin + t * direction: (body, t, normal) or None."""
        h = self._ensure_world().query_rays([[origin[0], origin[1], direction[0], direction[1]]], max_t, layer_mask)[0]
        return None if h["body"] == 0xFFFFFFFF else (int(h["body"]), float(h["t"]), (float(h["normal"][0]), float(h["normal"][1])))

    def tick_and_read(self, frame_dt):
        """A render loop's frame: starts the read-back of the poses as they are now, runs the next tick while
        they travel, and returns them as [n][4] = (x, y, cos, sin) (one frame behind the simulation)."""
        w = self._ensure_world()
        w.readback_begin()
        w.step(frame_dt)
        return w.readback_wait()

    def poses(self):
        pos, rot = self._world.poses()
        return pos, np.arctan2(rot[:, 1], rot[:, 0])

    def velocities(self):
        return self._world.velocities()

    def stats(self):
        return self._world.stats()
