"""Slab domain decomposition: partition logic, logical slabs in one process, two
ranks over gloo on the CPU, and -- on the GPU -- logical slabs with device-to-device halos.

The CPU tests run the decomposition on the CPU oracle's engine: they check the driver and the
exchange protocol, not the HIP kernels (those are covered by the gpu-marked tests below)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from starframe_amd import scenes
from starframe_amd.slab import LocalSlabs, default_halo, plan_slabs

DT = 1.0 / 60.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_scene():
    return scenes.circles_container(n=600, width=60, floor_hw=8.0, wall_hh=8.0)


def test_plan_owns_every_dynamic_body_once_and_mirrors_halos():
    s = scenes.mixed_pile(n=4000, width=200)
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    plans, bounds = plan_slabs(s, 4)
    owned = np.concatenate([p.ids[:p.n_owned] for p in plans])
    assert np.array_equal(np.sort(owned), np.nonzero(dyn)[0])
    counts = [p.n_owned for p in plans]
    assert max(counts) - min(counts) <= 1
    halo = default_halo(s)
    for r, p in enumerate(plans):
        x = s.pos[p.ids[:p.n_owned], 0]
        assert np.all(x >= p.lo) and np.all(x < p.hi)
        if r + 1 < len(plans):
            q = plans[r + 1]
            # what r sends right is what r+1 receives from the left, same bodies in the same order
            assert np.array_equal(p.ids[p.send_right], q.ids[q.recv_left])
            assert np.array_equal(q.ids[q.send_left], p.ids[p.recv_right])
            assert len(p.send_right) > 0
            # every body of r within reach of the boundary is ghosted
            near = p.ids[:p.n_owned][s.pos[p.ids[:p.n_owned], 0] > p.hi - 0.5 * halo]
            assert np.isin(near, q.ids[q.recv_left]).all()
        # fixed bodies are replicated where they are in reach: the floor everywhere, a wall only at its end
        fixed = np.nonzero(~dyn)[0]
        assert np.isin(fixed[0], p.ids)
        assert np.isin(fixed[1], p.ids) == (r == 0) and np.isin(fixed[2], p.ids) == (r == len(plans) - 1)
    assert len(plans[0].send_left) == 0 and len(plans[-1].send_right) == 0


def test_plan_rejects_too_thin_slabs_and_constraints():
    s = small_scene()
    with pytest.raises(ValueError):
        plan_slabs(s, 64)
    with pytest.raises(ValueError):
        plan_slabs(scenes.chains(n_chains=4, links=10), 2)


def test_free_fall_is_bit_identical_to_the_undecomposed_world():
    """Before anything touches, every body's arithmetic is independent of the decomposition."""
    from oracle.oracle import OracleWorld

    s = small_scene()
    L = LocalSlabs(s, 3, OracleWorld)
    W = OracleWorld.from_scene(s)
    for _ in range(3):
        L.step(DT)
        W.step(DT)
    assert W.stats().n_contacts == 0   # (pairs within the speculative margin have manifolds, nothing pushes)
    pos, rot, vel, ang = L.gather()
    assert np.array_equal(pos.view(np.uint32), W.poses()[0].view(np.uint32))
    assert np.array_equal(vel.view(np.uint32), W.velocities()[0].view(np.uint32))


def test_logical_slabs_match_undecomposed_pile_statistically():
    from oracle.oracle import OracleWorld

    s = small_scene()
    L = LocalSlabs(s, 3, OracleWorld, halo=1.5, rebalance_every=2)   # landing circles shoot sideways at 3 m/s: a wide halo, frequent re-cuts
    W = OracleWorld.from_scene(s)
    for _ in range(360):
        L.step(DT)
        W.step(DT)
    pos, rot, vel, ang = L.gather()
    ref = W.poses()[0]
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    assert np.isfinite(pos).all()
    assert abs(pos[dyn, 1].mean() - ref[dyn, 1].mean()) < 0.02          # pile height
    assert abs(pos[dyn, 1].max() - ref[dyn, 1].max()) < 0.15
    ke = lambda v: float((v[dyn] ** 2).sum())
    assert ke(vel) < 1.0 and ke(W.velocities()[0]) < 1.0                  # both (nearly) at rest: sum of v^2 over 600 bodies
    # nothing fell through the seams: every body is inside the container
    assert pos[dyn, 1].min() > -0.1 and np.abs(pos[dyn, 0]).max() < 8.5
    depth = max(w.stats().max_depth for w in L.worlds)
    assert depth < 0.05 and W.stats().max_depth < 0.05


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_ranks_over_gloo_equal_logical_slabs(tmp_path):
    """world_size 2, gloo, CPU: the distributed driver gives bit-for-bit what two logical slabs in
    one process give (same plan, same exchange order, another transport)."""
    from oracle.oracle import OracleWorld

    out = str(tmp_path / "dist.npz")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "slab_worker.py"), out, "41"], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    z = np.load(out)
    s = small_scene()
    L = LocalSlabs(s, 2, OracleWorld, halo=1.5, rebalance_every=2)
    for _ in range(41):   # odd: the last frame is not one that may rebuild the worlds (fresh worlds have no statistics)
        L.step(DT)
    pos, rot, vel, ang = L.gather()
    assert np.array_equal(z["pos"].view(np.uint32), pos.view(np.uint32))
    assert np.array_equal(z["vel"].view(np.uint32), vel.view(np.uint32))
    assert (L.worlds[0].stats().n_manifolds + L.worlds[1].stats().n_manifolds) > 100
    # the all-reduced reporting figures: every dynamic body counted once, energy of the gathered state
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    assert int(z["bodies"]) == int(dyn.sum())
    m = np.where(s.inv_mass > 0, 1.0 / np.where(s.inv_mass > 0, s.inv_mass, 1.0), 0.0)
    inertia = np.where(s.inv_inertia > 0, 1.0 / np.where(s.inv_inertia > 0, s.inv_inertia, 1.0), 0.0)
    ke = (0.5 * m * (vel.astype(np.float64) ** 2).sum(1)).sum() + (0.5 * inertia * ang.astype(np.float64) ** 2).sum()
    assert float(z["ke"]) == pytest.approx(ke, rel=1e-9)
    assert int(z["manifolds"]) == L.worlds[0].stats().n_manifolds + L.worlds[1].stats().n_manifolds
    assert float(z["max_depth"]) == pytest.approx(max(w.stats().max_depth for w in L.worlds), rel=1e-6)


def test_an_error_on_one_rank_stops_all_ranks_together(tmp_path):
    """the halo guard / capacity errors are rank-local.  Two ranks over gloo; a body of rank 0's slab shoots
    off and trips rank 0's guard in the first frames.  Rank 0 must not stop alone (rank 1 would wait for its halo for
    ever): both ranks raise SlabError at the same frame, the next collective check."""
    out = str(tmp_path / "trip")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "slab_worker.py"), out, "20", "trip"], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0     # (a hang ends the test by this timeout)
    r0 = open(out + ".rank0").read().splitlines()
    r1 = open(out + ".rank1").read().splitlines()
    assert r0[0] == r1[0] == "4", (r0, r1)               # the first check after the guard tripped, on both ranks
    assert "moved more than halo / 4" in r0[1]            # the rank that saw it says what it saw
    assert "another rank" in r1[1]                        # the other one stops with it


def drifting_scene():
    """The small pile with 2 m/s to the right: it slides against the right wall, bodies cross the cuts."""
    s = small_scene()
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    vel = np.zeros_like(s.pos)
    vel[dyn, 0] = 2.0
    return s.with_state(s.pos, s.rot, vel, s.angvel)


def test_rebalancing_recuts_the_slabs_when_bodies_migrate():
    """count exchange + migration.  A pile that slides sideways leaves its slabs; the driver
    notices (population imbalance > 10 % or a body moved a quarter of the halo), gathers the owned states
    and cuts again.  The run stays a valid simulation: nothing lost at the seams, same pile as undecomposed."""
    from oracle.oracle import OracleWorld

    s = drifting_scene()
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    L = LocalSlabs(s, 3, OracleWorld, halo=1.5, rebalance_every=2)
    frozen = LocalSlabs(s, 3, OracleWorld)            # never re-cut: its populations drift apart
    W = OracleWorld.from_scene(s)
    b0 = L.bounds.copy()
    for _ in range(240):
        L.step(DT)
        W.step(DT)
    from starframe_amd.slab import StaleHalo
    with pytest.raises(StaleHalo):                    # ... and never silently: the halo guard stops it
        for _ in range(60):
            frozen.step(DT)
    assert frozen.frames < 40
    assert L.replans >= 2
    assert not np.array_equal(b0, L.bounds)
    # right after a cut every dynamic body is owned once and the populations are equal
    owned = np.concatenate([p.ids[:p.n_owned] for p in L.plans])
    assert np.array_equal(np.sort(owned), np.nonzero(dyn)[0])
    assert L.rebalance(force=True)
    counts = [p.n_owned for p in L.plans]
    assert max(counts) - min(counts) <= 1
    assert not L.rebalance() and L.imbalance <= 0.10   # nothing moved since: no new cut
    pos, rot, vel, ang = L.gather()
    ref = W.poses()[0]
    assert np.isfinite(pos).all()
    assert pos[dyn, 1].min() > -0.1 and np.abs(pos[dyn, 0]).max() < 8.5     # nothing fell through a seam
    assert abs(pos[dyn, 1].mean() - ref[dyn, 1].mean()) < 0.05
    assert abs(pos[dyn, 0].mean() - ref[dyn, 0].mean()) < 0.15               # slid equally far
    assert max(w.stats().max_depth for w in L.worlds) < 0.05 and W.stats().max_depth < 0.05


def test_two_ranks_over_gloo_rebalance_like_logical_slabs(tmp_path):
    """world_size 2, gloo: the distributed count exchange (all-reduce) and migration (all-gather + new cut)
    take the same decisions at the same frames as the one-process driver: bit-identical states."""
    from oracle.oracle import OracleWorld

    out = str(tmp_path / "dist.npz")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "slab_worker.py"), out, "60", "drift"], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    z = np.load(out)
    L = LocalSlabs(drifting_scene(), 2, OracleWorld, halo=1.5, rebalance_every=2)
    for _ in range(60):
        L.step(DT)
    pos, rot, vel, ang = L.gather()
    assert int(z["replans"]) == L.replans and L.replans >= 1
    assert np.array_equal(z["bounds"], L.bounds)
    assert np.array_equal(z["pos"].view(np.uint32), pos.view(np.uint32))
    assert np.array_equal(z["vel"].view(np.uint32), vel.view(np.uint32))


@pytest.mark.gpu
def test_gpu_logical_slabs_match_oracle_slabs_and_undecomposed():
    """Four logical slabs on one MI355X, halos copied device to device through the C ABI:
    one tick equals the same decomposition on the oracle within 1e-5; 150 ticks settle to the
    same pile as the undecomposed GPU world."""
    from oracle.oracle import OracleWorld
    from starframe_amd.world import World
    from tests.helpers import rel_err

    s = scenes.circles_container(n=3000, width=150, floor_hw=20.0, wall_hh=8.0)
    G = LocalSlabs(s, 4, World, halo=1.5, rebalance_every=5)
    O = LocalSlabs(s, 4, OracleWorld)
    for _ in range(60):          # fall until the first layers touch
        G.step(DT)
    gp, gr, gv, ga = G.gather()
    s2 = s.with_state(gp, gr, gv, ga)
    G.close()
    G = LocalSlabs(s2, 4, World, halo=1.5, rebalance_every=5)   # the settling circles roll further than the halo guard allows
    O = LocalSlabs(s2, 4, OracleWorld, halo=1.5, rebalance_every=5)
    G.step(DT)
    O.step(DT)
    assert sum(w.stats().n_manifolds for w in G.worlds) > 500
    assert rel_err(G.gather()[0], O.gather()[0], floor=1.0) <= 1e-5
    assert rel_err(G.gather()[2], O.gather()[2], floor=1.0) <= 1e-4
    W = World.from_scene(s2)
    W.step(DT)
    for _ in range(150):
        G.step(DT)
        W.step(DT)
    pos = G.gather()[0]
    ref = W.poses()[0]
    dyn = (s.inv_mass > 0) | (s.inv_inertia > 0)
    assert np.isfinite(pos).all()
    assert abs(pos[dyn, 1].mean() - ref[dyn, 1].mean()) < 0.02
    assert pos[dyn, 1].min() > -0.1
    # deepest overlap of the still settling pile: no worse than the undecomposed world's by more than 3 cm
    assert max(w.stats().max_depth for w in G.worlds) < W.stats().max_depth + 0.03


@pytest.mark.gpu
def test_gpu_halo_guard_reports_a_stale_cut():
    """rebalance_every = 0 and a pile that slides sideways: the first kernel of every tick compares each body's x
    with where it was when the halo lists were cut (sfx_halo_guard, halo / 4); the tick after a body has moved
    further sets overflow bit 128 and the next sync -- or tick -- returns SFX_ERR_CAPACITY.  Nothing else changes:
    with the guard off the same run goes through."""
    from starframe_amd.world import SfxError, World

    L = LocalSlabs(drifting_scene(), 3, World)
    with pytest.raises(SfxError) as e:
        for _ in range(120):
            L.step(DT)
            for w in L.worlds:
                w.sync()
    assert e.value.code == -4 and "128" in str(e.value) or "capacity" in str(e.value)
    assert 2 < L.frames < 60
    L.close()
    L = LocalSlabs(drifting_scene(), 3, World)
    for w in L.worlds:
        w.halo_guard(0.0)
    for _ in range(60):
        L.step(DT)
    for w in L.worlds:
        w.sync()
    L.close()


@pytest.mark.gpu
def test_gpu_halo_host_path_equals_device_path():
    from starframe_amd.world import World

    s = scenes.circles_container(n=1000, width=100, floor_hw=14.0, wall_hh=6.0)
    A = LocalSlabs(s, 2, World, halo=3.0)      # a wide halo: the guard (halo / 4) outlasts the 30 frames without a re-cut
    B = LocalSlabs(s, 2, World, halo=3.0)
    B.device_copies = False      # through sfx_halo_download / sfx_halo_upload
    for _ in range(30):
        A.step(DT)
        B.step(DT)
    assert np.array_equal(A.gather()[0].view(np.uint32), B.gather()[0].view(np.uint32))


@pytest.mark.gpu
def test_gpu_torch_views_the_library_halo_buffers_and_single_rank_nccl():
    """The RCCL transport hands torch tensors that alias libsfx's device buffers to send/recv.
    Runs in a fresh process that imports torch BEFORE libsfx is loaded (as bench.py does for
    N > 1): both must share one HIP runtime, and a process that loaded libsfx first cannot
    initialise torch's afterwards."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "slab_gpu_worker.py")], env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "slab gpu worker ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", ["strips", "square"])
def test_gpu_overlapped_exchange_equals_the_serial_exchange(tiles):
    """sfx_halo_overlap (SURVEY.md section 5 "overlap it with interior-constraint solving"): the velocity sweep as a near and a
    far launch with the halo's pack and device-to-device copies on the halo streams behind the near launch.  Which launch
    runs a batch does not change what it computes: four logical slabs of a mixed pile tick to the SAME BITS with overlap on
    and off, on strips (two phases) and on square tiles (four phases, the closure over three earlier phases), and the near
    list is a proper part of the batches (something is left to overlap with)."""
    from starframe_amd.world import World

    kw = dict(tile_cells_y=1 << 20) if tiles == "strips" else dict(tile_cells=4, tile_cells_y=4)
    s = scenes.mixed_pile(n=6000, width=120)
    W = World.from_scene(s, linear_damping=3.3, angular_damping=1.0)
    for _ in range(250):
        W.step(DT)
    pos, rot = W.poses()
    vel, ang = W.velocities()
    W.close()
    s2 = s.with_state(pos, rot, vel, ang)
    A = LocalSlabs(s2, 4, World, halo=1.5, overlap=False, **kw)
    B = LocalSlabs(s2, 4, World, halo=1.5, overlap=True, **kw)
    assert sum(len(p.recv_idx) for p in B.plans) > 100
    for f in range(12):
        A.step(DT)
        B.step(DT)
    a, b = A.gather(), B.gather()
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert sum(w.stats().n_manifolds for w in B.worlds) > 5000
    assert all(w.stats().overflow == 0 for w in B.worlds)
    near = [w.debug_near_batches() for w in B.worlds]
    assert all(0 < n < total for n, total in near), near
    A.close()
    B.close()


@pytest.mark.gpu
def test_gpu_overlap_in_a_world_alone_on_its_device_and_with_a_fallback_list():
    """The same with ONE world on the device (its fused launches start workgroup b on batch b instead of taking tickets): a
    pile whose left edge is 'sent' onto ghosts that sit in a second pile further right (a halo copied to itself) -- not a
    physical scene, a deterministic one.  Overlap on and off give the same bits; so does a world with pairs on the fallback
    list (a large dynamic body), whose ticks are not split."""
    from starframe_amd.world import World

    for big in (False, True):
        s = scenes.mixed_pile(n=4000, width=100)
        W = World.from_scene(s, linear_damping=3.3, angular_damping=1.0)
        for _ in range(200):
            W.step(DT)
        pos, rot = W.poses()
        W.close()
        s2 = s.with_state(pos, rot)
        if big:   # one circle on top of the pile grows beyond a grid cell: its pairs go to the fallback list
            import copy
            s2 = copy.copy(s2)
            col = s2.colliders.copy()
            circ = np.flatnonzero((col["type"][: s2.n] == 1) & (s2.inv_mass > 0))
            k = circ[np.argmax(s2.pos[circ, 1])]
            col["p"][k, 0] = 2.5
            s2.colliders = col
        dyn = np.flatnonzero(s2.inv_mass > 0)
        order = dyn[np.argsort(s2.pos[dyn, 0])]
        send, recv = order[:150].astype(np.uint32), order[-150:].astype(np.uint32)
        out = []
        for overlap in (False, True):
            w = World.from_scene(s2, tile_cells_y=1 << 20, **(dict(cell_size=0.3) if big else {}))
            w.halo_set(send, recv)
            if overlap:
                w.halo_overlap(True)
            for _ in range(6):
                w.tick_begin(DT)
                for _ in range(s2.substeps):
                    w.tick_substep()
                    w.halo_pack()
                    w.halo_copy_from(w, 0, 0, 150)
                    w.halo_unpack()
            st = w.stats()
            assert st.overflow == 0 and st.n_manifolds > 3000
            if big:
                assert st.n_fallback > 0
            n, total = w.debug_near_batches()
            out.append((w.poses(), w.velocities(), n, total))
            w.close()
        (p0, v0, _, _), (p1, v1, n, total) = out
        for x, y in zip(p0 + v0, p1 + v1):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        assert (n == 0xFFFFFFFF) if big else (0 < n < total), (big, n, total)
