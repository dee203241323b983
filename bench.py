#!/usr/bin/env python3
"""bench.py -- body-substeps/s of the libsfx physics tick on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one physics tick (one frame: broadphase + colouring + `substeps` substeps) of the
workload; value = dynamic bodies x substeps x K / seconds, summed over all ranks.

N = 1 : the headline config (BASELINE.json configs[2]): 100k mixed circle/rect/capsule pile,
        4 substeps, settled on the GPU before anything is timed.
N > 1 : weak scaling, one process per GPU, BASELINE.json configs[4]: a granular slab of 500 k circles per
        GPU (4 M at N = 8), N times as wide, cut into N slabs on x (starframe_amd/slab.py); every
        substep the boundary bodies go to the two neighbours with plain RCCL ncclSend / ncclRecv
        (libsfx_rccl.so) straight from the library's device buffers on the world's stream.
        value = all bodies x substeps x K / max-over-ranks seconds; scaling_reference = rank 0's
        share of the same settled slab alone on one GPU (the denominator of an efficiency figure:
        the N = 1 line of this script is the 100 k pile, another workload).

The JSON line also carries
  roofline     : the kernel the tick spends most time in (one of the contact-sweep kernels k_narrow,
                 k_tile_pos, k_tile_vel): algorithmic bytes per launch / mean launch time from HIP
                 events recorded by libsfx on its own stream, against the 8 TB/s HBM peak; the other
                 two under roofline.other_kernels;
  cpu_baseline : this repository's single-threaded CPU restatement (oracle/, kind "port")
                 timed on this host on a bounded number of frames of the same settled scene;
                 cpu_baseline.openmp: the same source built with -fopenmp on all host threads.
                 It is NOT starframe's Rust tick, which was unavailable.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DT = 1.0 / 60.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is achievable


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def settle(scene, device, frames=600, damping=3.3, angular=1.0):
    """Let the lattice fall into a pile on the GPU (damped, so nothing tunnels), return the
    scene with the settled poses and zero velocities.  Not timed."""
    from starframe_amd.world import World

    w = World.from_scene(scene, linear_damping=damping, angular_damping=angular, device=device)
    for _ in range(frames):
        w.step(DT)
    pos, rot = w.poses()
    w.close()
    return scene.with_state(pos, rot, name=scene.name + "_settled")


def penetration_mm(w):
    """Distribution of the positive contact-point depths at the current poses (slow parity hook: the narrowphase of every
    candidate pair, downloaded), in millimetres: what "pile" means physically beside the throughput."""
    w.broadphase(DT)
    c = w.contacts()
    d = np.concatenate([c["depth"][:, 0][c["count"] > 0], c["depth"][:, 1][c["count"] > 1]])
    d = d[d > 0]
    if not len(d):
        return {"points": 0}
    p50, p90, p99 = np.percentile(d, [50, 90, 99])
    return {"points": int(len(d)), "p50": round(1e3 * float(p50), 2), "p90": round(1e3 * float(p90), 2),
            "p99": round(1e3 * float(p99), 2), "max": round(1e3 * float(d.max()), 2)}


def kernel_bytes(st, derive_bodies=0, fused_narrow=False):
    """Algorithmic (compulsory) bytes of ONE launch of each contact-sweep kernel (one launch per substep),
    from the tick's measured counts.  DESIGN.md "Kernels": P = solver pairs, M = touching pairs (manifolds),
    M2 = of which with two points; record planes A..G are 16 bytes each (sfx_device.h ContactArrays).
      k_narrow   : P x (16 order (ids + slot) + 2 x (16 pose + 16 shape) read, 16 plane A written)
                   + M x (2 x (16 velocity + 8 mass) read; planes B, C, D = 48 written) + M2 x 48 (planes E, F, G)
      k_tile_pos : P x (4 pair indices + 16 plane A read) + M x (16 plane B + 2 x (16 displacement + 8 mass) read;
                   8 lambda + 2 x 16 displacement written) + M2 x 48
      k_tile_vel : P x (4 + 16 + 8 lambda read) + M x (planes B, C, D = 48 + 2 x (16 velocity + 8 mass) read;
                   2 x 16 velocity written) + M2 x 48
                   + derive_bodies x 64: inside sfx_step the kernel also applies the position sweep's displacement and
                   derives the velocities (DESIGN.md "Velocities derived where they are first read"): pose, start-of-substep
                   pose and displacement of every dynamic body read once, its pose written once
    fused_narrow: the narrowphase runs inside k_tile_pos (libsfx's default; DESIGN.md "Narrowphase inside the position sweep")."""
    P, M, M2 = float(st.n_solver_pairs), float(st.n_manifolds), float(st.n_manifolds2)
    kb = {
        "narrow": P * 96.0 + M * 96.0 + M2 * 48.0,
        "solve_pos": P * 20.0 + M * 104.0 + M2 * 48.0,
        "solve_vel": P * 28.0 + M * 128.0 + M2 * 48.0 + 64.0 * derive_bodies,
    }
    if fused_narrow:
        # k_tile_pos computes the records of its batches' pairs itself (no k_narrow launch): the narrowphase's bytes are
        # its own now, and what a kernel that kept the records on chip between the two stages would not read again is NOT
        # counted as compulsory -- the pair indices, plane A, B and E..G of the position solve and the masses (read once)
        kb["solve_pos"] = kb.pop("narrow") + M * (2 * 16.0 + 8.0 + 2 * 16.0)   # + displacements read, lambda and displacements written
    return kb


FUSED_POS_NAME = ("k_tile_pos (narrowphase of the batch's own pairs, then the XPBD manifold block solves; all tile phases in one "
                  "launch, bodies in LDS)")
KERNEL_NAMES = {
    "narrow": "k_narrow (manifolds of all solver pairs, one pair per lane)",
    "solve_pos": "k_tile_pos (XPBD manifold block solves, all tile phases in one launch, bodies in LDS)",
    "solve_vel": "k_tile_vel (restitution + friction per manifold, all tile phases in one launch, bodies in LDS)",
}


def tick_bytes_per_body_substep(st, n_dyn, substeps, fused_narrow=False, integrate_on_load=False):
    """SURVEY.md 8d accounting for the whole tick, from measured pair / contact counts."""
    kb = kernel_bytes(st, n_dyn, fused_narrow)
    integrate = 16 + 16 + 8 + 16 + 16 + 16 + 16     # pose, vel, mass read; prev, pose, vel, zero displacement written
    if integrate_on_load:
        # no k_integrate (DESIGN.md "Integration on load"): the sweeps integrate the records they read anyway; what is left
        # is the zero displacement written where a body is derived and one k_tick_free per tick (mass and velocity read) --
        # per substep where the tick is driven substep by substep (integrate_on_load == "substep": the slab drivers)
        integrate = 16 + (8 + 16) / (1 if integrate_on_load == "substep" else substeps)
    derive = 0                                      # applied and derived where the velocity solve loads a body: in kernel_bytes
    p = st.n_pairs / n_dyn
    broad = (16 + 16 + 16 + 16 + 4 + 8 + 4 + 4 + 28 + 16 * 5 + p * (8 + 8 + 2 + 16 + 16)) / substeps
    return integrate + derive + sum(kb.values()) / n_dyn + broad


def survey_scene_leg(device, bodies, steps, warmup, settle_frames=1500):
    """SURVEY.md 8d's own lattice for the headline config -- 400 wide (250 rows: a pile twice as deep as the bench scene's,
    about 55 layers) -- settled and timed the same way, reported beside the headline as config.survey_scene (VERDICT r2
    item 4).  Deeper piles compress more under XPBD at 4 substeps (DESIGN.md "Deep piles") and need 2-D tiles."""
    from starframe_amd import scenes
    from starframe_amd.world import World

    sc = scenes.mixed_pile(n=bodies, width=400)
    sc = settle(sc, device, frames=settle_frames)
    w = World.from_scene(sc, device=device)
    for _ in range(warmup):
        w.step(DT)
    w.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step(DT)
    w.sync()
    el = time.perf_counter() - t0
    st = w.stats()
    pen = penetration_mm(w)
    vel, _ = w.velocities()
    dyn = np.asarray(sc.inv_mass) > 0
    pos, _ = w.poses()
    w.close()
    return {"scene": f"lattice 400 wide x {bodies // 400} rows, pitch 0.30 m, settled {settle_frames} damped ticks",
            "value": sc.n_dynamic() * sc.substeps * steps / el, "unit": "body-substeps/s", "ms_per_step": 1e3 * el / steps,
            "steps": steps, "warmup": warmup, "pairs": st.n_pairs, "manifolds": st.n_manifolds, "colors": st.n_colors,
            "tiles": st.n_tiles, "pile_height_m": round(float(pos[dyn, 1].max()), 2),
            "v_rms": round(float(np.sqrt((vel[dyn] ** 2).sum(1).mean())), 4), "penetration_mm": pen}


def cpu_baseline(scene, budget_s=15.0):
    """Single-threaded CPU restatement (oracle) on the same settled scene, sequential
    Gauss-Seidel order (what a CPU engine does).  Bounded sample."""
    from oracle.oracle import ORDER_SEQUENTIAL, OracleWorld

    w = OracleWorld.from_scene(scene)
    t0 = time.perf_counter()
    w.step(DT, ORDER_SEQUENTIAL)
    first = time.perf_counter() - t0
    frames = int(max(1, min(200, budget_s / max(first, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(frames):
        w.step(DT, ORDER_SEQUENTIAL)
    el = time.perf_counter() - t0
    n_dyn = scene.n_dynamic()
    w.close()
    return {
        "openmp": cpu_baseline_openmp(scene, n_dyn),
        "value": n_dyn * scene.substeps * frames / el,
        "unit": "body-substeps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{frames} ticks of the same settled {scene.name} scene ({n_dyn} bodies, {scene.substeps} substeps), "
                  f"sequential Gauss-Seidel order; this repo's CPU restatement, not starframe's Rust tick (unavailable)",
        "ms_per_step": 1e3 * el / frames,
        "cpu_model": cpu_model(),
    }


def cpu_baseline_openmp(scene, n_dyn, budget_s=8.0):
    """The same restatement built with -fopenmp (oracle/libsfx_oracle_omp.so; SURVEY.md 8d "OpenMP-coloured variant,
    reported separately"): coloured order, the units of one phase, the manifolds and the per-body loops side by side on
    the host's threads, same bits as the serial build.  Its broadphase stays serial."""
    from oracle.oracle import ORDER_COLORED, OracleWorld, load

    try:
        threads = int(load(True).sfo_threads())
        w = OracleWorld.from_scene(scene, openmp=True)
    except Exception as e:  # the variant is optional: no libgomp, no figure
        return {"error": str(e)}
    t0 = time.perf_counter()
    w.step(DT, ORDER_COLORED)
    first = time.perf_counter() - t0
    frames = int(max(1, min(200, budget_s / max(first, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(frames):
        w.step(DT, ORDER_COLORED)
    el = time.perf_counter() - t0
    w.close()
    return {"value": n_dyn * scene.substeps * frames / el, "unit": "body-substeps/s", "cores": threads, "kind": "port",
            "sample": f"{frames} ticks, coloured order, OpenMP over the units of a phase", "ms_per_step": 1e3 * el / frames}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def load_traffic(cls):
    """HBM bytes per launch of a contact-sweep kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_solve.json, written by scripts/pmc_report.py), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_solve.json")
    try:
        with open(p) as f:
            return json.load(f)[cls]["hbm_bytes_per_launch"]
    except Exception:
        return None


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* as torchrun sets them), relay rank 0's JSON line, return non-zero when any rank failed.  The parent makes no
    GPU call and never falls through to the N = 1 workload.  The first rank to fail ends the others (by their process
    ids): ranks left alone would wait for it in a collective for ever."""
    import subprocess

    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc, line = 0, b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                try:
                    if r == 0:
                        out, _ = procs[0].communicate(timeout=0.5)   # drains the pipe while waiting
                        line = out or b""
                    else:
                        procs[r].wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if procs[r].returncode != 0:
                    rc = rc or procs[r].returncode or 1
                    print(f"bench.py: rank {r} exited with {procs[r].returncode}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if rc == 0:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    return rc


def dry_run(args, rank, world):
    """Everything of the launch that needs no GPU: the environment, the rendezvous (gloo) and one collective."""
    fail = os.environ.get("SFX_BENCH_FAIL_RANK")
    if fail is not None and int(fail) == rank:
        raise SystemExit(3)
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([1.0])
        dist.all_reduce(t)
        assert int(t.item()) == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--bodies", type=int, default=100_000)
    ap.add_argument("--width", type=int, default=1000, help="bodies per lattice row of the N = 1 pile (SURVEY.md 8d: 400)")
    ap.add_argument("--slab-bodies", type=int, default=500_000, help="N > 1: circles per GPU of the granular slab (configs[4]: 500 k)")
    ap.add_argument("--no-single-reference", dest="single_reference", action="store_false",
                    help="N > 1: skip rank 0's one-GPU run of its share of the slab (reported as scaling_reference)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-survey-scene", action="store_true",
                    help="N = 1: skip the second, untimed-for-value leg on SURVEY.md 8d's 400-wide lattice (config.survey_scene)")
    ap.add_argument("--settle-frames", type=int, default=600)
    ap.add_argument("--live-events", default="solve_pos",
                    help="comma-separated kernel classes bracketed with HIP events INSIDE the timed region (the roofline's "
                         "launch time when the dominant kernel is among them); 'off' = none")
    ap.add_argument("--live-every", type=int, default=8,
                    help="bracket the --live-events launches in every n-th timed tick (an event record costs the stream "
                         "3.4 us: every tick would take 3.4 %% off the value, every 8th takes 0.4 %%)")
    ap.add_argument("--force-slab", action="store_true",
                    help="rehearse the N > 1 code path (slab driver over torch.distributed nccl) with one rank")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that SHARE GPU 0, process group over gloo, halos through host memory: everything of the "
                         "N > 1 path except RCCL (cuts, ghosts, guard, re-cuts, gather, timing); the line says so and is no result")
    ap.add_argument("--dry-run", action="store_true",
                    help="argument handling, self-launch and rendezvous only (gloo, no GPU, no library): prints a line with "
                         "n_gpus and \"dry_run\": true; what tests/test_bench_launch.py runs on the CPU")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`, the way the driver starts the N = 1 line): this process becomes
        # the launcher -- it never touches the GPU -- and runs one child per GPU with the torchrun environment
        raise SystemExit(self_launch(args.gpus))
    rank, local_rank, world = dist_env()
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}: start N ranks (torchrun) or none (bench.py starts them itself)")
    if args.dry_run:
        return dry_run(args, rank, world)
    # stdout carries the one JSON line and nothing else: what libraries print to file descriptor 1 (RCCL's version
    # banner at communicator creation) goes to stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from starframe_amd import scenes
    from starframe_amd.world import World

    device = local_rank
    dist = None
    slab = None
    setup = None
    if world > 1 or args.force_slab:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            device = 0
            os.environ["SFX_SHARED_DEVICE"] = "1"   # several processes' fused launches on one GPU: batches by ticket
            torch.cuda.set_device(device)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))

        from starframe_amd.slab import DistSlab

        # BASELINE.json configs[4] in its weak-scaling form: a granular slab of `slab_bodies` circles per
        # GPU (500 k: 4 M at N = 8), 100 lattice rows, `world` times as wide; settled as slabs with damping (untimed, one
        # cut with a wide halo), relaxed without damping, then cut again on the state at rest
        per_gpu = args.slab_bodies
        scene = scenes.granular_slab(n=world * per_gpu, width=world * (per_gpu // 100))
        scene_note = f"granular_slab: lattice {world * (per_gpu // 100)} wide x 100 rows of circles r ~ U[0.05, 0.10], pitch 0.22 m"
        setup = {}
        t_setup = time.perf_counter()

        def untimed(sc, frames, what, **kw):
            # ONE cut with a 12 m halo and no re-cut (a re-cut gathers every body on every rank and rebuilds the worlds:
            # at 4 M bodies ~5 s of host work each, and re-cutting every 50 frames made the untimed part the longest
            # thing in an N = 8 run); the halo guard is off while the circles roll into place, and what it would have
            # checked is checked once at the end: no body may have travelled more than halo / 4 in x
            t = time.perf_counter()
            pre = DistSlab(sc, World, dist, device_index=device, halo=12.0, **kw)
            pre.world.halo_guard(0.0)
            for _ in range(frames):
                pre.step(DT)
            pre.sync()
            _, moved = pre.drift()
            if moved > pre.halo / 4:
                raise SystemExit(f"bench: a body travelled {moved:.2f} m in x while the pile {what} (limit {pre.halo / 4:.1f} m): "
                                 f"the halo of the untimed set-up is too narrow for this scene")
            pos, rot, _, _ = pre.gather()
            pre.close()
            setup[what] = round(time.perf_counter() - t, 2)
            return sc.with_state(pos, rot, name=scene.name + "_settled")

        scene = untimed(scene, args.settle_frames, "settled", linear_damping=3.3, angular_damping=1.0)
        scene = untimed(scene, 120, "relaxed")    # ... without damping, so that the cut below is made on a pile at rest
        single = None
        if args.single_reference:   # rank 0's share of the same workload on one GPU alone: the denominator of a scaling figure
            if rank == 0:
                dyn = np.nonzero((scene.inv_mass > 0) | (scene.inv_inertia > 0))[0]
                order = dyn[np.argsort(scene.pos[dyn, 0], kind="stable")][:per_gpu]
                fixed = np.nonzero(~((scene.inv_mass > 0) | (scene.inv_inertia > 0)))[0]
                alone = scene.subset(np.concatenate([np.sort(order), fixed]), name="granular_slab_one_gpu")
                w1 = World.from_scene(alone, device=device)
                for _ in range(20):
                    w1.step(DT)
                w1.sync()
                t1 = time.perf_counter()
                for _ in range(100):
                    w1.step(DT)
                w1.sync()
                single = {"n_gpus": 1, "value": per_gpu * scene.substeps * 100 / (time.perf_counter() - t1), "unit": "body-substeps/s",
                          "workload": f"the leftmost {per_gpu} bodies of the same settled slab alone on one GPU (no halo, an open right side), 100 ticks"}
                w1.close()
            dist.barrier()
        # an 8 m halo (0.7 % more bodies per slab): surface circles roll, and the halo guard stops the run -- loudly --
        # when a body has moved halo / 4 = 2 m since this cut; no re-cut inside the timed region (it rebuilds the worlds)
        t = time.perf_counter()
        slab = DistSlab(scene, World, dist, device_index=device, halo=8.0)
        setup["final_cut"] = round(time.perf_counter() - t, 2)
        setup["total"] = round(time.perf_counter() - t_setup, 2)
        print(f"bench: rank {rank} set-up {setup}", file=sys.stderr, flush=True)
        w = slab.world
        n_dyn = scene.n_dynamic()            # all ranks together
        n_local = slab.plan.n_owned
        halo = len(slab.plan.send_idx)
        parallelism = (f"{world} slabs on x, {n_local} owned + {len(slab.plan.recv_idx)} ghost bodies on rank 0, "
                       f"halo of {halo} bodies x 32 B to <= 2 neighbours per substep over "
                       f"{'libsfx_rccl (plain rccl.h ncclSend / ncclRecv on the world stream)' if slab.native else 'torch.distributed nccl on the library buffers' if slab.on_device else 'host staging'}")
        step = lambda: slab.step(DT)
    else:
        scene = scenes.mixed_pile(n=args.bodies, width=args.width)
        scene_note = (f"lattice {args.width} wide x {args.bodies // args.width} rows, pitch 0.30 m, settled {args.settle_frames} damped ticks "
                      f"(SURVEY.md 8d names a 400-wide lattice: figures for it in profiles/README.md)")
        parallelism = "1 GPU"
        scene = settle(scene, device, frames=args.settle_frames)
        n_dyn = scene.n_dynamic()
        w = World.from_scene(scene, device=device)
        step = lambda: w.step(DT)

    sync = slab.sync if slab is not None else w.sync   # (slabs: the ranks also agree that none of them has failed)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events around the launches of the dominant kernel class over exactly the timed steps (recorded by libsfx on
    # the stream the kernels run on; nothing is collected or synchronised until the region has ended)
    live = [] if args.live_events == "off" else [c for c in args.live_events.split(",") if c]
    w.kernel_times(reset=True)
    every = max(1, args.live_every)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if live and i % every == 0:
            w.set_profiling_classes(live)          # a host-side flag: no device work, no synchronisation
        elif live and i % every == 1:
            w.set_profiling_classes(["stats"])     # a class no tick kernel belongs to = paused (0 would collect and synchronise)
        step()
    sync()
    elapsed = time.perf_counter() - t0
    kt_live = w.kernel_times(reset=True) if live else {}
    w.set_profiling(False)
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    st = w.stats()

    # second leg: the same ticks again with HIP events around EVERY launch (per-kernel table; kept out of `value`)
    prof_frames = max(8, min(args.steps, 60))
    w.set_profiling(True)
    w.kernel_times(reset=True)
    for _ in range(prof_frames):
        step()
    kt = w.kernel_times()
    w.set_profiling(False)
    st2 = w.stats()
    pen = penetration_mm(w) if slab is None else None   # after everything timed: it runs a broadphase of its own
    v_rms = None
    if slab is None:   # how much the "settled" pile still moves (DESIGN.md "Jitter"): root mean square speed of the dynamic bodies
        vv, _ = w.velocities()
        v_rms = round(float(np.sqrt((vv[np.asarray(scene.inv_mass) > 0] ** 2).sum(1).mean())), 4)
    derive_bodies = scene.n_dynamic() if slab is None else 0   # sfx_step derives on load; the slab driver steps substep by substep
    fused_narrow = kt["narrow"]["launches"] == 0     # no k_narrow launch: the position sweep computed the records
    names = dict(KERNEL_NAMES, solve_pos=FUSED_POS_NAME) if fused_narrow else KERNEL_NAMES
    kb = kernel_bytes(st2, derive_bodies, fused_narrow)
    per_kernel = {}
    for cls, nbytes in kb.items():
        k = kt[cls]
        if not k["launches"]:
            continue
        per_launch = k["launches"] / (prof_frames * scene.substeps)   # 1 when fused; tile phases when not
        dur = 1e-3 * k["total_ms"] / k["launches"]
        per_kernel[cls] = {"bytes_per_launch": nbytes / per_launch, "launch_us": dur * 1e6,
                           "achieved": nbytes / per_launch / dur / 1e9, "ms_per_step": k["total_ms"] / prof_frames}
    dom = max(per_kernel, key=lambda c: per_kernel[c]["ms_per_step"])   # the kernel the tick spends most time in
    sp = dict(per_kernel[dom])
    measured_over = f"{prof_frames} further ticks after the timed region (the dominant kernel was not among --live-events)"
    if dom in kt_live and kt_live[dom]["launches"]:
        k = kt_live[dom]
        nbytes = kernel_bytes(st, derive_bodies, fused_narrow)[dom]
        live_ticks = len(range(0, args.steps, every))
        per_launch = k["launches"] / (live_ticks * scene.substeps)
        dur = 1e-3 * k["total_ms"] / k["launches"]
        sp = {"bytes_per_launch": nbytes / per_launch, "launch_us": dur * 1e6, "achieved": nbytes / per_launch / dur / 1e9,
              "ms_per_step": k["total_ms"] / live_ticks}
        measured_over = (f"every {every}th of the {args.steps} timed ticks ({k['launches']} launches inside the timed region)"
                         if every > 1 else f"the {args.steps} timed ticks ({k['launches']} launches)")
    achieved = sp["achieved"]
    kernel_ms = {k: round(v["total_ms"] / prof_frames, 4) for k, v in kt.items() if v["launches"]}
    value = n_dyn * scene.substeps * args.steps / elapsed
    n_rank = slab.plan.n_owned if slab is not None else n_dyn
    # one k_tick_free per tick instead of k_integrate per substep -- or, substep by substep (the slab drivers), one per substep
    # and no k_derive
    integrate_on_load = (True if kt["integrate"]["launches"] == prof_frames else
                         "substep" if kt["integrate"]["launches"] == prof_frames * scene.substeps and kt["derive_vel"]["launches"] == 0 else False)
    bpbs = tick_bytes_per_body_substep(st2, n_rank, scene.substeps, fused_narrow, integrate_on_load)

    out = {
        "metric": "body-substeps/sec (100k-body pile, 4 substeps)" if slab is None else
                  f"body-substeps/sec (granular slab, {args.slab_bodies // 1000}k bodies per GPU, 4 substeps)",
        "value": value,
        "unit": "body-substeps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"mixed_pile: {n_dyn} circle/rect/capsule bodies settled in a static container, "
                         f"{scene.substeps} substeps, dt=1/60 (BASELINE.json configs[2])") if slab is None else
                        (f"granular_slab: {n_dyn} circles settled in a static container, cut into {world} slabs on x, "
                         f"{scene.substeps} substeps, dt=1/60 (BASELINE.json configs[4] at {args.slab_bodies} bodies per GPU: 4 M at 8 GPUs)"),
            "scene": scene_note,
            "penetration_mm": pen,
            "v_rms": v_rms,
            "bodies_per_gpu": n_rank,
            "substeps": scene.substeps,
            "parallelism": parallelism,
            **({"setup_s": setup} if setup else {}),     # untimed: cut, settle, gather, relax, final cut (wall seconds on rank 0)
            "pairs": st.n_pairs,
            "manifolds": st.n_manifolds,
            "colors": st.n_colors,
            "tiles": st.n_tiles,
            "fallback_pairs": st.n_fallback,
            "launches_per_step": st.launches,
            "integration_on_load": bool(integrate_on_load),   # no k_integrate: DESIGN.md "Integration on load" (SFX_FUSE_TICK=0 keeps it)
        },
        "roofline": {
            "bound": "hbm",
            "kernel": names[dom],
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": load_traffic(dom) if slab is None else None,   # the PMC passes were taken on the N = 1 pile
            "bytes_per_launch": sp["bytes_per_launch"],
            "launch_us": sp["launch_us"],
            "launches_per_step": kt[dom]["launches"] / prof_frames,
            "launch_us_second_leg": per_kernel[dom]["launch_us"],
            "other_kernels": {names[c].split(" ")[0]: {"achieved": round(v["achieved"], 1), "frac": round(v["achieved"] / HBM_PEAK_GBS, 4),
                                                                "launch_us": round(v["launch_us"], 2), "bytes_per_launch": v["bytes_per_launch"],
                                                                "traffic": load_traffic(c) if slab is None else None}
                              for c, v in per_kernel.items() if c != dom},
            "tick_bytes_per_body_substep": bpbs,
            "tick_frac": (value / world) * bpbs / (HBM_PEAK_GBS * 1e9),
            "kernel_ms_per_step": kernel_ms,
            "note": "launch time of the dominant kernel from hipEvents recorded by libsfx around its launches on the stream they "
                    f"run on, over {measured_over}; other_kernels and kernel_ms_per_step from a second leg of {prof_frames} "
                    "ticks with events around every launch; traffic = 2 x FETCH_SIZE + WRITE_SIZE from "
                    "rocprofv3 --pmc passes of the native driver on the same settled scene (profiles/)",
        },
    }
    if slab is not None and single is not None:
        out["scaling_reference"] = single
    if args.rehearse_on_one_gpu:
        out["rehearsal"] = f"{world} ranks sharing ONE GPU over gloo with host-staged halos: a test of the N > 1 code path, not a measurement"
    if dist is not None:
        dist.barrier()
    w.close()
    if slab is None and not args.no_survey_scene and args.bodies == 100_000 and args.width != 400:
        out["config"]["survey_scene"] = survey_scene_leg(device, args.bodies, args.steps, args.warmup)
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(scene)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
