#!/usr/bin/env python3
"""bench.py -- body-substeps/s of the libsfx physics tick on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one physics tick (one frame: broadphase + colouring + `substeps` substeps) of the
workload; value = dynamic bodies x substeps x K / seconds, summed over all ranks.

N = 1 : the headline config (BASELINE.json configs[2]): 100k mixed circle/rect/capsule pile,
        4 substeps, settled on the GPU before anything is timed.
N > 1 : weak scaling, one process per GPU: a pile N times as wide (N x 100k bodies) cut into N
        slabs on x (starframe_amd/slab.py); every substep the boundary bodies go to the two
        neighbours with RCCL send/recv (torch.distributed "nccl") straight from the library's
        device buffers.  value = all bodies x substeps x K / max-over-ranks seconds.

The JSON line also carries
  roofline     : the dominant kernel (k_tile_pos: narrowphase + XPBD manifold block solves of one
                 phase of solver tiles, bodies in LDS), algorithmic bytes per launch / mean launch
                 time from HIP events recorded by libsfx on its own stream, against the 8 TB/s
                 HBM peak;
  cpu_baseline : this repository's single-threaded CPU restatement (oracle/, kind "port")
                 timed on this host on a bounded number of frames of the same settled scene.
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


def solve_pos_bytes(st):
    """Algorithmic (compulsory) bytes of the four k_tile_pos launches of ONE substep.
    DESIGN.md "Kernels": per candidate pair 16 B pair record + 2 x (16 B pose + 16 B shape) read
    + 8 B count written; per touching pair additionally 2 x (8 B mass + 16 B velocity) read,
    48 B record + 16 B lambdas + 2 x 16 B pose written; 16 B more for a second contact point."""
    return st.n_pairs * 88.0 + st.n_manifolds * 144.0 + st.n_manifolds2 * 16.0


def tick_bytes_per_body_substep(st, n_dyn, substeps):
    """SURVEY.md 8d accounting for the whole tick, from measured pair / contact counts."""
    p, m, m2 = st.n_pairs / n_dyn, st.n_manifolds / n_dyn, st.n_manifolds2 / n_dyn
    integrate = 16 + 16 + 8 + 16 + 16 + 16          # pose, vel, mass read; prev, pose, vel written
    derive = 16 + 16 + 8 + 16
    pos = p * 88 + m * 144 + m2 * 16
    vel = p * (16 + 8) + m * (64 + 2 * (16 + 16 + 8) + 2 * 16) + m2 * 16
    broad = (16 + 16 + 16 + 16 + 4 + 8 + 4 + 4 + 28 + 16 * 5 + p * (8 + 8 + 2 + 16)) / substeps
    return integrate + derive + pos + vel + broad


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
        "value": n_dyn * scene.substeps * frames / el,
        "unit": "body-substeps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{frames} ticks of the same settled {scene.name} scene ({n_dyn} bodies, {scene.substeps} substeps), "
                  f"sequential Gauss-Seidel order; this repo's CPU restatement, not starframe's Rust tick (unavailable)",
        "ms_per_step": 1e3 * el / frames,
    }


def load_traffic():
    """HBM bytes per k_solve_pos launch from the committed rocprofv3 --pmc passes, if any."""
    p = os.path.join(ROOT, "profiles", "pmc_solve_pos.json")
    try:
        with open(p) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--bodies", type=int, default=100_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--settle-frames", type=int, default=600)
    ap.add_argument("--force-slab", action="store_true",
                    help="rehearse the N > 1 code path (slab driver over torch.distributed nccl) with one rank")
    args = ap.parse_args()
    rank, local_rank, world = dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}")

    from starframe_amd import scenes
    from starframe_amd.world import World

    device = local_rank
    dist = None
    slab = None
    if world > 1 or args.force_slab:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))

        from starframe_amd.slab import DistSlab

        # one pile, `world` times as wide; settled as slabs with damping, then re-cut on the settled state
        scene = scenes.mixed_pile(n=world * args.bodies, width=world * 1000)
        pre = DistSlab(scene, World, dist, device_index=device, linear_damping=3.3, angular_damping=1.0)
        for _ in range(args.settle_frames):
            pre.step(DT)
        pos, rot, _, _ = pre.gather()
        pre.close()
        scene = scene.with_state(pos, rot, name=scene.name + "_settled")
        slab = DistSlab(scene, World, dist, device_index=device)
        w = slab.world
        n_dyn = scene.n_dynamic()            # all ranks together
        n_local = slab.plan.n_owned
        halo = len(slab.plan.send_idx)
        parallelism = (f"{world} slabs on x, {n_local} owned + {len(slab.plan.recv_idx)} ghost bodies on rank 0, "
                       f"halo of {halo} bodies x 32 B to <= 2 neighbours per substep over "
                       f"{'RCCL device buffers' if slab.on_device else 'host staging'}")
        step = lambda: slab.step(DT)
    else:
        scene = scenes.mixed_pile(n=args.bodies)
        parallelism = "1 GPU"
        scene = settle(scene, device, frames=args.settle_frames)
        n_dyn = scene.n_dynamic()
        w = World.from_scene(scene, device=device)
        step = lambda: w.step(DT)

    def barrier():
        w.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    w.sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    st = w.stats()

    # roofline leg: the same ticks again with per-launch HIP events (kept out of `value`)
    prof_frames = max(8, min(args.steps, 60))
    w.set_profiling(True)
    w.kernel_times(reset=True)
    for _ in range(prof_frames):
        step()
    kt = w.kernel_times()
    w.set_profiling(False)
    st2 = w.stats()
    sp = kt["solve_pos"]
    launches_per_substep = sp["launches"] / (prof_frames * scene.substeps)
    dur_s = 1e-3 * sp["total_ms"] / max(sp["launches"], 1)
    bytes_per_launch = solve_pos_bytes(st2) / max(launches_per_substep, 1)
    achieved = bytes_per_launch / dur_s / 1e9
    kernel_ms = {k: round(v["total_ms"] / prof_frames, 4) for k, v in kt.items() if v["launches"]}
    value = n_dyn * scene.substeps * args.steps / elapsed
    n_rank = slab.plan.n_owned if slab is not None else n_dyn
    bpbs = tick_bytes_per_body_substep(st2, n_rank, scene.substeps)

    out = {
        "metric": "body-substeps/sec (100k-body pile, 4 substeps)",
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
            "workload": f"mixed_pile: {n_dyn} circle/rect/capsule bodies settled in a static container, "
                        f"{scene.substeps} substeps, dt=1/60 (BASELINE.json configs[2])",
            "bodies_per_gpu": n_rank,
            "substeps": scene.substeps,
            "parallelism": parallelism,
            "pairs": st.n_pairs,
            "manifolds": st.n_manifolds,
            "colors": st.n_colors,
            "tiles": st.n_tiles,
            "fallback_pairs": st.n_fallback,
            "launches_per_step": st.launches,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_tile_pos (narrowphase + XPBD manifold block solves of one tile phase, bodies in LDS; "
                      "4 launches per substep)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": load_traffic(),
            "bytes_per_launch": bytes_per_launch,
            "launch_us": dur_s * 1e6,
            "launches_per_step": sp["launches"] / prof_frames,
            "tick_bytes_per_body_substep": bpbs,
            "tick_frac": (value / world) * bpbs / (HBM_PEAK_GBS * 1e9),
            "kernel_ms_per_step": kernel_ms,
            "note": "launch time from hipEvents recorded by libsfx around every launch on its own stream, "
                    f"over {prof_frames} further ticks after the timed region",
        },
    }
    if dist is not None:
        dist.barrier()
    w.close()
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(scene)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
