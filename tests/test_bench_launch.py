"""bench.py started plainly with --gpus N must run N ranks itself: never the N = 1 workload under
an N > 1 flag.  --dry-run stops before any device or library call, so this runs on the CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_plain_start_with_gpus_2_launches_two_ranks():
    r = run(["--gpus", "2", "--dry-run", "--steps", "7"])
    assert r.returncode == 0, r.stderr
    line = last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 7 and line["dry_run"] is True


def test_plain_start_with_one_gpu_stays_one_process():
    r = run(["--dry-run"])
    assert r.returncode == 0, r.stderr
    assert last_json(r.stdout)["n_gpus"] == 1


def test_a_failing_rank_fails_the_launch_and_prints_no_line():
    r = run(["--gpus", "2", "--dry-run"], SFX_BENCH_FAIL_RANK="1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_that_contradicts_gpus_is_refused():
    r = run(["--gpus", "1", "--dry-run"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    r = run(["--gpus", "2", "--dry-run"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
