"""The substeps of a tick as one launch (sfx_kernels.h k_tile_tick, DESIGN.md "The substeps of a tick as one launch"): sfx_step
runs the position and velocity sweeps of all substeps as the items of ONE launch, integrates bodies where they are loaded and
lets a batch wait for the batches around it instead of the device for the end of a launch; SFX_FUSE_TICK=0 keeps a launch per
stage.  Same functions in the same order per body and pair: the two must end in the same bits -- on soups (with moving
non-dynamic bodies), square tiles and strips, with a fallback list, and when a world changes between the two paths from tick
to tick.  (The parity tests against the oracle run on the default.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(mode):
    e = dict(os.environ, SFX_FUSE_TICK=str(mode))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tick_worker.py")], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_one_launch_tick_equals_a_launch_per_stage():
    staged, fused = worker(0), worker(1)
    assert staged["tick_launches"] == 0 and fused["tick_launches"] > 0
    assert fused["solver_pairs"] > 50_000, fused
    for a, b in zip(staged["per_world"], fused["per_world"]):
        assert a == b, (a, b)
    assert fused["fallback"] > 0, fused   # the fallback list inside the launch was exercised
