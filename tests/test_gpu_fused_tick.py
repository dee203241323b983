"""Integration on load and the substeps of a tick as one launch (DESIGN.md sections of those names; sfx_host.hip tick_fused).
sfx_step runs the substeps of a world without joints, further colliders or halos WITHOUT k_integrate (SFX_FUSE_TICK=3, the
default): the narrowphase integrates the bodies it loads, the velocity sweep the ones it derives, one k_tick_free steps the
bodies no batch holds.  SFX_FUSE_TICK=1 runs the sweeps of all substeps as the items of ONE launch (a batch waits for the
batches around it, not the device for the end of a launch), 2 the same per substep behind a narrowphase launch; 0 keeps
k_integrate and a launch per stage.  Same functions in the same order per body and pair: all four must end in the same
bits -- on soups (with moving non-dynamic bodies), square tiles and strips, with a fallback list, and when a world changes
between the paths from tick to tick.  (The parity tests against the oracle run on the default.)"""
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
def test_integration_on_load_and_one_launch_ticks_equal_a_launch_per_stage():
    staged, one, per_substep, on_load = worker(0), worker(1), worker(2), worker(3)
    assert staged["tick_launches"] == 0 and on_load["tick_launches"] == 0
    assert one["tick_launches"] == one["ticks"] and per_substep["tick_launches"] == 4 * per_substep["ticks"]
    assert staged["integrate_launches"] == 4 * staged["ticks"] and on_load["integrate_launches"] == on_load["ticks"]   # k_integrate per substep / one k_tick_free
    assert on_load["solver_pairs"] > 50_000, on_load
    for w in (one, per_substep, on_load):
        for a, b in zip(staged["per_world"], w["per_world"]):
            assert a == b, (a, b)
    assert on_load["fallback"] > 0, on_load   # the fallback list's integration on load was exercised
