"""The narrowphase inside the position sweep (sfx_kernels.h narrow_batch, DESIGN.md "Narrowphase inside the position sweep"):
libsfx's default computes every batch's records in the workgroup that solves the batch; SFX_FUSE_NARROW=0 keeps the
separate k_narrow launch, 2 fuses the later phases only.  Same function on the same inputs: the three must end in the
same bits -- on soups with compounds and fallback lists, on square tiles and strips, with joints, through sfx_step and
substep by substep.  (The parity tests against the oracle run on the default.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(mode):
    e = dict(os.environ, SFX_FUSE_NARROW=str(mode))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "narrow_worker.py")], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_fused_narrowphase_equals_the_separate_launch():
    sep, fused, later = worker(0), worker(1), worker(2)
    assert sep["narrow_launches"] > 0 and later["narrow_launches"] > 0 and fused["narrow_launches"] == 0
    assert fused["solver_pairs"] > 20_000 and fused["fallback"] > 0, fused   # the fallback list's records come from the sweep too
    assert fused["state"] == sep["state"] == later["state"]
