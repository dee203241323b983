"""SURVEY.md section 5 "race detection": the colouring invariant -- within one colour no body is written twice -- checked
ON THE DEVICE, inside the real fused launches.  libsfx_races.so is libsfx built with
-DSFX_DEBUG_RACES: every write of a body's LDS record in a colour step of k_tile_pos / k_tile_vel stamps the body with the
colour by an atomic exchange and reports a second writer as overflow bit 256.  The workers run in processes of their own
(the library is chosen per process by SFX_LIBRARY)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RACES = os.path.join(ROOT, "starframe_amd", "csrc", "libsfx_races.so")


def worker(which, **env):
    assert os.path.exists(RACES), "libsfx_races.so is missing: run __graft_entry__.build()"
    e = dict(os.environ, SFX_LIBRARY=RACES, **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "race_worker.py"), which], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_no_colour_writes_a_body_twice_on_the_pile():
    r = worker("pile")
    assert r["ticks"] >= 200 and r["solver_pairs"] > 300_000
    assert r["overflow"] == 0, r


def test_no_colour_writes_a_body_twice_on_random_soups():
    r = worker("soups")
    assert r["ticks"] == 72
    assert r["overflow"] & 256 == 0, r


def test_the_detector_reports_a_second_writer():
    """The report path itself: with SFX_RACE_SELFTEST=1 every write is stamped twice, which must come back as bit 256."""
    r = worker("soups", SFX_RACE_SELFTEST="1")
    assert r["overflow"] & 256, r
