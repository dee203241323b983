"""Joints by component (sfx_kernels.h k_joint_groups, DESIGN.md "Joints by component"): the connected components of the joint
graph share no body, so one workgroup takes a group of whole components through ALL joint colours of a substep in one launch
(workgroup barriers between the colours) where SFX_JOINT_GROUPS=0 keeps a launch per colour.  Per body the same solves in the
same order: the two must end in the same bits -- chains, damped springs, hinges and pins in soups -- and a world with a
component larger than a group keeps the launch per colour.  (The parity tests against the oracle run on the default.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(groups):
    e = dict(os.environ, SFX_JOINT_GROUPS=str(groups))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "joint_worker.py")], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])["per_world"]


@pytest.mark.gpu
def test_joints_by_component_equal_a_launch_per_colour():
    per_colour, grouped = worker(0), worker(1)
    fewer = 0
    for a, b in zip(per_colour, grouped):
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3], (a, b)   # same world, same joint colours, same bits
        assert b[2] <= a[2]
        fewer += b[2] < a[2]
    assert grouped[0][2] < per_colour[0][2] and grouped[1][2] < per_colour[1][2]   # the chains took the one launch
    assert grouped[2][2] == per_colour[2][2]                                       # a 400-joint component does not fit a group
    assert fewer >= 3
