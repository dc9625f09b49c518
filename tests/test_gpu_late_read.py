"""Stress test behind DESIGN.md section 7's "late read" (round 5: once, a one-rank cap_dmp factor was read while its first step was still
running, after the plan's info query and a device synchronisation had both returned).  Round 6 makes that structurally impossible - every
plan's info query drains the plan's own helper streams, and the multi-rank plans create their streams and events with the plan, never inside
the first factor call - and this test hammers exactly the situation: 3 x 700 fresh one-rank plans (cap_dmp, cap_dist, cap_mpchol), each
factored once and read at once, next to three idle peer processes holding contexts on the GPU.  One differing read fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fresh_plans_factored_once_and_read_at_once_beside_idle_peers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "late_read_worker.py"), "700", "3"], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert r.stdout.count("0 late or wrong reads") == 3, r.stdout[-3000:]
