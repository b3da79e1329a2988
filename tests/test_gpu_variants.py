"""Tuning knobs must not change results: ring depth, non-temporal loads, tiles per range (1 .. 960)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env", [
    {"SCFQ_RING": "3", "SCFQ_NT": "0"}, {"SCFQ_RING": "4", "SCFQ_NT": "1"}, {"SCFQ_RING": "2", "SCFQ_NT": "0"},
    {"SCFQ_TILES_PER_RANGE": "1"}, {"SCFQ_TILES_PER_RANGE": "2"}, {"SCFQ_TILES_PER_RANGE": "3"},
    {"SCFQ_TILES_PER_RANGE": "7"}, {"SCFQ_TILES_PER_RANGE": "960"}, {},
])
def test_knobs_do_not_change_results(gpu, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_variant_check.py"), "7", "45"], env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "variant ok" in r.stdout
