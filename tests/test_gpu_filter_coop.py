"""The cooperative AC-RANSAC kernel (a long pair spread over several workgroups, kernels_filter_coop.hip): identical to the
one-workgroup-per-pair kernel at 1 / 2 / 4 / 8 slices per pair, with fewer workers than slices, and with the invariant checks of the
developer build on (every model evaluated in full, histogram bound <= NFA, slice counts == full-pass counts)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cooperative_filter_equals_the_one_workgroup_kernel():
    env = {k: v for k, v in os.environ.items() if not k.startswith("R3DM_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "filter_coop_check.py"), "7", "5000", "9000"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "identical" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
