"""The two marching kernels of the detector (regard3d_amd/csrc/fed_march.inc: K FED steps per pass in registers; level_head.inc: Gaussian
+ derivatives + determinant + conductivity per pass through LDS rings) share their control flow with CPU emulations: the same template
code compiled by g++ with a 64-float struct in place of a wavefront, compared bit for bit with the plain per-pixel passes they replace
(/root/reference/src/thirdparty/fast-akaze/AKAZEFeatures.cpp:300-367 as restated in oracle/akaze.c) -- strips, bands, image borders,
tiny images.  No GPU needed: this is how the index arithmetic of both kernels was validated before they first ran."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["fed_march_emul", "level_head_emul"])
def test_marching_kernel_control_flow_on_the_cpu(tmp_path, name):
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", exe, src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "identical" in r.stdout, r.stdout[-3000:]
