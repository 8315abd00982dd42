"""The .feat writer's number formatting (regard3d_amd/csrc/fmt_g6.hpp) against printf("%g") in the "C" locale as std::to_chars
implements it -- SIOPointFeature::operator<< of the reference writes "x y scale orientation" with the stream's default float
formatting (src/keypointSet.hpp:61-67) -- and the value a reader parses back from that text (std::from_chars): random values of the
writer's ranges, every float around the decade boundaries, ties of the sixth digit, carries, zeros, infinities, any bit pattern."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_percent_g_is_printf_percent_g(tmp_path):
    exe = str(tmp_path / "fmt_g6_test")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "fmt_g6_test.cpp")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "1500000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-3000:]
