"""The short form of the f64 square root (crucible_amd/csrc/pathtrace.hpp r_sqrt) returns the compiler's bits; `as i32` written as the
conversion instruction returns what the guarded cast returns."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_short_sqrt_and_as_i32_match_their_long_forms(tmp_path):
    exe = tmp_path / "sqrt_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                    "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-function", "-Wno-unused-value", "-I", os.path.join(ROOT, "crucible_amd", "csrc"),
                    "-o", str(exe), os.path.join(ROOT, "tests", "sqrt_check.hip")], check=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "\nmismatches 0 of" in "\n" + r.stdout and "as_i32 mismatches 0 of" in r.stdout
