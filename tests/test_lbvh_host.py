"""The per-node functions of the device LBVH builder (crucible_amd/csrc/lbvh.hpp) are host/device: this test
compiles them for the host and checks Karras' construction on 4000 random key sets (unique keys, heavy duplicates,
all keys equal, clustered keys) -- no GPU involved.  The device kernels call the same functions."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_karras_topology_on_the_host(tmp_path):
    exe = str(tmp_path / "lbvh_check")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "crucible_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "lbvh_check.cpp")], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok 4000 trees"), out.stdout
    # Morton keys: origin -> 0, far corner -> all 63 bits, +x alone -> every third bit from the top
    assert "key(0)=0 key(1)=7fffffffffffffff key(x)=4924924924924924" in out.stdout
