"""The C-ABI library: loads without a GPU, exports exactly what include/crucible_hip.h
declares, and the ctypes mirror matches the C struct layouts.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from crucible_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "crucible_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"CR_API\s+[\w\s\*]+?\b(cr_\w+)\s*\(", text)))


def test_header_symbols_match_python_table():
    assert declared_symbols() == sorted(A.SYMBOLS)


def test_library_exports_every_declared_symbol(hiplib):
    for name in declared_symbols():
        assert hasattr(hiplib, name), name
    assert hiplib.cr_abi_version() == A.CR_ABI_VERSION


def test_struct_layouts_match_header():
    structs = ["CrPrimitive", "CrMaterial", "CrTexture", "CrImage", "CrKeyframe", "CrSceneDesc", "CrCameraDesc",
               "CrRenderParams", "CrStats", "CrGroupStats"]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "crucible_hip.h"\nint main(){\n'
    for s in structs:
        src += f'printf("{s} %zu\\n", sizeof({s}));\n'
        for fname, _ in getattr(A, s)._fields_:
            src += f'printf("{s}.{fname} %zu\\n", offsetof({s}, {fname}));\n'
    src += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        lines = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in lines if l)
    for s in structs:
        cls = getattr(A, s)
        assert int(got[s]) == C.sizeof(cls), s
        for fname, _ in cls._fields_:
            assert int(got[f"{s}.{fname}"]) == getattr(cls, fname).offset, f"{s}.{fname}"


def test_create_without_gpu_fails_loudly(hiplib):
    """No CPU fallback: on a box without a HIP device cr_create reports CR_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = hiplib.cr_create(0, C.byref(h))
    assert rc == A.CR_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in hiplib.cr_last_error(None)


def test_group_entry_points_without_gpu(hiplib):
    """The multi-GPU entry points: the shard arithmetic is pure (no device needed); creating a group without a HIP
    device fails like cr_create; RCCL is not a link-time dependency (it is dlopen'ed when a group needs it)."""
    import torch
    from crucible_amd.distributed import shard_range
    from crucible_amd.group import shard
    for spp, world in ((512, 8), (512, 1), (10, 3), (4, 8), (0, 2), (1024, 6), (7, 7)):
        ranges = [shard(spp, r, world) for r in range(world)]
        assert ranges == [shard_range(r, world, spp) for r in range(world)]
        assert ranges[0][0] == 0 and sum(n for _, n in ranges) == spp
        assert all(ranges[r][0] + ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
        assert max(n for _, n in ranges) - min(n for _, n in ranges) <= 1
    b, n = C.c_int32(), C.c_int32()
    for bad in ((8, 3, 3), (8, -1, 3), (8, 0, 0), (-1, 0, 1)):
        assert hiplib.cr_group_shard(*bad, C.byref(b), C.byref(n)) == A.CR_ERR_INVALID_ARG
    assert hiplib.cr_group_size(None) == 0 and hiplib.cr_group_rank(None) == -1 and not hiplib.cr_group_handle(None, 0)
    hiplib.cr_group_destroy(None)
    g = C.c_void_p()
    assert hiplib.cr_group_create(None, 0, C.byref(g)) == A.CR_ERR_INVALID_ARG
    two = (C.c_int32 * 2)(0, 0)
    assert hiplib.cr_group_create(two, 2, C.byref(g)) == A.CR_ERR_INVALID_ARG and b"twice" in hiplib.cr_group_last_error(None)
    assert hiplib.cr_group_create_rank(0, 3, 2, None, C.byref(g)) == A.CR_ERR_INVALID_ARG
    if not torch.cuda.is_available():
        one = (C.c_int32 * 1)(0)
        assert hiplib.cr_group_create(one, 1, C.byref(g)) == A.CR_ERR_NO_DEVICE and not g.value
        assert hiplib.cr_group_create_rank(0, 0, 1, None, C.byref(g)) == A.CR_ERR_NO_DEVICE
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "crucible_amd", "libcrucible_hip.so")]).decode()
    assert "rccl" not in out and "nccl" not in out


def test_null_arguments_are_rejected(hiplib):
    assert hiplib.cr_create(0, None) == A.CR_ERR_INVALID_ARG
    assert hiplib.cr_upload_scene(None, None) == A.CR_ERR_INVALID_ARG
    assert hiplib.cr_synchronize(None) == A.CR_ERR_INVALID_ARG
    assert hiplib.cr_write_ppm(None, None, 0, 1, 1) == A.CR_ERR_INVALID_ARG
    hiplib.cr_destroy(None)   # no-op


def test_product_never_touches_the_oracle():
    """The package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "crucible_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile", ".map")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in (r"^\s*(from|import)\s+oracle", r"liboracle", r"oracle[/\\.]", r"crucible_oracle", r"oracle_\w+\("):
                    assert not re.search(pat, text, re.M), (os.path.join(dirpath, f), pat)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libcrucible_hip.so")]).decode()
    assert "oracle" not in out
