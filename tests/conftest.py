import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The suite's bit-exact tests run in the parity mode: CR_SUM_DEFAULT means the reference's summation order for every
# handle created in this session (and in the CLI subprocesses the tests start).  tests/test_gpu_relaxed.py asks for
# CR_SUM_RELAXED -- the library's own default -- explicitly.
os.environ.setdefault("CRUCIBLE_SUM_ORDER", "reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracles():
    from crucible_amd import _abi as A
    from oracle.oracle import Oracle
    return {A.CR_REAL_F64: Oracle(A.CR_REAL_F64), A.CR_REAL_F32: Oracle(A.CR_REAL_F32)}


@pytest.fixture(scope="session")
def o64(oracles):
    from crucible_amd import _abi as A
    return oracles[A.CR_REAL_F64]


@pytest.fixture(scope="session")
def o32(oracles):
    from crucible_amd import _abi as A
    return oracles[A.CR_REAL_F32]


@pytest.fixture(scope="session")
def hiplib():
    """The built C-ABI library (loads without a GPU; compute calls need one)."""
    lib_path = os.path.join(ROOT, "crucible_amd", "libcrucible_hip.so")
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build()
    from crucible_amd.renderer import load_library
    return load_library()


@pytest.fixture(scope="session")
def renderer(hiplib):
    from crucible_amd.renderer import Renderer
    r = Renderer(0)   # raises without a GPU: gpu tests must not silently fall back
    yield r
    r.close()
