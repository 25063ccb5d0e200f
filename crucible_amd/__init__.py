"""crucible_amd -- MI355X (gfx950) path-tracing integrator standing in for
kylittle/Crucible's per-pixel render loop (Camera::render and everything under it).

  include/crucible_hip.h        the C ABI (the drop-in boundary)
  crucible_amd/csrc/            HIP kernels + the C ABI implementation
  crucible_amd/scene.py         host-side mirror of Crucible's Scene/Camera builder API
  crucible_amd/demo_builder.py  seeded restatements of Crucible's demo scenes
  crucible_amd/renderer.py      ctypes binding of the library
"""
from . import _abi  # noqa: F401
from ._abi import CR_REAL_F32, CR_REAL_F64  # noqa: F401
