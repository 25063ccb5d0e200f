"""ctypes binding of libcrucible_hip.so (include/crucible_hip.h).

This is the only route to pixels in the package: if the HIP library is missing
or no GPU is present, construction raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRUCIBLE_HIP_LIB") or os.path.join(_HERE, "libcrucible_hip.so")   # override: diagnostic builds only
_lib = None


class CrucibleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"crucible_hip error {code}: {msg}")
        self.code = code


def load_library():
    """dlopen the in-tree library and bind every symbol the header declares."""
    global _lib
    if _lib is None:
        # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64; if this
        # library pulled in /opt/rocm's copies first, torch would later load a second runtime that finds no
        # GPU.  Importing torch first makes both share one (measured on the MI355X box, ROCm 7.2 + torch 2.10).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "or `make -C crucible_amd/csrc`")
        _lib = A.bind(C.CDLL(LIB_PATH))
        if _lib.cr_abi_version() != A.CR_ABI_VERSION:
            raise RuntimeError("libcrucible_hip.so ABI version mismatch")
    return _lib


def np_real(real_type):
    return np.float64 if real_type == A.CR_REAL_F64 else np.float32


class Renderer:
    """One CrHandle (one HIP device).  Mirrors the call `Camera::render(&skybox, &world, fname)`
    (reference src/camera/mod.rs:270) split into upload_scene / render / write_ppm."""

    def __init__(self, device=0, sum_order=A.CR_SUM_DEFAULT):
        self.lib = load_library()
        # CrRenderParams.sum_order of this renderer's renders unless a call names its own: CR_SUM_DEFAULT is the
        # library's choice (relaxed), CR_SUM_REFERENCE_ORDER the parity mode the bit-exact tests run in
        self.sum_order = sum_order
        h = C.c_void_p()
        rc = self.lib.cr_create(device, C.byref(h))
        if rc != A.CR_OK:
            raise CrucibleError(rc, (self.lib.cr_last_error(None) or b"").decode())
        self.h = h

    def _check(self, rc):
        if rc != A.CR_OK:
            raise CrucibleError(rc, (self.lib.cr_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.lib.cr_destroy(self.h)
            self.h = None

    def __del__(self):
        if getattr(self, "h", None):
            self.close()

    def upload_scene(self, flat):
        self._check(self.lib.cr_upload_scene(self.h, C.byref(flat.desc)))

    def render(self, cam, *, seed, real_type=A.CR_REAL_F32, sample_begin=0, sample_count=None, output_sum=False,
               want_stats=True, sum_order=None):
        """Render into a host array (H, W, 3) of f32/f64.  Returns (image, stats dict)."""
        cd = cam.desc()
        p = cam.params(seed, real_type, sample_begin, sample_count, output_sum, self.sum_order if sum_order is None else sum_order)
        out = np.empty((cam.image_height, cam.image_width, 3), dtype=np_real(real_type))
        st = A.CrStats()
        self._check(self.lib.cr_render_host(self.h, C.byref(cd), C.byref(p), out.ctypes.data_as(C.c_void_p),
                                            C.byref(st) if want_stats else None))
        return out, st.as_dict()

    def render_device(self, cam, d_ptr, *, seed, real_type=A.CR_REAL_F32, sample_begin=0, sample_count=None,
                      output_sum=False, want_stats=False, sum_order=None):
        """Render into device memory at `d_ptr` (W*H*3 reals).  Asynchronous unless want_stats."""
        cd = cam.desc()
        p = cam.params(seed, real_type, sample_begin, sample_count, output_sum, self.sum_order if sum_order is None else sum_order)
        st = A.CrStats()
        self._check(self.lib.cr_render_device(self.h, C.byref(cd), C.byref(p), C.c_void_p(d_ptr),
                                              C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def export_bvh(self, real_type=A.CR_REAL_F32):
        """The wrapper tree the device walks: (boxes (n, 6) f64, children (n, 2) i32, split_axis (n,) i32), see
        cr_export_bvh."""
        n = C.c_int32()
        self._check(self.lib.cr_export_bvh(self.h, real_type, None, None, None, 0, C.byref(n)))
        boxes = np.zeros((max(1, n.value), 6), dtype=np.float64)
        kids = np.zeros((max(1, n.value), 2), dtype=np.int32)
        axis = np.full(max(1, n.value), -1, dtype=np.int32)
        self._check(self.lib.cr_export_bvh(self.h, real_type, boxes.ctypes.data_as(C.c_void_p),
                                           kids.ctypes.data_as(C.c_void_p), axis.ctypes.data_as(C.c_void_p), n.value,
                                           C.byref(n)))
        return boxes[:n.value], kids[:n.value], axis[:n.value]

    def last_kernel_ms(self):
        ms = C.c_double()
        self._check(self.lib.cr_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def synchronize(self):
        self._check(self.lib.cr_synchronize(self.h))

    def stream(self):
        return self.lib.cr_stream(self.h)

    def write_ppm(self, path, img):
        img = np.ascontiguousarray(img)
        rt = A.CR_REAL_F64 if img.dtype == np.float64 else A.CR_REAL_F32
        rc = self.lib.cr_write_ppm(path.encode(), img.ctypes.data_as(C.c_void_p), rt, img.shape[1], img.shape[0])
        if rc != A.CR_OK:
            raise CrucibleError(rc, "cannot write " + path)

    def write_image(self, path, img):
        """P3 (.ppm, the reference's format), P6 (.pbm6 / .p6.ppm) or PNG by extension; same bytes per channel."""
        img = np.ascontiguousarray(img)
        rt = A.CR_REAL_F64 if img.dtype == np.float64 else A.CR_REAL_F32
        fn = self.lib.cr_write_png if path.endswith(".png") else (self.lib.cr_write_ppm_binary if path.endswith(".p6.ppm") else self.lib.cr_write_ppm)
        rc = fn(path.encode(), img.ctypes.data_as(C.c_void_p), rt, img.shape[1], img.shape[0])
        if rc != A.CR_OK:
            raise CrucibleError(rc, "cannot write " + path)


def quantize_rgb8(img):
    """impl Display for Color (reference src/utils.rs:422-437) over a whole image -> uint8 (H, W, 3)."""
    lib = load_library()
    img = np.ascontiguousarray(img)
    rt = A.CR_REAL_F64 if img.dtype == np.float64 else A.CR_REAL_F32
    out = np.empty(img.shape, dtype=np.uint8)
    rc = lib.cr_quantize_rgb8(img.ctypes.data_as(C.c_void_p), rt, img.size // 3, out.ctypes.data_as(C.c_void_p))
    if rc != A.CR_OK:
        raise CrucibleError(rc, "quantize")
    return out
