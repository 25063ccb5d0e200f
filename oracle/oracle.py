"""ctypes loader for the CPU oracle (oracle/crucible_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under crucible_amd/ imports this.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from crucible_amd import _abi as A  # noqa: E402  (struct layouts of include/crucible_hip.h only)


def build(force=False):
    libs = [os.path.join(HERE, f"liboracle_{t}.so") for t in ("f64", "f32")]
    src = os.path.join(HERE, "crucible_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "crucible_hip.h")
    stale = force or any(not os.path.exists(l) or os.path.getmtime(l) < max(os.path.getmtime(src), os.path.getmtime(hdr))
                         for l in libs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return libs


class Oracle:
    """One precision of the oracle.  real_type: A.CR_REAL_F64 (the reference's) or A.CR_REAL_F32."""

    def __init__(self, real_type=A.CR_REAL_F64):
        build()
        self.real_type = real_type
        name = "liboracle_f64.so" if real_type == A.CR_REAL_F64 else "liboracle_f32.so"
        self.lib = C.CDLL(os.path.join(HERE, name))
        self.real = C.c_double if real_type == A.CR_REAL_F64 else C.c_float
        self.np_real = np.float64 if real_type == A.CR_REAL_F64 else np.float32
        L, R, P = self.lib, self.real, C.c_void_p
        assert L.oracle_real_type() == real_type
        L.oracle_scene_create.restype = P
        L.oracle_scene_create.argtypes = [C.POINTER(A.CrSceneDesc)]
        L.oracle_scene_destroy.argtypes = [P]
        L.oracle_render.argtypes = [P, C.POINTER(A.CrCameraDesc), C.POINTER(A.CrRenderParams), C.c_int64, C.c_int64, P,
                                    C.c_int32, C.POINTER(A.CrStats)]
        for n in ("oracle_dot", "oracle_length", "oracle_degrees_to_radians", "oracle_radians_to_degrees",
                  "oracle_interval_size", "oracle_interval_proportion"):
            getattr(L, n).restype = R
        L.oracle_dot.argtypes = [P, P]
        L.oracle_length.argtypes = [P]
        L.oracle_degrees_to_radians.argtypes = [R]
        L.oracle_radians_to_degrees.argtypes = [R]
        L.oracle_interval_size.argtypes = [R, R]
        for n in ("contains", "surrounds", "is_greater", "is_less"):
            getattr(L, "oracle_interval_" + n).argtypes = [R, R, R]
        L.oracle_interval_proportion.argtypes = [R, R, R]
        L.oracle_color_valid.argtypes = [R, R, R]
        L.oracle_color_display.argtypes = [C.c_double, C.c_double, C.c_double, P]
        L.oracle_color_scale.argtypes = [R, P, P]
        L.oracle_color_div.argtypes = [P, R, P]
        L.oracle_refract.argtypes = [P, P, R, P]
        L.oracle_ray_at.argtypes = [P, P, R, P]
        L.oracle_average_samples.argtypes = [P, C.c_int32, P]
        L.oracle_timeline_eval.argtypes = [P, C.POINTER(A.CrKeyframe), C.c_int32, C.c_int32, R, P]
        L.oracle_aabb_hit.argtypes = [P, P, P, R, R]
        L.oracle_sphere_hit.argtypes = [P, P, P, R, R, P]
        L.oracle_triangle_hit.argtypes = [P, P, P, R, R, P]
        L.oracle_world_hit.argtypes = [P, P, P, R, R, R, P, P]
        L.oracle_scatter.argtypes = [P, C.c_int32, P, P, P, C.c_uint64, C.c_uint32, C.c_uint32, P]
        L.oracle_texture_value.argtypes = [P, C.c_int32, R, R, P, P]
        L.oracle_sky.argtypes = [P, P, P]
        L.oracle_camera_ray.argtypes = [C.POINTER(A.CrCameraDesc), C.POINTER(A.CrRenderParams), C.c_uint32, C.c_uint32,
                                        C.c_uint32, P]
        L.oracle_rng_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, P]
        L.oracle_rng_u64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, P]
        L.oracle_bvh_dump.argtypes = [P, P, P, C.c_int32]
        L.oracle_set_tree.argtypes = [P, P, P, P, C.c_int32]
        L.oracle_set_tree.restype = C.c_int32
        L.oracle_use_list.argtypes = [P]
        L.oracle_use_list.restype = None
        L.oracle_set_faithful.argtypes = [C.c_int32]
        L.oracle_set_faithful.restype = None
        L.oracle_set_libm.argtypes = [C.c_int32]
        L.oracle_set_libm.restype = None
        L.oracle_trig.argtypes = [P, C.c_int32, P, P, P]
        L.oracle_trig.restype = None

    # ---- helpers
    def arr(self, x):
        return np.ascontiguousarray(x, dtype=self.np_real)

    def _p(self, a):
        return a.ctypes.data_as(C.c_void_p)

    def vec_fn(self, name, *ins, n_out=3):
        ins = [self.arr(i) for i in ins]
        out = np.zeros(n_out, dtype=self.np_real)
        getattr(self.lib, name)(*[self._p(i) for i in ins], self._p(out))
        return out

    def set_faithful(self, on):
        """Evaluate timelines, update_bb, material clones and the pixel hand-out the way the reference does at every
        hit (bench.py's `faithful` CPU baseline); results are unchanged, only slower."""
        self.lib.oracle_set_faithful(1 if on else 0)

    # ---- atan2 / asin / acos: the build's defined functions (default) or glibc's (what the Rust binary would call here)
    def set_libm(self, use_glibc):
        self.lib.oracle_set_libm(1 if use_glibc else 0)

    def trig(self, y, x):
        """(atan2(y, x), asin(y), acos(y)) elementwise with the currently selected implementation."""
        inp = self.arr(np.stack([np.asarray(y), np.asarray(x)], axis=1))
        outs = [np.zeros(len(inp), dtype=self.np_real) for _ in range(3)]
        self.lib.oracle_trig(self._p(inp), len(inp), *[self._p(o) for o in outs])
        return outs

    # ---- scene
    def scene_create(self, flat):
        h = self.lib.oracle_scene_create(C.byref(flat.desc))
        assert h
        return h

    def scene_destroy(self, h):
        self.lib.oracle_scene_destroy(h)

    def set_tree(self, scene_h, boxes, kids, axis=None):
        """Walk this wrapper tree (cr_export_bvh's output) instead of the reference-built one."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64)
        kids = np.ascontiguousarray(kids, dtype=np.int32)
        axis = None if axis is None else np.ascontiguousarray(axis, dtype=np.int32)
        rc = self.lib.oracle_set_tree(scene_h, boxes.ctypes.data, kids.ctypes.data,
                                      None if axis is None else axis.ctypes.data, len(kids))
        assert rc == 0, "malformed wrapper tree"

    def render(self, scene_h, cam, *, seed, sample_begin=0, sample_count=None, output_sum=False, pix_begin=0,
               pix_end=None, n_threads=None):
        """Returns (H*W*3 array reshaped (n_pix,3) of reals, stats dict)."""
        cd = cam.desc()
        p = cam.params(seed, self.real_type, sample_begin, sample_count, output_sum)
        n_pix_total = cam.image_width * cam.image_height
        pix_end = n_pix_total if pix_end is None else pix_end
        out = np.zeros((pix_end - pix_begin, 3), dtype=self.np_real)
        st = A.CrStats()
        if n_threads is None:
            n_threads = os.cpu_count() or 1
        rc = self.lib.oracle_render(scene_h, C.byref(cd), C.byref(p), pix_begin, pix_end, self._p(out), n_threads,
                                    C.byref(st))
        assert rc == 0, rc
        return out, st.as_dict()

    def render_image(self, scene, *, seed, n_threads=None, tree=None, linear_list=False, **kw):
        """tree: walk this exported wrapper tree; linear_list: no BVH at all (HitList::hit over the visible
        primitives) -- the ground truth for closest hits."""
        flat = scene.flatten()
        h = self.scene_create(flat)
        try:
            if tree is not None:
                self.set_tree(h, *tree)
            if linear_list:
                self.lib.oracle_use_list(h)
            out, st = self.render(h, scene.scene_cam, seed=seed, n_threads=n_threads, **kw)
        finally:
            self.scene_destroy(h)
        cam = scene.scene_cam
        return out.reshape(cam.image_height, cam.image_width, 3), st
