"""Cost of refit_boxes on a large tree: the 1,000,001-sphere scene with one keyframed sphere (so the scene counts as
animated and every wrapper is refitted), tiny image so the path trace itself is negligible."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from crucible_amd import _abi as A
from crucible_amd.demo_builder import million_spheres
from crucible_amd.renderer import Renderer

for mode, name in ((A.CR_BVH_REFERENCE, "reference"), (A.CR_BVH_SAH, "sah")):
    sc = million_spheres(1, scene_seed=1, image_width=16, samples=1)
    sc.bvh_mode = mode
    flat = sc.flatten()
    keys = (A.CrKeyframe * 1)(A.CrKeyframe(A.CR_KEY_TX, A.CR_KEY_LERP, 0.0, 1.0, 0.5, 0.0))
    flat.keys = keys
    flat.desc.keys = keys
    flat.desc.n_keys = 1
    flat.prims[5].key_first, flat.prims[5].key_count = 0, 1
    r = Renderer(0)
    t0 = time.perf_counter(); r.upload_scene(flat); 
    for rt, tag in ((A.CR_REAL_F32, "f32"), (A.CR_REAL_F64, "f64")):
        res = {}
        for refit in (False, True, False, True, True):
            sc.scene_cam.refit_boxes = refit
            t0 = time.perf_counter()
            img, st = r.render(sc.scene_cam, seed=1, real_type=rt)
            res.setdefault(refit, []).append((time.perf_counter() - t0) * 1e3)
        print(name, tag, "entries", st["bvh_entries"], "render wall ms without refit %.2f, with refit %.2f" % (min(res[False][1:] or res[False]), min(res[True][1:])))
    r.close()
