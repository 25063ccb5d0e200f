import sys, time, os
sys.path.insert(0, os.getcwd())
from crucible_amd.demo_builder import million_spheres
from crucible_amd.renderer import Renderer
from crucible_amd import _abi as A
t=time.time(); sc = million_spheres(1, scene_seed=1, image_width=64, samples=1); print("gen %.2fs" % (time.time()-t))
sc.bvh_mode = {"sah": A.CR_BVH_SAH, "ordered": A.CR_BVH_SAH_ORDERED, "lbvh": A.CR_BVH_LBVH}.get(os.environ.get("BVH"), A.CR_BVH_REFERENCE)
r = Renderer(0)
t=time.time(); r.upload_scene(sc.flatten()); print("upload %.2fs" % (time.time()-t))
for rt in (A.CR_REAL_F32, A.CR_REAL_F64):
    t=time.time(); img, st = r.render(sc.scene_cam, seed=1, real_type=rt); print("first render (incl. BVH build) %.2fs upload_ms=%.0f kernel_ms=%.2f entries=%d" % (time.time()-t, st["upload_ms"], st["kernel_ms"], st["bvh_entries"]))
    t=time.time(); img, st = r.render(sc.scene_cam, seed=1, real_type=rt); print("second render %.3fs" % (time.time()-t))
