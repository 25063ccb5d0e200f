"""A/B tool of the experiments under profiles/experiments/: renders one workload five times in this process and prints the
Msamples/s of each render, the frame's md5 and the walk statistics.  LIB=/path/to/libcrucible_hip.so picks another build of the
library (the knobs are read at cr_create), BVH=sah|ordered|lbvh another tree.
usage: ab_render.py book1|teapot|movie|million f64|f32 WIDTH SPP"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import crucible_amd.renderer as R
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene, load_teapot, procedural_sky, teapot_orbit_movie, million_spheres
if os.environ.get("LIB"):
    R.LIB_PATH = os.environ["LIB"]
workload, real, w, spp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
sc = million_spheres(1, scene_seed=1, image_width=w, samples=spp) if workload == "million" else book1_end_scene(1, scene_seed=1, image_width=w, samples=spp) if workload == "book1" else (teapot_orbit_movie(1, image_width=w, samples=spp) if workload == "movie" else load_teapot(1, image_width=w, samples=spp, sky=procedural_sky()))
sc.bvh_mode = {'sah': A.CR_BVH_SAH, 'ordered': A.CR_BVH_SAH_ORDERED, 'lbvh': A.CR_BVH_LBVH}.get(os.environ.get('BVH'), A.CR_BVH_REFERENCE)
flat = sc.flatten()
rt = A.CR_REAL_F32 if real == "f32" else A.CR_REAL_F64
cam = sc.scene_cam
out = torch.empty((cam.image_height, cam.image_width, 3), dtype=torch.float32 if real == "f32" else torch.float64, device="cuda")
r = R.Renderer(0)
r.upload_scene(flat)
ts = []
for rep in range(5):
    r.render_device(cam, out.data_ptr(), seed=0xC0FFEE, real_type=rt)
    ts.append(r.last_kernel_ms())
n = cam.image_width * cam.image_height * spp
st = r.render_device(cam, out.data_ptr(), seed=0xC0FFEE, real_type=rt, want_stats=True)
import hashlib
h = hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:10]
print("wide" if os.environ.get("CRUCIBLE_WIDE") else "", os.environ.get("BVH",""), "img", h, "mean %.6f" % float(out.mean()), "node/seg %.1f prim/seg %.2f" % (st["node_tests"] / st["segments"], st["prim_tests"] / st["segments"]), end=" ")
print(os.environ.get("LIB", "new"), workload, real, " ".join(f"{n / t / 1e3:.0f}" for t in ts), flush=True)
