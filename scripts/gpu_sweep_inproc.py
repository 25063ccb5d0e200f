"""Sweep of the megakernel's wave-scheduling knobs in ONE process (the knobs are read at cr_create).
usage: gpu_sweep_inproc.py WORKLOAD REAL WIDTH SPP "round,exit;round,exit;..." [ENV=VAL ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene, load_teapot, million_spheres, procedural_sky
from crucible_amd.renderer import Renderer

workload, real, w, spp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
combos = [tuple(c.split(",")) for c in sys.argv[5].split(";")]
for kv in sys.argv[6:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
if workload == "book1":
    sc = book1_end_scene(1, scene_seed=1, image_width=w, samples=spp)
elif workload == "teapot":
    sc = load_teapot(1, image_width=w, samples=spp, sky=procedural_sky())
else:
    sc = million_spheres(1, scene_seed=1, image_width=w, samples=spp)
sc.bvh_mode = {"sah": A.CR_BVH_SAH, "ordered": A.CR_BVH_SAH_ORDERED, "lbvh": A.CR_BVH_LBVH}.get(os.environ.get("BVH"), A.CR_BVH_REFERENCE)
flat = sc.flatten()
rt = A.CR_REAL_F32 if real == "f32" else A.CR_REAL_F64
cam = sc.scene_cam
out = torch.empty((cam.image_height, cam.image_width, 3), dtype=torch.float32 if real == "f32" else torch.float64, device="cuda")
for rnd, ex in combos:
    os.environ["CRUCIBLE_WALK_ROUND"] = rnd
    os.environ["CRUCIBLE_WALK_EXIT"] = ex
    r = Renderer(0)
    r.upload_scene(flat)
    best = 1e30
    for rep in range(3):
        r.render_device(cam, out.data_ptr(), seed=0xC0FFEE, real_type=rt)
        best = min(best, r.last_kernel_ms())
    n = cam.image_width * cam.image_height * spp
    st = r.render_device(cam, out.data_ptr(), seed=0xC0FFEE, real_type=rt, want_stats=True)   # counters (and [diag] lines of a diagnostic build)
    print(f"{workload} {real} round={rnd} exit={ex}: {best:.2f} ms  {n / best / 1e3:.1f} Msamples/s  "
          f"seg/sample {st['segments'] / st['samples']:.3f} node/seg {st['node_tests'] / st['segments']:.2f} prim/seg {st['prim_tests'] / st['segments']:.3f}", flush=True)
    r.close()
