"""Quick throughput probe: book1 at a given size/spp, f32 and optionally f64."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene
from crucible_amd.renderer import Renderer

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
modes = sys.argv[3] if len(sys.argv) > 3 else "f32"
sc = book1_end_scene(1, scene_seed=1, image_width=w, samples=spp)
sc.bvh_mode = {"sah": A.CR_BVH_SAH, "ordered": A.CR_BVH_SAH_ORDERED, "lbvh": A.CR_BVH_LBVH}.get(os.environ.get("BVH"), A.CR_BVH_REFERENCE)
r = Renderer(0)
r.upload_scene(sc.flatten())
for name in modes.split(","):
    rt = A.CR_REAL_F32 if name == "f32" else A.CR_REAL_F64
    for rep in range(2):
        img, st = r.render(sc.scene_cam, seed=0xC0FFEE, real_type=rt)
    S, N, P = st["segments"], st["node_tests"], st["prim_tests"]
    esz = 32 if name == "f32" else 64
    B = S * 96 + N * esz + P * 16 + img.shape[0] * img.shape[1] * 12
    print(name, "%dx%d@%d kernel %.2f ms  %.1f Msamples/s  seg/sample %.2f node/seg %.1f prim/seg %.2f  alg %.1f GB/s lds=%d mean=%s" % (
        img.shape[1], img.shape[0], spp, st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3, S / st["samples"], N / S, P / S,
        B / st["kernel_ms"] / 1e6, st["scene_in_lds"], img.mean(axis=(0, 1))))
