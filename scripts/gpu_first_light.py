"""First-light check on the GPU box: HIP path vs oracle on a small book1 render."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene
from crucible_amd.renderer import Renderer
from oracle.oracle import Oracle

w = int(sys.argv[1]) if len(sys.argv) > 1 else 128
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = book1_end_scene(1, scene_seed=1, image_width=w, samples=spp)
flat = sc.flatten()
r = Renderer(0)
r.upload_scene(flat)
for rt, name in ((A.CR_REAL_F32, "f32"), (A.CR_REAL_F64, "f64")):
    t = time.time(); img, st = r.render(sc.scene_cam, seed=0xC0FFEE, real_type=rt); dt = time.time() - t
    o = Oracle(rt)
    ref, ost = o.render_image(sc, seed=0xC0FFEE)
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
    print(name, "gpu wall %.3fs kernel %.3f ms" % (dt, st["kernel_ms"]), "max|d|=%g" % d.max(), "n_diff=%d" % (d > 0).sum(),
          "bit_equal=%s" % np.array_equal(img, ref))
    print("  gpu stats", {k: st[k] for k in ("samples", "segments", "node_tests", "prim_tests", "bvh_entries", "scene_in_lds")})
    print("  ora stats", {k: ost[k] for k in ("samples", "segments", "node_tests", "prim_tests", "bvh_entries")})
    print("  Msamples/s (kernel): %.1f" % (st["samples"] / st["kernel_ms"] / 1e3))
