"""Seeded random scenes beyond the committed ones (tests/test_gpu_fuzz.py's generator), HIP library vs oracle: images and
counters must be equal in the parity mode (CR_SUM_REFERENCE_ORDER); the library default (CR_SUM_RELAXED) is rendered too and must
have the same counters, the same NaN verdict and a frame within 1e-12 (f64) / n_samples * 2^-22 (f32).  usage: fuzz_campaign.py FIRST_SEED COUNT [lists] [wrappers] [hostile|big|camera]   -- prints one line per mismatch and a summary.
Test infrastructure (it imports the oracle); not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from crucible_amd import _abi as A
from crucible_amd.renderer import Renderer
from oracle.oracle import Oracle
from test_gpu_fuzz import random_scene, hostile_scene, big_scene, COUNTERS

first, count = int(sys.argv[1]), int(sys.argv[2])
lists = "lists" in sys.argv[3:]
hostile = "hostile" in sys.argv[3:]
big = "big" in sys.argv[3:]
wrappers = "wrappers" in sys.argv[3:]
camera = "camera" in sys.argv[3:]


oracles = {A.CR_REAL_F64: Oracle(A.CR_REAL_F64), A.CR_REAL_F32: Oracle(A.CR_REAL_F32)}
r = Renderer(0)
bad, nan_scenes, t0 = 0, 0, time.time()
for seed in range(first, first + count):
    try:
        sc = hostile_scene(seed, lists=lists, wrappers=wrappers, degenerate_camera=True) if camera else (big_scene if big else hostile_scene if hostile else random_scene)(seed, lists=lists, wrappers=wrappers)
    except ValueError as e:   # the mirror's own argument checks (negative radius, fuzz > 1, ...)
        continue
    variant = seed % 3
    sc.scene_cam.refit_boxes = variant == 1
    if variant == 2:
        sc.bvh_mode = [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH][(seed // 3) % 3]
    for rt in (A.CR_REAL_F64, A.CR_REAL_F32):
        try:
            r.upload_scene(sc.flatten())
            try:
                img, st = r.render(sc.scene_cam, seed=seed, real_type=rt, sum_order=A.CR_SUM_REFERENCE_ORDER)
                gpu_nan = False
            except Exception as e:
                if getattr(e, "code", None) != A.CR_ERR_NAN:
                    raise
                gpu_nan = True
            try:
                fast, fst = r.render(sc.scene_cam, seed=seed, real_type=rt, sum_order=A.CR_SUM_RELAXED)
                fast_nan = False
            except Exception as e:
                if getattr(e, "code", None) != A.CR_ERR_NAN:
                    raise
                fast_nan = True
            if fast_nan != gpu_nan:
                bad += 1; print(f"MISMATCH seed {seed} rt {rt}: NaN verdict relaxed={fast_nan} reference order={gpu_nan}", flush=True)
            elif not gpu_nan:
                tol = 1e-12 if rt == A.CR_REAL_F64 else sc.scene_cam.samples * 2.0 ** -22
                d = float(np.abs(fast.astype(np.float64) - img.astype(np.float64)).max()) if img.size else 0.0
                if not d <= tol:
                    bad += 1; print(f"MISMATCH seed {seed} rt {rt} variant {variant}: relaxed frame off by {d}", flush=True)
                for k in COUNTERS:
                    if fst[k] != st[k]:
                        bad += 1; print(f"MISMATCH seed {seed} rt {rt} variant {variant}: relaxed counter {k} {fst[k]} vs {st[k]}", flush=True)
            tree = r.export_bvh(rt) if variant == 2 else None
            empty = tree is not None and len(tree[1]) == 0   # no visible primitive: the opt-in trees have no wrapper at all
            if empty:
                tree = None
            ref, rst = oracles[rt].render_image(sc, seed=seed, tree=tree, linear_list=empty if variant == 2 else False)
            if gpu_nan or rst["nan_pixels"]:
                nan_scenes += 1
                if gpu_nan != (rst["nan_pixels"] > 0):
                    bad += 1; print(f"MISMATCH seed {seed} rt {rt}: NaN policy gpu={gpu_nan} oracle={rst['nan_pixels']}", flush=True)
                continue
            if not np.array_equal(img, ref):
                bad += 1; print(f"MISMATCH seed {seed} rt {rt} variant {variant}: {(img != ref).any(axis=2).sum()} pixels differ", flush=True)
            for k in COUNTERS:
                if st[k] != rst[k]:
                    bad += 1; print(f"MISMATCH seed {seed} rt {rt} variant {variant}: counter {k} {st[k]} vs {rst[k]}", flush=True)
        except Exception as e:
            bad += 1; print(f"ERROR seed {seed} rt {rt}: {type(e).__name__}: {e}", flush=True)
    if (seed - first) % (10 if big else 100) == (9 if big else 99):
        print(f"... {seed - first + 1} scenes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {count} scenes x 2 precisions from seed {first}{' with lists' if lists else ''}{' with wrappers' if wrappers else ''}{' hostile' if hostile else ''}{' big' if big else ''}{' degenerate cameras' if camera else ''}: {bad} mismatches, {nan_scenes} NaN-policy renders, {time.time() - t0:.0f} s")
r.close()
