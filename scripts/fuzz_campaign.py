"""Seeded random scenes beyond the committed ones (tests/test_gpu_fuzz.py's generator), HIP library vs oracle: images and
counters must be equal.  usage: fuzz_campaign.py FIRST_SEED COUNT [lists]   -- prints one line per mismatch and a summary.
Test infrastructure (it imports the oracle); not part of the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from crucible_amd import _abi as A
from crucible_amd.renderer import Renderer
from oracle.oracle import Oracle
from test_gpu_fuzz import random_scene, COUNTERS

first, count = int(sys.argv[1]), int(sys.argv[2])
lists = "lists" in sys.argv[3:]
hostile = "hostile" in sys.argv[3:]


def hostile_scene(seed):
    """Degenerate inputs on purpose: axis-aligned camera rays (zero direction components: Aabb::hit's compare/select
    form), coincident and zero-radius spheres (ties, empty boxes), axis-flat and zero-area triangles, huge and tiny
    coordinates, scatter_prob 0 / negative / > 1 (division by zero, complements: the NaN policy), fuzz 1, ior 1,
    deep checker chains, 1x1 images, depth 0."""
    from crucible_amd.scene import (LERP, LOCAL, NERP, CheckerTexture, Dielectric, HitList, Lambertian, Metal, Scene, SolidColor,
                                    Sphere, Triangle)
    rs = np.random.RandomState(seed)
    u = rs.uniform
    width = int(rs.choice([1, 2, 9, 33]))
    sc = Scene.new_image(float(rs.choice([1.0, 16.0 / 9.0, 0.5])), width, 24.0, float(rs.choice([0.0, 180.0, 360.0])), 1)
    cam = sc.scene_cam
    cam.set_samples(int(rs.randint(1, 4)))
    cam.set_max_depth(int(rs.choice([0, 1, 2, 6, 50])))
    axis_aligned = rs.rand() < 0.5
    scale = float(rs.choice([1.0, 1.0, 1e-6, 1e6]))
    if axis_aligned:
        cam.look_from((0.0, 0.0, 5.0 * scale))
        cam.look_at((0.0, 0.0, 0.0))
        cam.set_vfov(float(rs.choice([1e-9, 1.0, 40.0])))
    else:
        cam.look_from(tuple(u(-6, 6, 3) * scale))
        cam.look_at(tuple(u(-1, 1, 3) * scale))
        cam.set_vfov(u(5, 120))
    cam.set_defocus_angle(float(rs.choice([0.0, 0.0, 1.0])))
    cam.set_focus_dist(5.0 * scale)

    def texture(depth):
        if depth == 0 or rs.rand() < 0.3:
            return SolidColor(tuple(rs.choice([0.0, 1.0, 0.5], 3)))
        return CheckerTexture.new_from_textures(float(rs.choice([1e-9, 0.3, 1e9])), texture(depth - 1), texture(depth - 1) if rs.rand() < 0.3 else SolidColor((0.2, 0.8, 0.1)))

    def material():
        k = rs.randint(0, 4)
        if k == 0:
            return Lambertian.new_from_texture(texture(int(rs.choice([0, 1, 3, 12]))), float(rs.choice([1.0, 0.5, 0.0, -0.5, 2.0, 1e-300])))
        if k == 1:
            return Metal.new(tuple(rs.choice([0.0, 1.0, 0.7], 3)), float(rs.choice([0.0, 1.0, 0.5])))
        if k == 2:
            return Dielectric.new(float(rs.choice([1.0, 1.5, 0.0, 1e-9, 1e9, -1.5])))
        return Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)

    elems = []
    centre = tuple(u(-1, 1, 3) * scale)
    for k in range(int(rs.randint(0, 9))):
        kind = rs.randint(0, 7)
        if kind == 0:
            elems.append(Sphere.new(centre, float(rs.choice([0.0, 0.5, 1.0])) * scale, material()))        # coincident / zero radius
        elif kind == 1:
            elems.append(Sphere.new(tuple(u(-2, 2, 3) * scale), u(0.1, 1.5) * scale, material()))
        elif kind == 2:
            z = float(rs.choice([0.0, 1.0])) * scale
            elems.append(Triangle.new((-scale, -scale, z), (scale, -scale, z), (0.0, scale, z), material()))  # axis-flat
        elif kind == 3:
            p = tuple(u(-1, 1, 3) * scale)
            elems.append(Triangle.new(p, p, tuple(u(-1, 1, 3) * scale), material()))                         # zero area
        elif kind == 4:
            elems.append(Triangle.new(*(tuple(u(-2, 2, 3) * scale) for _ in range(3)), material()))
        elif kind == 5:
            elems.append(Sphere.new((0.0, -1000.0 * scale, 0.0), 1000.0 * scale, material()))
        else:
            elems.append(Sphere.new(tuple(u(-2, 2, 3) * scale), 0.7 * scale, material()))
            if rs.rand() < 0.5:
                elems[-1].timeline.translate_point(tuple(u(-1, 1, 3) * scale), float(rs.choice([0.0, 1e-9, 0.01])), LERP if rs.rand() < 0.5 else NERP, LOCAL)
            else:
                elems[-1].timeline.scale_sphere(float(rs.choice([0.0, 2.0])) * scale, float(rs.choice([0.0, 0.01])), LERP if rs.rand() < 0.5 else NERP)
    if lists and elems:
        cut = len(elems) // 2
        l = HitList.new(elems[:cut]) if rs.rand() < 0.5 else HitList.default()
        if not l.objs:
            for e in elems[:cut]:
                l.add(e)
        sc.add_element(l, "l")
        elems = elems[cut:]
    for k, e in enumerate(elems):
        if rs.rand() < 0.1:
            e.hide = True
        sc.add_element(e, f"e{k}")
        if e.hide:
            sc.hide_element(f"e{k}")
    return sc

oracles = {A.CR_REAL_F64: Oracle(A.CR_REAL_F64), A.CR_REAL_F32: Oracle(A.CR_REAL_F32)}
r = Renderer(0)
bad, nan_scenes, t0 = 0, 0, time.time()
for seed in range(first, first + count):
    try:
        sc = hostile_scene(seed) if hostile else random_scene(seed, lists=lists)
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
                img, st = r.render(sc.scene_cam, seed=seed, real_type=rt)
                gpu_nan = False
            except Exception as e:
                if getattr(e, "code", None) != A.CR_ERR_NAN:
                    raise
                gpu_nan = True
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
    if (seed - first) % 100 == 99:
        print(f"... {seed - first + 1} scenes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {count} scenes x 2 precisions from seed {first}{' with lists' if lists else ''}{' hostile' if hostile else ''}: {bad} mismatches, {nan_scenes} NaN-policy renders, {time.time() - t0:.0f} s")
r.close()
