"""Seeded random scenes through the whole path: every feature the ABI carries, mixed at random -- spheres and
triangles, all three materials, solid / nested checker / image textures, default or spherical sky, hidden primitives,
translate / radius / ScaleX-Y-Z keys (LERP and NERP) on primitives and camera, defocus on or off, odd image sizes,
shallow and deep paths, HitList elements (seeds from 100) -- rendered by the HIP library and by the oracle: bit-equal images and equal work counters, f64
and f32, with the reference's tree and (every third scene) with refit or an exported opt-in tree."""
import numpy as np
import pytest

from crucible_amd import _abi as A
from crucible_amd.renderer import CrucibleError
from crucible_amd.scene import (LERP, LOCAL, NERP, WORLD, BVHWrapper, CheckerTexture, Dielectric, HitList, ImageTexture, Lambertian, Metal,
                                RTWImage, Scene, SolidColor, Sphere, Triangle)

pytestmark = pytest.mark.gpu

COUNTERS = ("segments", "node_tests", "prim_tests", "texel_fetches")
REALS = [(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")]


def random_scene(seed, lists=False, wrappers=False):
    rs = np.random.RandomState(seed)
    u = rs.uniform
    width = int(rs.choice([17, 32, 45, 64, 73]))
    sc = Scene.new_image(float(rs.choice([16.0 / 9.0, 1.0, 2.35])), width, float(rs.choice([1, 24, 30])), float(rs.choice([90, 180, 360])), 1)
    cam = sc.scene_cam
    cam.set_samples(int(rs.randint(1, 6)))
    cam.set_max_depth(int(rs.choice([1, 3, 8, 50])))
    cam.look_from((u(-9, 9), u(0.5, 6), u(4, 10)))
    cam.look_at((u(-1, 1), u(0, 1.5), u(-1, 1)))
    cam.set_vfov(u(20, 70))
    cam.set_defocus_angle(float(rs.choice([0.0, 0.4, 2.0])))
    cam.set_focus_dist(u(4, 12))
    cam.frame = int(rs.randint(0, 3))
    img = RTWImage(rs.randint(0, 256, size=(int(rs.randint(2, 12)), int(rs.randint(2, 20)), 3)).astype(np.uint8))

    def colour():
        return tuple(u(0, 1, 3))

    def texture(depth=0):
        k = rs.randint(0, 4 if depth < 3 else 2)
        if k == 0:
            return SolidColor(colour())
        if k == 1:
            return ImageTexture(img)
        return CheckerTexture.new_from_textures(u(0.2, 2.0), texture(depth + 1), texture(depth + 1))

    def material():
        k = rs.randint(0, 5)
        if k <= 1:
            return Lambertian.new_from_texture(texture(), float(rs.choice([1.0, 1.0, 0.7, 0.35])))
        if k == 2:
            return Lambertian.new_from_color(colour(), 1.0)
        if k == 3:
            return Metal.new(colour(), float(rs.choice([0.0, 0.1, 0.6, 1.0])))
        return Dielectric.new(float(rs.choice([1.5, 1.0 / 1.5, 2.4])))

    sc.add_element(Sphere.new((0.0, -200.0, 0.0), 200.0, material()), "ground")
    n = int(rs.randint(0, 14))
    names = []
    for k in range(n):
        if rs.rand() < 0.55:
            sc.add_element(Sphere.new((u(-4, 4), u(0.2, 2.0), u(-4, 3)), u(0.15, 1.1), material()), f"s{k}")
            names.append((f"s{k}", "sphere"))
        else:
            c = np.array([u(-4, 4), u(0.0, 2.0), u(-4, 3)])
            a, b, d = (tuple(c + u(-1.2, 1.2, 3)) for _ in range(3))
            sc.add_element(Triangle.new(a, b, d, material()), f"t{k}")
            names.append((f"t{k}", "triangle"))
    for alias, kind in names:
        r = rs.rand()
        if r < 0.2:
            sc.hide_element(alias)
        elif r < 0.6:
            for _ in range(int(rs.randint(1, 4))):
                key, interp = float(rs.choice([0.02, 0.5, 1.0, 1.7, 2.5])), (LERP if rs.rand() < 0.6 else NERP)
                what = rs.randint(0, 3)
                try:
                    if what == 0:
                        sc.translate_point(tuple(u(-1.5, 1.5, 3)), key, interp, LOCAL if rs.rand() < 0.5 else WORLD, alias)
                    elif kind == "sphere":
                        sc.scale_r(u(0.1, 1.4), key, interp, alias)
                    else:
                        which = rs.randint(0, 5)
                        if which == 0:
                            sc.scale_x(u(-0.5, 2.0), key, interp, alias)
                        elif which == 1:
                            sc.scale_y(u(-0.5, 1.0), key, interp, alias)
                        elif which == 2:
                            sc.scale_z(u(0.2, 2.0), key, interp, alias)
                        elif which == 3:
                            sc.scale_point(tuple(u(0.3, 1.8, 3)), key, interp, alias)
                        else:
                            sc.scale_all_uniform(u(0.3, 1.8), key, interp, alias)
                except ValueError:
                    pass   # "Missing transform data": a key earlier than an existing one of its kind, as in the reference
    if lists:   # HitList elements (tests/test_gpu_lists.py): grown by add(), from new(vec), nested; hidden and keyed objects
        for li in range(int(rs.randint(1, 5))):
            objs = []
            for k in range(int(rs.randint(0, 7))):
                if rs.rand() < 0.6:
                    o = Sphere.new((u(-4, 4), u(0.2, 2.0), u(-4, 3)), u(0.15, 0.9), material())
                else:
                    c = np.array([u(-4, 4), u(0.0, 2.0), u(-4, 3)])
                    o = Triangle.new(*(tuple(c + u(-1.2, 1.2, 3)) for _ in range(3)), material())
                o.hide = rs.rand() < 0.15
                if rs.rand() < 0.4:   # objects of a list have no alias: their timelines are filled directly
                    key, interp = float(rs.choice([0.02, 0.5, 1.0, 1.7])), (LERP if rs.rand() < 0.6 else NERP)
                    if rs.rand() < 0.5:
                        o.timeline.translate_point(tuple(u(-1.5, 1.5, 3)), key, interp, LOCAL)
                    elif isinstance(o, Sphere):
                        o.timeline.scale_sphere(u(0.1, 1.2), key, interp)
                    else:
                        o.timeline.scale_point(tuple(u(0.3, 1.8, 3)), key, interp)
                objs.append(o)
            how = rs.randint(0, 3)
            if how == 0:
                l = HitList.new(objs)
            else:
                l = HitList.default()
                cut = len(objs) // 2 if how == 2 else len(objs)
                for o in objs[:cut]:
                    l.add(o)
                if how == 2:
                    inner = HitList.default()
                    for o in objs[cut:]:
                        inner.add(o)
                    l.add(inner)
            sc.add_element(l, f"list{li}")
    if wrappers:   # BVHWrapper elements (tests/test_gpu_lists.py): wrappers of 0..8 objects, some hidden (dropped), some keyed
        for wi in range(int(rs.randint(1, 4))):
            objs = []
            for k in range(int(rs.randint(0, 9))):
                if rs.rand() < 0.6:
                    o = Sphere.new((u(-4, 4), u(0.2, 2.0), u(-4, 3)), u(0.15, 0.9), material())
                else:
                    c = np.array([u(-4, 4), u(0.0, 2.0), u(-4, 3)])
                    o = Triangle.new(*(tuple(c + u(-1.2, 1.2, 3)) for _ in range(3)), material())
                o.hide = rs.rand() < 0.15
                if rs.rand() < 0.3:
                    key, interp = float(rs.choice([0.02, 0.5, 1.0, 1.7])), (LERP if rs.rand() < 0.6 else NERP)
                    o.timeline.translate_point(tuple(u(-1.5, 1.5, 3)), key, interp, LOCAL)
                objs.append(o)
            sc.add_element(BVHWrapper.new_wrapper(HitList.new(objs)), f"wrap{wi}")
    if rs.rand() < 0.5:
        sc.load_spherical_skybox(RTWImage(rs.randint(40, 256, size=(8, 16, 3)).astype(np.uint8)))
    if rs.rand() < 0.4:
        sc.cam_translate_point(tuple(u(-9, 9, 3) * np.array([1, 0.3, 1]) + np.array([0, 3, 6])), float(rs.choice([0.03, 1.0, 2.0])), LERP, WORLD, "from")
    if rs.rand() < 0.25:
        sc.cam_translate_point((u(-1, 1), u(0, 1), u(-1, 1)), float(rs.choice([0.02, 1.5])), NERP, WORLD, "at")
    return sc


def hostile_scene(seed, lists=False, wrappers=False, degenerate_camera=False):
    """Degenerate inputs on purpose: axis-aligned camera rays (zero direction components: Aabb::hit's compare/select
    form), coincident and zero-radius spheres (ties, empty boxes), axis-flat and zero-area triangles, huge and tiny
    coordinates, scatter_prob 0 / negative / > 1 (division by zero, complements: the NaN policy), fuzz 1, ior 1,
    deep checker chains, 1x1 images, depth 0."""
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
    if degenerate_camera:   # no basis (look_from == look_at, or vup along the view), zero focus distance, half-turn field of view
        k = seed % 4
        if k == 0:
            cam.look_at(cam.look_from_tl.start_pos)
        elif k == 1:
            cam.look_from((0.0, 5.0 * scale, 0.0))
            cam.look_at((0.0, 0.0, 0.0))            # vup = +y is along the view direction
        elif k == 2:
            cam.set_focus_dist(0.0)
        else:
            cam.set_vfov(float(rs.choice([0.0, 180.0, 360.0])))

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
    if wrappers and elems:
        cut = max(1, len(elems) // 2)
        sc.add_element(BVHWrapper.new_wrapper(HitList.new(elems[:cut])), "w")
        elems = elems[cut:]
    for k, e in enumerate(elems):
        if rs.rand() < 0.1:
            e.hide = True
        sc.add_element(e, f"e{k}")
        if e.hide:
            sc.hide_element(f"e{k}")
    return sc


def big_scene(seed, lists=False, wrappers=False):
    """Hundreds to tens of thousands of primitives: trees that do not fit in LDS (the top-levels window, the all-global
    kernels, the f32 entry point for more than 65 536 wrappers), long list elements, keyed primitives among static ones."""
    rs = np.random.RandomState(seed)
    u = rs.uniform
    n = int(rs.choice([300, 1500, 6000, 40000]))
    sc = Scene.new_image(16.0 / 9.0, int(rs.choice([24, 40])), 24.0, float(rs.choice([180.0, 360.0])), 1)
    cam = sc.scene_cam
    cam.set_samples(int(rs.randint(1, 3)))
    cam.set_max_depth(int(rs.choice([2, 8, 50])))
    cam.look_from((u(-3, 3), u(1, 6), u(8, 14)))
    cam.look_at((0.0, 0.5, 0.0))
    cam.set_vfov(u(20, 50))
    cam.set_defocus_angle(float(rs.choice([0.0, 0.6])))
    cam.frame = int(rs.randint(0, 2))
    mats = [Lambertian.new_from_color(tuple(u(0.1, 0.9, 3)), 1.0), Metal.new(tuple(u(0.5, 1.0, 3)), 0.1), Dielectric.new(1.5),
            Lambertian.new_from_texture(CheckerTexture.new_from_color(0.3, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)]
    sc.add_element(Sphere.new((0.0, -1000.0, 0.0), 1000.0, mats[3]), "ground")
    span = 2.0 + n ** 0.5 * 0.25
    members = []
    keyed = []
    for k in range(n):
        m = mats[int(rs.randint(0, 3))]
        if rs.rand() < 0.7:
            o = Sphere.new((u(-span, span), u(0.05, 0.4), u(-span, span)), u(0.03, 0.2), m)
        else:
            c = np.array([u(-span, span), u(0.0, 0.6), u(-span, span)])
            o = Triangle.new(*(tuple(c + u(-0.3, 0.3, 3)) for _ in range(3)), m)
        if rs.rand() < 0.01:
            o.timeline.translate_point(tuple(u(-0.5, 0.5, 3)), float(rs.choice([0.01, 0.5, 1.5])), LERP if rs.rand() < 0.5 else NERP, LOCAL)
        if (lists or wrappers) and rs.rand() < 0.3:
            o.hide = rs.rand() < 0.05
            members.append(o)
        else:
            sc.add_element(o, f"p{k}")
            if rs.rand() < 0.02:
                sc.hide_element(f"p{k}")
    if wrappers and members:
        half = len(members) // 2
        sc.add_element(BVHWrapper.new_wrapper(HitList.new(members[:half])), "w0")
        members = members[half:]
    if lists:
        cuts = sorted(rs.randint(0, len(members) + 1, size=3))
        for li, (a, b) in enumerate(zip([0] + cuts, cuts + [len(members)])):
            if rs.rand() < 0.3:
                l = HitList.new(members[a:b])
            else:
                l = HitList.default()
                for o in members[a:b]:
                    l.add(o)
            sc.add_element(l, f"l{li}")
    return sc


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", list(range(36)) + list(range(100, 124)) + list(range(200, 224)))
def test_random_scene_bit_exact(renderer, oracles, rt, tag, seed):
    sc = random_scene(1000 + seed, lists=seed >= 100, wrappers=seed >= 200)   # seeds from 100: with HitList elements, from 200: and BVHWrapper elements
    variant = seed % 3
    if variant == 1:
        sc.scene_cam.refit_boxes = True
    if variant == 2:
        sc.bvh_mode = [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH][(seed // 3) % 3]
    renderer.upload_scene(sc.flatten())
    try:
        img, st = renderer.render(sc.scene_cam, seed=4000 + seed, real_type=rt)
        gpu_nan = False
    except Exception as e:   # CR_ERR_NAN: the reference would panic in Color::new; the oracle must see the same pixels
        assert getattr(e, "code", None) == A.CR_ERR_NAN, e
        gpu_nan = True
    tree = renderer.export_bvh(rt) if variant == 2 else None
    ref, rst = oracles[rt].render_image(sc, seed=4000 + seed, tree=tree)
    if gpu_nan:
        assert rst["nan_pixels"] > 0
        return
    assert rst["nan_pixels"] == 0
    assert np.array_equal(img, ref), f"seed {seed}: {(img != ref).any(axis=2).sum()} pixels differ"
    for k in COUNTERS:
        assert st[k] == rst[k], (seed, k, st[k], rst[k])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", [300052, 309589, 400541] + list(range(300000, 300045)) + list(range(1200000, 1200008)))
def test_hostile_scene_bit_exact(renderer, oracles, rt, tag, seed):
    """Degenerate inputs (hostile_scene): the same equalities.  300052 / 309589: zero-length radius keys at the frame time
    under refit_boxes (0/0 at the key's own start -- the refit rule skips that sample); 400541: an opt-in tree over a
    scene whose only element is a list without visible objects.  scripts/fuzz_campaign.py runs tens of thousands more."""
    try:
        sc = hostile_scene(seed, lists=seed >= 400000 or (seed < 300045 and seed % 2 == 1), wrappers=seed >= 1200000,
                           degenerate_camera=seed >= 1200000)   # from 1200000: no camera basis / zero focus distance / half-turn fov
    except ValueError:
        pytest.skip("the mirror's own argument checks reject this scene")
    variant = seed % 3
    sc.scene_cam.refit_boxes = variant == 1
    if variant == 2:
        sc.bvh_mode = [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH][(seed // 3) % 3]
    renderer.upload_scene(sc.flatten())
    try:
        img, st = renderer.render(sc.scene_cam, seed=seed, real_type=rt)
        gpu_nan = False
    except Exception as e:
        assert getattr(e, "code", None) == A.CR_ERR_NAN, e
        gpu_nan = True
    tree = renderer.export_bvh(rt) if variant == 2 else None
    empty = tree is not None and len(tree[1]) == 0
    ref, rst = oracles[rt].render_image(sc, seed=seed, tree=None if empty else tree, linear_list=empty)
    assert gpu_nan == (rst["nan_pixels"] > 0)
    if gpu_nan:
        return
    assert np.array_equal(img, ref), f"seed {seed}: {(img != ref).any(axis=2).sum()} pixels differ"
    for k in COUNTERS:
        assert st[k] == rst[k], (seed, k, st[k], rst[k])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(500000, 500012))
def test_big_scene_bit_exact(renderer, oracles, rt, tag, seed):
    """big_scene: trees outside LDS, with and without lists, refit and the opt-in trees by the same rota as above."""
    sc = big_scene(seed, lists=seed % 2 == 1)
    variant = seed % 3
    sc.scene_cam.refit_boxes = variant == 1
    if variant == 2:
        sc.bvh_mode = [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH][(seed // 3) % 3]
    renderer.upload_scene(sc.flatten())
    img, st = renderer.render(sc.scene_cam, seed=seed, real_type=rt)
    tree = renderer.export_bvh(rt) if variant == 2 else None
    ref, rst = oracles[rt].render_image(sc, seed=seed, tree=tree)
    assert np.array_equal(img, ref), f"seed {seed}: {(img != ref).any(axis=2).sum()} pixels differ"
    for k in COUNTERS:
        assert st[k] == rst[k], (seed, k, st[k], rst[k])


def _mutations(flat, rs):
    """(label, edit) pairs: every edit makes the descriptor invalid in a way cr_upload_scene must answer with a status
    code (include/crucible_hip.h: 'status codes instead of panics')."""
    d = flat.desc
    out = []
    nan, inf = float("nan"), float("inf")
    if d.n_prims:
        i = int(rs.randint(0, d.n_prims))
        p = flat.prims[i]
        if p.kind != A.CR_PRIM_LIST:
            out += [("prim kind", lambda: setattr(p, "kind", int(rs.choice([7, -1])))),
                    ("prim material", lambda: setattr(p, "material", int(rs.choice([d.n_materials, -1, 1 << 30])))),
                    ("prim key range", lambda: (setattr(p, "key_first", d.n_keys), setattr(p, "key_count", 1))),
                    ("prim key count", lambda: setattr(p, "key_count", -1)),
                    ("prim coordinate", lambda: p.v.__setitem__(int(rs.randint(0, 3)), float(rs.choice([nan, inf, -inf]))))]
            if p.kind == A.CR_PRIM_SPHERE:
                out.append(("negative radius", lambda: p.v.__setitem__(3, -0.5)))
    if d.n_materials:
        m = flat.materials[int(rs.randint(0, d.n_materials))]
        out += [("material kind", lambda: setattr(m, "kind", 9)), ("material param", lambda: setattr(m, "param", nan))]
        if m.kind == A.CR_MAT_LAMBERTIAN:
            out.append(("material texture", lambda: setattr(m, "texture", int(rs.choice([d.n_textures, -1])))))
        if m.kind == A.CR_MAT_METAL:
            out += [("metal fuzz", lambda: setattr(m, "param", float(rs.choice([1.5, -0.1])))), ("metal albedo", lambda: m.albedo.__setitem__(1, 1.5))]
    if d.n_textures:
        j = int(rs.randint(0, d.n_textures))
        x = flat.textures[j]
        out.append(("texture kind", lambda: setattr(x, "kind", 5)))
        if x.kind == A.CR_TEX_CHECKER:
            out += [("checker child", lambda: setattr(x, "even", int(rs.choice([j, d.n_textures, -1])))), ("checker child", lambda: setattr(x, "odd", j))]
        elif x.kind == A.CR_TEX_IMAGE:
            out.append(("texture image", lambda: setattr(x, "image", int(rs.choice([d.n_images, -1])))))
        else:
            out.append(("solid colour", lambda: x.color.__setitem__(0, float(rs.choice([-0.1, 1.1, nan])))))
    if d.n_keys:
        k = flat.keys[int(rs.randint(0, d.n_keys))]
        out += [("key channel", lambda: setattr(k, "channel", int(rs.choice([9, -1])))), ("key interp", lambda: setattr(k, "interp", 3))]
    if d.n_images:
        out.append(("image size", lambda: setattr(flat.images[0], "width", 0)))
    out += [("negative count", lambda: setattr(d, str(rs.choice(["n_prims", "n_materials", "n_textures", "n_images", "n_keys"])), -1)),
            ("sky kind", lambda: setattr(d, "sky_kind", 3)), ("bvh mode", lambda: setattr(d, "bvh_mode", 9)),
            ("sky image", lambda: (setattr(d, "sky_kind", A.CR_SKY_SPHERICAL), setattr(d, "sky_image", d.n_images)))]
    if d.n_prims:
        out.append(("null array", lambda: setattr(d, "prims", None)))
    if d.n_materials:
        out.append(("null array", lambda: setattr(d, "materials", None)))
    return out


@pytest.mark.parametrize("seed", range(700000, 700060))
def test_mutated_descriptors_are_refused(renderer, oracles, seed):
    """One invalid field at a time in an otherwise valid random descriptor: cr_upload_scene answers CR_ERR_INVALID_ARG (or
    _UNSUPPORTED), never crashes, and the handle keeps working -- the scene it held still renders the oracle's image."""
    rs = np.random.RandomState(seed)
    sc = random_scene(1000 + seed % 60, lists=seed % 2 == 1)
    good = sc.flatten()
    renderer.upload_scene(good)
    for _ in range(6):
        flat = sc.flatten()
        options = _mutations(flat, rs)
        label, edit = options[int(rs.randint(0, len(options)))]
        edit()
        with pytest.raises(CrucibleError) as e:
            renderer.upload_scene(flat)
        assert e.value.code in (A.CR_ERR_INVALID_ARG, A.CR_ERR_UNSUPPORTED), (label, e.value)
    try:
        img, _ = renderer.render(sc.scene_cam, seed=seed, real_type=A.CR_REAL_F32)
    except CrucibleError as e:
        assert e.code == A.CR_ERR_NAN
        return
    ref, rst = oracles[A.CR_REAL_F32].render_image(sc, seed=seed)
    assert np.array_equal(img, ref)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(800000, 800016))
def test_random_sample_shards_and_groups(renderer, oracles, monkeypatch, rt, tag, seed):
    """SURVEY 8(e) on random scenes: a random partition of the sample range into shards (empty ones included), each
    rendered as a per-pixel SUM and compared with the oracle's sum of the same samples bit for bit; then the library's
    own group over a random member count (members sharing this box's one device) against its shard sums added in member order."""
    from crucible_amd.group import RenderGroup, shard
    monkeypatch.setenv("CRUCIBLE_GROUP_SAME_DEVICE", "1")
    rs = np.random.RandomState(seed)
    sc = random_scene(1000 + seed % 97, lists=seed % 2 == 1)
    cam = sc.scene_cam
    spp = int(rs.randint(1, 10))
    cam.set_samples(spp)
    flat = sc.flatten()
    renderer.upload_scene(flat)
    cuts = sorted(int(c) for c in rs.randint(0, spp + 1, size=int(rs.randint(0, 4))))
    h = oracles[rt].scene_create(flat)
    try:
        for b, e in zip([0] + cuts, cuts + [spp]):
            try:
                part, st = renderer.render(cam, seed=seed, real_type=rt, sample_begin=b, sample_count=e - b, output_sum=True)
            except CrucibleError as err:
                assert err.code == A.CR_ERR_NAN
                return
            ref, rst = oracles[rt].render(h, cam, seed=seed, sample_begin=b, sample_count=e - b, output_sum=True)
            assert np.array_equal(part.reshape(-1, 3), ref), (seed, b, e)
            for k in COUNTERS:
                assert st[k] == rst[k], (seed, b, e, k)
    finally:
        oracles[rt].scene_destroy(h)
    members = int(rs.randint(1, 7))
    total = None
    for m in range(members):
        b, n = shard(spp, m, members)
        part, _ = renderer.render(cam, seed=seed, real_type=rt, sample_begin=b, sample_count=n, output_sum=True)
        total = part.copy() if total is None else total + part
    expect = total / total.dtype.type(spp)
    g = RenderGroup.local([0] * members)
    try:
        g.upload_scene(flat)
        try:
            img, gst = g.render(cam, seed=seed, real_type=rt)
        except CrucibleError as err:
            assert err.code == A.CR_ERR_NAN
            return
        assert gst["members"] == members
        assert np.array_equal(img, expect, equal_nan=True)
    finally:
        g.close()


def test_independent_handles_render_concurrently(oracles):
    """one handle = one caller thread at a time, independent handles may run concurrently.  Four host threads,
    each with its own handle on the one device, each rendering its own stream of random scenes (f64 and f32 alternating)
    while the others do the same: every frame equals the oracle's."""
    import threading
    from crucible_amd.renderer import Renderer
    results, errors = {}, []

    def work(tid):
        try:
            r = Renderer(0)
            try:
                for k in range(12):
                    seed = 3000 + 100 * tid + k
                    sc = random_scene(seed, lists=k % 2 == 1, wrappers=k % 3 == 2)
                    rt = REALS[(tid + k) % 2][0]
                    r.upload_scene(sc.flatten())
                    try:
                        img, st = r.render(sc.scene_cam, seed=seed, real_type=rt)
                    except CrucibleError as e:
                        if e.code != A.CR_ERR_NAN:
                            raise
                        img, st = None, None
                    results[(tid, k)] = (seed, rt, k, img, st)
            finally:
                r.close()
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert len(results) == 48
    for (tid, k), (seed, rt, kk, img, st) in sorted(results.items()):
        sc = random_scene(seed, lists=kk % 2 == 1, wrappers=kk % 3 == 2)
        ref, rst = oracles[rt].render_image(sc, seed=seed)
        if img is None:
            assert rst["nan_pixels"] > 0
            continue
        assert np.array_equal(img, ref), (tid, k)
        for c in COUNTERS:
            assert st[c] == rst[c], (tid, k, c)
