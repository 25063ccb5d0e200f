"""Small scenes used by the parity tests and by tests/golden/make_golden.py."""
import numpy as np

from crucible_amd.scene import (LERP, LOCAL, NERP, WORLD, BVHWrapper, CheckerTexture, Dielectric, HitList, ImageTexture, Lambertian,
                                Metal, RTWImage, Scene, SolidColor, Sphere, Triangle)


def small_image(w=16, h=8, seed=3):
    rs = np.random.RandomState(seed)
    return RTWImage(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8))


def mixed_scene(width=64, samples=4, sky=True, animate=False, depth=12):
    """Triangles + spheres, all three materials, solid / checker / nested checker / image textures,
    spherical sky, defocus blur; optional keyframes on camera, spheres and a triangle pair."""
    sc = Scene.new_image(16.0 / 9.0, width, 24, 180.0, 1)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(depth)
    cam.look_from((6.0, 2.5, 5.0))
    cam.look_at((0.0, 0.6, 0.0))
    cam.set_vfov(35.0)
    cam.set_defocus_angle(0.8)
    cam.set_focus_dist(7.5)
    img = small_image()
    nested = CheckerTexture.new_from_textures(1.5, CheckerTexture.new_from_color(0.25, (0.9, 0.1, 0.1), (0.1, 0.1, 0.9)),
                                              ImageTexture(img))
    sc.add_element(Sphere.new((0.0, -100.0, 0.0), 100.0, Lambertian.new_from_texture(nested, 1.0)), "ground")
    sc.add_element(Sphere.new((0.0, 1.0, 0.0), 1.0, Dielectric.new(1.5)), "glass")
    sc.add_element(Sphere.new((0.0, 1.0, 0.0), 0.6, Dielectric.new(1.0 / 1.5)), "bubble")
    sc.add_element(Sphere.new((-2.2, 0.8, 0.5), 0.8, Lambertian.new_from_texture(ImageTexture(img), 0.85)), "globe")
    sc.add_element(Sphere.new((2.2, 0.7, -0.3), 0.7, Metal.new((0.8, 0.6, 0.2), 0.3)), "brass")
    sc.add_element(Sphere.new((1.0, 0.3, 2.0), 0.3, Metal.new((0.9, 0.9, 0.9), 0.0)), "mirror")
    sc.add_element(Sphere.new((-1.0, 0.25, 2.2), 0.25, Lambertian.new_from_color((0.2, 0.7, 0.3), 0.6)), "matte")
    # a quad (two triangles, one alias each) and a tetrahedron behind the spheres
    m_quad = Metal.new((0.7, 0.7, 0.9), 0.1)
    sc.add_element(Triangle.new((-3.0, 0.0, -2.5), (3.0, 0.0, -2.5), (3.0, 3.0, -2.5), m_quad), "quad_a")
    sc.add_element(Triangle.new((-3.0, 0.0, -2.5), (3.0, 3.0, -2.5), (-3.0, 3.0, -2.5), m_quad), "quad_b")
    m_tet = Lambertian.new_from_texture(CheckerTexture.new_from_color(0.4, (0.9, 0.9, 0.2), (0.2, 0.2, 0.2)), 1.0)
    p = [(3.2, 0.0, 1.5), (4.2, 0.0, 1.2), (3.7, 0.0, 2.3), (3.7, 1.0, 1.7)]
    for k, (a, b, c) in enumerate([(0, 1, 3), (1, 2, 3), (2, 0, 3), (0, 2, 1)]):
        sc.add_element(Triangle.new(p[a], p[b], p[c], m_tet), f"tet{k}")
    # an axis-aligned flat triangle alone in its BVH leaf never hits (zero-thickness box, bvh.rs:126): keep one
    sc.add_element(Triangle.new((-4.0, 0.01, 3.0), (-3.0, 0.01, 3.0), (-3.5, 0.01, 4.0), Metal.new((1.0, 0.2, 0.2), 0.0)), "flat")
    if sky:
        rs = np.random.RandomState(11)
        sc.load_spherical_skybox(RTWImage(rs.randint(60, 256, size=(16, 32, 3)).astype(np.uint8)))
    if animate:
        sc.cam_translate_point((7.0, 3.0, 4.0), 0.02, LERP, WORLD, "from")
        sc.cam_translate_point((0.2, 0.5, 0.0), 0.015, LERP, WORLD, "at")
        sc.translate_point((0.3, 0.2, 0.0), 0.01, LERP, LOCAL, "brass")
        sc.translate_point((0.0, 0.1, 0.1), 0.012, NERP, LOCAL, "mirror")
        sc.scale_r(0.4, 0.02, LERP, "matte")
        sc.scale_r(0.9, 0.005, NERP, "globe")
        sc.translate_point((0.0, 0.3, 0.0), 0.018, LERP, LOCAL, "tet0")
    return sc


def few_spheres(n, width=48, samples=3):
    """n = 0, 1, 2, 3 ... primitives: the BVH edge cases (empty list, span-1 root, span-2 root, first sort)."""
    sc = Scene.new_image(16.0 / 9.0, width, 24, 180.0, 1)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(8)
    cam.look_from((0.0, 1.0, 6.0))
    cam.look_at((0.0, 0.5, 0.0))
    cam.set_vfov(40.0)
    mats = [Lambertian.new_from_color((0.8, 0.3, 0.3), 1.0), Metal.new((0.8, 0.8, 0.8), 0.1), Dielectric.new(1.5),
            Lambertian.new_from_texture(CheckerTexture.new_from_color(0.5, (0.1, 0.1, 0.1), (0.9, 0.9, 0.9)), 1.0),
            Lambertian.new_from_texture(SolidColor((0.3, 0.3, 0.9)), 0.5)]
    for k in range(n):
        x = (k - (n - 1) / 2.0) * 1.3
        sc.add_element(Sphere.new((x, 0.5 + 0.1 * (k % 3), -0.2 * k), 0.5, mats[k % len(mats)]), f"s{k}")
    return sc


def moving_scene(width=96, samples=6, frame=0, null_motion=False, depth=8):
    """Keyframed primitives that leave their construction-time boxes: 1 fps with a 360 degree shutter, so frame f
    draws ray times in [f, f + 1].  A sphere sweeping 6 units (LERP), one jumping mid-shutter (NERP), one growing
    (radius LERP) then snapping small (radius NERP), a late mover (frame 2), a moving triangle pair, static
    neighbours.  null_motion: the same keys with zero offsets / unchanged radii (refit must then change nothing)."""
    sc = Scene.new_image(16.0 / 9.0, width, 1, 360.0, 1)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(depth)
    cam.look_from((0.0, 3.0, 9.0))
    cam.look_at((0.0, 0.7, 0.0))
    cam.set_vfov(35.0)
    cam.frame = frame
    k = 0.0 if null_motion else 1.0
    ground = Lambertian.new_from_texture(CheckerTexture.new_from_color(0.8, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)
    sc.add_element(Sphere.new((0.0, -100.0, 0.0), 100.0, ground), "ground")
    sc.add_element(Sphere.new((-3.0, 0.5, 0.0), 0.5, Lambertian.new_from_color((0.8, 0.2, 0.2), 1.0)), "runner")
    sc.add_element(Sphere.new((0.0, 0.5, -2.0), 0.5, Metal.new((0.8, 0.8, 0.9), 0.05)), "jumper")
    sc.add_element(Sphere.new((2.5, 0.3, 1.5), 0.3, Lambertian.new_from_color((0.2, 0.3, 0.8), 1.0)), "grower")
    sc.add_element(Sphere.new((-2.0, 0.4, 2.5), 0.4, Dielectric.new(1.5)), "late")
    sc.add_element(Sphere.new((3.5, 0.6, -1.0), 0.6, Metal.new((0.9, 0.7, 0.3), 0.2)), "still_a")
    sc.add_element(Sphere.new((-4.0, 0.7, -1.5), 0.7, Lambertian.new_from_color((0.3, 0.7, 0.3), 1.0)), "still_b")
    m_tri = Metal.new((0.7, 0.7, 0.9), 0.1)
    sc.add_element(Triangle.new((-1.0, 0.0, 3.0), (0.0, 0.0, 3.2), (-0.5, 1.2, 3.1), m_tri), "tri_a")
    sc.add_element(Triangle.new((0.0, 0.0, 3.2), (1.0, 0.0, 3.0), (0.5, 1.2, 3.1), m_tri), "tri_b")
    sc.translate_point((6.0 * k, 0.0, 0.0), 1.0, LERP, LOCAL, "runner")
    sc.translate_point((0.0, 1.5 * k, 0.5 * k), 0.5, NERP, LOCAL, "jumper")
    sc.scale_r(0.3 + 0.9 * k, 0.75, LERP, "grower")
    sc.scale_r(0.3 + 0.2 * k, 1.5, NERP, "grower")
    sc.translate_point((0.0, 0.0, -3.0 * k), 2.0, NERP, LOCAL, "late")
    sc.translate_point((4.0 * k, 0.5 * k, 0.0), 3.0, LERP, LOCAL, "late")
    for alias in ("tri_a", "tri_b"):
        sc.translate_point((1.5 * k, 0.8 * k, -1.0 * k), 1.0, LERP, LOCAL, alias)
    return sc


def scaled_scene(width=96, samples=6, frame=0, depth=8):
    """Triangles under every non-sphere scale builder (scene_animator.rs:38-229): ScaleX / ScaleY / ScaleZ keys, LERP and
    NERP, scale_point and scale_all_uniform (whose Z key wins, timeline/mod.rs:249-255), mixed with translations; 1 fps
    with a 360 degree shutter, so frame f draws ray times in [f, f + 1] and the keys change inside the exposure.
    Default sky, solid and checker textures only: nothing goes through acos/atan2/asin, so renders are bit-exact."""
    sc = Scene.new_image(16.0 / 9.0, width, 1, 360.0, 1)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(depth)
    cam.look_from((1.0, 3.0, 9.0))
    cam.look_at((0.5, 1.0, 0.0))
    cam.set_vfov(38.0)
    cam.frame = frame
    ground = Lambertian.new_from_texture(CheckerTexture.new_from_color(0.8, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)
    sc.add_element(Sphere.new((0.0, -100.0, 0.0), 100.0, ground), "ground")
    sc.add_element(Sphere.new((-3.0, 0.6, 1.0), 0.6, Dielectric.new(1.5)), "glass")
    sc.add_element(Sphere.new((3.4, 0.5, 1.5), 0.5, Metal.new((0.8, 0.8, 0.9), 0.0)), "mirror")
    m_quad = Metal.new((0.7, 0.7, 0.9), 0.1)
    sc.add_element(Triangle.new((0.5, 0.0, -2.0), (2.5, 0.0, -2.0), (2.5, 2.0, -2.2), m_quad), "quad_a")
    sc.add_element(Triangle.new((0.5, 0.0, -2.0), (2.5, 2.0, -2.2), (0.5, 2.0, -2.2), m_quad), "quad_b")
    m_tet = Lambertian.new_from_texture(CheckerTexture.new_from_color(0.4, (0.9, 0.9, 0.2), (0.2, 0.2, 0.2)), 1.0)
    p = [(1.0, 0.0, 1.5), (2.0, 0.0, 1.2), (1.5, 0.0, 2.3), (1.5, 1.0, 1.7)]
    for k, (a, b, c) in enumerate([(0, 1, 3), (1, 2, 3), (2, 0, 3), (0, 2, 1)]):
        sc.add_element(Triangle.new(p[a], p[b], p[c], m_tet), f"tet{k}")
    m_fin = Lambertian.new_from_color((0.8, 0.3, 0.2), 1.0)
    sc.add_element(Triangle.new((-1.5, 0.0, 0.5), (-0.5, 0.0, 0.8), (-1.0, 1.5, 0.6), m_fin), "fin")
    for alias in ("quad_a", "quad_b"):
        sc.scale_x(1.5, 1.0, LERP, alias)                 # X wins until the Y key starts
        sc.scale_y(0.3, 0.5, NERP, alias)                 # from t = 0.5: y' = 0.3 * x + y (row 1, column 0)
        sc.scale_y(-0.2, 2.5, LERP, alias)
    for k in range(4):
        sc.scale_all_uniform(1.3, 1.0, LERP, f"tet{k}")   # X, Y, Z keys over [0, 1]: Z is last and wins
        sc.translate_point((0.4, 0.0, -0.5), 1.5, LERP, LOCAL, f"tet{k}")
        sc.scale_z(0.8, 2.0, NERP, f"tet{k}")
    sc.scale_point((0.5, 2.0, 1.5), 0.25, NERP, "fin")
    sc.translate_point((-0.5, 0.3, 0.0), 0.75, NERP, LOCAL, "fin")
    sc.scale_x(2.0, 1.75, LERP, "fin")
    return sc


def list_scene(width=96, samples=6, frame=0, depth=8, variant="mixed"):
    """HitList elements among ordinary ones (Scene::add_element keeps a list as one object of the BVH build,
    scene/mod.rs:164-166, bvhwrapper.rs:18-22): lists grown by add() (their box is the union, hidden objects
    included), a list from HitList::new(vec) (box stays Aabb::default(): sorted last, its wrapper box never shrinks the
    interval), an empty list, lists of one and two objects, a list inside a list, a hidden object, keyed objects
    (their timelines are filled before they join the list: objects of a list have no alias).
    variant "only_lists": every element is a list; "one_list": the world is a single add()-built list (a span-1
    root: the reference walks the list twice)."""
    sc = Scene.new_image(16.0 / 9.0, width, 1, 360.0, 1)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(depth)
    cam.look_from((0.5, 3.0, 9.5))
    cam.look_at((0.0, 0.8, 0.0))
    cam.set_vfov(36.0)
    cam.frame = frame
    ground = Lambertian.new_from_texture(CheckerTexture.new_from_color(0.8, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)
    red, blue = Lambertian.new_from_color((0.8, 0.2, 0.2), 1.0), Lambertian.new_from_color((0.2, 0.3, 0.8), 0.9)
    steel, brass, glass = Metal.new((0.8, 0.8, 0.9), 0.05), Metal.new((0.9, 0.7, 0.3), 0.2), Dielectric.new(1.5)

    # a row of spheres grown by add(); the third is hidden, the fourth moves inside the exposure
    row = HitList.default()
    for k in range(5):
        s = Sphere.new((-4.0 + 1.1 * k, 0.45, 1.0 + 0.3 * k), 0.45, [red, steel, blue, brass, glass][k])
        s.hide = k == 2
        if k == 3:
            s.timeline.translate_point((0.0, 0.8, 0.0), 0.5, LERP, LOCAL)
            s.timeline.scale_sphere(0.6, 1.5, NERP)
        row.add(s)
    # a small mesh (tetrahedron) built the way load_obj does, one triangle scaled by a key
    p = [(1.0, 0.0, 1.5), (2.2, 0.0, 1.2), (1.6, 0.0, 2.5), (1.6, 1.3, 1.7)]
    mesh = HitList.default()
    for k, (a, b, c) in enumerate([(0, 1, 3), (1, 2, 3), (2, 0, 3), (0, 2, 1)]):
        t = Triangle.new(p[a], p[b], p[c], Lambertian.new_from_texture(CheckerTexture.new_from_color(0.4, (0.9, 0.9, 0.2), (0.2, 0.2, 0.2)), 1.0))
        if k == 1:
            t.timeline.scale_x(1.2, 1.0, LERP)
        mesh.add(t)
    # HitList::new(vec): the box stays empty
    loose = HitList.new([Sphere.new((3.2, 0.6, -0.5), 0.6, steel), Sphere.new((3.0, 1.6, -0.6), 0.35, red),
                         Triangle.new((2.0, 0.0, -2.0), (4.5, 0.0, -2.2), (3.2, 2.5, -2.4), brass)])
    # a list inside a list, both grown by add()
    inner = HitList.default()
    inner.add(Sphere.new((-1.0, 0.3, 3.0), 0.3, glass))
    inner.add(Sphere.new((-0.3, 0.3, 3.3), 0.3, blue))
    outer = HitList.default()
    outer.add(Sphere.new((-1.8, 0.35, 3.4), 0.35, brass))
    outer.add(inner)
    outer.add(Triangle.new((-2.5, 0.0, 2.4), (-1.2, 0.0, 2.2), (-1.8, 1.1, 2.3), steel))
    single = HitList.default()
    single.add(Sphere.new((0.2, 0.5, 0.0), 0.5, steel))
    pair = HitList.default()
    pair.add(Sphere.new((0.4, 1.6, -2.5), 0.5, red))
    pair.add(Sphere.new((-0.8, 1.4, -2.5), 0.4, glass))

    if variant == "one_list":
        sc.add_element(row, "row")
        return sc
    if variant != "only_lists":
        sc.add_element(Sphere.new((0.0, -100.0, 0.0), 100.0, ground), "ground")
        sc.add_element(Sphere.new((-3.2, 0.7, -1.5), 0.7, blue), "still")
        sc.add_element(Triangle.new((-5.0, 0.0, -3.0), (-2.5, 0.0, -3.2), (-3.8, 2.2, -3.1), steel), "fin")
        sc.add_element(Sphere.new((4.6, 0.4, 1.8), 0.4, red), "hidden_top")
        sc.hide_element("hidden_top")
        sc.translate_point((0.0, 0.0, 1.5), 1.0, LERP, LOCAL, "still")
    sc.add_element(row, "row")
    sc.add_element(mesh, "mesh")
    sc.add_element(HitList.default(), "nothing")
    sc.add_element(loose, "loose")
    sc.add_element(outer, "outer")
    sc.add_element(single, "single")
    sc.add_element(pair, "pair")
    return sc


def wrapped_scene(width=96, samples=6, frame=0, depth=8, variant="mixed"):
    """BVHWrapper elements (`scene.add_element(BVHWrapper::new_wrapper(list), ..)`, scene/mod.rs:161-163): the outer build
    sorts a wrapper by its root box and a leaf wrapper that holds it walks into it (bvhwrapper.rs:96-126).
    "mixed": wrappers beside spheres, a triangle and a list -- leaves holding (primitive, wrapper), (wrapper, list) ...;
    "only": the world is one wrapper (a span-1 root: the reference walks it twice); "pair": two wrappers (a span-2 root
    of two sub-trees); "small": wrappers of one and of two objects, one without a visible object (an empty list)."""
    sc = Scene.new_image(16.0 / 9.0, width, 1, 360.0, 1)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(depth)
    cam.look_from((0.5, 3.0, 9.5))
    cam.look_at((0.0, 0.8, 0.0))
    cam.set_vfov(36.0)
    cam.frame = frame
    rs = np.random.RandomState(12)
    mats = [Lambertian.new_from_color((0.8, 0.2, 0.2), 1.0), Metal.new((0.8, 0.8, 0.9), 0.05), Dielectric.new(1.5),
            Lambertian.new_from_texture(CheckerTexture.new_from_color(0.4, (0.9, 0.9, 0.2), (0.2, 0.2, 0.2)), 1.0)]

    def cluster(cx, cz, n, keyed=False, hidden=False):
        objs = []
        for k in range(n):
            if k % 4 == 3:
                c = np.array([cx + rs.uniform(-1, 1), rs.uniform(0.2, 1.0), cz + rs.uniform(-1, 1)])
                o = Triangle.new(*(tuple(c + rs.uniform(-0.4, 0.4, 3)) for _ in range(3)), mats[k % 4])
            else:
                o = Sphere.new((cx + rs.uniform(-1.2, 1.2), rs.uniform(0.2, 0.9), cz + rs.uniform(-1.2, 1.2)), rs.uniform(0.15, 0.35), mats[k % 4])
            if keyed and k == 1:
                o.timeline.translate_point((0.0, 0.9, 0.0), 0.5, LERP, LOCAL)
            if hidden and k == 2:
                o.hide = True      # new_wrapper drops it
            objs.append(o)
        return objs
    big = BVHWrapper.new_wrapper(HitList.new(cluster(-2.0, 0.5, 17, keyed=True, hidden=True)))
    other = BVHWrapper.new_wrapper(HitList.new(cluster(2.5, -0.5, 9)))
    if variant == "only":
        sc.add_element(big, "big")
        return sc
    if variant == "pair":
        sc.add_element(big, "big")
        sc.add_element(other, "other")
        return sc
    ground = Lambertian.new_from_texture(CheckerTexture.new_from_color(0.8, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)
    sc.add_element(Sphere.new((0.0, -100.0, 0.0), 100.0, ground), "ground")
    if variant == "small":
        sc.add_element(BVHWrapper.new_wrapper(HitList.new(cluster(-2.0, 1.0, 1))), "one")
        sc.add_element(BVHWrapper.new_wrapper(HitList.new(cluster(1.5, 0.0, 2))), "two")
        gone = cluster(0.0, 2.0, 3)
        for o in gone:
            o.hide = True
        sc.add_element(BVHWrapper.new_wrapper(HitList.new(gone)), "none")
        return sc
    sc.add_element(big, "big")
    sc.add_element(Sphere.new((0.0, 0.6, 2.5), 0.6, mats[2]), "glass")
    sc.add_element(other, "other")
    row = HitList.default()
    for o in cluster(-0.5, -2.5, 5):
        row.add(o)
    sc.add_element(row, "row")
    sc.add_element(Triangle.new((-5.0, 0.0, -3.0), (-2.5, 0.0, -3.2), (-3.8, 2.2, -3.1), mats[1]), "fin")
    sc.add_element(BVHWrapper.new_wrapper(HitList.new(cluster(4.5, 2.0, 3))), "third")
    sc.translate_point((0.0, 0.0, 1.0), 1.0, LERP, LOCAL, "glass")
    return sc
