"""Analytic and property checks of the oracle restatement (oracle/crucible_oracle.c).  The reference pins none of
this, so these tests check the restatement against closed-form answers, against a brute-force
scan, and against the reference's documented quirks (each cited)."""
import math

import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene
from crucible_amd.scene import (BVHWrapper, CheckerTexture, Dielectric, HitList, ImageTexture, Lambertian, Metal, RTWImage, Scene, Sphere,
                                Triangle)

INF = float("inf")


@pytest.fixture(params=[A.CR_REAL_F64, A.CR_REAL_F32], ids=["f64", "f32"])
def o(request, oracles):
    return oracles[request.param]


def sphere_hit(o, cr, orig, d, tmin=0.001, tmax=INF):
    out = np.zeros(10, dtype=o.np_real)
    hit = o.lib.oracle_sphere_hit(o._p(o.arr(cr)), o._p(o.arr(orig)), o._p(o.arr(d)), tmin, tmax, o._p(out))
    return hit, out


def tri_hit(o, abc, orig, d, tmin=0.001, tmax=INF):
    out = np.zeros(10, dtype=o.np_real)
    hit = o.lib.oracle_triangle_hit(o._p(o.arr(abc)), o._p(o.arr(orig)), o._p(o.arr(d)), tmin, tmax, o._p(out))
    return hit, out


def test_sphere_hit_closed_form(o):   # sphere.rs:60-105
    hit, r = sphere_hit(o, [0, 0, 0, 1], [0, 0, -5], [0, 0, 1])
    assert hit and r[0] == 4.0 and list(r[1:4]) == [0, 0, -1] and list(r[4:7]) == [0, 0, -1] and r[9] == 1   # outward normal, front face
    hit, r = sphere_hit(o, [0, 0, 0, 1], [0, 0, 0], [0, 0, 2])           # from inside: second root, flipped normal, unnormalised dir
    assert hit and r[0] == 0.5 and list(r[4:7]) == [0, 0, -1] and r[9] == 0
    assert not sphere_hit(o, [0, 0, 0, 1], [0, 2, -5], [0, 0, 1])[0]     # discriminant < 0
    assert not sphere_hit(o, [0, 0, 0, 1], [0, 0, -5], [0, 0, 1], 0.001, 4.0)[0]   # strict surrounds: t == tmax rejected, far root beyond
    hit, r = sphere_hit(o, [0, 0, 0, 1], [0, 0, -5], [0, 0, 1], 4.5, INF)
    assert hit and r[0] == 6.0                                           # near root outside the interval -> far root


def test_sphere_uv(o):   # get_sphere_uv, sphere.rs:41-46: u = (atan2(-z, x) + pi) / 2pi, v = acos(-y) / pi
    hit, r = sphere_hit(o, [0, 0, 0, 1], [5, 0, 0], [-1, 0, 0])          # hits (1,0,0)
    assert hit and abs(r[7] - 0.5) < 1e-6 and abs(r[8] - 0.5) < 1e-6
    hit, r = sphere_hit(o, [0, 0, 0, 1], [0, 5, 0], [0, -1, 0])          # north pole: v = acos(-1)/pi = 1
    assert hit and abs(r[8] - 1.0) < 1e-6


def test_triangle_hit(o):   # triangle.rs:84-140, Moller-Trumbore, two-sided
    abc = [0, 0, 0, 1, 0, 0, 0, 1, 0]
    hit, r = tri_hit(o, abc, [0.25, 0.25, 1], [0, 0, -1])
    assert hit and r[0] == 1.0 and list(r[1:4]) == [0.25, 0.25, 0] and list(r[4:7]) == [0, 0, 1] and r[9] == 1 and r[7] == r[8] == 0
    hit, r = tri_hit(o, abc, [0.25, 0.25, -1], [0, 0, 1])                # from behind: normal flips
    assert hit and list(r[4:7]) == [0, 0, -1] and r[9] == 0
    assert not tri_hit(o, abc, [0.75, 0.75, 1], [0, 0, -1])[0]           # u + v > 1
    assert not tri_hit(o, abc, [-0.1, 0.2, 1], [0, 0, -1])[0]            # u < 0
    assert not tri_hit(o, abc, [0.25, 0.25, 1], [1, 0, 0])[0]            # parallel: |det| < EPSILON
    assert tri_hit(o, abc, [1.0, 0.0, 1], [0, 0, -1])[0]                 # vertex b: u == 1 is inside (0.0..=1.0)
    assert not tri_hit(o, [0, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 1], [0, 0, -1])[0]   # degenerate


def test_aabb_quirks(o):   # bvh.rs:96-132
    box = o.arr([0, 1, 0, 1, 0, 1])
    hit = lambda b, orig, d, tmin=0.001, tmax=INF: o.lib.oracle_aabb_hit(o._p(o.arr(b)), o._p(o.arr(orig)), o._p(o.arr(d)), tmin, tmax)
    assert hit(box, [0.5, 0.5, -1], [0, 0, 1]) and not hit(box, [1.5, 0.5, -1], [0, 0, 1])
    assert not hit(box, [0.5, 0.5, -1], [0, 0, 1], 0.001, 1.0)           # max <= min: touching the entry face is a miss
    flat = [0, 1, 0, 1, 0.5, 0.5]                                        # zero thickness: never hit (`ray_t.max() <= ray_t.min()`)
    assert not hit(flat, [0.5, 0.5, -1], [0, 0, 1])
    assert hit(box, [0.5, 0.5, 0.5], [0, 0, 1]) and hit(box, [0.5, 0.5, 0.5], [1, 0, 0])   # origin inside, zero components
    assert not hit(box, [0.5, 2.0, 0.5], [1, 0, 0])                      # parallel to the y slab, outside it


def test_vector_helpers(o):   # utils.rs:149-163
    r = o.vec_fn("oracle_reflect", [1, -1, 0], [0, 1, 0])
    assert list(r) == [1, 1, 0]
    v = o.arr([math.sin(0.5), -math.cos(0.5), 0.0])
    out = np.zeros(3, dtype=o.np_real)
    o.lib.oracle_refract(o._p(v), o._p(o.arr([0, 1, 0])), 1.0 / 1.5, o._p(out))
    tol = 1e-12 if o.real_type == A.CR_REAL_F64 else 1e-6
    assert abs(out[0] - math.sin(0.5) / 1.5) < tol                       # Snell: sin(theta_t) = eta * sin(theta_i)
    assert abs(np.linalg.norm(out) - 1.0) < tol and out[1] < 0


def test_color_ops(o):   # utils.rs:445-607
    assert list(o.vec_fn("oracle_color_mul", [0.5, 1.0, 0.25], [0.5, 0.5, 4.0])) == [0.25, 0.5, 1.0]     # clamped product
    out = np.zeros(3, dtype=o.np_real)
    o.lib.oracle_color_div(o._p(o.arr([0.2, 0.4, 0.9])), 0.5, o._p(out))
    assert list(out) == [o.np_real(2.0) * o.np_real(0.2), o.np_real(2.0) * o.np_real(0.4), 1.0]           # (1/rhs) * c, saturating
    o.lib.oracle_color_scale(-0.5, o._p(o.arr([1.0, 0.0, 0.0])), o._p(out))
    assert list(out) == [0.0, 0.5, 0.5]                                  # negative scalar: complement first (utils.rs:546-549)


def build(prims, sky=None, **cam):
    sc = Scene.new_image(16 / 9, 32, 24, 180.0, 1)
    for k, p in enumerate(prims):
        sc.add_element(p, f"p{k}")
    if sky is not None:
        sc.load_spherical_skybox(sky)
    return sc


def test_bvh_equals_brute_force(o):
    """BVHWrapper::hit (bvhwrapper.rs:96-126) must return the closest hit of a linear scan (HitList::hit, hitlist.rs:51-65)
    for random scenes of spheres and (non axis-flat) triangles."""
    rs = np.random.RandomState(7)
    m = Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)
    for n in (1, 2, 3, 7, 64, 200):
        prims, geo = [], []
        for k in range(n):
            if k % 3 == 2:
                p = rs.uniform(-4, 4, size=(3, 3))
                prims.append(Triangle.new(*p, m)); geo.append(("t", p.reshape(-1)))
            else:
                c, r = rs.uniform(-4, 4, size=3), rs.uniform(0.1, 1.0)
                prims.append(Sphere.new(c, r, m)); geo.append(("s", np.array([*c, r])))
        h = o.scene_create(build(prims).flatten())
        try:
            for _ in range(200):
                orig, d = rs.uniform(-6, 6, size=3), rs.normal(size=3)
                best = INF
                for kind, g in geo:
                    hit, r = (sphere_hit if kind == "s" else tri_hit)(o, g, orig, d, 0.001, best)
                    if hit:
                        best = float(r[0])
                out = np.zeros(10, dtype=o.np_real); mat = np.zeros(1, dtype=np.int32)
                hit = o.lib.oracle_world_hit(h, o._p(o.arr(orig)), o._p(o.arr(d)), 0.0, 0.001, INF, o._p(out), mat.ctypes.data)
                assert (hit == 1) == (best < INF)
                if hit:
                    assert float(out[0]) == best
        finally:
            o.scene_destroy(h)


def test_bvh_topology_matches_reference_shape(o64):
    """N primitives give a full tree of wrappers: 484 -> 511, 6321 -> 8191."""
    sc = book1_end_scene(1, scene_seed=1, image_width=16, samples=1)
    h = o64.scene_create(sc.flatten())
    try:
        boxes = np.zeros((600, 6)); kids = np.zeros((600, 2), dtype=np.int32)
        n = o64.lib.oracle_bvh_dump(h, boxes.ctypes.data, kids.ctypes.data, 600)
        assert n == 511
        leaves = kids[:n][(kids[:n] >= 0).all(axis=1)]
        assert sorted(set(leaves.reshape(-1))) == list(range(484))       # every primitive in exactly one leaf wrapper
        root = boxes[0]
        assert root[2] == -2000.0 and root[0] == -1000.0 and root[1] == 1000.0   # the ground sphere's box dominates
    finally:
        o64.scene_destroy(h)


def scatter(o, h, mat, d, normal, front, sample):
    rec = o.arr([1.0, 0.0, 0.0, 0.0, *normal, 0.3, 0.6, 1.0 if front else 0.0])
    out = np.zeros(10, dtype=o.np_real)
    some = o.lib.oracle_scatter(h, mat, o._p(o.arr([0, 5, 0])), o._p(o.arr(d)), o._p(rec), 99, 3, sample, o._p(out))
    return some, out


def test_materials(o):
    sc = build([Sphere.new((0, 0, 0), 1.0, Lambertian.new_from_color((0.2, 0.4, 0.8), 0.5)),
                Sphere.new((3, 0, 0), 1.0, Metal.new((0.9, 0.8, 0.7), 0.0)),
                Sphere.new((6, 0, 0), 1.0, Dielectric.new(1.5)),
                Sphere.new((9, 0, 0), 1.0, Lambertian.new_from_color((0.9, 0.9, 0.9), 1.0))])
    h = o.scene_create(sc.flatten())
    try:
        kept = 0
        for s in range(400):
            some, out = scatter(o, h, 0, [0, -1, 0], [0, 1, 0], True, s)
            assert list(out[0:3]) == [o.np_real(2.0) * o.np_real(0.2), o.np_real(2.0) * o.np_real(0.4), 1.0]   # tex / prob, clamped (lambertian.rs:51-54)
            assert out[7] > -1e-6 and abs(np.linalg.norm(out[6:9] - np.array([0, 1, 0])) - 1.0) < 1e-5           # normal + unit vector
            assert out[9] >= 4 and (out[9] - 1) % 3 == 0                 # 3 draws per rejection round + the roulette draw
            kept += some
        assert 150 < kept < 250                                          # scatter_prob = 0.5
        some, out = scatter(o, h, 3, [0, -1, 0], [0, 1, 0], True, 1)
        assert some == 1                                                 # prob 1.0 always scatters (the draw is still consumed)
        some, out = scatter(o, h, 1, [1, -1, 0], [0, 1, 0], True, 0)     # fuzz 0: mirror direction, but the unit vector is still drawn
        assert some == 1 and list(out[0:3]) == list(o.arr([0.9, 0.8, 0.7])) and out[9] >= 3
        assert np.allclose(out[6:9], np.array([1, 1, 0]) / math.sqrt(2), atol=1e-6)
        some, _ = scatter(o, h, 1, [1, 1, 0], [0, 1, 0], True, 0)        # reflected into the surface: absorbed (metal.rs:37-41)
        assert some == 0
        # dielectric: attenuation (1,1,1), always Some; total internal reflection from inside at a grazing angle draws nothing
        some, out = scatter(o, h, 2, [1, -0.2, 0], [0, 1, 0], False, 0)
        assert some == 1 and list(out[0:3]) == [1, 1, 1] and out[9] == 0 and out[7] > 0
        some, out = scatter(o, h, 2, [0, -1, 0], [0, 1, 0], True, 0)     # normal incidence from outside: one draw, mostly refracts straight
        assert some == 1 and out[9] == 1
    finally:
        o.scene_destroy(h)


def test_textures_and_sky(o):
    img = RTWImage(np.arange(4 * 2 * 3, dtype=np.uint8).reshape(2, 4, 3))
    chk = CheckerTexture.new_from_color(0.5, (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    sc = build([Sphere.new((0, 0, 0), 1.0, Lambertian.new_from_texture(chk, 1.0)),
                Sphere.new((3, 0, 0), 1.0, Lambertian.new_from_texture(ImageTexture(img), 1.0))], sky=img)
    flat = sc.flatten()
    h = o.scene_create(flat)
    try:
        kinds = [flat.textures[i].kind for i in range(flat.desc.n_textures)]
        t_chk, t_img = kinds.index(A.CR_TEX_CHECKER), kinds.index(A.CR_TEX_IMAGE)
        tv = lambda t, u, v, p: (lambda out: (o.lib.oracle_texture_value(h, t, u, v, o._p(o.arr(p)), o._p(out)), list(out))[1])(np.zeros(3, dtype=o.np_real))
        assert tv(t_chk, 0, 0, [0.1, 0.1, 0.1]) == [1, 0, 0]             # floor(2*0.1)*3 = 0 -> even
        assert tv(t_chk, 0, 0, [0.6, 0.1, 0.1]) == [0, 0, 1]             # 1 -> odd
        assert tv(t_chk, 0, 0, [-0.1, 0.1, 0.1]) == [0, 0, 1]            # -1 % 2 == -1 in Rust: odd (checker_texture.rs:44)
        assert tv(t_chk, 0, 0, [-0.6, 0.1, 0.1]) == [1, 0, 0]            # -2 -> even
        px = lambda x, y: [c / 255.0 for c in img.rgb8[y, x]]
        close = lambda a, b: np.allclose(a, b, atol=1e-6)
        assert close(tv(t_img, 0.0, 1.0, [0, 0, 0]), px(0, 0))            # v flipped: v = 1 is the top row (image_texture.rs:25)
        assert close(tv(t_img, 0.99, 0.0, [0, 0, 0]), px(3, 1))
        assert close(tv(t_img, 1.0, 0.0, [0, 0, 0]), px(3, 1))            # i = W clamps to W - 1 (img_loader.rs:72)
        assert close(tv(t_img, -5.0, 7.0, [0, 0, 0]), px(0, 0))           # u, v clamped to [0, 1]
        sky = lambda d: (lambda out: (o.lib.oracle_sky(h, o._p(o.arr(d)), o._p(out)), list(out))[1])(np.zeros(3, dtype=o.np_real))
        assert close(sky([0, 1, 0]), px(2, 0))                           # phi = pi/2 -> v = 1 -> top row; theta = 0 -> u = 0.5
        assert close(sky([0, -1, 0]), px(2, 1))
        assert close(sky([-1e-9, 0, -1]), px(0, 0)) or close(sky([-1e-9, 0, -1]), px(0, 1))   # theta -> -pi: u -> 0
    finally:
        o.scene_destroy(h)
    sc = build([])                                                       # default sky: (1-a)*white + a*(0.5,0.7,1.0), a = 0.5(y+1)
    h = o.scene_create(sc.flatten())
    try:
        out = np.zeros(3, dtype=o.np_real)
        o.lib.oracle_sky(h, o._p(o.arr([0, 2, 0])), o._p(out))
        assert np.allclose(out, [0.5, 0.7, 1.0], atol=1e-6)
        o.lib.oracle_sky(h, o._p(o.arr([0, -3, 0])), o._p(out))
        assert np.allclose(out, [1, 1, 1], atol=1e-6)
    finally:
        o.scene_destroy(h)


def test_camera_ray_geometry(o):
    """Pixel centres map onto the focus plane: rays of the four corners and the centre (rendering_compute.rs)."""
    sc = build([])
    cam = sc.scene_cam
    cam.look_from((0, 0, 5)); cam.look_at((0, 0, 0)); cam.set_vfov(90.0); cam.set_focus_dist(5.0)
    cd, p = cam.desc(), cam.params(1, o.real_type)
    out = np.zeros(8, dtype=o.np_real)
    W, H = cam.image_width, cam.image_height
    o.lib.oracle_camera_ray(cd, p, W // 2, H // 2, 0, o._p(out))
    assert list(out[0:3]) == [0, 0, 5] and out[7] == 3                   # no defocus: origin = look_from; draws: time, x, y
    target = out[0:3] + out[3:6]                                         # unnormalised direction ends on the focus plane
    assert abs(target[2]) < 1e-5 and abs(target[0]) < 10.0 / H and abs(target[1]) < 10.0 / H   # one pixel = 10/H in x and y
    o.lib.oracle_camera_ray(cd, p, 0, 0, 0, o._p(out))
    target = out[0:3] + out[3:6]
    assert target[0] < -5.0 * (W / H) * 0.9 and target[1] > 5.0 * 0.9    # upper-left corner: -x, +y
    cam.set_defocus_angle(10.0)
    cd = cam.desc()
    o.lib.oracle_camera_ray(cd, p, 3, 4, 2, o._p(out))
    r = math.hypot(out[0], out[1])
    assert out[7] >= 5 and (out[7] - 3) % 2 == 0 and r <= 5.0 * math.tan(math.radians(5.0)) + 1e-6 and out[2] == 5   # lens disk
    assert 0.0 <= out[6] <= (180.0 / 360.0) / 24.0                       # time in [0, shutter_length]


@pytest.mark.parametrize("frame", [0, 2])
def test_refit_boxes_rule_against_the_linear_list(o64, frame):
    """refit_boxes (not in the reference; crucible_amd/csrc/refit.hpp): with the wrapper boxes re-derived for the
    frame, the BVH gives the linear list's closest hits (HitList::hit, no boxes); with the reference's stale
    construction-time boxes (bvhwrapper.rs:47-50) the moving primitives are clipped."""
    import scenes
    sc = scenes.moving_scene(96, 3, frame=frame)
    sc.scene_cam.refit_boxes = False
    truth, tst = o64.render_image(sc, seed=9, linear_list=True)
    stale, _ = o64.render_image(sc, seed=9)
    sc.scene_cam.refit_boxes = True
    fitted, fst = o64.render_image(sc, seed=9)
    assert (fitted == truth).all(axis=2).mean() >= 0.995
    assert (stale == truth).all(axis=2).mean() < 0.97
    assert fst["segments"] == pytest.approx(tst["segments"], rel=2e-3)


def test_refit_boxes_identity_without_motion(o64):
    import scenes
    sc = scenes.moving_scene(64, 2, frame=0, null_motion=True)
    sc.scene_cam.refit_boxes = False
    a, sa = o64.render_image(sc, seed=9)
    sc.scene_cam.refit_boxes = True
    b, sb = o64.render_image(sc, seed=9)
    assert np.array_equal(a, b) and sa["node_tests"] == sb["node_tests"]


def test_rng_streams_look_uniform_and_independent(o64):
    """One xorshift64* stream per (seed, pixel, sample), keyed by SplitMix64's finaliser: first draws of neighbouring
    keys must be uniform and uncorrelated (a weak key derivation would show up exactly here: the scheduler hands
    neighbouring pixels and consecutive samples to neighbouring lanes)."""
    n_pix, n_smp, n_draw = 64, 64, 6
    u = np.zeros((n_pix, n_smp, n_draw))
    buf = np.zeros(n_draw)
    for p in range(n_pix):
        for s in range(n_smp):
            o64.lib.oracle_rng_uniforms(1234, p, s, n_draw, buf.ctypes.data)
            u[p, s] = buf
    n = u.size
    assert abs(u.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n)
    assert abs(u.var() - 1 / 12) < 0.002
    assert 0.0 <= u.min() and u.max() < 1.0
    hist = np.histogram(u, bins=16, range=(0, 1))[0]
    assert ((hist - n / 16) ** 2 / (n / 16)).sum() < 45          # chi-square, 15 dof: p ~ 1e-4
    c = lambda a, b: abs(np.corrcoef(a.ravel(), b.ravel())[0, 1])
    lim = 5 / np.sqrt(n_pix * n_smp)
    assert c(u[:, :-1, 0], u[:, 1:, 0]) < lim        # consecutive samples of a pixel
    assert c(u[:-1, :, 0], u[1:, :, 0]) < lim        # neighbouring pixels
    assert c(u[:, :, 0], u[:, :, 1]) < lim and c(u[:, :, 1], u[:, :, 2]) < lim   # consecutive draws of a stream
    # a different seed gives a different stream
    o64.lib.oracle_rng_uniforms(1235, 0, 0, n_draw, buf.ctypes.data)
    assert not np.array_equal(buf, u[0, 0])


@pytest.mark.parametrize("frame", [0, 1, 2])
def test_refit_boxes_with_scale_keys_against_the_linear_list(o64, frame):
    """Triangles under ScaleX/ScaleY/ScaleZ keys move along products of piecewise-linear functions; the refit rule
    samples their translate and scale parts independently (refit.hpp) and must still enclose them."""
    import scenes
    sc = scenes.scaled_scene(96, 3, frame=frame)
    sc.scene_cam.refit_boxes = False
    truth, _ = o64.render_image(sc, seed=9, linear_list=True)
    stale, _ = o64.render_image(sc, seed=9)
    sc.scene_cam.refit_boxes = True
    fitted, _ = o64.render_image(sc, seed=9)
    assert (fitted == truth).all(axis=2).mean() >= 0.995
    assert (stale == truth).all(axis=2).mean() < 0.98


def _ulp_distance(u, v):
    it = np.int64 if u.dtype == np.float64 else np.int32
    top = 63 if u.dtype == np.float64 else 31
    ui, vi = u.view(it).astype(np.int64), v.view(it).astype(np.int64)
    ui = np.where(ui < 0, -(ui & ((1 << top) - 1)), ui)
    vi = np.where(vi < 0, -(vi & ((1 << top) - 1)), vi)
    return np.abs(ui - vi)


def test_defined_trig_functions_against_glibc(oracles):
    """atan2 / asin / acos are DEFINED by the build (DESIGN.md "software trigonometry") so that oracle and device
    agree bit for bit; the reference calls the platform libm.  The definition stays within 2 ulp of glibc over the
    ranges the path uses and wider, and agrees on the special values that select a quadrant."""
    for rt, o in oracles.items():
        rs = np.random.RandomState(5)
        n = 200000
        y = np.concatenate([rs.uniform(-1, 1, n), rs.uniform(-1, 1, n) * 10.0 ** rs.uniform(-8, 8, n)])
        x = np.concatenate([rs.uniform(-1, 1, n), rs.uniform(-1, 1, n) * 10.0 ** rs.uniform(-8, 8, n)])
        special_y = [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.4375, 0.6875, 1.1875, 2.4375, 1.0, -1.0, np.inf, -np.inf, np.nan, 2.0]
        special_x = [1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 0.5, 1.0, 1.0, 1.0, 1.0, 1.0, -0.0, 0.0, 1.0, -np.inf, 1.0, 1.0]
        y, x = np.concatenate([y, special_y]), np.concatenate([x, special_x])
        o.set_libm(False)
        mine = o.trig(y, x)
        o.set_libm(True)
        try:
            libm = o.trig(y, x)
        finally:
            o.set_libm(False)
        for name, a, b in zip(("atan2", "asin", "acos"), mine, libm):
            assert np.array_equal(np.isnan(a), np.isnan(b)), name
            ok = ~np.isnan(a)
            assert _ulp_distance(a[ok], b[ok]).max() <= 2, (name, rt)
            k = len(special_y)
            tail = ~np.isnan(a[-k:])
            assert np.array_equal(np.signbit(a[-k:][tail]), np.signbit(b[-k:][tail])), name


def test_images_do_not_depend_on_which_trig_is_used(o64):
    """A last-ulp difference in u or v moves a texel only when u*W sits within an ulp of an integer: on the mixed
    scene (sphere u,v + spherical sky) the defined functions and glibc give the same image."""
    import scenes
    sc = scenes.mixed_scene(64, 4)
    a, sa = o64.render_image(sc, seed=3)
    o64.set_libm(True)
    try:
        b, sb = o64.render_image(sc, seed=3)
    finally:
        o64.set_libm(False)
    assert (a == b).all(axis=2).mean() >= 0.999 and sa["segments"] == sb["segments"] and sa["texel_fetches"] == sb["texel_fetches"]


def test_faithful_evaluation_is_the_same_arithmetic(oracles):
    """bench.py's second CPU baseline (SURVEY 8(d) "faithful"): timelines through 4x4 matrices of closures evaluated
    at every hit (timeline/mod.rs:90-96,233-263), the dead update_bb calls (bvhwrapper.rs:104-106), material
    reference counts and the per-pixel mutex.  It must change the time, never a bit of the image."""
    import scenes
    from crucible_amd.demo_builder import book1_end_scene
    for rt, o in oracles.items():
        for sc in (book1_end_scene(1, scene_seed=2, image_width=96, samples=3), scenes.mixed_scene(48, 3, animate=True),
                   scenes.scaled_scene(64, 3, frame=1), scenes.moving_scene(64, 3, frame=0)):
            a, sa = o.render_image(sc, seed=11, n_threads=2)
            o.set_faithful(True)
            try:
                b, sb = o.render_image(sc, seed=11, n_threads=2)
            finally:
                o.set_faithful(False)
            assert np.array_equal(a, b)
            assert all(sa[k] == sb[k] for k in ("segments", "node_tests", "prim_tests", "texel_fetches"))


# ---- HitList as a scene element (scene/mod.rs:164-166, bvhwrapper.rs:18-22, hitlist.rs)
def test_list_element_random_rays_equal_brute_force(o):
    """A tree whose leaves hold add()-built lists (true boxes, hidden objects included) still returns the closest hit
    of a linear scan over every visible object, for random static scenes."""
    rs = np.random.RandomState(21)
    m = Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)
    for trial in range(6):
        sc = build([])
        geo = []
        for li in range(rs.randint(1, 6)):
            objs = []
            for k in range(rs.randint(0, 9)):
                if rs.rand() < 0.35:
                    p = rs.uniform(-4, 4, size=(3, 3))
                    objs.append(Triangle.new(*p, m)); g = ("t", p.reshape(-1))
                else:
                    c, r = rs.uniform(-4, 4, size=3), rs.uniform(0.1, 1.0)
                    objs.append(Sphere.new(c, r, m)); g = ("s", np.array([*c, r]))
                objs[-1].hide = rs.rand() < 0.15
                if not objs[-1].hide:
                    geo.append(g)
            l = HitList.default()
            for ob in objs:
                l.add(ob)
            sc.add_element(l, f"l{li}")
            if rs.rand() < 0.6:
                c, r = rs.uniform(-4, 4, size=3), rs.uniform(0.1, 1.0)
                sc.add_element(Sphere.new(c, r, m), f"s{li}"); geo.append(("s", np.array([*c, r])))
        h = o.scene_create(sc.flatten())
        try:
            for _ in range(300):
                orig, d = rs.uniform(-6, 6, size=3), rs.normal(size=3)
                best = INF
                for kind, g in geo:
                    hit, r = (sphere_hit if kind == "s" else tri_hit)(o, g, orig, d, 0.001, best)
                    if hit:
                        best = float(r[0])
                out = np.zeros(10, dtype=o.np_real); mat = np.zeros(1, dtype=np.int32)
                hit = o.lib.oracle_world_hit(h, o._p(o.arr(orig)), o._p(o.arr(d)), 0.0, 0.001, INF, o._p(out), mat.ctypes.data)
                assert (hit == 1) == (best < INF)
                if hit:
                    assert float(out[0]) == best
        finally:
            o.scene_destroy(h)


@pytest.mark.parametrize("variant", ["mixed", "only_lists", "one_list"])
def test_list_scene_refit_agrees_with_the_linear_list(o64, variant):
    """Keyed objects inside lists leave the construction-time boxes like any other primitive; with refit_boxes the
    tree returns the linear list's closest hits, and the stale boxes clip."""
    sc = scenes.list_scene(64, 3, frame=1, variant=variant)
    truth, _ = o64.render_image(sc, seed=9, linear_list=True)
    sc.scene_cam.refit_boxes = True
    fitted, _ = o64.render_image(sc, seed=9)
    sc.scene_cam.refit_boxes = False
    stale, _ = o64.render_image(sc, seed=9)
    assert (fitted == truth).all(axis=2).mean() >= 0.995
    assert (stale == truth).all(axis=2).mean() < 0.995


def test_empty_box_list_is_seen_only_through_its_sibling(o64):
    """HitList::new(vec) keeps Aabb::default() (hitlist.rs:13-18).  The empty box adds nothing to the parent wrapper
    (utils.rs:631-635), so the list is scanned exactly when the ray crosses the OTHER child's box -- and alone in
    the world its wrapper box is empty, which never shrinks the interval (bvh.rs:96-130): everything is scanned."""
    m = Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)
    far = Sphere.new((50.0, 0.0, 0.0), 1.0, m)

    def world_hit(sc, orig, d):
        h = o64.scene_create(sc.flatten())
        try:
            out = np.zeros(10); mat = np.zeros(1, dtype=np.int32)
            hit = o64.lib.oracle_world_hit(h, o64._p(o64.arr(orig)), o64._p(o64.arr(d)), 0.0, 0.001, INF, o64._p(out), mat.ctypes.data)
            return hit, float(out[0])
        finally:
            o64.scene_destroy(h)
    sc = build([])
    sc.add_element(HitList.new([Sphere.new((0.0, 0.0, 0.0), 1.0, m)]), "loose")
    sc.add_element(far, "far")
    assert world_hit(sc, (0.0, 0.0, 5.0), (0.0, 0.0, -1.0)) == (0, 0.0)            # straight at the list's sphere: culled
    hit, t = world_hit(sc, (-5.0, 0.0, 0.0), (1.0, 0.0, 0.0))                      # through both: the near one wins
    assert hit == 1 and t == 4.0
    alone = build([])
    alone.add_element(HitList.new([Sphere.new((0.0, 0.0, 0.0), 1.0, m)]), "loose")
    hit, t = world_hit(alone, (0.0, 0.0, 5.0), (0.0, 0.0, -1.0))
    assert hit == 1 and t == 4.0
    grown = build([])
    l = HitList.default()
    l.add(Sphere.new((0.0, 0.0, 0.0), 1.0, m))
    grown.add_element(l, "grown")
    grown.add_element(far, "far")
    hit, t = world_hit(grown, (0.0, 0.0, 5.0), (0.0, 0.0, -1.0))
    assert hit == 1 and t == 4.0


def test_span_one_list_is_walked_twice_and_counted_once(o64):
    """A world of one list is a span-1 root whose two children are the same list (bvhwrapper.rs:56-58): the reference
    scans it twice; the second scan cannot change the result (every t is outside the shrunk interval).  The work
    counter the device is compared with counts one scan."""
    sc = scenes.list_scene(48, 2, variant="one_list")
    h = o64.scene_create(sc.flatten())
    try:
        boxes = np.zeros((4, 6)); kids = np.zeros((4, 2), dtype=np.int32)
        assert o64.lib.oracle_bvh_dump(h, boxes.ctypes.data, kids.ctypes.data, 4) == 1
        assert kids[0, 0] == kids[0, 1] == 0      # the list record is prims[0]
    finally:
        o64.scene_destroy(h)
    img, st = o64.render_image(sc, seed=4)
    visible = 4                                    # five spheres, one hidden
    assert st["prim_tests"] <= visible * st["node_tests"]


@pytest.mark.parametrize("variant", ["mixed", "only", "pair", "small"])
def test_wrapper_element_scene_refit_agrees_with_the_linear_list(o64, variant):
    """BVHWrapper elements (scene/mod.rs:161-163): with refit_boxes every tree level -- the element's own included --
    follows the keyed objects and the closest hits are the linear list's."""
    sc = scenes.wrapped_scene(64, 3, frame=1, variant=variant)
    truth, _ = o64.render_image(sc, seed=9, linear_list=True)
    sc.scene_cam.refit_boxes = True
    fitted, _ = o64.render_image(sc, seed=9)
    assert (fitted == truth).all(axis=2).mean() >= 0.995


def test_wrapper_element_random_rays_equal_brute_force(o):
    """Static objects: a world holding wrapper elements returns the closest hit of a linear scan."""
    rs = np.random.RandomState(33)
    m = Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)
    for trial in range(6):
        sc = build([])
        geo = []
        for wi in range(rs.randint(1, 4)):
            objs = []
            for k in range(rs.randint(1, 12)):
                c, r = rs.uniform(-4, 4, size=3), rs.uniform(0.1, 1.0)
                objs.append(Sphere.new(c, r, m))
                objs[-1].hide = rs.rand() < 0.15
                if not objs[-1].hide:
                    geo.append(("s", np.array([*c, r])))
            sc.add_element(BVHWrapper.new_wrapper(HitList.new(objs)), f"w{wi}")
            if rs.rand() < 0.6:
                c, r = rs.uniform(-4, 4, size=3), rs.uniform(0.1, 1.0)
                sc.add_element(Sphere.new(c, r, m), f"s{wi}"); geo.append(("s", np.array([*c, r])))
        h = o.scene_create(sc.flatten())
        try:
            for _ in range(300):
                orig, d = rs.uniform(-6, 6, size=3), rs.normal(size=3)
                best = INF
                for kind, g in geo:
                    hit, r = sphere_hit(o, g, orig, d, 0.001, best)
                    if hit:
                        best = float(r[0])
                out = np.zeros(10, dtype=o.np_real); mat = np.zeros(1, dtype=np.int32)
                hit = o.lib.oracle_world_hit(h, o._p(o.arr(orig)), o._p(o.arr(d)), 0.0, 0.001, INF, o._p(out), mat.ctypes.data)
                assert (hit == 1) == (best < INF)
                if hit:
                    assert float(out[0]) == best
        finally:
            o.scene_destroy(h)
