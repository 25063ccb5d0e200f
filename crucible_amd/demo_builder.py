"""Seeded restatements of Crucible's demo scenes (the benchmark definitions).

Reference: src/demo_builder/demo_images.rs:14-242, demo_movies.rs:12-128.  The
reference builds them from an unseeded rand::rng(); here every draw comes from
a SplitMix64 stream seeded by `scene_seed` (uniform = (u>>11) * 2^-53,
random_range(lo..hi) = lo + (hi-lo)*u), in the reference's draw order, so the
scenes are reproducible.  Scene *parameters* are the reference's.
"""
import math

import numpy as np

from .scene import (LERP, WORLD, CheckerTexture, Dielectric, ImageTexture, Lambertian, Metal, RTWImage, Scene, Sphere)

_MASK = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15


class SceneRng:
    def __init__(self, seed):
        self.s = seed & _MASK

    def u64(self):
        self.s = (self.s + _GAMMA) & _MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def random(self):
        return (self.u64() >> 11) * (2.0 ** -53)

    def random_range(self, lo, hi):
        return lo + (hi - lo) * self.random()


def _clamp01(x):
    return min(max(x, 0.0), 1.0)


def _checker_ground():
    return Lambertian.new_from_texture(CheckerTexture.new_from_color(0.32, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)), 1.0)


def _small_sphere_material(rng, choose_mat):
    """demo_images.rs:58-82"""
    if choose_mat < 0.8:
        c1 = (rng.random(), rng.random(), rng.random())          # Color::random_color()
        c2 = (rng.random(), rng.random(), rng.random())
        albedo = tuple(_clamp01(a * b) for a, b in zip(c1, c2))   # Color * Color (utils.rs:582-596)
        return Lambertian.new_from_color(albedo, 1.0)
    if choose_mat < 0.95:
        albedo = tuple(rng.random_range(0.5, 1.0) for _ in range(3))   # random_color_range(0.5, 1.0)
        fuzz = rng.random_range(0.0, 0.5)
        return Metal.new(albedo, fuzz)
    return Dielectric.new(1.5)


def book1_end_scene(threads=1, scene_seed=1, image_width=400, samples=500):
    """demo_images.rs:14-109 (RTIOW book-1 final scene)."""
    sc = Scene.new_image(16.0 / 9.0, image_width, 24, 180.0, threads)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(50)
    cam.look_from((13.0, 2.0, 3.0))
    cam.look_at((0.0, 0.0, 0.0))
    cam.set_vfov(20.0)
    cam.set_defocus_angle(0.6)
    cam.set_focus_dist(10.0)
    sc.add_element(Sphere.new((0.0, -1000.0, 0.0), 1000.0, _checker_ground()), "ground")
    rng = SceneRng(scene_seed)
    counter = 0
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = rng.random()
            center = (a + 0.9 * rng.random(), 0.2, b + 0.9 * rng.random())
            d = (center[0] - 4.0, center[1] - 0.2, center[2] - 0.0)
            if math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) > 0.9:   # (center - (4,0.2,0)).length()
                sc.add_element(Sphere.new(center, 0.2, _small_sphere_material(rng, choose_mat)), f"small{counter}")
                counter += 1
    sc.add_element(Sphere.new((0.0, 1.0, 0.0), 1.0, Dielectric.new(1.5)), "large_dielectric")
    sc.add_element(Sphere.new((-4.0, 1.0, 0.0), 1.0, Lambertian.new_from_color((0.4, 0.2, 0.1), 1.0)), "large_lambertian")
    sc.add_element(Sphere.new((4.0, 1.0, 0.0), 1.0, Metal.new((0.7, 0.6, 0.5), 0.0)), "large_metal")
    return sc


def checkered_spheres(threads=1, image_width=400, samples=500):
    """demo_images.rs:112-152"""
    sc = Scene.new_image(16.0 / 9.0, image_width, 24, 180.0, threads)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(50)
    cam.look_from((13.0, 2.0, 3.0))
    cam.look_at((0.0, 0.0, 0.0))
    cam.set_vfov(20.0)
    cam.set_defocus_angle(0.6)
    cam.set_focus_dist(10.0)
    checker = CheckerTexture.new_from_color(0.32, (0.2, 0.3, 0.1), (0.9, 0.9, 0.9))
    sc.add_element(Sphere.new((0.0, -10.0, 0.0), 10.0, Lambertian.new_from_texture(checker, 1.0)), "bottom_sphere")
    sc.add_element(Sphere.new((0.0, 10.0, 0.0), 10.0, Lambertian.new_from_texture(checker, 1.0)), "top_sphere")
    return sc


def procedural_sky(width=2048, height=1024, seed=7):
    """Equirectangular RGB8 map standing in for the missing assets/garden.hdr
    (.MISSING_LARGE_BLOBS): vertical gradient, a sun disc and seeded low-frequency
    noise so neighbouring texels differ.  The reference clamps HDR maps to 8-bit at
    load (img_loader.rs:28), so an LDR map exercises the same path."""
    rs = np.random.RandomState(seed)
    v = np.linspace(0.0, 1.0, height)[:, None]           # 0 = top row (sky), 1 = bottom (ground)
    u = np.linspace(0.0, 1.0, width)[None, :]
    top = np.array([0.25, 0.45, 0.85])
    horizon = np.array([0.85, 0.8, 0.7])
    ground = np.array([0.25, 0.3, 0.2])
    t = np.clip(v * 2.0, 0, 1)[..., None]
    b = np.clip(v * 2.0 - 1.0, 0, 1)[..., None]
    img = (1 - t) * top + t * horizon
    img = np.where(v[..., None] > 0.5, (1 - b) * horizon + b * ground, img) * np.ones((1, width, 1))
    sun = np.exp(-(((u - 0.7) * 2.0) ** 2 + ((v - 0.25) ** 2)) * 900.0)[..., None]
    img = img + sun * np.array([1.0, 0.95, 0.8])
    coarse = rs.rand(height // 32 + 1, width // 32 + 1, 3)
    noise = np.kron(coarse, np.ones((32, 32, 1)))[:height, :width] * 0.08
    return RTWImage((np.clip(img + noise, 0, 1) * 255.0).astype(np.uint8))


def load_teapot(threads=1, image_width=400, samples=200, sky=None):
    """demo_images.rs:155-200; `sky` adds the spherical environment map of BASELINE config 3."""
    sc = Scene.new_image(16.0 / 9.0, image_width, 24, 180.0, threads)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(50)
    cam.look_from((13.0, 10.0, 3.0))
    cam.look_at((0.0, 0.0, 0.0))
    cam.set_vfov(20.0)
    cam.set_defocus_angle(0.6)
    cam.set_focus_dist(10.0)
    sc.load_asset("teapot.obj", "teapot", 0.5, (0.0, 0.0, 0.0), Metal.new((0.8, 0.3, 0.5), 0.05))
    sc.add_element(Sphere.new((0.0, -1000.0, 0.0), 1000.0, _checker_ground()), "ground")
    if sky is not None:
        sc.load_spherical_skybox(sky)
    return sc


def scaled_teapot(threads=1, image_width=400, samples=200, sky=None):
    """The teapot with every non-sphere scale builder on it (scene_animator.rs:38-229) plus a translation: what
    demo_movies::moving_teapot (demo_movies.rs:125) is after -- it calls scale_r on the mesh, which the reference's
    own type check rejects.  Keys fall inside the first frames' shutter intervals so a still image shows them."""
    from .scene import LERP, LOCAL, NERP
    sc = load_teapot(threads, image_width, samples, sky)
    sc.scale_x(1.4, 0.012, LERP, "teapot")
    sc.scale_y(0.25, 0.016, NERP, "teapot")
    sc.translate_point((0.0, 0.4, 0.3), 0.02, LERP, LOCAL, "teapot")
    sc.scale_all_uniform(1.2, 0.05, LERP, "teapot")
    sc.scale_z(0.7, 0.09, LERP, "teapot")
    return sc


def teapot_as_list(threads=1, image_width=400, samples=200):
    """Not in the reference's demos: the teapot handed to the scene as ONE HitList element (what
    `scene.add_element(Hittables::HitList(load_obj(..)), ..)` gives, scene/mod.rs:164-166) next to a HitList::new(vec)
    list (empty box), a list inside a list with a hidden object, and ordinary elements."""
    from .scene import BVHWrapper, HitList, load_obj
    sc = Scene.new_image(16.0 / 9.0, image_width, 24, 180.0, threads)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(50)
    cam.look_from((13.0, 10.0, 3.0))
    cam.look_at((0.0, 0.0, 0.0))
    cam.set_vfov(20.0)
    cam.set_defocus_angle(0.6)
    cam.set_focus_dist(10.0)
    sc.add_element(load_obj("teapot.obj", 0.5, (0.0, 0.0, 0.0), Metal.new((0.8, 0.3, 0.5), 0.05)), "teapot")
    sc.add_element(Sphere.new((0.0, -1000.0, 0.0), 1000.0, _checker_ground()), "ground")
    glass, matte = Dielectric.new(1.5), Lambertian.new_from_color((0.2, 0.4, 0.8), 1.0)
    sc.add_element(HitList.new([Sphere.new((2.5, 0.5, 2.0), 0.5, glass), Sphere.new((3.4, 0.3, 1.2), 0.3, matte)]), "loose")
    inner = HitList.default()
    inner.add(Sphere.new((-2.0, 0.4, 2.5), 0.4, matte))
    hidden = Sphere.new((-2.0, 1.2, 2.5), 0.4, glass)
    hidden.hide = True
    inner.add(hidden)
    outer = HitList.default()
    outer.add(Sphere.new((-3.0, 0.5, 1.5), 0.5, Metal.new((0.8, 0.8, 0.8), 0.0)))
    outer.add(inner)
    sc.add_element(outer, "outer")
    sc.add_element(HitList.default(), "nothing")
    # ... and a pre-built BVHWrapper as an element (scene/mod.rs:161-163)
    sc.add_element(BVHWrapper.new_wrapper(HitList.new([Sphere.new((1.5, 0.3, 3.0), 0.3, matte), Sphere.new((0.6, 0.3, 3.4), 0.3, glass),
                                                       Sphere.new((-0.4, 0.3, 3.6), 0.3, matte)])), "wrapped")
    return sc


def earth(threads=1, image_width=400, samples=500, image=None):
    """demo_images.rs:202-221; `image` replaces earthmap.jpg with an in-memory RTWImage."""
    sc = Scene.new_image(16.0 / 9.0, image_width, 24, 180.0, threads)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(50)
    cam.look_from((0.0, 0.0, 12.0))
    cam.look_at((0.0, 0.0, 0.0))
    cam.set_vfov(20.0)
    tex = ImageTexture(image if image is not None else "earthmap.jpg")
    sc.add_element(Sphere.new((0.0, 0.0, 0.0), 2.0, Lambertian.new_from_texture(tex, 1.0)), "earth")
    return sc


def garden_skybox(threads=1, image_width=1920, samples=500, sky=None):
    """demo_images.rs:223-242 with the procedural map in place of garden.hdr."""
    sc = Scene.new_image(16.0 / 9.0, image_width, 24, 180.0, threads)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(50)
    cam.look_from((0.0, 0.0, -12.0))
    cam.look_at((0.0, 0.0, 0.0))
    cam.set_vfov(40.0)
    sc.add_element(Sphere.new((0.0, 0.0, 0.0), 2.0, Metal.new((0.8, 0.8, 0.8), 0.05)), "metal_ball")
    sc.load_spherical_skybox(sky if sky is not None else procedural_sky())
    return sc


def _splitmix_stream(seed, n):
    """First n outputs of the SceneRng stream as float64 uniforms (vectorised: the stream is counter-based)."""
    k = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _MASK) + k * np.uint64(_GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


class _PrebuiltScene(Scene):
    """A Scene whose flat description was generated directly (a million Python Sphere objects buy nothing)."""

    def flatten(self):
        self._flat.desc.bvh_mode = self.bvh_mode
        return self._flat


def million_spheres(threads=1, scene_seed=1, half_extent=500, image_width=3840, samples=256):
    """BASELINE config 4 (no reference scene exists: the book1 recipe widened):
    (2*half_extent)^2 small spheres of radius 0.2 at (a + 0.9U, 0.2, b + 0.9U), a,b in [-h, h), with book1's
    80/15/5 material mix and draw order (demo_images.rs:48-82) and no exclusion zone, on the r=1000 checker ground.
    Generated straight into the flat C-ABI arrays; identical to building it sphere by sphere with SceneRng."""
    from . import _abi as A
    from .scene import FlatScene
    sc = _PrebuiltScene.new_image(16.0 / 9.0, image_width, 24, 180.0, threads)
    cam = sc.scene_cam
    cam.set_samples(samples)
    cam.set_max_depth(50)
    cam.look_from((13.0, 6.0, 3.0))
    cam.look_at((0.0, 0.0, 0.0))
    cam.set_vfov(40.0)
    n = (2 * half_extent) ** 2
    U = _splitmix_stream(scene_seed, 9 * n)
    # walk the stream: 3 draws per sphere + 6 (lambertian) | 4 (metal) | 0 (glass)
    off = np.empty(n, dtype=np.int64)
    o = 0
    ul = U.tolist()
    for i in range(n):
        off[i] = o
        c = ul[o]
        o += 9 if c < 0.8 else (7 if c < 0.95 else 3)
    choose = U[off]
    aa, bb = np.divmod(np.arange(n), 2 * half_extent)
    cx = (aa - half_extent) + 0.9 * U[off + 1]
    cz = (bb - half_extent) + 0.9 * U[off + 2]
    lam, met = choose < 0.8, (choose >= 0.8) & (choose < 0.95)
    prims = np.zeros(n + 1, dtype=np.dtype([("kind", "<i4"), ("material", "<i4"), ("flags", "<i4"), ("key_first", "<i4"),
                                            ("key_count", "<i4"), ("_pad", "<i4"), ("v", "<f8", 9)]))
    prims["v"][0, :4] = (0.0, -1000.0, 0.0, 1000.0)
    prims["v"][1:, 0], prims["v"][1:, 1], prims["v"][1:, 2], prims["v"][1:, 3] = cx, 0.2, cz, 0.2
    prims["material"] = np.arange(n + 1)
    mats = np.zeros(n + 1, dtype=np.dtype([("kind", "<i4"), ("texture", "<i4"), ("albedo", "<f8", 3), ("param", "<f8")]))
    texs = np.zeros(3 + int(lam.sum()), dtype=np.dtype([("kind", "<i4"), ("even", "<i4"), ("odd", "<i4"), ("image", "<i4"),
                                                        ("color", "<f8", 3), ("inv_scale", "<f8")]))
    # textures 0,1: the checker's solids; 2: the checker; then one solid per lambertian sphere
    texs["even"], texs["odd"], texs["image"] = -1, -1, -1
    texs["color"][0], texs["color"][1] = (0.2, 0.3, 0.1), (0.9, 0.9, 0.9)
    texs["kind"][2], texs["even"][2], texs["odd"][2], texs["inv_scale"][2] = A.CR_TEX_CHECKER, 0, 1, 1.0 / 0.32
    mats["kind"][0], mats["texture"][0], mats["param"][0] = A.CR_MAT_LAMBERTIAN, 2, 1.0
    li = np.nonzero(lam)[0]
    c1 = np.stack([U[off[li] + 3 + k] for k in range(3)], axis=1)
    c2 = np.stack([U[off[li] + 6 + k] for k in range(3)], axis=1)
    texs["color"][3:] = np.clip(c1 * c2, 0.0, 1.0)
    mats["kind"][1 + li], mats["texture"][1 + li], mats["param"][1 + li] = A.CR_MAT_LAMBERTIAN, 3 + np.arange(len(li)), 1.0
    mi = np.nonzero(met)[0]
    mats["kind"][1 + mi], mats["texture"][1 + mi] = A.CR_MAT_METAL, -1
    mats["albedo"][1 + mi] = np.stack([0.5 + (1.0 - 0.5) * U[off[mi] + 3 + k] for k in range(3)], axis=1)
    mats["param"][1 + mi] = 0.0 + (0.5 - 0.0) * U[off[mi] + 6]
    gi = np.nonzero(~lam & ~met)[0]
    mats["kind"][1 + gi], mats["texture"][1 + gi], mats["param"][1 + gi] = A.CR_MAT_DIELECTRIC, -1, 1.5
    mats["albedo"][1 + gi] = 1.0
    flat = FlatScene.__new__(FlatScene)
    flat._np = (prims, mats, texs)
    flat.prims = (A.CrPrimitive * len(prims)).from_buffer(prims)
    flat.materials = (A.CrMaterial * len(mats)).from_buffer(mats)
    flat.textures = (A.CrTexture * len(texs)).from_buffer(texs)
    flat.images = (A.CrImage * 1)()
    flat.keys = (A.CrKeyframe * 1)()
    flat._image_arrays = []
    flat.desc = A.CrSceneDesc(len(prims), len(mats), len(texs), 0, 0, A.CR_SKY_DEFAULT, -1, 0, flat.prims, flat.materials,
                              flat.textures, flat.images, flat.keys)
    sc._flat = flat
    sc.elements = range(n + 1)   # only len() is meaningful
    return sc


def teapot_orbit_movie(threads=1, image_width=1920, samples=512, frame_rate=24, duration=10.0, radius=13.0, sky=None):
    """BASELINE config 5: the teapot scene with the camera walking first_movie's square
    (demo_movies.rs:33-60) scaled to `radius`, at = origin."""
    sc = load_teapot(threads, image_width, samples, sky if sky is not None else procedural_sky())
    sc.duration = duration
    sc.frame_rate = frame_rate
    sc.scene_cam.frame_rate = float(frame_rate)
    sc.scene_cam.look_from((0.0, 10.0, -radius))
    q = duration / 4.0
    for k, p in enumerate([(radius, 10.0, 0.0), (0.0, 10.0, radius), (-radius, 10.0, 0.0), (0.0, 10.0, -radius)]):
        sc.cam_translate_point(p, q * (k + 1), LERP, WORLD, "from")
    return sc
