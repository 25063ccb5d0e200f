"""Host-side mirror of Crucible's Scene / Camera builder API, flattened to the C ABI.

Names, argument meaning and error behaviour follow the reference:
  Scene        src/scene/mod.rs:75-347, src/scene/scene_animator.rs
  Camera       src/camera/mod.rs:66-263
  Sphere       src/objects/sphere.rs:25-39      Triangle  src/objects/triangle.rs:23-46
  Lambertian / Metal / Dielectric   src/materials/{lambertian,metal,dielectric}.rs
  SolidColor / CheckerTexture / ImageTexture   src/textures/*.rs
  load_obj     src/asset_loader/obj_loader.rs:21-143
Where the reference panics this raises; where it returns io::Error the C ABI
returns CR_ERR_IO.  Nothing here computes pixels: `Scene.flatten()` produces the
CrSceneDesc / CrCameraDesc the library (or, in tests, the oracle) consumes, and
`Scene.render_scene()` is Camera::render over the HIP library + the PPM writer.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _abi as A
from .timeline import LERP, LOCAL, NERP, WORLD, TransformTimeline  # noqa: F401


def _color(c):
    r, g, b = (float(x) for x in c)
    # Color::new asserts 0 <= c <= 1 (utils.rs:345-350)
    for name, v in (("R", r), ("G", g), ("B", b)):
        if not (0.0 <= v <= 1.0):
            raise ValueError(f"{name} must be between 0.0 and 1.0. Got {v}")
    return (r, g, b)


# ------------------------------------------------------------------ textures
class SolidColor:
    def __init__(self, albedo):
        self.albedo = _color(albedo)

    new_from_color = classmethod(lambda cls, c: cls(c))
    new_from_rgb = classmethod(lambda cls, r, g, b: cls((r, g, b)))


class CheckerTexture:
    def __init__(self, scale, even, odd):
        self.inv_scale = 1.0 / float(scale)   # checker_texture.rs:23,31
        self.even, self.odd = even, odd

    @classmethod
    def new_from_textures(cls, scale, even, odd):
        return cls(scale, even, odd)

    @classmethod
    def new_from_color(cls, scale, c1, c2):
        return cls(scale, SolidColor(c1), SolidColor(c2))


class RTWImage:
    """Decoded RGB8 image (img_loader.rs:17-55).  `rgb8` is HxWx3 uint8."""

    def __init__(self, rgb8):
        arr = np.ascontiguousarray(rgb8, dtype=np.uint8)
        assert arr.ndim == 3 and arr.shape[2] == 3
        self.rgb8 = arr

    @classmethod
    def new(cls, filename):
        path = build_asset_path(filename)
        if path.lower().endswith(".hdr"):
            return cls(decode_radiance(open(path, "rb").read()))
        from PIL import Image   # the reference uses the `image` crate; decoders may differ by an LSB
        with Image.open(path) as im:
            return cls(np.asarray(im.convert("RGB")))

    def width(self):
        return self.rgb8.shape[1]

    def height(self):
        return self.rgb8.shape[0]


def decode_radiance(data):
    """See _decode_radiance; a file that ends early is a ValueError like every other malformed input."""
    try:
        return _decode_radiance(data)
    except IndexError:
        raise ValueError("truncated Radiance data") from None


def _decode_radiance(data):
    """Radiance .hdr (RGBE) -> HxWx3 uint8 the way the reference keeps images (`image` crate
    decode, then `to_rgb8()`, img_loader.rs:24-28): float = mantissa * 2^(e - 136) (0 when e == 0), then
    round(clamp(x, 0, 1) * 255).  The crate is not vendored: parity unpinned.  Flat, new-style RLE and old-style
    repeat-marker scanlines; `-Y H +X W` and `+Y H +X W`.  Same decoder as RTWImage::load_hdr in host/crucible.hpp."""
    pos = 0

    def line():
        nonlocal pos
        end = data.find(b"\n", pos)
        if end < 0:
            raise ValueError("truncated Radiance header")
        out = data[pos:end]
        pos = end + 1
        return out

    if not line().startswith(b"#?"):
        raise ValueError("not a Radiance file")
    while line():
        pass
    res = line().split()
    if len(res) != 4 or res[0] not in (b"-Y", b"+Y") or res[2] != b"+X":
        raise ValueError(f"unsupported Radiance resolution line: {res}")
    try:
        H, W = int(res[1]), int(res[3])
    except ValueError:
        raise ValueError(f"unsupported Radiance resolution line: {res}") from None
    if H < 1 or W < 1:
        raise ValueError(f"unsupported Radiance resolution line: {res}")
    if W * H > len(data) * 128:   # every pixel takes at least one byte of the file (a run covers 127 at most)
        raise ValueError("truncated Radiance data")
    data = data + bytes(8)         # reads past the end see zeros and fail the run checks below instead of raising IndexError
    buf = np.frombuffer(data, dtype=np.uint8)
    n_real = len(data) - 8
    rgbe = np.zeros((H, W, 4), dtype=np.uint8)
    for row in range(H):
        if 8 <= W < 32768 and data[pos] == 2 and data[pos + 1] == 2 and ((data[pos + 2] << 8) | data[pos + 3]) == W:
            pos += 4
            for c in range(4):
                x = 0
                while x < W:
                    n = data[pos]
                    pos += 1
                    if n > 128:
                        n -= 128
                        if x + n > W:
                            raise ValueError("bad Radiance run")
                        rgbe[row, x:x + n, c] = data[pos]
                        pos += 1
                    else:
                        if n == 0 or x + n > W:
                            raise ValueError("bad Radiance run")
                        rgbe[row, x:x + n, c] = buf[pos:pos + n]
                        pos += n
                    x += n
        else:
            x, shift = 0, 0
            while x < W:
                q = buf[pos:pos + 4]
                pos += 4
                if q[0] == 1 and q[1] == 1 and q[2] == 1 and x > 0:
                    n = int(q[3]) << shift
                    if x + n > W:
                        raise ValueError("bad Radiance run")
                    rgbe[row, x:x + n] = rgbe[row, x - 1]
                    x += n
                    shift += 8
                else:
                    rgbe[row, x] = q
                    x += 1
                    shift = 0
    if pos > n_real:
        raise ValueError("truncated Radiance data")
    if res[0] == b"+Y":
        rgbe = rgbe[::-1]
    e = rgbe[..., 3].astype(np.int32)
    f = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(np.float32)
    v = np.clip(rgbe[..., :3].astype(np.float32) * f[..., None], np.float32(0.0), np.float32(1.0))
    x = v * np.float32(255.0)
    r = np.floor(x)
    return (r + (x - r >= np.float32(0.5))).astype(np.uint8)   # f32::round: halves away from zero


class ImageTexture:
    def __init__(self, image):
        self.image = image if isinstance(image, RTWImage) else RTWImage.new(image)

    new = classmethod(lambda cls, filename: cls(filename))


# ------------------------------------------------------------------ materials
class Lambertian:
    def __init__(self, tex, prob):
        self.tex, self.scatter_prob = tex, float(prob)

    @classmethod
    def new_from_color(cls, c, prob):
        return cls(SolidColor(c), prob)

    @classmethod
    def new_from_texture(cls, tex, prob):
        return cls(tex, prob)


class Metal:
    def __init__(self, c, fuzz):
        # metal.rs:21-23
        if fuzz > 1.0:
            raise ValueError("A metal cannot have a fuzz factor above 1.0")
        if fuzz < 0.0:
            raise ValueError("A metal cannot have a fuzz factor below 0.0")
        self.albedo, self.fuzz = _color(c), float(fuzz)

    new = classmethod(lambda cls, c, fuzz: cls(c, fuzz))


class Dielectric:
    def __init__(self, refraction_index):
        self.refraction_index = float(refraction_index)

    new = classmethod(lambda cls, ri: cls(ri))


# ------------------------------------------------------------------ objects
class Sphere:
    def __init__(self, center, radius, mat):
        if not radius >= 0.0:
            raise ValueError("Cannot make a sphere with negative radius")   # sphere.rs:26
        self.id, self.hide = 0, False
        self.center, self.radius, self.mat = tuple(float(c) for c in center), float(radius), mat
        self.timeline = TransformTimeline.new_sphere(self.center, self.radius)

    new = classmethod(lambda cls, c, r, m: cls(c, r, m))


class Triangle:
    def __init__(self, a, b, c, mat):
        self.id, self.hide = 0, False
        self.a, self.b, self.c = (tuple(float(x) for x in p) for p in (a, b, c))
        self.mat = mat
        # a_/b_/c_timeline only ever receive the same calls (scene_animator.rs); one is kept
        self.timeline = TransformTimeline(self.a)

    new = classmethod(lambda cls, a, b, c, m: cls(a, b, c, m))


class HitList:
    """objects/hitlist.rs.  As a scene element (scene/mod.rs:164-166) the BVH build treats it as one object whose
    box is whatever `add` accumulated -- `new(objs)` leaves Aabb::default() (:13-18), `clear` keeps the old box
    (:20-22) -- and a leaf wrapper holding it scans every object in order (:51-65).  A list inside a list is passed
    as its objects spliced in place (the inner box is never read by a hit), which needs the outer box to be either
    untouched by `add` or the union of everything spliced; other mixtures raise at flatten()."""

    def __init__(self, objs=None):
        self.id, self.hide, self.timeline = -1, False, None
        self.objs = list(objs or [])
        self._boxed = []          # what add() has folded into the box
        self._stale_box = False   # clear() after add(): the box no longer describes the objects

    new = classmethod(lambda cls, objs: cls(objs))
    default = classmethod(lambda cls: cls())

    def clear(self):
        self._stale_box = self._stale_box or bool(self._boxed)
        self.objs = []

    def add(self, obj):
        self.objs.append(obj)
        self._boxed.append(obj)

    def get_objs(self):
        return self.objs

    def __iter__(self):
        return iter(self.objs)

    def __len__(self):
        return len(self.objs)

    def __getitem__(self, i):
        return self.objs[i]

    def _box_is_union(self):
        """The box equals the union of the boxes of every sphere/triangle under this list."""
        if self._stale_box or len(self._boxed) != len(self.objs) or any(a is not b for a, b in zip(self._boxed, self.objs)):
            return False
        return all(o._box_is_union() for o in self.objs if isinstance(o, HitList))

    def _box_is_default(self):
        return not self._boxed and not self._stale_box

    def spliced(self):
        """(objects in visiting order, empty_box flag) -- or ValueError when the box is neither form."""
        out = []

        def walk(l):
            for o in l.objs:
                if isinstance(o, HitList):
                    walk(o)
                elif isinstance(o, (Sphere, Triangle)):
                    out.append(o)
                else:
                    raise ValueError("a HitList element may hold spheres, triangles and lists (a BVHWrapper is not representable)")
        walk(self)
        if self._box_is_default():
            return out, True
        if self._box_is_union():
            return out, False
        # a box folded from only some of the objects, or from inner lists whose own boxes are not their union
        raise ValueError("this HitList's box is neither Aabb::default() nor the union of its objects: not representable")


class BVHWrapper:
    """objects/bvhwrapper.rs as a scene element (scene/mod.rs:161-163): `BVHWrapper.new_wrapper(list)` keeps the list's
    visible spheres and triangles (bvhwrapper.rs:16-26) -- the library rebuilds the tree from them with the reference's
    own algorithm -- and answers an empty HitList when there is none (:28-30).  Lists or wrappers inside the list are
    not representable."""

    def __init__(self, objs):
        self.id, self.hide, self.timeline = -1, False, None
        self.objs = objs

    @classmethod
    def new_wrapper(cls, lst):
        objs = []
        for o in (lst.get_objs() if isinstance(lst, HitList) else lst):
            if not isinstance(o, (Sphere, Triangle)):
                raise ValueError("a BVHWrapper element may hold spheres and triangles only")
            if not o.hide:
                objs.append(o)
        return cls(objs) if objs else HitList.default()


def build_asset_path(asset_filename):
    """asset_loader/mod.rs:6-41: $ASSET_DIR is prepended verbatim; else `assets/` up to six levels up."""
    folder = os.environ.get("ASSET_DIR")
    if folder is not None:
        return folder + asset_filename
    here = os.path.dirname(os.path.abspath(__file__))
    for base in ("", "..", "../..", "../../..", "../../../..", "../../../../..", "../../../../../.."):
        p = os.path.join(base, "assets", asset_filename)
        if os.path.exists(p):
            return p
    p = os.path.join(here, "..", "assets", asset_filename)   # repo checkout
    if os.path.exists(p):
        return p
    raise FileNotFoundError("Could not find the asset " + asset_filename)


def load_obj(file, scale, shift, mat, strict=True):
    """obj_loader.rs:21-143: `v`/`f` lines only, 1-based, triangles only, scale*p + shift.
    strict=False (not in the reference) additionally tolerates comments, `vn`/`vt`/`o`/`g`/`s`/
    `usemtl`/`mtllib` lines, `v/vt/vn` face corners, negative indices and polygons (fan-triangulated)."""
    path = build_asset_path(file)
    if not path.endswith(".obj"):
        raise ValueError("Expected an obj file.")
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            parts = line.split()
            if not parts:
                continue
            if parts[0] == "v":
                if len(parts) != 4 and (strict or len(parts) < 4):
                    raise ValueError("Invalid number of coordinates for a vertex")
                verts.append(tuple(float(x) for x in parts[1:4]))
            elif parts[0] == "f":
                if strict:
                    if len(parts) != 4:
                        raise ValueError("The asset loader only supports triangularized images")
                    faces.append(tuple(int(x) for x in parts[1:]))
                else:
                    idx = [int(c.split("/")[0]) for c in parts[1:]]
                    idx = [i if i > 0 else len(verts) + 1 + i for i in idx]
                    if len(idx) < 3:
                        raise ValueError("a face needs at least three corners")
                    for k in range(1, len(idx) - 1):
                        faces.append((idx[0], idx[k], idx[k + 1]))
            elif strict or not (parts[0].startswith("#") or parts[0] in ("vn", "vt", "vp", "o", "g", "s", "usemtl", "mtllib", "l")):
                raise ValueError("Unsupported OBJ file")
    verts = [tuple(scale * p[k] + shift[k] for k in range(3)) for p in verts]
    model = HitList.default()   # obj_loader.rs:22,137: one add() per face
    for f in faces:
        model.add(Triangle(verts[f[0] - 1], verts[f[1] - 1], verts[f[2] - 1], mat))
    return model


# ------------------------------------------------------------------ camera
class Camera:
    """Camera::new + setters (camera/mod.rs:106-263)."""

    def __init__(self, aspect_ratio, image_width, frame_rate, shutter_angle, thread_count):
        self.aspect_ratio = float(aspect_ratio)
        self.image_width = int(image_width)
        self.image_height = max(1, int(self.image_width / self.aspect_ratio))   # Viewport::new :37-38
        self.vfov_degrees = 90.0
        self.look_from_tl = TransformTimeline((0.0, 0.0, 0.0))
        self.look_at_tl = TransformTimeline((0.0, 0.0, 0.0))
        self.vup = (0.0, 1.0, 0.0)
        self.defocus_angle_degrees = 0.0
        self.focus_dist = 10.0
        self.samples, self.max_depth = 10, 10
        self.thread_count = thread_count
        self.frame_rate, self.frame, self.shutter_angle = float(frame_rate), 0, float(shutter_angle)
        self.refit_boxes = False   # True: CrRenderParams.refit_boxes (wrapper boxes follow keyframed primitives)

    def next_frame(self):
        self.frame += 1

    def look_from(self, loc):
        self.look_from_tl = TransformTimeline(loc)

    def look_at(self, loc):
        self.look_at_tl = TransformTimeline(loc)

    def set_vup(self, vup):
        self.vup = tuple(float(x) for x in vup)

    def set_vfov(self, vfov_degrees):
        self.vfov_degrees = float(vfov_degrees)

    def set_hfov(self, hfov_degrees):   # camera/mod.rs:220-228
        hfov = hfov_degrees * math.pi / 180.0
        self.set_vfov((math.atan(math.tan(hfov / (2.0 * self.aspect_ratio))) * 2.0) * 180.0 / math.pi)

    def set_samples(self, s):
        if not s > 0:
            raise ValueError(f"The camera must have a positive number of samples. {s} is invalid.")
        self.samples = int(s)

    def set_max_depth(self, md):
        self.max_depth = int(md)

    def set_defocus_angle(self, da_degree):
        self.defocus_angle_degrees = float(da_degree)

    def set_focus_dist(self, fd):
        self.focus_dist = float(fd)

    def set_threads(self, threads):
        self.thread_count = threads

    def desc(self):
        fk, ak = self.look_from_tl.keyframes(), self.look_at_tl.keyframes()
        fa = (A.CrKeyframe * max(1, len(fk)))(*fk)
        aa = (A.CrKeyframe * max(1, len(ak)))(*ak)
        d = A.CrCameraDesc(self.image_width, self.image_height, self.vfov_degrees, self.defocus_angle_degrees,
                           self.focus_dist, (C.c_double * 3)(*self.look_from_tl.start_pos),
                           (C.c_double * 3)(*self.look_at_tl.start_pos), (C.c_double * 3)(*self.vup),
                           len(fk), len(ak), fa, aa)
        d._keep = (fa, aa)
        return d

    def params(self, seed, real_type, sample_begin=0, sample_count=None, output_sum=False, sum_order=A.CR_SUM_DEFAULT):
        n = self.samples if sample_count is None else sample_count
        return A.CrRenderParams(self.samples, sample_begin, n, self.max_depth, seed, self.frame, real_type,
                                self.frame_rate, self.shutter_angle, 1 if output_sum else 0, 1 if self.refit_boxes else 0,
                                sum_order, 0)


# ------------------------------------------------------------------ scene
class FlatScene:
    """CrSceneDesc plus the buffers it points into."""

    def __init__(self, prims, materials, textures, images, keys, sky_kind, sky_image):
        self.prims = (A.CrPrimitive * max(1, len(prims)))(*prims)
        self.materials = (A.CrMaterial * max(1, len(materials)))(*materials)
        self.textures = (A.CrTexture * max(1, len(textures)))(*textures)
        self._image_arrays = images
        imgs = [A.CrImage(im.shape[1], im.shape[0], im.ctypes.data_as(C.POINTER(C.c_uint8))) for im in images]
        self.images = (A.CrImage * max(1, len(imgs)))(*imgs)
        self.keys = (A.CrKeyframe * max(1, len(keys)))(*keys)
        self.desc = A.CrSceneDesc(len(prims), len(materials), len(textures), len(images), len(keys), sky_kind,
                                  sky_image, 0, self.prims, self.materials, self.textures, self.images, self.keys)


class Scene:
    def __init__(self, aspect_ratio, image_width, frame_rate, shutter_angle, thread_count, duration=None):
        self.scene_cam = Camera(aspect_ratio, image_width, float(frame_rate), shutter_angle, thread_count)
        self.elements = []
        self.skybox = None          # Skybox::Default
        self._aliases = {"cam": (0, "Camera")}   # id_vendor.rs: "cam" reserved as id 0
        self._next_id = 1
        self.duration = duration
        self.frame_rate = int(frame_rate)
        self.seed = 0xC0FFEE
        self.real_type = A.CR_REAL_F32
        self.device = 0
        self.bvh_mode = A.CR_BVH_REFERENCE   # A.CR_BVH_SAH: the quality builder (include/crucible_hip.h)

    @classmethod
    def new_image(cls, aspect_ratio, image_width, frame_rate, shutter_angle, thread_count):
        return cls(aspect_ratio, image_width, frame_rate, shutter_angle, thread_count)

    @classmethod
    def new_movie(cls, aspect_ratio, image_width, frame_rate, shutter_angle, thread_count, duration):
        return cls(aspect_ratio, image_width, frame_rate, shutter_angle, thread_count, duration)

    def _vend_id(self, alias, otype):
        if alias in self._aliases:
            raise ValueError(f"This {otype}'s alias collides with another name in the scene! Try changing {alias} to a new name.")
        self._aliases[alias] = (self._next_id, otype)
        self._next_id += 1
        return self._aliases[alias][0]

    def load_default_skybox(self):
        self.skybox = None

    def load_spherical_skybox(self, file):
        self.skybox = file if isinstance(file, RTWImage) else RTWImage.new(file)

    def add_element(self, element, alias):
        if isinstance(element, (HitList, BVHWrapper)):   # scene/mod.rs:161-166: added as it is, the alias is not registered
            self.elements.append(element)
            return
        element.id = self._vend_id(alias, "Sphere" if isinstance(element, Sphere) else "Triangle")
        self.elements.append(element)

    def load_asset(self, asset_path, alias, scale, shift, mat, strict=True):
        mesh_id = self._vend_id(alias, "TriangleMesh")
        for t in load_obj(asset_path, scale, shift, mat, strict=strict):
            t.id = mesh_id
            self.elements.append(t)

    def _lookup(self, alias, invalid=()):
        if alias not in self._aliases:
            raise KeyError(f"Could not find an object with the alias: `{alias}`. Are you sure you spelled it right?")
        oid, otype = self._aliases[alias]
        if otype in invalid:
            raise ValueError(f"this transform cannot apply to a {otype}")
        return oid

    def show_element(self, alias):
        self._set_visibility(alias, False)

    def hide_element(self, alias):
        self._set_visibility(alias, True)

    def _set_visibility(self, alias, hide):
        if alias not in self._aliases:
            print(f"WARNING: The element `{alias}` does not exist. Are you sure you typed the right name?")
            return
        oid = self._aliases[alias][0]
        for e in self.elements:
            if e.id == oid:
                e.hide = hide

    # scene_animator.rs
    def translate_point(self, p, keyframe, it, space, alias):
        oid = self._lookup(alias, invalid=("Camera",))
        for e in self.elements:
            if e.id == oid:
                e.timeline.translate_point(p, keyframe, it, space)

    def scale_r(self, r, keyframe, it, alias):
        oid = self._lookup(alias, invalid=("Camera", "TriangleMesh", "Triangle"))   # scene_animator.rs:140-150
        for e in self.elements:
            if e.id == oid:
                e.timeline.scale_sphere(r, keyframe, it)

    def _scale_non_sphere(self, method, value, keyframe, it, alias, what):
        # scene_animator.rs:38-229: ScaleX/Y/Z/All type-check against Sphere only and touch Triangles
        oid = self._lookup(alias, invalid=("Sphere",))
        for e in self.elements:
            if e.id == oid and isinstance(e, Triangle):
                getattr(e.timeline, method)(value, keyframe, it)

    def scale_x(self, x, keyframe, it, alias):
        self._scale_non_sphere("scale_x", x, keyframe, it, alias, "ScaleX")

    def scale_y(self, y, keyframe, it, alias):
        self._scale_non_sphere("scale_y", y, keyframe, it, alias, "ScaleY")

    def scale_z(self, z, keyframe, it, alias):
        self._scale_non_sphere("scale_z", z, keyframe, it, alias, "ScaleZ")

    def scale_point(self, p, keyframe, it, alias):
        self._scale_non_sphere("scale_point", tuple(float(c) for c in p), keyframe, it, alias, "ScaleAll")

    def scale_all_uniform(self, v, keyframe, it, alias):   # scene_animator.rs:217-219
        self.scale_point((v, v, v), keyframe, it, alias)

    def cam_translate_point(self, p, keyframe, it, space, which):
        tl = self.scene_cam.look_from_tl if which == "from" else self.scene_cam.look_at_tl
        tl.translate_point(p, keyframe, it, space)

    # ---- flatten to the C ABI
    def flatten(self):
        materials, textures, images, keys, prims = [], [], [], [], []
        tex_ids, mat_ids, img_ids = {}, {}, {}

        def image_id(img):
            if id(img) not in img_ids:
                img_ids[id(img)] = len(images)
                images.append(img.rgb8)
            return img_ids[id(img)]

        def texture_id(t):
            if id(t) in tex_ids:
                return tex_ids[id(t)]
            if isinstance(t, SolidColor):
                rec = A.CrTexture(A.CR_TEX_SOLID, -1, -1, -1, (C.c_double * 3)(*t.albedo), 0.0)
            elif isinstance(t, CheckerTexture):
                e, o = texture_id(t.even), texture_id(t.odd)
                rec = A.CrTexture(A.CR_TEX_CHECKER, e, o, -1, (C.c_double * 3)(0, 0, 0), t.inv_scale)
            else:
                rec = A.CrTexture(A.CR_TEX_IMAGE, -1, -1, image_id(t.image), (C.c_double * 3)(0, 0, 0), 0.0)
            tex_ids[id(t)] = len(textures)
            textures.append(rec)
            return tex_ids[id(t)]

        def material_id(m):
            if id(m) in mat_ids:
                return mat_ids[id(m)]
            if isinstance(m, Lambertian):
                rec = A.CrMaterial(A.CR_MAT_LAMBERTIAN, texture_id(m.tex), (C.c_double * 3)(0, 0, 0), m.scatter_prob)
            elif isinstance(m, Metal):
                rec = A.CrMaterial(A.CR_MAT_METAL, -1, (C.c_double * 3)(*m.albedo), m.fuzz)
            else:
                rec = A.CrMaterial(A.CR_MAT_DIELECTRIC, -1, (C.c_double * 3)(1, 1, 1), m.refraction_index)
            mat_ids[id(m)] = len(materials)
            materials.append(rec)
            return mat_ids[id(m)]

        def emit(e, extra_flags=0):
            ks = e.timeline.keyframes()
            v = (C.c_double * 9)()
            if isinstance(e, Sphere):
                v[0:4] = [*e.center, e.radius]
                kind = A.CR_PRIM_SPHERE
            else:
                v[0:9] = [*e.a, *e.b, *e.c]
                kind = A.CR_PRIM_TRIANGLE
            prims.append(A.CrPrimitive(kind, material_id(e.mat), (A.CR_PRIM_HIDDEN if e.hide else 0) | extra_flags,
                                       len(keys), len(ks), 0, v))
            keys.extend(ks)

        for e in self.elements:
            if isinstance(e, BVHWrapper):   # the wrapper record, then its objects (include/crucible_hip.h CR_PRIM_BVH)
                v = (C.c_double * 9)()
                v[0:2] = [len(prims) + 1, len(e.objs)]
                prims.append(A.CrPrimitive(A.CR_PRIM_BVH, 0, 0, 0, 0, 0, v))
                for o in e.objs:
                    emit(o, A.CR_PRIM_MEMBER)
            elif isinstance(e, HitList):   # the list record, then its objects (include/crucible_hip.h CR_PRIM_LIST)
                objs, empty_box = e.spliced()
                v = (C.c_double * 9)()
                v[0:2] = [len(prims) + 1, len(objs)]
                prims.append(A.CrPrimitive(A.CR_PRIM_LIST, 0, A.CR_LIST_EMPTY_BOX if empty_box else 0, 0, 0, 0, v))
                for o in objs:
                    emit(o, A.CR_PRIM_MEMBER)
            else:
                emit(e)
        sky_kind, sky_image = A.CR_SKY_DEFAULT, -1
        if self.skybox is not None:
            sky_kind, sky_image = A.CR_SKY_SPHERICAL, image_id(self.skybox)
        flat = FlatScene(prims, materials, textures, images, keys, sky_kind, sky_image)
        flat.desc.bvh_mode = self.bvh_mode
        return flat

    # ---- Camera::render over the HIP library (scene/mod.rs:283-347)
    def compute_frame_count(self):
        return int(math.ceil(self.duration * self.frame_rate))   # scene/mod.rs:324-330

    def render_scene(self, fname):
        if self.duration is not None:
            return self.render_movie(fname)
        return self.render_image(fname)

    def render_image(self, fname, renderer=None):
        from .renderer import Renderer
        own = renderer is None
        r = renderer or Renderer(self.device)
        try:
            r.upload_scene(self.flatten())
            img, stats = r.render(self.scene_cam, seed=self.seed, real_type=self.real_type)
            r.write_ppm(fname + ".ppm", img)
            print(f"Successful render! Image stored at: {fname}.ppm")
            return stats
        finally:
            if own:
                r.close()

    def mp4_command(self, fname, padding, ext=".ppm"):
        """movie_maker::make_mp4's ffmpeg argument vector (scene/movie_maker.rs:6-33).  The reference runs it after
        the last frame; here it is returned to the caller (a process that holds the GPU does not exec others)."""
        return ["ffmpeg", "-framerate", str(self.frame_rate), "-i", f"{fname}/artifacts/image%0{padding}d{ext}",
                "-vf", "scale=trunc(iw/2)*2:trunc(ih/2)*2", "-c:v", "libx264", "-pix_fmt", "yuv420p", "-crf", "25",
                f"{fname}/movie.mp4"]

    def render_movie(self, fname):
        from .renderer import Renderer
        os.mkdir(fname)   # scene/mod.rs:296: fails if it exists
        os.mkdir(os.path.join(fname, "artifacts"))
        frames = self.compute_frame_count()
        digits = len(str(frames))
        r = Renderer(self.device)
        try:
            r.upload_scene(self.flatten())
            for frame in range(frames):
                img, _ = r.render(self.scene_cam, seed=self.seed, real_type=self.real_type)
                r.write_ppm(os.path.join(fname, "artifacts", f"image{frame:0{digits}d}.ppm"), img)
                self.scene_cam.next_frame()
        finally:
            r.close()
