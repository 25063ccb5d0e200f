"""Host side of the boundary: scene builder mirror, keyframe flattening, demo scenes,
OBJ loader, PPM writer / byte quantisation (cr_write_ppm needs no GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A
from crucible_amd.demo_builder import SceneRng, book1_end_scene, load_teapot, procedural_sky
from crucible_amd.scene import (LERP, LOCAL, NERP, WORLD, BVHWrapper, Camera, CheckerTexture, HitList, Lambertian, Metal, Scene,
                                Sphere, Triangle, load_obj)
from crucible_amd.timeline import TransformTimeline


def test_viewport_height_matches_reference_rule():   # Viewport::new, src/camera/mod.rs:37-38
    for w, h in ((400, 225), (1920, 1080), (3840, 2160), (64, 36), (1, 1)):
        assert Camera(16.0 / 9.0, w, 24.0, 180.0, 1).image_height == h


def test_book1_is_deterministic_and_has_reference_shape():
    a = book1_end_scene(1, scene_seed=1).flatten()
    b = book1_end_scene(1, scene_seed=1).flatten()
    assert a.desc.n_prims == b.desc.n_prims == 484   # <= 1 + 22*22 + 3 (demo_images.rs:48-84)
    assert bytes(a.prims) == bytes(b.prims) and bytes(a.materials) == bytes(b.materials)
    c = book1_end_scene(1, scene_seed=2).flatten()
    assert bytes(a.prims) != bytes(c.prims)
    p0 = a.prims[0]   # ground sphere first (demo_images.rs:35-42)
    assert list(p0.v[0:4]) == [0.0, -1000.0, 0.0, 1000.0]
    last = a.prims[a.desc.n_prims - 1]
    assert list(last.v[0:4]) == [4.0, 1.0, 0.0, 1.0]
    kinds = [a.materials[a.prims[i].material].kind for i in range(1, a.desc.n_prims - 3)]
    frac = [kinds.count(k) / len(kinds) for k in (0, 1, 2)]
    assert 0.7 < frac[0] < 0.9 and 0.08 < frac[1] < 0.22 and 0.01 < frac[2] < 0.1   # 80/15/5 mix


def test_scene_rng_is_splitmix64():
    r = SceneRng(0)   # published SplitMix64 test vector for seed 0
    assert [r.u64() for _ in range(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_alias_collision_raises():   # id_vendor.rs:51-75 / scene/mod.rs:170-174
    sc = Scene.new_image(16 / 9, 32, 24, 180.0, 1)
    m = Metal.new((0.5, 0.5, 0.5), 0.0)
    sc.add_element(Sphere.new((0, 0, 0), 1.0, m), "a")
    with pytest.raises(ValueError):
        sc.add_element(Sphere.new((1, 0, 0), 1.0, m), "a")
    with pytest.raises(ValueError):
        sc.add_element(Sphere.new((1, 0, 0), 1.0, m), "cam")   # reserved


def test_constructor_checks_mirror_reference_asserts():
    with pytest.raises(ValueError):
        Metal.new((0.5, 0.5, 0.5), 1.5)          # metal.rs:21
    with pytest.raises(ValueError):
        Sphere.new((0, 0, 0), -1.0, None)        # sphere.rs:26
    with pytest.raises(ValueError):
        Lambertian.new_from_color((2.0, 0, 0), 1.0)   # Color::new, utils.rs:345
    with pytest.raises(ValueError):
        Camera(16 / 9, 32, 24.0, 180.0, 1).set_samples(0)   # camera/mod.rs:235


def test_hide_and_show():
    sc = scenes.few_spheres(3)
    sc.hide_element("s1")
    f = sc.flatten()
    assert [f.prims[i].flags for i in range(3)] == [0, A.CR_PRIM_HIDDEN, 0]
    sc.show_element("s1")
    assert sc.flatten().prims[1].flags == 0


def test_keyframe_flattening_lerp_world():
    """first_movie's camera walk (demo_movies.rs:33-60): World LERP keys become offsets."""
    tl = TransformTimeline((0.0, 0.0, -12.0))
    tl.translate_point((12.0, 0.0, 0.0), 2.5, LERP, WORLD)
    tl.translate_point((0.0, 0.0, 12.0), 5.0, LERP, WORLD)
    ks = tl.keyframes()
    assert len(ks) == 6
    x = [k for k in ks if k.channel == A.CR_KEY_TX]
    z = [k for k in ks if k.channel == A.CR_KEY_TZ]
    assert (x[0].t0, x[0].t1, x[0].a) == (0.0, 2.5, 12.0)       # 12 - start 0
    assert (x[1].t0, x[1].t1, x[1].a) == (2.5, 5.0, -12.0)      # 0 - previous end 12
    assert (z[0].a, z[1].a) == (12.0, 12.0)                     # 0-(-12), 12-0
    assert all(k.interp == A.CR_KEY_LERP for k in ks)


def test_keyframe_flattening_nerp_and_radius():
    tl = TransformTimeline.new_sphere((1.0, 2.0, 3.0), 0.5)
    tl.translate_x(4.0, 2.0, NERP, LOCAL)
    tl.scale_sphere(2.0, 3.0, LERP)
    tl.scale_sphere(1.0, 1.0, NERP)      # inserted earlier: list stays sorted by start
    ks = tl.keyframes()
    assert [(k.channel, k.interp, k.t0, k.t1, k.a, k.b) for k in ks] == [
        (A.CR_KEY_TX, A.CR_KEY_NERP, 2.0, 2.0, 4.0, 0.0),
        (A.CR_KEY_RADIUS, A.CR_KEY_LERP, 0.0, 3.0, 0.5, 2.0),
        (A.CR_KEY_RADIUS, A.CR_KEY_NERP, 1.0, 1.0, 1.0, 0.0)]
    with pytest.raises(AssertionError):
        tl.translate_x(1.0, -1.0, NERP, LOCAL)   # keyframe before the animation start


def test_scale_r_rejects_meshes():   # scene_animator.rs:140-150
    sc = scenes.mixed_scene(32, 1)
    with pytest.raises(ValueError):
        sc.scale_r(0.5, 1.0, LERP, "quad_a")
    with pytest.raises(KeyError):
        sc.scale_r(0.5, 1.0, LERP, "nope")


def test_flatten_shares_materials_and_orders_textures():
    sc = scenes.mixed_scene(32, 1)
    f = sc.flatten()
    d = f.desc
    assert d.n_prims == len(sc.elements) and d.sky_kind == A.CR_SKY_SPHERICAL
    assert f.prims[7].material == f.prims[8].material       # the quad's two triangles share m_quad
    for i in range(d.n_textures):                           # children before parents
        t = f.textures[i]
        if t.kind == A.CR_TEX_CHECKER:
            assert 0 <= t.even < i and 0 <= t.odd < i
    assert f.textures[0].kind == A.CR_TEX_SOLID
    inv = [f.textures[i].inv_scale for i in range(d.n_textures) if f.textures[i].kind == A.CR_TEX_CHECKER]
    assert 1.0 / 0.25 in inv and 1.0 / 1.5 in inv


def test_obj_loader(tmp_path, monkeypatch):   # obj_loader.rs:66-143
    (tmp_path / "t.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n\nf 1 2 3\nf 1 3 4\n")
    monkeypatch.setenv("ASSET_DIR", str(tmp_path) + "/")
    tris = load_obj("t.obj", 2.0, (1.0, 0.0, -1.0), None)
    assert len(tris) == 2
    assert tris[0].a == (1.0, 0.0, -1.0) and tris[0].b == (3.0, 0.0, -1.0) and tris[1].c == (1.0, 0.0, 1.0)
    (tmp_path / "bad.obj").write_text("v 0 0 0\nvn 0 0 1\n")
    with pytest.raises(ValueError):
        load_obj("bad.obj", 1.0, (0, 0, 0), None)            # "Unsupported OBJ file"
    (tmp_path / "quad.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3 4\n")
    with pytest.raises(ValueError):
        load_obj("quad.obj", 1.0, (0, 0, 0), None)           # triangles only


def test_obj_loader_tolerant_mode(tmp_path, monkeypatch):   # SURVEY 8(f) row 4; the reference panics on all of these
    (tmp_path / "m.obj").write_text("# comment\nmtllib a.mtl\no thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n"
                                    "s off\nusemtl m\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4 -3 -2\n")
    monkeypatch.setenv("ASSET_DIR", str(tmp_path) + "/")
    with pytest.raises(ValueError):
        load_obj("m.obj", 1.0, (0, 0, 0), None)
    tris = load_obj("m.obj", 1.0, (0, 0, 0), None, strict=False)
    assert [(t.a, t.b, t.c) for t in tris] == [((0, 0, 0), (1, 0, 0), (1, 1, 0)), ((0, 0, 0), (1, 1, 0), (0, 1, 0)),
                                               ((0, 0, 0), (1, 0, 0), (1, 1, 0))]


def test_teapot_asset_loads():
    sc = load_teapot(1, image_width=32, samples=1)
    assert len(sc.elements) == 6320 + 1     # assets/teapot.obj: 3644 v / 6320 f, + ground


def test_ppm_writer_and_quantiser(hiplib, tmp_path):
    """Camera::render's file layout (camera/mod.rs:286,306-311) and Display for Color (utils.rs:422-437)."""
    for dtype, rt in ((np.float64, A.CR_REAL_F64), (np.float32, A.CR_REAL_F32)):
        img = np.array([[[0.529, 0.616, 0.730], [0.0, 1.0, 0.25]], [[0.5, 0.04, 0.9999], [1.0, 0.0, 1e-9]]], dtype=dtype)
        path = str(tmp_path / f"o_{rt}.ppm")
        assert hiplib.cr_write_ppm(path.encode(), img.ctypes.data_as(C.c_void_p), rt, 2, 2) == A.CR_OK
        lines = open(path).read().split("\n")
        assert lines[:3] == ["P3", "2 2", "255"]
        assert lines[3] == "185 200 217"                     # color_display_test, utils.rs:781
        assert lines[4] == "0 255 127"
        expect = [[int(255.0 * np.sqrt(float(c))) for c in px] for px in img.reshape(-1, 3)]
        assert [[int(x) for x in l.split()] for l in lines[3:7]] == expect
        assert lines[7] == ""
        q = np.zeros((4, 3), dtype=np.uint8)
        assert hiplib.cr_quantize_rgb8(img.ctypes.data_as(C.c_void_p), rt, 4, q.ctypes.data_as(C.c_void_p)) == A.CR_OK
        assert q.tolist() == expect
    assert hiplib.cr_write_ppm(b"/nonexistent_dir/x.ppm", img.ctypes.data_as(C.c_void_p), rt, 2, 2) == A.CR_ERR_IO
    nan = np.array([[[np.nan, -1.0, 2.0]]])
    q = np.zeros((1, 3), dtype=np.uint8)
    hiplib.cr_quantize_rgb8(nan.ctypes.data_as(C.c_void_p), A.CR_REAL_F64, 1, q.ctypes.data_as(C.c_void_p))
    assert q.tolist() == [[0, 0, 255]]                       # `as u32` saturates, NaN -> 0; 255*sqrt(2) = 360 -> byte clamp


def test_ppm_writer_formats_whole_frames_in_chunks(hiplib, tmp_path):
    """The P3 writer formats 65536 pixels at a time into memory (one fprintf per pixel took as long as rendering the frame): a frame that is
    no multiple of the chunk, every byte value, and channels no Color could hold (> 1: printed as `as u32` would, not clamped; NaN and
    negatives: 0) must come out as the line-by-line `writeln!(file, "{color}")` of camera/mod.rs:306-311 would."""
    rs = np.random.RandomState(11)
    w, h = 517, 331                                     # 171127 pixels: two full chunks and a ragged one
    img = rs.uniform(0, 1, size=(h, w, 3))
    img.reshape(-1)[:256] = (np.arange(256) / 255.0) ** 2 * (1 + 1e-12)     # every byte value, just above its boundary
    img[5, 7] = (4.0, 1e12, 1e300)                      # 510, 255000000, u32::MAX (saturating cast)
    img[6, 7] = (np.nan, -0.5, -0.0)
    for dtype, rt in ((np.float64, A.CR_REAL_F64), (np.float32, A.CR_REAL_F32)):
        with np.errstate(over="ignore"):
            a = np.ascontiguousarray(img.astype(dtype))    # 1e300 -> inf in f32
        path = str(tmp_path / f"big_{rt}.ppm")
        assert hiplib.cr_write_ppm(path.encode(), a.ctypes.data_as(C.c_void_p), rt, w, h) == A.CR_OK
        lines = open(path).read().split("\n")
        assert lines[:3] == ["P3", f"{w} {h}", "255"] and lines[-1] == "" and len(lines) == 3 + w * h + 1
        with np.errstate(invalid="ignore", over="ignore"):
            v = 255.0 * np.sqrt(a.astype(np.float64).reshape(-1, 3))
        v = np.where(np.isnan(v) | (v <= 0), 0.0, np.minimum(v, 4294967295.0))
        expect = v.astype(np.uint64)
        got = np.array([l.split() for l in lines[3:-1]], dtype=np.uint64)
        assert np.array_equal(got, expect)
        assert lines[3 + 5 * w + 7] == ("510 255000000 4294967295" if dtype == np.float64 else lines[3 + 5 * w + 7])
        assert lines[3 + 6 * w + 7] == "0 0 0"


def test_png_and_binary_ppm_carry_the_same_bytes(hiplib, tmp_path):
    """P6 and PNG output quantise exactly like the P3 writer."""
    from PIL import Image
    rs = np.random.RandomState(4)
    img = rs.uniform(0, 1, size=(37, 53, 3))
    img[0, 0] = (0.529, 0.616, 0.730)
    expect = (255.0 * np.sqrt(img)).astype(np.uint8)
    for dtype, rt in ((np.float64, A.CR_REAL_F64), (np.float32, A.CR_REAL_F32)):
        a = img.astype(dtype)
        exp = (255.0 * np.sqrt(a.astype(np.float64))).astype(np.uint8)
        png, p6 = str(tmp_path / f"o{rt}.png"), str(tmp_path / f"o{rt}.p6.ppm")
        assert hiplib.cr_write_png(png.encode(), a.ctypes.data_as(C.c_void_p), rt, 53, 37) == A.CR_OK
        assert hiplib.cr_write_ppm_binary(p6.encode(), a.ctypes.data_as(C.c_void_p), rt, 53, 37) == A.CR_OK
        assert np.array_equal(np.asarray(Image.open(png).convert("RGB")), exp)
        raw = open(p6, "rb").read()
        assert raw.startswith(b"P6\n53 37\n255\n") and np.array_equal(np.frombuffer(raw[len(b"P6\n53 37\n255\n"):], dtype=np.uint8).reshape(37, 53, 3), exp)
    assert tuple(expect[0, 0]) == (185, 200, 217)
    assert hiplib.cr_write_png(b"/nonexistent_dir/x.png", img.ctypes.data_as(C.c_void_p), A.CR_REAL_F64, 53, 37) == A.CR_ERR_IO


def test_procedural_sky_is_seeded():
    a, b = procedural_sky(64, 32, seed=7), procedural_sky(64, 32, seed=7)
    assert np.array_equal(a.rgb8, b.rgb8) and a.rgb8.shape == (32, 64, 3)


def test_movie_frame_count():   # scene/mod.rs:324-330
    sc = Scene.new_movie(16 / 9, 32, 24, 180.0, 1, 10.0)
    assert sc.compute_frame_count() == 240
    sc.duration = 0.51
    assert sc.compute_frame_count() == 13


# ---------------------------------------------------------------- Radiance .hdr
def _rgbe_from_float(img):
    """float RGB -> RGBE bytes (the standard Radiance encoding: shared exponent of the largest channel)."""
    img = np.asarray(img, dtype=np.float64)
    m = img.max(axis=2)
    e = np.where(m > 1e-32, np.frexp(np.maximum(m, 1e-38))[1], 0)
    scale = np.where(m > 1e-32, np.ldexp(256.0, -e), 0.0)
    out = np.zeros(img.shape[:2] + (4,), dtype=np.uint8)
    out[..., :3] = np.clip(np.floor(img * scale[..., None]), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    return out


def _rle_channel(vals):
    out, i, n = bytearray(), 0, len(vals)
    while i < n:
        run = 1
        while i + run < n and run < 127 and vals[i + run] == vals[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, vals[i]])
            i += run
        else:
            j = i
            while j < n and j - i < 128:
                r = 1
                while j + r < n and r < 4 and vals[j + r] == vals[j]:
                    r += 1
                if r >= 4:
                    break
                j += 1
            out += bytes([j - i]) + bytes(vals[i:j])
            i = j
    return bytes(out)


def write_hdr(path, rgbe, mode="rle", flip=False):
    h, w = rgbe.shape[:2]
    rows = rgbe[::-1] if flip else rgbe
    body = bytearray()
    for row in rows:
        if mode == "rle":
            body += bytes([2, 2, w >> 8, w & 255])
            for c in range(4):
                body += _rle_channel(row[:, c].tolist())
        elif mode == "flat":
            body += row.tobytes()
        else:   # old-style repeat markers
            x = 0
            while x < w:
                body += row[x].tobytes()
                run = 1
                while x + run < w and (row[x + run] == row[x]).all():
                    run += 1
                x += 1
                rep = min(run - 1, 255)
                if rep >= 2 and not (row[x - 1][:3] == 1).all():
                    body += bytes([1, 1, 1, rep])
                    x += rep
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + (b"+Y" if flip else b"-Y") +
                f" {h} +X {w}\n".encode() + bytes(body))


def _expected_rgb8(rgbe):
    e = rgbe[..., 3].astype(np.int64)
    f = np.where(e > 0, 2.0 ** (e - 136.0), 0.0)
    x = np.clip(rgbe[..., :3].astype(np.float64) * f[..., None], 0.0, 1.0) * 255.0   # exact in f64 for 8-bit mantissas
    return np.where(x - np.floor(x) >= 0.5, np.floor(x) + 1, np.floor(x)).astype(np.uint8)


@pytest.mark.parametrize("mode,flip", [("rle", False), ("flat", False), ("old", False), ("rle", True)])
def test_radiance_decoder(tmp_path, monkeypatch, mode, flip):
    from crucible_amd.scene import RTWImage
    rs = np.random.RandomState(5)
    img = rs.rand(9, 40, 3) * rs.choice([0.02, 0.5, 1.0, 3.0], size=(9, 40, 1))
    img[2, 5:30] = img[2, 5]          # long runs for the RLE paths
    img[4, :] = 0.0                   # e == 0 pixels
    img[6, 10:20] = (0.25, 0.5, 1.0)
    rgbe = _rgbe_from_float(img)
    write_hdr(str(tmp_path / "t.hdr"), rgbe, mode, flip)
    monkeypatch.setenv("ASSET_DIR", str(tmp_path) + "/")
    got = RTWImage.new("t.hdr").rgb8
    assert got.shape == (9, 40, 3)
    assert np.array_equal(got, _expected_rgb8(rgbe))
    assert got.max() == 255 and got.min() == 0     # clamped highlights, black row


def test_mp4_command_is_the_references_ffmpeg_call():
    """scene/movie_maker.rs:6-33: the argument list, with the frame pattern of render_movie (scene/mod.rs:295-322)."""
    from crucible_amd.demo_builder import book1_end_scene
    sc = book1_end_scene(1, scene_seed=1, image_width=32, samples=1)
    sc.frame_rate = 24
    assert sc.mp4_command("out", 3) == ["ffmpeg", "-framerate", "24", "-i", "out/artifacts/image%03d.ppm", "-vf",
                                        "scale=trunc(iw/2)*2:trunc(ih/2)*2", "-c:v", "libx264", "-pix_fmt", "yuv420p",
                                        "-crf", "25", "out/movie.mp4"]


def test_mirror_rejects_boxes_it_cannot_describe():
    """HitList as a scene element (hitlist.rs:13-27): the descriptor can say "empty box" or "union of the objects";
    add() after new(vec), clear() after add() and an add()ed inner list with an empty box are neither."""
    m = Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)
    mixed = HitList.new([Sphere.new((0, 0, 0), 1.0, m)])
    mixed.add(Sphere.new((2, 0, 0), 1.0, m))
    with pytest.raises(ValueError):
        mixed.spliced()
    cleared = HitList.default()
    cleared.add(Sphere.new((0, 0, 0), 1.0, m))
    cleared.clear()
    with pytest.raises(ValueError):
        cleared.spliced()
    outer = HitList.default()
    outer.add(HitList.new([Sphere.new((0, 0, 0), 1.0, m)]))   # the inner box is empty, the spliced union is not
    with pytest.raises(ValueError):
        outer.spliced()
    fine = HitList.new([HitList.new([Triangle.new((0, 0, 0), (1, 0, 0), (0, 1, 0), m)]), Sphere.new((0, 0, 0), 1.0, m)])
    objs, empty = fine.spliced()
    assert empty and [type(o).__name__ for o in objs] == ["Triangle", "Sphere"]


def test_bvh_wrapper_element_mirror():
    """BVHWrapper::new_wrapper(list) as a scene element (bvhwrapper.rs:15-32): hidden objects are dropped, no visible object
    gives the empty list, anything but spheres and triangles inside is refused; flatten() emits the record and its objects."""
    m = Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)
    a, b, c = Sphere.new((0, 0, 0), 1.0, m), Sphere.new((2, 0, 0), 1.0, m), Triangle.new((0, 0, 0), (1, 0, 0), (0, 1, 0), m)
    b.hide = True
    w = BVHWrapper.new_wrapper(HitList.new([a, b, c]))
    assert isinstance(w, BVHWrapper) and w.objs == [a, c]
    b2 = Sphere.new((2, 0, 0), 1.0, m)
    b2.hide = True
    assert isinstance(BVHWrapper.new_wrapper(HitList.new([b2])), HitList)
    with pytest.raises(ValueError):
        BVHWrapper.new_wrapper(HitList.new([a, HitList.default()]))
    sc = Scene.new_image(1.0, 8, 24, 180.0, 1)
    sc.add_element(w, "w")
    sc.add_element(Sphere.new((5, 0, 0), 1.0, m), "s")
    flat = sc.flatten()
    kinds = [(p.kind, p.flags) for p in flat.prims]
    assert kinds == [(A.CR_PRIM_BVH, 0), (A.CR_PRIM_SPHERE, A.CR_PRIM_MEMBER), (A.CR_PRIM_TRIANGLE, A.CR_PRIM_MEMBER), (A.CR_PRIM_SPHERE, 0)]
    assert (flat.prims[0].v[0], flat.prims[0].v[1]) == (1.0, 2.0)


def test_malformed_radiance_files_are_value_errors(tmp_path, monkeypatch):
    """Truncated files, corrupted headers and runs, sizes the file cannot hold: ValueError, never IndexError / MemoryError
    (the C++ loader answers the same inputs with its `panic:` exit, tests/test_cpp_host.py)."""
    from crucible_amd.scene import decode_radiance
    rs = np.random.RandomState(2)
    img = rs.rand(8, 40, 3)
    good = {}
    for mode in ("rle", "flat", "old"):
        p = str(tmp_path / f"g_{mode}.hdr")
        write_hdr(p, _rgbe_from_float(img), mode, False)
        good[mode] = open(p, "rb").read()
        assert decode_radiance(good[mode]).shape == (8, 40, 3)
    for it in range(300):
        b = bytearray(good[("rle", "flat", "old")[it % 3]])
        k = it % 5
        if k == 0:
            b = b[:rs.randint(0, len(b))]
        elif k == 1:
            for _ in range(rs.randint(1, 8)):
                b[rs.randint(0, len(b))] = rs.randint(0, 256)
        elif k == 2:
            for _ in range(rs.randint(1, 6)):
                b[rs.randint(0, 60)] = rs.randint(0, 256)
        elif k == 3:
            b += bytes(rs.randint(0, 256, size=rs.randint(1, 50)).tolist())
        else:
            b = b.replace(b"-Y 8 +X 40", [b"-Y 99999999 +X 99999999", b"-Y -1 +X 40", b"-Y 8 +X 0", b"+X 40 -Y 8", b"-Y 8"][it % 25 // 5])
        try:
            out = decode_radiance(bytes(b))
            assert out.dtype == np.uint8 and out.ndim == 3
        except ValueError:
            pass
