"""The C++ host-side mirror (crucible_amd/host/crucible.hpp + crucible_render CLI) against the Python mirror:
both must flatten the demo scenes to the same bytes, and the CLI's PPM must equal the Python path's."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene, checkered_spheres, load_teapot, scaled_teapot, teapot_as_list

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "crucible_amd", "host", "crucible_render")


@pytest.fixture(scope="module")
def cli(hiplib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "crucible_amd", "host"), "all"])
    return CLI


def py_dump(sc):
    f = sc.flatten()
    d = f.desc
    hdr = np.array([d.n_prims, d.n_materials, d.n_textures, d.n_images, d.n_keys, d.sky_kind, d.sky_image], dtype=np.int32).tobytes()
    out = (hdr + bytes(f.prims)[:d.n_prims * C.sizeof(A.CrPrimitive)] + bytes(f.materials)[:d.n_materials * C.sizeof(A.CrMaterial)] +
           bytes(f.textures)[:d.n_textures * C.sizeof(A.CrTexture)] + bytes(f.keys)[:d.n_keys * C.sizeof(A.CrKeyframe)])
    for im in f._image_arrays:
        out += np.array([im.shape[1], im.shape[0]], dtype=np.int32).tobytes() + im.tobytes()
    return out


@pytest.mark.parametrize("world,build", [(1, lambda: book1_end_scene(1, scene_seed=7)), (2, lambda: checkered_spheres(1)),
                                         (3, lambda: load_teapot(1)), (6, lambda: scaled_teapot(1)),
                                         (7, lambda: teapot_as_list(1))])
def test_cpp_and_python_mirrors_flatten_identically(cli, tmp_path, world, build):
    out = str(tmp_path / "d.bin")
    subprocess.check_call([cli, "--world", str(world), "--scene-seed", "7", "--dump-desc", out], cwd=ROOT, stderr=subprocess.DEVNULL)
    assert open(out, "rb").read() == py_dump(build())


@pytest.mark.parametrize("mode,flip", [("rle", False), ("flat", False), ("old", False), ("rle", True)])
def test_cpp_and_python_decode_radiance_identically(cli, tmp_path, monkeypatch, mode, flip):
    """world 5 = demo_images::garden_skybox (demo_images.rs:223-242) with a Radiance map given on the command line:
    RTWImage::load_hdr (C++) and decode_radiance (Python) must produce the same RGB8 sky."""
    from test_host_logic import _rgbe_from_float, write_hdr
    from crucible_amd.demo_builder import garden_skybox
    from crucible_amd.scene import RTWImage
    rs = np.random.RandomState(8)
    img = rs.rand(16, 64, 3) * rs.choice([0.05, 0.7, 2.0], size=(16, 64, 1))
    img[3, 8:40] = img[3, 8]
    img[9] = 0.0
    hdr = str(tmp_path / "sky.hdr")
    write_hdr(hdr, _rgbe_from_float(img), mode, flip)
    out = str(tmp_path / "d.bin")
    subprocess.check_call([cli, "--world", "5", "--sky", hdr, "--dump-desc", out], cwd=ROOT, stderr=subprocess.DEVNULL)
    monkeypatch.setenv("ASSET_DIR", str(tmp_path) + "/")
    assert open(out, "rb").read() == py_dump(garden_skybox(1, sky=RTWImage.new("sky.hdr")))


def test_cli_world5_needs_a_sky(cli):
    r = subprocess.run([cli, "--world", "5", "--file", "/tmp/x"], cwd=ROOT, capture_output=True)
    assert r.returncode == 101 and b"garden.hdr is not shipped" in r.stderr


def test_cpp_movie_keyframes_match_python(cli, tmp_path):
    from crucible_amd.scene import LERP, WORLD
    out = str(tmp_path / "m.bin")
    # camera keys are not part of the scene dump; the movie's scene (book1) must still flatten identically
    subprocess.check_call([cli, "--world", "1", "--movie", "--seconds", "2", "--rate", "4", "--dump-desc", out], cwd=ROOT,
                          stderr=subprocess.DEVNULL)
    assert open(out, "rb").read() == py_dump(book1_end_scene(1, scene_seed=1))
    sc = book1_end_scene(1)
    sc.cam_translate_point((3.0, 2.0, 13.0), 2.0, LERP, WORLD, "from")
    ks = sc.scene_cam.look_from_tl.keyframes()
    assert [(k.channel, k.t0, k.t1, k.a) for k in ks] == [(0, 0.0, 2.0, -10.0), (1, 0.0, 2.0, 0.0), (2, 0.0, 2.0, 10.0)]


def test_cli_errors_like_the_reference(cli):
    r = subprocess.run([cli, "--world", "1", "--movie", "--file", "/tmp/x"], cwd=ROOT, capture_output=True)
    assert r.returncode == 101 and b"You must provide a frame rate" in r.stderr   # main.rs:47-49 expect()


@pytest.mark.gpu
@pytest.mark.parametrize("bvh,mode", [("reference", A.CR_BVH_REFERENCE), ("sah", A.CR_BVH_SAH)])
def test_cli_renders_the_same_ppm_as_python(cli, renderer, tmp_path, bvh, mode):
    stem = str(tmp_path / "cli")
    subprocess.check_call([cli, "--file", stem, "--world", "1", "--width", "64", "--samples", "3", "--real", "f64",
                           "--bvh", bvh], cwd=ROOT)
    sc = book1_end_scene(1, scene_seed=1, image_width=64, samples=3)
    sc.real_type = A.CR_REAL_F64
    sc.bvh_mode = mode
    sc.render_image(str(tmp_path / "py"), renderer=renderer)
    assert open(stem + ".ppm").read() == open(str(tmp_path / "py") + ".ppm").read()


@pytest.mark.gpu
def test_cli_movie_writes_frames(cli, tmp_path):
    stem = str(tmp_path / "mov")
    subprocess.check_call([cli, "--file", stem, "--world", "1", "--movie", "--seconds", "0.5", "--rate", "4", "--width", "32",
                           "--samples", "2"], cwd=ROOT)
    frames = sorted(os.listdir(os.path.join(stem, "artifacts")))   # render_movie, scene/mod.rs:295-322
    assert frames == ["image0.ppm", "image1.ppm"]
    a, b = (open(os.path.join(stem, "artifacts", f)).read() for f in frames)
    assert a != b and a.startswith("P3\n32 18\n255\n")


@pytest.mark.gpu
def test_cli_movie_prints_the_references_ffmpeg_command(cli, tmp_path):
    """movie_maker::make_mp4 (scene/movie_maker.rs:6-33): same arguments, handed to the caller instead of executed."""
    stem = str(tmp_path / "mov")
    r = subprocess.run([cli, "--file", stem, "--world", "1", "--movie", "--seconds", "0.5", "--rate", "4", "--width", "32",
                        "--samples", "1"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0
    want = (f"ffmpeg -framerate 4 -i {stem}/artifacts/image%01d.ppm -vf 'scale=trunc(iw/2)*2:trunc(ih/2)*2' -c:v libx264 "
            f"-pix_fmt yuv420p -crf 25 {stem}/movie.mp4")
    assert want in r.stderr, r.stderr


@pytest.mark.gpu
def test_cli_movie_frame_formats_hold_the_same_pixels(cli, tmp_path):
    """--format p6 / png, written by the helper thread while the next frame renders: same bytes
    per channel as the reference's ASCII P3, frame for frame."""
    import zlib
    pix = {}
    for fmt in ("ppm", "p6", "png"):
        stem = str(tmp_path / fmt)
        subprocess.check_call([cli, "--file", stem, "--world", "1", "--movie", "--seconds", "1", "--rate", "3", "--width", "32",
                               "--samples", "2", "--format", fmt], cwd=ROOT)
        names = sorted(os.listdir(os.path.join(stem, "artifacts")))
        assert names == [f"image{k}.{'png' if fmt == 'png' else 'ppm'}" for k in range(3)]
        out = []
        for n in names:
            raw = open(os.path.join(stem, "artifacts", n), "rb").read()
            if fmt == "ppm":
                out.append(np.array(raw.split()[4:], dtype=np.int64).astype(np.uint8).tobytes())
            elif fmt == "p6":
                assert raw.startswith(b"P6\n32 18\n255\n")
                out.append(raw[len(b"P6\n32 18\n255\n"):])
            else:
                assert raw[:8] == b"\x89PNG\r\n\x1a\n"
                pos, idat = 8, b""
                while pos < len(raw):
                    ln = int.from_bytes(raw[pos:pos + 4], "big")
                    if raw[pos + 4:pos + 8] == b"IDAT":
                        idat += raw[pos + 8:pos + 8 + ln]
                    pos += 12 + ln
                rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(18, 1 + 32 * 3)
                assert (rows[:, 0] == 0).all()        # filter type None on every scanline
                out.append(rows[:, 1:].tobytes())
        pix[fmt] = out
    assert pix["ppm"] == pix["p6"] == pix["png"] and len(set(pix["ppm"])) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("force_rccl", [False, True], ids=["direct", "rccl"])
def test_cli_group_path_renders_the_same_files(cli, tmp_path, monkeypatch, force_rccl):
    """--gpus N / --group: the compiled host drives cr_group_create + cr_group_render_host for a still (samples split
    over the devices, one RCCL reduce inside the library) and one host thread per device for a movie (frame f on device
    f % N).  With the one device of this box the files must equal the single-handle path's byte for byte."""
    if force_rccl:
        monkeypatch.setenv("CRUCIBLE_GROUP_FORCE_RCCL", "1")
    a, b = str(tmp_path / "plain"), str(tmp_path / "group")
    common = ["--world", "1", "--width", "64", "--samples", "3", "--real", "f64"]
    subprocess.check_call([cli, "--file", a] + common, cwd=ROOT)
    subprocess.check_call([cli, "--file", b, "--gpus", "1", "--group"] + common, cwd=ROOT)
    assert open(a + ".ppm").read() == open(b + ".ppm").read()
    ma, mb = str(tmp_path / "mplain"), str(tmp_path / "mgroup")
    movie = ["--world", "1", "--movie", "--seconds", "0.75", "--rate", "4", "--width", "32", "--samples", "2"]
    subprocess.check_call([cli, "--file", ma] + movie, cwd=ROOT)
    r = subprocess.run([cli, "--file", mb, "--group"] + movie, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "ffmpeg -framerate 4" in r.stderr
    names = sorted(os.listdir(os.path.join(ma, "artifacts")))
    assert names == sorted(os.listdir(os.path.join(mb, "artifacts"))) == ["image0.ppm", "image1.ppm", "image2.ppm"]
    for n in names:
        assert open(os.path.join(ma, "artifacts", n)).read() == open(os.path.join(mb, "artifacts", n)).read(), n


def test_cli_refuses_malformed_radiance_files(cli, tmp_path):
    """The C++ loader on truncated / corrupted / oversized-header files: the `panic:` exit (101) or a clean load, never a
    signal and never an allocation of what the header promises."""
    from test_host_logic import _rgbe_from_float, write_hdr
    rs = np.random.RandomState(3)
    hdr = str(tmp_path / "g.hdr")
    write_hdr(hdr, _rgbe_from_float(rs.rand(8, 40, 3)), "rle", False)
    good = open(hdr, "rb").read()
    cases = [good[:20], good[:len(good) // 2], good.replace(b"-Y 8 +X 40", b"-Y 99999999 +X 99999999"), good.replace(b"-Y 8 +X 40", b"-Y 8 +X 0"),
             good.replace(b"#?", b"??"), good + b"garbage"]
    for k, data in enumerate(cases):
        f = str(tmp_path / f"m{k}.hdr")
        open(f, "wb").write(data)
        r = subprocess.run([cli, "--world", "5", "--sky", f, "--dump-desc", str(tmp_path / "o.bin")], cwd=ROOT, capture_output=True)
        assert r.returncode in (0, 101), (k, r.returncode, r.stderr[-200:])
        assert b"bad_alloc" not in r.stderr


@pytest.mark.gpu
def test_cli_timing_lines_and_sum_orders(cli, tmp_path):
    """--repeat N --timing: one JSON line of wall-clock phases per render_scene call (the shape of the reference's criterion
    benchmark); --sum-order picks CrRenderParams.sum_order, and both orders write the same P3 file here."""
    import json
    a, b = str(tmp_path / "ref"), str(tmp_path / "rel")
    common = ["--world", "1", "--width", "96", "--samples", "6", "--real", "f64"]
    r = subprocess.run([cli, "--file", a, "--sum-order", "reference", "--repeat", "2", "--timing"] + common, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-300:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert [l["run"] for l in lines] == [0, 1] and all(l["width"] == 96 and l["height"] == 54 and l["samples"] == 6 for l in lines)
    for l in lines:
        assert l["total_ms"] >= l["render_ms"] + l["write_ms"] > 0 and l["kernel_ms"] > 0 and l["bvh_build_ms"] >= 0
    env = dict(os.environ, CRUCIBLE_SUM_ORDER="reference")   # the flag wins over the environment
    subprocess.check_call([cli, "--file", b, "--sum-order", "relaxed"] + common, cwd=ROOT, env=env)
    assert open(a + ".ppm").read() == open(a + "_1.ppm").read() == open(b + ".ppm").read()
