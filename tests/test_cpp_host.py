"""The C++ host-side mirror (crucible_amd/host/crucible.hpp + crucible_render CLI) against the Python mirror:
both must flatten the demo scenes to the same bytes, and the CLI's PPM must equal the Python path's."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene, checkered_spheres, load_teapot

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
    return (hdr + bytes(f.prims)[:d.n_prims * C.sizeof(A.CrPrimitive)] + bytes(f.materials)[:d.n_materials * C.sizeof(A.CrMaterial)] +
            bytes(f.textures)[:d.n_textures * C.sizeof(A.CrTexture)] + bytes(f.keys)[:d.n_keys * C.sizeof(A.CrKeyframe)])


@pytest.mark.parametrize("world,build", [(1, lambda: book1_end_scene(1, scene_seed=7)), (2, lambda: checkered_spheres(1)),
                                         (3, lambda: load_teapot(1))])
def test_cpp_and_python_mirrors_flatten_identically(cli, tmp_path, world, build):
    out = str(tmp_path / "d.bin")
    subprocess.check_call([cli, "--world", str(world), "--scene-seed", "7", "--dump-desc", out], cwd=ROOT, stderr=subprocess.DEVNULL)
    assert open(out, "rb").read() == py_dump(build())


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
