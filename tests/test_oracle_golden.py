"""The oracle against the committed golden vectors (tests/golden/*.npz, written by
tests/golden/make_golden.py).  Bit-exact: the fixtures pin the restatement."""
import os

import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene, checkered_spheres

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 0xC0FFEE


@pytest.fixture(params=[(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")], ids=["f64", "f32"])
def og(request, oracles):
    rt, tag = request.param
    return oracles[rt], tag


def test_vectors(og):
    import importlib.util
    o, tag = og
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    got = mg.vectors(o)
    ref = np.load(os.path.join(GOLD, f"vectors_{tag}.npz"))
    assert set(got) == set(ref.files)
    for k in ref.files:
        assert np.array_equal(np.asarray(got[k]), ref[k], equal_nan=True), k


IMAGE_SCENES = {
    "book1_64x36_spp4": lambda: book1_end_scene(1, scene_seed=1, image_width=64, samples=4),
    "checkered_48x27_spp3": lambda: checkered_spheres(1, image_width=48, samples=3),
    "mixed_64x36_spp4": lambda: scenes.mixed_scene(64, 4),
    "mixed_anim_48x27_spp4": lambda: scenes.mixed_scene(48, 4, animate=True),
    "mixed_nosky_40x22_spp3": lambda: scenes.mixed_scene(40, 3, sky=False),
}


@pytest.mark.parametrize("name", sorted(IMAGE_SCENES))
def test_images(og, name):
    o, tag = og
    ref = np.load(os.path.join(GOLD, f"images_{tag}.npz"))
    img, st = o.render_image(IMAGE_SCENES[name](), seed=SEED)
    assert img.dtype == ref[name].dtype
    assert np.array_equal(img, ref[name])
    assert [st["segments"], st["node_tests"], st["prim_tests"], st["texel_fetches"]] == list(ref[name + "_stats"])


def test_thread_count_does_not_change_pixels(o64):
    sc = book1_end_scene(1, scene_seed=1, image_width=48, samples=2)
    a, _ = o64.render_image(sc, seed=3, n_threads=1)
    b, _ = o64.render_image(sc, seed=3, n_threads=7)
    assert np.array_equal(a, b)


def test_f32_and_f64_agree_statistically(o64, o32):
    """Same uniforms (f32 draws are truncations of the f64 draws): images agree except where a
    path flipped a branch; the mean image difference is small."""
    sc = book1_end_scene(1, scene_seed=1, image_width=64, samples=16)
    a, _ = o64.render_image(sc, seed=5)
    b, _ = o32.render_image(sc, seed=5)
    d = np.abs(a - b.astype(np.float64))
    assert d.mean() < 5e-3
    assert np.median(d) < 1e-4
