"""SURVEY 8(a) row a16 for non-sphere primitives: ScaleX / ScaleY / ScaleZ keyframes on triangles
(scene_animator.rs:38-229 -> transform_builder.rs:101-346 -> timeline/mod.rs:233-263), evaluated per hit on the
device.  Bit-exact against the oracle (image and work counters), f64 and f32, with the reference's stale wrapper
boxes and with refit_boxes, and in every BVH mode; the Scene API's type checks are status codes at upload."""
import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A
from crucible_amd.renderer import CrucibleError

pytestmark = pytest.mark.gpu

REALS = [(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")]
COUNTERS = ("segments", "node_tests", "prim_tests", "texel_fetches")
SEED = 90210


def render(renderer, sc, rt, mode=A.CR_BVH_REFERENCE, refit=False):
    sc.bvh_mode = mode
    sc.scene_cam.refit_boxes = refit
    renderer.upload_scene(sc.flatten())
    return renderer.render(sc.scene_cam, seed=SEED, real_type=rt)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("refit", [False, True], ids=["stale-boxes", "refit"])
@pytest.mark.parametrize("frame", [0, 1, 2, 4])
def test_scale_keys_bit_exact(renderer, oracles, rt, tag, refit, frame):
    sc = scenes.scaled_scene(96, 4, frame=frame)
    img, st = render(renderer, sc, rt, refit=refit)
    ref, rst = oracles[rt].render_image(sc, seed=SEED)
    assert np.array_equal(img, ref), f"differing px = {(img != ref).any(axis=2).sum()}"
    for k in COUNTERS:
        assert st[k] == rst[k], (k, st[k], rst[k])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
def test_scale_keys_in_the_other_bvh_modes(renderer, oracles, rt, tag, mode):
    sc = scenes.scaled_scene(80, 3, frame=1)
    img, st = render(renderer, sc, rt, mode=mode, refit=True)
    ref, rst = oracles[rt].render_image(sc, seed=SEED, tree=renderer.export_bvh(rt))
    assert np.array_equal(img, ref)
    for k in COUNTERS:
        assert st[k] == rst[k], (k, st[k], rst[k])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
def test_refit_encloses_scaled_triangles(renderer, oracles, rt, tag):
    """Ground truth = the oracle's linear list (no boxes): the refitted boxes lose nothing, the stale ones clip."""
    sc = scenes.scaled_scene(128, 4, frame=1)
    truth, _ = oracles[rt].render_image(sc, seed=SEED, linear_list=True)
    fitted, _ = render(renderer, sc, rt, refit=True)
    stale, _ = render(renderer, sc, rt, refit=False)
    assert (fitted == truth).all(axis=2).mean() >= 0.995
    assert (stale == truth).all(axis=2).mean() < 0.98


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
def test_scaled_teapot_demo(renderer, oracles, rt, tag):
    """6320 triangles sharing nine keys (three translate, six scale), default sky: bit-exact."""
    from crucible_amd.demo_builder import scaled_teapot
    sc = scaled_teapot(1, image_width=96, samples=3)
    for frame in (0, 2):
        sc.scene_cam.frame = frame
        img, st = render(renderer, sc, rt)
        ref, rst = oracles[rt].render_image(sc, seed=SEED)
        assert np.array_equal(img, ref)
        for k in COUNTERS:
            assert st[k] == rst[k], (k, st[k], rst[k])


def test_scale_key_type_checks(renderer):
    """ScaleX/Y/Z on a sphere and ScaleR on a triangle are panics in the Scene API (scene_animator.rs:39-41,140-150):
    CR_ERR_INVALID_ARG here; camera keys are translations only."""
    sc = scenes.scaled_scene(32, 1)
    flat = sc.flatten()
    tri = next(i for i in range(flat.desc.n_prims) if flat.prims[i].kind == A.CR_PRIM_TRIANGLE)
    sph = next(i for i in range(flat.desc.n_prims) if flat.prims[i].kind == A.CR_PRIM_SPHERE)
    k0 = flat.prims[tri].key_first
    saved = flat.keys[k0].channel
    flat.keys[k0].channel = A.CR_KEY_RADIUS
    with pytest.raises(CrucibleError) as e:
        renderer.upload_scene(flat)
    assert e.value.code == A.CR_ERR_INVALID_ARG and "ScaleR" in str(e.value)
    flat.keys[k0].channel = saved
    flat.prims[sph].key_first, flat.prims[sph].key_count = flat.prims[tri].key_first, flat.prims[tri].key_count
    with pytest.raises(CrucibleError) as e:
        renderer.upload_scene(flat)
    assert e.value.code == A.CR_ERR_INVALID_ARG and "Spheres" in str(e.value)
    flat.prims[sph].key_count = 0
    flat.keys[k0].channel = 7
    with pytest.raises(CrucibleError):
        renderer.upload_scene(flat)
    flat.keys[k0].channel = saved
    renderer.upload_scene(flat)
    cam = sc.scene_cam
    cam.look_from_tl.translate_x(1.0, 0.5, "LERP", "Local")
    cd = cam.desc()
    cd.from_keys[0].channel = A.CR_KEY_SCALE_X
    import ctypes as C
    p = cam.params(1, A.CR_REAL_F32)
    out = np.zeros((cam.image_height, cam.image_width, 3), dtype=np.float32)
    rc = renderer.lib.cr_render_host(renderer.h, C.byref(cd), C.byref(p), out.ctypes.data_as(C.c_void_p), None)
    assert rc == A.CR_ERR_INVALID_ARG
