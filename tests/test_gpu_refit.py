"""CrRenderParams.refit_boxes -- wrapper boxes re-derived per frame on the device.

The reference never recomputes wrapper boxes (src/objects/bvhwrapper.rs:47-50,102-106), so a keyframed primitive is
clipped where it leaves its construction-time box; refit_boxes = 0 reproduces that, refit_boxes = 1 does not.  Nothing
of the reference's pins the refitted boxes.  Pinned here: (a) bit-exact against the oracle applying the same rule to
the same tree; (b) against ground truth -- the oracle's linear list (HitList::hit, no boxes at all) -- the refitted
render agrees except for box-grazing rays, while the stale boxes visibly do not.
"""
import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A

pytestmark = pytest.mark.gpu

REALS = [(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")]
COUNTERS = ("segments", "node_tests", "prim_tests", "texel_fetches")
SEED = 4242


def render(renderer, sc, rt, mode, refit):
    sc.bvh_mode = mode
    sc.scene_cam.refit_boxes = refit
    renderer.upload_scene(sc.flatten())
    return renderer.render(sc.scene_cam, seed=SEED, real_type=rt)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_REFERENCE, A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH],
                         ids=["reference", "sah", "ordered", "lbvh"])
@pytest.mark.parametrize("frame", [0, 1, 2, 5])
def test_refit_bit_exact_against_oracle_refit(renderer, oracles, rt, tag, mode, frame):
    sc = scenes.moving_scene(96, 4, frame=frame)
    img, st = render(renderer, sc, rt, mode, True)
    tree = renderer.export_bvh(rt) if mode != A.CR_BVH_REFERENCE else None
    ref, rst = oracles[rt].render_image(sc, seed=SEED, tree=tree)
    assert np.array_equal(img, ref), f"differing px = {(img != ref).any(axis=2).sum()}"
    for k in COUNTERS:
        assert st[k] == rst[k], (k, st[k], rst[k])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_REFERENCE, A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH],
                         ids=["reference", "sah", "ordered", "lbvh"])
def test_stale_boxes_still_match_the_oracle(renderer, oracles, rt, tag, mode):
    """refit_boxes = 0 on the same scene: the reference's (clipped) image, bit-exact as before."""
    sc = scenes.moving_scene(96, 4, frame=0)
    img, st = render(renderer, sc, rt, mode, False)
    tree = renderer.export_bvh(rt) if mode != A.CR_BVH_REFERENCE else None
    ref, rst = oracles[rt].render_image(sc, seed=SEED, tree=tree)
    assert np.array_equal(img, ref)
    for k in COUNTERS:
        assert st[k] == rst[k]


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("frame", [0, 2])
def test_refit_agrees_with_the_linear_list_and_stale_boxes_do_not(renderer, oracles, rt, tag, frame):
    sc = scenes.moving_scene(128, 4, frame=frame)
    sc.scene_cam.refit_boxes = False
    truth, tst = oracles[rt].render_image(sc, seed=SEED, linear_list=True)
    fitted, fst = render(renderer, sc, rt, A.CR_BVH_REFERENCE, True)
    stale, _ = render(renderer, sc, rt, A.CR_BVH_REFERENCE, False)
    same_fitted = (fitted == truth).all(axis=2).mean()
    same_stale = (stale == truth).all(axis=2).mean()
    assert same_fitted >= 0.995, same_fitted
    assert same_stale < 0.97, same_stale            # the moving primitives are clipped by their old boxes
    assert abs(int(fst["segments"]) - int(tst["segments"])) <= 0.002 * int(tst["segments"]) + 4


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_REFERENCE, A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH],
                         ids=["reference", "sah", "ordered", "lbvh"])
def test_refit_of_motionless_keys_changes_nothing(renderer, rt, tag, mode):
    """Keys with zero offsets: the refit kernels run (the scene has primitive keys) and must reproduce the
    construction-time boxes exactly."""
    sc = scenes.moving_scene(80, 3, frame=0, null_motion=True)
    a, sa = render(renderer, sc, rt, mode, False)
    b, sb = render(renderer, sc, rt, mode, True)
    assert np.array_equal(a, b)
    for k in COUNTERS:
        assert sa[k] == sb[k]


def test_refit_is_per_render_not_sticky(renderer):
    """A refitted render must not leave its boxes behind for the next render with refit_boxes = 0."""
    sc = scenes.moving_scene(64, 3, frame=0)
    stale1, _ = render(renderer, sc, A.CR_REAL_F32, A.CR_BVH_REFERENCE, False)
    sc.scene_cam.refit_boxes = True
    fitted, _ = renderer.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F32)
    sc.scene_cam.refit_boxes = False
    stale2, _ = renderer.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F32)
    assert np.array_equal(stale1, stale2) and not np.array_equal(stale1, fitted)
