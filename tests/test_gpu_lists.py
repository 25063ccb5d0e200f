"""a HitList as a scene element (Scene::add_element keeps it, scene/mod.rs:164-166; the BVH build
treats it as one object, bvhwrapper.rs:18-22; a leaf wrapper that holds it scans its objects in order with the
shrinking interval and no box test, hitlist.rs:51-65).  Bit-exact against the oracle (image and work counters), f64
and f32, every pipeline, stale boxes and refit, and in the opt-in trees (where a list's visible objects are ordinary
primitives); cr_export_bvh names the list as the leaf's child; the descriptor rules are status codes."""
import ctypes as C

import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A
from crucible_amd.renderer import CrucibleError
from crucible_amd.scene import HitList, Lambertian, Scene, Sphere, Triangle

pytestmark = pytest.mark.gpu

REALS = [(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")]
COUNTERS = ("segments", "node_tests", "prim_tests", "texel_fetches")
SEED = 60606


def render(renderer, sc, rt, mode=A.CR_BVH_REFERENCE, refit=False):
    sc.bvh_mode = mode
    sc.scene_cam.refit_boxes = refit
    flat = sc.flatten()
    renderer.upload_scene(flat)
    return renderer.render(sc.scene_cam, seed=SEED, real_type=rt)


def same(img, st, ref, rst):
    assert np.array_equal(img, ref), f"differing px = {(img != ref).any(axis=2).sum()}"
    for k in COUNTERS:
        assert st[k] == rst[k], (k, st[k], rst[k])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("refit", [False, True], ids=["stale-boxes", "refit"])
@pytest.mark.parametrize("variant,frame", [("mixed", 0), ("mixed", 1), ("mixed", 3), ("only_lists", 0), ("only_lists", 1),
                                           ("one_list", 0), ("one_list", 1)])
def test_list_elements_bit_exact(renderer, oracles, rt, tag, refit, variant, frame):
    sc = scenes.list_scene(96, 4, frame=frame, variant=variant)
    img, st = render(renderer, sc, rt, refit=refit)
    ref, rst = oracles[rt].render_image(sc, seed=SEED)
    same(img, st, ref, rst)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("pipeline", ["wavefront", "queue", "pixel-granular"])
def test_list_elements_in_the_other_pipelines(oracles, monkeypatch, rt, tag, pipeline):
    from crucible_amd.renderer import Renderer
    if pipeline == "pixel-granular":
        monkeypatch.setenv("CRUCIBLE_SAMPLE_GRANULAR", "0")
    else:
        monkeypatch.setenv("CRUCIBLE_PIPELINE", pipeline)
    r = Renderer(0)
    try:
        sc = scenes.list_scene(80, 3, frame=1)
        img, st = render(r, sc, rt)
        ref, rst = oracles[rt].render_image(sc, seed=SEED)
        same(img, st, ref, rst)
    finally:
        r.close()


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("env", [{"CRUCIBLE_LDS_LIMIT": "0", "CRUCIBLE_LDS_TOP_KB": "0"}, {"CRUCIBLE_LDS_LIMIT": "0"},
                                 {"CRUCIBLE_LDS_LIMIT": "0", "CRUCIBLE_LDS_TOP_KB": "1"}], ids=["global", "top-levels", "top-1KB"])
def test_list_elements_outside_lds(oracles, monkeypatch, rt, tag, env):
    """The scene-in-LDS path is what the small scenes above take; the same scene with the tree in HBM (and with only
    its top levels staged) reads the leaf runs the same way."""
    from crucible_amd.renderer import Renderer
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = Renderer(0)
    try:
        sc = scenes.list_scene(80, 3, frame=0)
        img, st = render(r, sc, rt)
        assert st["scene_in_lds"] != 1   # 1 = the whole scene staged in LDS
        ref, rst = oracles[rt].render_image(sc, seed=SEED)
        same(img, st, ref, rst)
    finally:
        r.close()


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
@pytest.mark.parametrize("variant", ["mixed", "only_lists"])
def test_list_elements_in_the_other_bvh_modes(renderer, oracles, rt, tag, mode, variant):
    """There a list's visible objects are primitives of the tree: the exported tree names them, never the list."""
    sc = scenes.list_scene(80, 3, frame=1, variant=variant)
    img, st = render(renderer, sc, rt, mode=mode, refit=True)
    boxes, kids, axis = renderer.export_bvh(rt)
    flat = sc.flatten()
    named = sorted({~c for c in kids.ravel() if c < 0})
    want = [i for i, p in enumerate(flat.prims) if p.kind != A.CR_PRIM_LIST and not (p.flags & A.CR_PRIM_HIDDEN)]
    assert named == want
    ref, rst = oracles[rt].render_image(sc, seed=SEED, tree=(boxes, kids, axis))
    same(img, st, ref, rst)
    truth, _ = oracles[rt].render_image(sc, seed=SEED, linear_list=True)
    assert (img == truth).all(axis=2).mean() >= 0.98   # the tetrahedron's axis-flat face has a zero-thickness box: lost when alone in a leaf (bvh.rs:126)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("variant", ["mixed", "only_lists", "one_list"])
def test_export_names_the_list(renderer, oracles, rt, tag, variant):
    """CR_BVH_REFERENCE: the exported tree is the oracle's, wrapper for wrapper, and a list is one child."""
    sc = scenes.list_scene(32, 1, variant=variant)
    sc.bvh_mode = A.CR_BVH_REFERENCE
    flat = sc.flatten()
    renderer.upload_scene(flat)
    boxes, kids, axis = renderer.export_bvh(rt)
    o = oracles[rt]
    h = o.scene_create(flat)
    try:
        cap = len(kids) + 8
        oboxes = np.zeros((cap, 6), dtype=o.np_real)
        okids = np.zeros((cap, 2), dtype=np.int32)
        n = o.lib.oracle_bvh_dump(h, oboxes.ctypes.data, okids.ctypes.data, cap)
    finally:
        o.scene_destroy(h)
    assert n == len(kids)
    assert np.array_equal(boxes, oboxes[:n].astype(np.float64))
    assert np.array_equal(np.where(kids >= 0, -1, ~kids), okids[:n])
    named = {~c for c in kids.ravel() if c < 0}
    lists = {i for i, p in enumerate(flat.prims) if p.kind == A.CR_PRIM_LIST}
    assert lists <= named and not any(flat.prims[i].flags & A.CR_PRIM_MEMBER for i in named)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
def test_refit_with_lists_agrees_with_the_linear_list(renderer, oracles, rt, tag):
    sc = scenes.list_scene(128, 4, frame=1)
    truth, _ = oracles[rt].render_image(sc, seed=SEED, linear_list=True)
    fitted, _ = render(renderer, sc, rt, refit=True)
    stale, _ = render(renderer, sc, rt, refit=False)
    assert (fitted == truth).all(axis=2).mean() >= 0.995
    assert (stale == truth).all(axis=2).mean() < 0.98


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
def test_a_list_is_its_objects_when_nothing_clips(renderer, oracles, rt, tag):
    """Static objects with true boxes: the list element grown by add() and the same objects added one by one give
    the same closest hits (different trees, the same image).  The same objects through HitList::new keep an empty
    box, are scanned only when the ray crosses the ground's box (tests/test_oracle_properties.py) and so lose the
    bounces that start on the ground -- the reference's image, pinned against the oracle."""
    def build(how):
        sc = Scene.new_image(16.0 / 9.0, 96, 1, 360.0, 1)
        cam = sc.scene_cam
        cam.set_samples(4)
        cam.set_max_depth(6)
        cam.look_from((0.0, 2.5, 8.0))
        cam.look_at((0.0, 0.5, 0.0))
        cam.set_vfov(40.0)
        sc.add_element(Sphere.new((0.0, -100.0, 0.0), 100.0, Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)), "ground")
        rs = np.random.RandomState(4)
        objs = [Sphere.new((rs.uniform(-4, 4), 0.3, rs.uniform(-3, 3)), 0.3, Lambertian.new_from_color(tuple(rs.uniform(0.1, 0.9, 3)), 1.0))
                for _ in range(24)]
        if how == "flat":
            for k, o in enumerate(objs):
                sc.add_element(o, f"s{k}")
        elif how == "add":
            l = HitList.default()
            for o in objs:
                l.add(o)
            sc.add_element(l, "l")
        else:
            sc.add_element(HitList.new(objs), "l")
        return sc
    imgs = [render(renderer, build(how), rt)[0] for how in ("flat", "add", "new")]
    assert np.array_equal(imgs[0], imgs[1])
    assert not np.array_equal(imgs[0], imgs[2])
    ref, _ = oracles[rt].render_image(build("new"), seed=SEED)
    assert np.array_equal(imgs[2], ref)
    # refit_boxes re-derives every wrapper box from what is under it, also in a scene without a single key: the
    # empty-box list gets its objects' box and nothing is lost any more (found by scripts/fuzz_campaign.py, seed 80431:
    # the device used to skip the refit when no primitive was keyed)
    sc = build("new")
    fitted, fst = render(renderer, sc, rt, refit=True)
    ref, rst = oracles[rt].render_image(sc, seed=SEED)
    same(fitted, fst, ref, rst)
    assert np.array_equal(fitted, imgs[0])


def test_descriptor_rules(renderer):
    sc = scenes.list_scene(32, 1)
    good = sc.flatten()
    renderer.upload_scene(good)

    def broken(edit):
        flat = sc.flatten()
        edit(flat)
        with pytest.raises(CrucibleError) as e:
            renderer.upload_scene(flat)
        assert e.value.code == A.CR_ERR_INVALID_ARG
    lists = [i for i, p in enumerate(good.prims) if p.kind == A.CR_PRIM_LIST and p.v[1] > 0]
    li = lists[0]

    def past_the_end(f):
        f.prims[li].v[1] = len(f.prims) + 1

    def fractional(f):
        f.prims[li].v[0] = f.prims[li].v[0] + 0.5

    def unflagged_object(f):
        f.prims[int(f.prims[li].v[0])].flags &= ~A.CR_PRIM_MEMBER

    def orphan(f):
        f.prims[li].v[1] -= 1     # the last object keeps its flag but belongs to no list

    def shared(f):
        f.prims[lists[1]].v[0] = f.prims[li].v[0]

    def hidden_list(f):
        f.prims[li].flags |= A.CR_PRIM_HIDDEN

    def list_in_list(f):
        first = int(f.prims[li].v[0])
        f.prims[first].kind = A.CR_PRIM_LIST
    for edit in (past_the_end, fractional, unflagged_object, orphan, shared, hidden_list, list_in_list):
        broken(edit)
    renderer.upload_scene(good)   # the handle still takes a good scene


# ---- BVHWrapper elements (scene/mod.rs:161-163)
@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("refit", [False, True], ids=["stale-boxes", "refit"])
@pytest.mark.parametrize("variant,frame", [("mixed", 0), ("mixed", 1), ("only", 0), ("pair", 0), ("pair", 1), ("small", 0)])
def test_wrapper_elements_bit_exact(renderer, oracles, rt, tag, refit, variant, frame):
    sc = scenes.wrapped_scene(96, 4, frame=frame, variant=variant)
    img, st = render(renderer, sc, rt, refit=refit)
    ref, rst = oracles[rt].render_image(sc, seed=SEED)
    same(img, st, ref, rst)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("env", [{"CRUCIBLE_PIPELINE": "wavefront"}, {"CRUCIBLE_PIPELINE": "queue"}, {"CRUCIBLE_SAMPLE_GRANULAR": "0"},
                                 {"CRUCIBLE_LDS_LIMIT": "0", "CRUCIBLE_LDS_TOP_KB": "0"}, {"CRUCIBLE_LDS_LIMIT": "0", "CRUCIBLE_LDS_TOP_KB": "1"}],
                         ids=["wavefront", "queue", "pixel-granular", "global", "top-1KB"])
def test_wrapper_elements_in_the_other_pipelines(oracles, monkeypatch, rt, tag, env):
    from crucible_amd.renderer import Renderer
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = Renderer(0)
    try:
        sc = scenes.wrapped_scene(80, 3, frame=1)
        img, st = render(r, sc, rt)
        ref, rst = oracles[rt].render_image(sc, seed=SEED)
        same(img, st, ref, rst)
    finally:
        r.close()


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
def test_wrapper_elements_in_the_other_bvh_modes(renderer, oracles, rt, tag, mode):
    """There the wrappers' visible objects are primitives of the one tree."""
    sc = scenes.wrapped_scene(80, 3, frame=1)
    img, st = render(renderer, sc, rt, mode=mode, refit=True)
    tree = renderer.export_bvh(rt)
    flat = sc.flatten()
    named = sorted({~c for c in tree[1].ravel() if c < 0})
    assert named == [i for i, p in enumerate(flat.prims) if p.kind in (A.CR_PRIM_SPHERE, A.CR_PRIM_TRIANGLE) and not (p.flags & A.CR_PRIM_HIDDEN)]
    ref, rst = oracles[rt].render_image(sc, seed=SEED, tree=tree)
    same(img, st, ref, rst)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
def test_a_wrapper_is_its_objects_when_nothing_clips(renderer, oracles, rt, tag):
    """Static objects: wrapped, listed or added one by one, the closest hits are the same; and the reference-mode tree of a
    scene with a wrapper element cannot be exported (its records are not two-children wrappers)."""
    from crucible_amd.scene import BVHWrapper

    def build(how):
        sc = Scene.new_image(16.0 / 9.0, 96, 1, 360.0, 1)
        cam = sc.scene_cam
        cam.set_samples(4)
        cam.set_max_depth(6)
        cam.look_from((0.0, 2.5, 8.0))
        cam.look_at((0.0, 0.5, 0.0))
        cam.set_vfov(40.0)
        sc.add_element(Sphere.new((0.0, -100.0, 0.0), 100.0, Lambertian.new_from_color((0.5, 0.5, 0.5), 1.0)), "ground")
        rs = np.random.RandomState(4)
        objs = [Sphere.new((rs.uniform(-4, 4), 0.3, rs.uniform(-3, 3)), 0.3, Lambertian.new_from_color(tuple(rs.uniform(0.1, 0.9, 3)), 1.0))
                for _ in range(24)]
        if how == "flat":
            for k, o in enumerate(objs):
                sc.add_element(o, f"s{k}")
        else:
            sc.add_element(BVHWrapper.new_wrapper(HitList.new(objs)), "w")
        return sc
    flat, _ = render(renderer, build("flat"), rt)
    wrapped, _ = render(renderer, build("wrapped"), rt)
    assert np.array_equal(flat, wrapped)
    with pytest.raises(CrucibleError) as e:
        renderer.export_bvh(rt)
    assert e.value.code == A.CR_ERR_UNSUPPORTED
