"""the CR_BVH_SAH quality builder, the device-built CR_BVH_LBVH tree, and cr_export_bvh.

The reference builds only its median-split tree (src/objects/bvhwrapper.rs:46-78), so nothing of the reference's
pins the SAH topology.  What is pinned:
  * cr_export_bvh in CR_BVH_REFERENCE mode is the oracle's own tree, wrapper for wrapper;
  * in CR_BVH_SAH mode the device render is bit-exact (image and work counters) against the oracle walking the
    exported tree with BVHWrapper::hit's rules (oracle_set_tree) -- same bar as the parity mode, different tree;
  * the SAH tree is well formed and the image agrees with the parity mode's except for box-grazing rays.
"""
import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene, checkered_spheres, load_teapot, procedural_sky

pytestmark = pytest.mark.gpu

SEED = 0xC0FFEE
REALS = [(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")]
COUNTERS = ("segments", "node_tests", "prim_tests", "texel_fetches")


def upload(renderer, sc, mode):
    sc.bvh_mode = mode
    flat = sc.flatten()
    renderer.upload_scene(flat)
    return flat


def visible(flat):
    return [i for i in range(flat.desc.n_prims) if not (flat.prims[i].flags & A.CR_PRIM_HIDDEN)]


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("build", [lambda: book1_end_scene(1, scene_seed=2, image_width=32, samples=1),
                                   lambda: scenes.mixed_scene(32, 1), lambda: scenes.few_spheres(1),
                                   lambda: scenes.few_spheres(2), lambda: scenes.few_spheres(5)],
                         ids=["book1", "mixed", "one", "two", "five"])
def test_export_reference_mode_is_the_oracle_tree(renderer, oracles, rt, tag, build):
    sc = build()
    flat = upload(renderer, sc, A.CR_BVH_REFERENCE)
    boxes, kids, axis = renderer.export_bvh(rt)
    assert (axis == -1).all()
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
    # the oracle's dump marks wrapper children -1 and names primitives by list index
    mine = np.where(kids >= 0, -1, ~kids)
    assert np.array_equal(mine, okids[:n])


def check_tree(boxes, kids, vis):
    n = len(kids)
    seen = []
    reach = np.zeros(n, dtype=bool)
    reach[0] = True
    for k in range(n):
        assert reach[k], "wrapper not reachable from the root in walk order"
        for c in kids[k]:
            if c >= 0:
                assert k < c < n and not reach[c]
                reach[c] = True
                lo_ok = (boxes[c, 0::2] >= boxes[k, 0::2]).all() and (boxes[c, 1::2] <= boxes[k, 1::2]).all()
                assert lo_ok, "child box not inside its parent's"
        leaf = [~c for c in kids[k] if c < 0]
        assert len(leaf) in (0, 2), "a wrapper holds two wrappers or two primitives (one primitive: named twice)"
        seen += sorted(set(leaf))
    assert sorted(seen) == sorted(vis), "every visible primitive in exactly one leaf"
    assert n <= max(1, 2 * len(vis) - 1)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("build", [lambda: book1_end_scene(1, scene_seed=2, image_width=32, samples=1),
                                   lambda: scenes.mixed_scene(32, 1), lambda: scenes.few_spheres(3),
                                   lambda: scenes.few_spheres(9)], ids=["book1", "mixed", "three", "nine"])
@pytest.mark.parametrize("mode", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
def test_sah_tree_is_well_formed(renderer, rt, tag, build, mode):
    sc = build()
    flat = upload(renderer, sc, mode)
    boxes, kids, axis = renderer.export_bvh(rt)
    check_tree(boxes, kids, visible(flat))
    inner = kids[:, 0] >= 0
    if mode == A.CR_BVH_SAH_ORDERED:
        assert ((axis[inner] >= 0) & (axis[inner] <= 2)).all() and (axis[~inner] == -1).all()
    else:
        assert (axis == -1).all()


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("name,build,seed", [
    ("book1", lambda: book1_end_scene(1, scene_seed=3, image_width=160, samples=6), 77),
    ("checkered", lambda: checkered_spheres(1, image_width=64, samples=4), SEED),
    ("triangles_nosky", lambda: scenes.mixed_scene(40, 3, sky=False), SEED),
    ("few0", lambda: scenes.few_spheres(0), SEED), ("few1", lambda: scenes.few_spheres(1), SEED),
    ("few2", lambda: scenes.few_spheres(2), SEED), ("few3", lambda: scenes.few_spheres(3), SEED),
    ("few9", lambda: scenes.few_spheres(9), SEED),
])
@pytest.mark.parametrize("mode", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
def test_sah_mode_bit_exact_against_oracle_on_the_same_tree(renderer, oracles, rt, tag, name, build, seed, mode):
    sc = build()
    flat = upload(renderer, sc, mode)
    img, st = renderer.render(sc.scene_cam, seed=seed, real_type=rt)
    tree = renderer.export_bvh(rt) if visible(flat) else None
    ref, rst = oracles[rt].render_image(sc, seed=seed, tree=tree)
    assert np.array_equal(img, ref), f"differing px = {(img != ref).any(axis=2).sum()}"
    for k in COUNTERS:
        assert st[k] == rst[k], (k, st[k], rst[k])
    if tree is not None:
        assert st["bvh_entries"] == len(tree[1])
    print(name, tag, mode, "node tests", st["node_tests"])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
def test_sah_teapot_against_oracle(renderer, oracles, rt, tag, mode):
    """6320 triangles + image sky (atan2 / asin are the library's own defined functions, DESIGN.md "software
    trigonometry"): bit-exact like everything else."""
    sc = load_teapot(1, image_width=96, samples=2, sky=procedural_sky(256, 128))
    upload(renderer, sc, mode)
    img, st = renderer.render(sc.scene_cam, seed=5, real_type=rt)
    ref, rst = oracles[rt].render_image(sc, seed=5, tree=renderer.export_bvh(rt))
    assert np.array_equal(img, ref), (img != ref).any(axis=2).sum()
    for k in ("segments", "node_tests", "prim_tests", "texel_fetches"):
        assert st[k] == rst[k], (k, st[k], rst[k])


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
def test_sah_agrees_with_reference_topology(renderer, rt, tag, mode):
    """Same closest hits except for rays grazing a box face: nearly all pixels identical, far fewer box tests."""
    sc = book1_end_scene(1, scene_seed=1, image_width=320, samples=8)
    upload(renderer, sc, A.CR_BVH_REFERENCE)
    ref, rst = renderer.render(sc.scene_cam, seed=SEED, real_type=rt)
    upload(renderer, sc, mode)
    img, st = renderer.render(sc.scene_cam, seed=SEED, real_type=rt)
    same = (img == ref).all(axis=2).mean()
    assert same >= 0.995, same
    assert np.abs(img.astype(np.float64) - ref.astype(np.float64)).mean() < 1e-4
    if mode != A.CR_BVH_LBVH:   # the Morton tree is built for speed of construction, not of traversal
        assert st["node_tests"] < 0.8 * rst["node_tests"], (st["node_tests"], rst["node_tests"])


def test_unknown_bvh_mode_is_rejected(renderer):
    from crucible_amd.renderer import CrucibleError
    sc = scenes.few_spheres(2)
    sc.bvh_mode = 7
    with pytest.raises(CrucibleError):
        renderer.upload_scene(sc.flatten())
    sc.bvh_mode = A.CR_BVH_REFERENCE
    renderer.upload_scene(sc.flatten())


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("scene", ["book1", "teapot", "million", "movie"])
def test_opt_in_trees_render_the_reference_image_on_the_baseline_scenes(renderer, rt, tag, scene):
    """For static primitives with true boxes any tree returns the same closest hits (up to exact ties in t, which
    random scenes do not produce): on the four BASELINE scenes -- book1, teapot + environment map, the 1M-sphere
    field, a movie frame (keyed camera) -- CR_BVH_SAH, _ORDERED and LBVH give the reference topology's frame bit for bit.
    bench.py repeats this check at full size (`opt_in_tree.image_identical_to_reference_topology`)."""
    from crucible_amd.demo_builder import load_teapot, million_spheres, procedural_sky, teapot_orbit_movie
    if scene == "book1":
        sc = book1_end_scene(1, scene_seed=1, image_width=320, samples=8)
    elif scene == "teapot":
        sc = load_teapot(1, image_width=256, samples=6, sky=procedural_sky())
    elif scene == "million":
        sc = million_spheres(1, scene_seed=1, image_width=256, samples=4)
    else:
        sc = teapot_orbit_movie(1, image_width=192, samples=6)
        sc.scene_cam.frame = 37
    frames = []
    for mode in (A.CR_BVH_REFERENCE, A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH):
        upload(renderer, sc, mode)
        img, _ = renderer.render(sc.scene_cam, seed=77, real_type=rt)
        frames.append(img)
    for f in frames[1:]:
        assert np.array_equal(frames[0], f)
