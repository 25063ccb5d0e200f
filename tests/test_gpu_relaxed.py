"""CrRenderParams.sum_order = CR_SUM_RELAXED (the library default): the SAME paths as the reference order -- same draws,
same walks, hence EQUAL work counters -- with the attenuations multiplied in path order and the finished samples added to
their pixel as 64-bit fixed-point integers (include/crucible_hip.h).  Nothing of the reference pins this mode; it is
pinned against the oracle's reference-order frame: per channel within 1e-12 in f64 (the documented bound is about
(2 * max_depth + samples) * 2^-53: below 1e-13 at depth 50 and 512 samples), equal quantised PPM bytes, and -- because integer adds commute -- frames that do
not depend on scheduling at all: two runs, every tile shape, every workgroup size and any split into shards agree bit
for bit.  The f32 mode is compared with its own f32 oracle within the f32 rounding of a sequential sum."""
import numpy as np
import pytest

import scenes
from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene, load_teapot, million_spheres, procedural_sky
from crucible_amd.renderer import CrucibleError, Renderer, quantize_rgb8

pytestmark = pytest.mark.gpu

SEED = 0xC0FFEE
COUNTERS = ("segments", "node_tests", "prim_tests", "texel_fetches")
REALS = [(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")]
RELAX = A.CR_SUM_RELAXED
TOL = {A.CR_REAL_F64: 1e-12,   # bound: about (2 * 50 + samples) * 2^-53 < 1e-13, means of samples in [0, 1]
       A.CR_REAL_F32: 4e-6}    # the f32 oracle adds its samples sequentially in f32 (n * 2^-24 per channel); relaxed f32 adds exactly


def _movie_frame(frame):
    """One frame of configs[4] (the teapot orbit: keyed camera, the CAMK kernels) at full width, 3 spp."""
    from crucible_amd.demo_builder import teapot_orbit_movie
    sc = teapot_orbit_movie(1, image_width=1920, samples=3)
    sc.scene_cam.frame = frame
    return sc


def relaxed(renderer, sc, rt, **kw):
    renderer.upload_scene(sc.flatten())
    return renderer.render(sc.scene_cam, seed=kw.pop("seed", SEED), real_type=rt, sum_order=RELAX, **kw)


def same_bytes(img, ref):
    """`impl Display for Color` bytes, (255 * sqrt(c)) as u32: equal, except where the reference's own value sits ON a
    byte boundary -- e.g. every sample of a pixel saw the texel k/255 through the attenuation k/255, so 255 * sqrt(c) is
    the integer k up to the last bit and a 1e-16 change of c moves the truncation.  There the bytes may differ by one."""
    qa, qb = quantize_rgb8(img), quantize_rgb8(ref)
    bad = qa != qb
    if bad.any():
        v = 255.0 * np.sqrt(ref[bad].astype(np.float64))
        assert np.abs(v - np.rint(v)).max() < 1e-9 and np.abs(qa[bad].astype(int) - qb[bad].astype(int)).max() == 1
        assert bad.sum() <= max(2, bad.size // 2000), bad.sum()


def check(img, st, ref, rst, rt):
    assert img.dtype == ref.dtype and img.shape == ref.shape
    for k in COUNTERS:
        assert st[k] == rst[k], (k, st[k], rst[k])   # path identity: the product never feeds a branch
    d = np.abs(img.astype(np.float64) - ref.astype(np.float64)).max() if img.size else 0.0
    assert d <= TOL[rt], d
    if rt == A.CR_REAL_F64:
        same_bytes(img, ref)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("name,build", [
    ("book1", lambda: book1_end_scene(1, scene_seed=3, image_width=160, samples=6)),
    ("mixed", lambda: scenes.mixed_scene(64, 4)),
    ("mixed-anim", lambda: scenes.mixed_scene(48, 4, animate=True)),
    ("mixed-nosky", lambda: scenes.mixed_scene(40, 3, sky=False)),
    ("empty", lambda: scenes.few_spheres(0)),
    ("three", lambda: scenes.few_spheres(3)),
])
def test_relaxed_against_the_oracle(renderer, oracles, rt, tag, name, build):
    sc = build()
    img, st = relaxed(renderer, sc, rt)
    ref, rst = oracles[rt].render_image(sc, seed=SEED)
    check(img, st, ref, rst, rt)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("width,samples", [(1, 1), (7, 5), (9, 1), (37, 3), (100, 7), (64, 100)])
def test_relaxed_ragged_sizes_and_sample_counts(renderer, oracles, rt, tag, width, samples):
    """Edge tiles, sample counts that are no multiple of the 4-sample group, chunks that straddle tiles."""
    sc = book1_end_scene(1, scene_seed=1, image_width=width, samples=samples)
    img, st = relaxed(renderer, sc, rt)
    ref, rst = oracles[rt].render_image(sc, seed=SEED)
    check(img, st, ref, rst, rt)


@pytest.mark.parametrize("depth", [0, 1, 2, 50])
def test_relaxed_depth_limits(renderer, o64, depth):
    sc = book1_end_scene(1, scene_seed=1, image_width=40, samples=3)
    sc.scene_cam.set_max_depth(depth)
    img, st = relaxed(renderer, sc, A.CR_REAL_F64)
    ref, rst = o64.render_image(sc, seed=SEED)
    check(img, st, ref, rst, A.CR_REAL_F64)
    if depth == 0:
        assert not img.any()


def test_relaxed_baseline_config0(renderer, o64):
    """BASELINE configs[0] (book1 400x225 @ 16 spp, depth 50) in the default mode, whole frame against the oracle."""
    sc = book1_end_scene(1, scene_seed=1, image_width=400, samples=16)
    sc.scene_cam.set_max_depth(50)
    img, st = relaxed(renderer, sc, A.CR_REAL_F64)
    ref, rst = o64.render_image(sc, seed=SEED)
    check(img, st, ref, rst, A.CR_REAL_F64)


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
def test_relaxed_is_deterministic_and_shards_add_up(renderer, rt, tag):
    """Integer sums: two runs are bit-identical; the raw sums of a split of the sample indices add up to the whole within
    the one rounding each u64 -> real conversion makes (a sum of n samples has up to 52 + log2(n) bits)."""
    sc = book1_end_scene(1, scene_seed=1, image_width=128, samples=12)
    cam = sc.scene_cam
    renderer.upload_scene(sc.flatten())
    a, sa = renderer.render(cam, seed=SEED, real_type=rt, sum_order=RELAX)
    b, sb = renderer.render(cam, seed=SEED, real_type=rt, sum_order=RELAX)
    assert np.array_equal(a, b) and all(sa[k] == sb[k] for k in COUNTERS)
    whole, _ = renderer.render(cam, seed=SEED, real_type=rt, sum_order=RELAX, output_sum=True)
    parts = [renderer.render(cam, seed=SEED, real_type=rt, sum_order=RELAX, sample_begin=s0, sample_count=n, output_sum=True)[0]
             for s0, n in ((0, 5), (5, 0), (5, 4), (9, 3))]
    if rt == A.CR_REAL_F64:
        assert np.abs(parts[0] + parts[1] + parts[2] + parts[3] - whole).max() <= 12 * 2.0 ** -50
        assert np.array_equal(whole / 12.0, a)
    else:
        assert np.abs(parts[0] + parts[2] + parts[3] - whole).max() <= 12 * 2.0 ** -22
    assert not parts[1].any()


@pytest.mark.parametrize("env", [{"CRUCIBLE_SG_TILE": "8x8"}, {"CRUCIBLE_SG_TILE": "2x2"}, {"CRUCIBLE_SG_TILE": "1x1", "CRUCIBLE_BLOCK": "512"},
                                 {"CRUCIBLE_BLOCK": "256"}, {"CRUCIBLE_SG_CHUNK": "64"}, {"CRUCIBLE_WALK_EXIT": "24"},
                                 {"CRUCIBLE_LDS_LIMIT": "4096"}, {"CRUCIBLE_LDS_LIMIT": "4096", "CRUCIBLE_LDS_TOP_KB": "0"},
                                 {"CRUCIBLE_LATENCY_ENTRIES": "1", "CRUCIBLE_LDS_LIMIT": "4096"}],
                         ids=["tile-8x8", "tile-2x2", "tile-1x1", "block-256", "chunk-64", "walk-exit-24", "tree-top-in-lds", "scene-in-global-memory",
                              "six-waves-per-simd"])
def test_relaxed_frame_does_not_depend_on_scheduling(renderer, monkeypatch, env):
    """Tile shapes, workgroup sizes, chunk sizes, residencies and the 6-wave entry point change which wave adds what
    when, and whether through its LDS slot or straight to the global sums -- never the frame."""
    scs = [book1_end_scene(1, scene_seed=1, image_width=100, samples=5), scenes.mixed_scene(56, 4, animate=True)]
    want = []
    for sc in scs:
        for rt, _ in REALS:
            want.append(relaxed(renderer, sc, rt))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    r = Renderer(0)
    try:
        i = 0
        for sc in scs:
            for rt, _ in REALS:
                img, st = relaxed(r, sc, rt)
                assert np.array_equal(img, want[i][0]) and all(st[k] == want[i][1][k] for k in COUNTERS)
                i += 1
    finally:
        r.close()


@pytest.mark.parametrize("limit", ["50000", "23000"], ids=["8-sample-launches", "4-sample-launches"])
def test_relaxed_sums_survive_several_launches(renderer, monkeypatch, limit):
    """A frame with more work items than one launch's 32-bit counter may hand out (a 4K frame beyond ~480 spp) runs as consecutive
    launches that keep adding to the same per-pixel sums.  CRUCIBLE_WORK_COUNTER_MAX shrinks the counter's range so that a 100 x 56
    frame at 21 spp needs three to six launches: the frame and the counters must be those of the one-launch render, bit for bit --
    also in the parity mode, whose batches it cuts the same way."""
    sc = book1_end_scene(1, scene_seed=1, image_width=100, samples=21)
    want = {}
    for rt, _ in REALS:
        want[rt] = (relaxed(renderer, sc, rt), (renderer.render(sc.scene_cam, seed=SEED, real_type=rt, sum_order=A.CR_SUM_REFERENCE_ORDER)))
    monkeypatch.setenv("CRUCIBLE_WORK_COUNTER_MAX", limit)
    r = Renderer(0)
    try:
        r.upload_scene(sc.flatten())
        for rt, _ in REALS:
            for k, order in enumerate((RELAX, A.CR_SUM_REFERENCE_ORDER)):
                img, st = r.render(sc.scene_cam, seed=SEED, real_type=rt, sum_order=order)
                assert np.array_equal(img, want[rt][k][0]) and all(st[c] == want[rt][k][1][c] for c in COUNTERS), (rt, order)
    finally:
        r.close()


def test_relaxed_is_the_library_default(monkeypatch, o64):
    """CR_SUM_DEFAULT resolves to the relaxed sums unless CRUCIBLE_SUM_ORDER=reference (which the test session sets for
    the bit-exact tests); the alternative pipelines are reference-order only."""
    sc = book1_end_scene(1, scene_seed=1, image_width=64, samples=9)
    ref, _ = o64.render_image(sc, seed=SEED)
    out = {}
    for mode in ("relaxed", "reference", None):
        if mode is None:
            monkeypatch.delenv("CRUCIBLE_SUM_ORDER", raising=False)
        else:
            monkeypatch.setenv("CRUCIBLE_SUM_ORDER", mode)
        r = Renderer(0)
        try:
            r.upload_scene(sc.flatten())
            out[mode] = r.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=A.CR_SUM_DEFAULT)[0]
            explicit = r.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=A.CR_SUM_REFERENCE_ORDER)[0]
            assert np.array_equal(explicit, ref)
        finally:
            r.close()
    assert np.array_equal(out["reference"], ref)
    assert np.array_equal(out[None], out["relaxed"]) and np.abs(out["relaxed"] - ref).max() <= 1e-12
    assert not np.array_equal(out["relaxed"], ref)   # a 9-term sum in another order: some last bits do differ
    monkeypatch.setenv("CRUCIBLE_PIPELINE", "wavefront")
    r = Renderer(0)
    try:
        r.upload_scene(sc.flatten())
        assert np.array_equal(r.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=A.CR_SUM_DEFAULT)[0], ref)
        with pytest.raises(CrucibleError) as e:
            r.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=RELAX)
        assert e.value.code == A.CR_ERR_UNSUPPORTED
    finally:
        r.close()
    with pytest.raises(CrucibleError) as e:
        renderer_bad = Renderer(0)
        try:
            renderer_bad.upload_scene(sc.flatten())
            renderer_bad.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=7)
        finally:
            renderer_bad.close()
    assert e.value.code == A.CR_ERR_INVALID_ARG


def test_relaxed_nan_policy(renderer):
    """A colour that is not a number sets the pixel's NaN flag: the frame reports CR_ERR_NAN like the reference order
    (the reference itself panics in Color::new)."""
    sc = scenes.few_spheres(2, width=24, samples=2)
    sc.scene_cam.look_from((1.0, 2.0, 3.0))
    sc.scene_cam.look_at((1.0, 2.0, 3.0))
    renderer.upload_scene(sc.flatten())
    for rt, _ in REALS:
        with pytest.raises(CrucibleError) as e:
            renderer.render(sc.scene_cam, seed=SEED, real_type=rt, sum_order=RELAX)
        assert e.value.code == A.CR_ERR_NAN
        sums, _ = renderer.render(sc.scene_cam, seed=SEED, real_type=rt, sum_order=RELAX, output_sum=True)
        assert np.isnan(sums).all()


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("bvh", [A.CR_BVH_SAH, A.CR_BVH_SAH_ORDERED, A.CR_BVH_LBVH], ids=["sah", "ordered", "lbvh"])
def test_relaxed_with_the_opt_in_trees(renderer, rt, tag, bvh):
    """Same frame and counters as the reference order on the same tree (the GPU's own reference-order render, which
    tests/test_gpu_bvh_modes.py pins against the oracle walking the exported tree)."""
    sc = book1_end_scene(1, scene_seed=2, image_width=96, samples=5)
    sc.bvh_mode = bvh
    renderer.upload_scene(sc.flatten())
    ref, rst = renderer.render(sc.scene_cam, seed=SEED, real_type=rt, sum_order=A.CR_SUM_REFERENCE_ORDER)
    img, st = renderer.render(sc.scene_cam, seed=SEED, real_type=rt, sum_order=RELAX)
    check(img, st, ref, rst, rt)


@pytest.mark.parametrize("name,build,rows", [
    ("C2-book1-1080p", lambda: book1_end_scene(1, scene_seed=1, image_width=1920, samples=8), (0, 540, 1079)),
    ("C3-teapot-1080p", lambda: load_teapot(1, image_width=1920, samples=4, sky=procedural_sky()), (300, 700)),
    ("C4-million-4k", lambda: million_spheres(1, scene_seed=1, image_width=3840, samples=2), (1200,)),
    ("C5-movie-frame-1080p", lambda: _movie_frame(97), (450,)),
])
def test_relaxed_full_size_frames(renderer, o64, name, build, rows):
    """The BASELINE frames at full size: equal counters and equal PPM bytes against the GPU's reference-order render of the
    same frame, within 1e-12 per channel, and rows of it against the oracle directly."""
    sc = build()
    cam = sc.scene_cam
    renderer.upload_scene(sc.flatten())
    ref, rst = renderer.render(cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=A.CR_SUM_REFERENCE_ORDER)
    img, st = renderer.render(cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=RELAX)
    check(img, st, ref, rst, A.CR_REAL_F64)
    assert img.min() >= 0.0 and img.max() <= 1.0 and st["nan_pixels"] == 0
    h = o64.scene_create(sc.flatten())
    try:
        for row in rows:
            want, _ = o64.render(h, cam, seed=SEED, pix_begin=row * cam.image_width, pix_end=(row + 1) * cam.image_width)
            assert np.abs(img[row] - want).max() <= 1e-12, row
    finally:
        o64.scene_destroy(h)


def test_relaxed_headline_sample_count(renderer, o64):
    """512 samples per pixel (the headline's count; S = 52 holds up to 2047): two rows of the 1920-wide book1 frame against
    the oracle's 512-term sequential sums."""
    sc = book1_end_scene(1, scene_seed=1, image_width=1920, samples=512)
    cam = sc.scene_cam
    renderer.upload_scene(sc.flatten())
    img, st = renderer.render(cam, seed=SEED, real_type=A.CR_REAL_F64, sum_order=RELAX)
    assert st["samples"] == 1920 * 1080 * 512 and st["nan_pixels"] == 0
    h = o64.scene_create(sc.flatten())
    try:
        for row in (3, 600):
            want, _ = o64.render(h, cam, seed=SEED, pix_begin=row * 1920, pix_end=(row + 1) * 1920)
            assert np.abs(img[row] - want).max() <= 1e-12, row
            same_bytes(img[row:row + 1], want.reshape(1, 1920, 3))
    finally:
        o64.scene_destroy(h)


def test_relaxed_many_samples_scale_down(renderer, o64):
    """Beyond 2047 samples per pixel the fixed-point scale drops (2^51 at 2048 ...), so the 64-bit sums cannot overflow."""
    sc = book1_end_scene(1, scene_seed=1, image_width=16, samples=5000)
    img, st = relaxed(renderer, sc, A.CR_REAL_F64)
    ref, rst = o64.render_image(sc, seed=SEED)
    check(img, st, ref, rst, A.CR_REAL_F64)


def test_relaxed_needs_no_sample_buffer():
    """The point of the mode for a drop-in: a 1080p render at 64 spp grows a fresh handle by tens of MB (24 B per pixel
    plus the scene), not by the 3 GiB of per-sample colours the reference order keeps."""
    import torch
    sc = book1_end_scene(1, scene_seed=1, image_width=1920, samples=64)
    r = Renderer(0)
    try:
        r.upload_scene(sc.flatten())
        out = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        r.render_device(sc.scene_cam, out.data_ptr(), seed=SEED, real_type=A.CR_REAL_F64, sum_order=RELAX, want_stats=True)
        free1, _ = torch.cuda.mem_get_info()
        assert free0 - free1 < 256 << 20, (free0 - free1) >> 20
    finally:
        r.close()
