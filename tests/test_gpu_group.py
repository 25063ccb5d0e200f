"""SURVEY 8(e) behind the C ABI: cr_group_* -- samples-per-pixel sharding + one RCCL reduce inside the library.
The GPU box has one device, so what can run here is: a one-member group (must be bit-identical to
cr_render_device), and the same group with the collective forced on (ncclCommInitAll / ncclCommInitRank with one
rank, ncclReduce in place, the divide kernel) -- which exercises the dlopen'ed RCCL table, the call sequence and the
sum -> mean step on real hardware.  More members are covered by the shard arithmetic (tests/test_abi.py), the gloo
test of the reduce (tests/test_distributed_cpu.py) and the shard-sum tests of test_gpu_parity.py."""
import numpy as np
import pytest

from crucible_amd import _abi as A
from crucible_amd.demo_builder import book1_end_scene
from crucible_amd.group import RenderGroup
from crucible_amd.renderer import CrucibleError

pytestmark = pytest.mark.gpu

REALS = [(A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")]
COUNTERS = ("segments", "node_tests", "prim_tests", "texel_fetches")
SEED = 0xC0FFEE


def group_image(g, sc, rt):
    import torch
    cam = sc.scene_cam
    t = torch.full((cam.image_height, cam.image_width, 3), -1.0, dtype=torch.float64 if rt == A.CR_REAL_F64 else torch.float32,
                   device="cuda:0")
    g.upload_scene(sc.flatten())
    st = g.render_device(cam, t.data_ptr(), seed=SEED, real_type=rt)
    return t.cpu().numpy(), st


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("mode", ["local", "local+rccl", "rank", "rank+rccl"])
def test_one_member_group_equals_render_device(renderer, oracles, monkeypatch, rt, tag, mode):
    sc = book1_end_scene(1, scene_seed=1, image_width=96, samples=5)
    single, sst = (renderer.upload_scene(sc.flatten()), renderer.render(sc.scene_cam, seed=SEED, real_type=rt))[1]
    ref, rst = oracles[rt].render_image(sc, seed=SEED)
    if mode.endswith("rccl"):
        monkeypatch.setenv("CRUCIBLE_GROUP_FORCE_RCCL", "1")
    g = RenderGroup.local([0]) if mode.startswith("local") else RenderGroup.rank(0, 0, 1, RenderGroup.unique_id() if mode.endswith("rccl") else None)
    try:
        assert (g.size, g.local_size, g.first_rank) == (1, 1, 0)
        img, st = group_image(g, sc, rt)
        assert st["used_rccl"] == (1 if mode.endswith("rccl") else 0) and st["members"] == 1
        assert np.array_equal(img, single) and np.array_equal(img, ref)      # sum of one shard / samples == the mean
        for k in COUNTERS:
            assert st[k] == sst[k] == rst[k], k
        assert st["samples"] == 96 * 54 * 5 and st["kernel_ms"] > 0
        again, _ = group_image(g, sc, rt)
        assert np.array_equal(again, img)
        host, hst = g.render(sc.scene_cam, seed=SEED, real_type=rt)        # cr_group_render_host
        assert np.array_equal(host, img) and hst["nan_pixels"] == 0
    finally:
        g.close()


def test_group_handles_drive_frame_sharding(renderer):
    """Movies shard whole frames (frame f -> member f % G, scene/mod.rs:307-316): the member handle is an ordinary
    CrHandle."""
    import ctypes as C
    import torch
    sc = book1_end_scene(1, scene_seed=1, image_width=64, samples=2)
    g = RenderGroup.local([0])
    try:
        g.upload_scene(sc.flatten())
        h = g.lib.cr_group_handle(g.g, 0)
        assert h and not g.lib.cr_group_handle(g.g, 1)
        cam = sc.scene_cam
        t = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.float32, device="cuda:0")
        cd, p = cam.desc(), cam.params(SEED, A.CR_REAL_F32)
        assert g.lib.cr_render_device(h, C.byref(cd), C.byref(p), C.c_void_p(t.data_ptr()), None) == A.CR_OK
        assert g.lib.cr_synchronize(h) == A.CR_OK
        renderer.upload_scene(sc.flatten())
        ref, _ = renderer.render(cam, seed=SEED, real_type=A.CR_REAL_F32)
        assert np.array_equal(t.cpu().numpy(), ref)
    finally:
        g.close()


@pytest.mark.parametrize("rt,tag", REALS, ids=["f64", "f32"])
@pytest.mark.parametrize("members,spp", [(2, 6), (4, 10), (8, 5)])
def test_several_members_on_one_device(renderer, monkeypatch, rt, tag, members, spp):
    """The whole multi-member path on the one GPU of this box: CRUCIBLE_GROUP_SAME_DEVICE lets the members share the
    device and replaces only the ncclReduce by an add kernel (RCCL refuses two ranks on one device).  Shard split,
    per-member handles and streams, sum in member order, divide by spp: the result equals the shard sums added in the
    same order bit for bit, and the 1-GPU image within re-association error; (8, 5) has empty shards."""
    from crucible_amd.group import shard
    monkeypatch.setenv("CRUCIBLE_GROUP_SAME_DEVICE", "1")
    sc = book1_end_scene(1, scene_seed=1, image_width=80, samples=spp)
    cam = sc.scene_cam
    renderer.upload_scene(sc.flatten())
    total = None
    for m in range(members):
        b, n = shard(spp, m, members)
        part, _ = renderer.render(cam, seed=SEED, real_type=rt, sample_begin=b, sample_count=n, output_sum=True)
        total = part.copy() if total is None else total + part
    expect = total / total.dtype.type(spp)
    single, sst = renderer.render(cam, seed=SEED, real_type=rt)
    g = RenderGroup.local([0] * members)
    try:
        assert (g.size, g.local_size) == (members, members)
        g.upload_scene(sc.flatten())
        img, st = g.render(cam, seed=SEED, real_type=rt)
        assert st["members"] == members and st["used_rccl"] == 0
        assert np.array_equal(img, expect)
        assert np.abs(img.astype(np.float64) - single).max() < (1e-5 if rt == A.CR_REAL_F32 else 1e-14)
        for k in COUNTERS:
            assert st[k] == sst[k], k      # the union of the shards is the 1-GPU sample set
        assert st["samples"] == 80 * 45 * spp
    finally:
        g.close()


@pytest.mark.parametrize("mode,failing", [("same-device-4", 2), ("same-device-4", 0), ("rank+rccl", 0), ("local+rccl", 0)])
def test_a_failing_member_fails_the_whole_group_without_a_hang(monkeypatch, mode, failing):
    """One member's render fails after it was launched (CRUCIBLE_GROUP_FAIL_MEMBER, a test hook): every member still
    reaches the agreement step -- with a communicator, the 4-byte ncclAllReduce(min) on the render streams -- nobody
    enters the ncclReduce, the call returns that member's error, and the SAME group renders correctly afterwards."""
    import torch
    sc = book1_end_scene(1, scene_seed=1, image_width=64, samples=8)
    if mode.startswith("same-device"):
        monkeypatch.setenv("CRUCIBLE_GROUP_SAME_DEVICE", "1")
        g = RenderGroup.local([0] * 4)
    else:
        monkeypatch.setenv("CRUCIBLE_GROUP_FORCE_RCCL", "1")
        g = RenderGroup.local([0]) if mode.startswith("local") else RenderGroup.rank(0, 0, 1, RenderGroup.unique_id())
    try:
        g.upload_scene(sc.flatten())
        good, st = g.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64)
        monkeypatch.setenv("CRUCIBLE_GROUP_FAIL_MEMBER", str(failing))
        torch.cuda.set_device(0)
        with pytest.raises(CrucibleError) as e:
            g.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64)
        assert e.value.code == A.CR_ERR_HIP and "injected" in str(e.value)
        t = torch.zeros((sc.scene_cam.image_height, sc.scene_cam.image_width, 3), dtype=torch.float64, device="cuda:0")
        with pytest.raises(CrucibleError):
            g.render_device(sc.scene_cam, t.data_ptr(), seed=SEED, real_type=A.CR_REAL_F64)
        monkeypatch.delenv("CRUCIBLE_GROUP_FAIL_MEMBER")
        again, st2 = g.render(sc.scene_cam, seed=SEED, real_type=A.CR_REAL_F64)       # the group is still usable
        assert np.array_equal(again, good) and all(st[k] == st2[k] for k in COUNTERS)
    finally:
        g.close()


def test_group_argument_errors_are_found_before_anything_is_launched(monkeypatch):
    """Bad arguments fail on every member identically, before a launch: no scene, a bad sample count, a missing output."""
    monkeypatch.setenv("CRUCIBLE_GROUP_SAME_DEVICE", "1")
    sc = book1_end_scene(1, scene_seed=1, image_width=32, samples=4)
    g = RenderGroup.local([0, 0])
    try:
        with pytest.raises(CrucibleError) as e:
            g.render(sc.scene_cam, seed=SEED)
        assert e.value.code == A.CR_ERR_NO_SCENE
        g.upload_scene(sc.flatten())
        cd, p = sc.scene_cam.desc(), sc.scene_cam.params(SEED, A.CR_REAL_F64)
        import ctypes as C
        assert g.lib.cr_group_render(g.g, C.byref(cd), C.byref(p), None, None) == A.CR_ERR_INVALID_ARG       # root without an output buffer
        assert g.lib.cr_group_render_host(g.g, C.byref(cd), C.byref(p), None, None) == A.CR_ERR_INVALID_ARG
        p.samples = 0
        out = np.zeros((sc.scene_cam.image_height, sc.scene_cam.image_width, 3))
        assert g.lib.cr_group_render_host(g.g, C.byref(cd), C.byref(p), out.ctypes.data_as(C.c_void_p), None) == A.CR_ERR_INVALID_ARG
        img, _ = g.render(sc.scene_cam, seed=SEED)
        assert img.max() > 0
    finally:
        g.close()


def test_group_calls_restore_the_callers_device(monkeypatch):
    """The cr_group_* entry points visit every member's device; the calling thread's current device is restored."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    monkeypatch.setenv("CRUCIBLE_GROUP_SAME_DEVICE", "1")
    sc = book1_end_scene(1, scene_seed=1, image_width=32, samples=4)
    dev = C.c_int(-1)
    assert hip.hipSetDevice(0) == 0
    g = RenderGroup.local([0, 0])
    try:
        g.upload_scene(sc.flatten())
        g.render(sc.scene_cam, seed=SEED)
        assert hip.hipGetDevice(C.byref(dev)) == 0 and dev.value == 0
    finally:
        g.close()
