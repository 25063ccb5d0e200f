"""ctypes binding of the multi-GPU entry points (include/crucible_hip.h, "several GPUs of one node").

A RenderGroup is the device-side replacement of the reference's worker pool (src/camera/cpu_threading.rs:25-115)
across GPUs: every member renders a range of sample indices of every pixel, one RCCL reduce adds the sums on the
root, the root divides by the sample count.  Two ways to build one:

    RenderGroup.local([0, 1, ...])            one process, several devices (ncclCommInitAll)
    RenderGroup.rank(device, rank, world, id)  one process per GPU (ncclCommInitRank); `id` = RenderGroup.unique_id()
                                               made on rank 0 and sent to every rank by the launcher's own means
"""
import ctypes as C

from . import _abi as A
from .renderer import CrucibleError, load_library


def shard(samples, member, n_members):
    """(begin, count) of `member`; cr_group_shard, callable without a GPU."""
    lib = load_library()
    b, n = C.c_int32(), C.c_int32()
    rc = lib.cr_group_shard(samples, member, n_members, C.byref(b), C.byref(n))
    if rc != A.CR_OK:
        raise ValueError("cr_group_shard: invalid arguments")
    return b.value, n.value


class RenderGroup:
    def __init__(self, handle):
        self.lib = load_library()
        self.g = handle

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = (C.c_uint8 * A.CR_GROUP_ID_BYTES)()
        rc = lib.cr_group_unique_id(buf)
        if rc != A.CR_OK:
            raise CrucibleError(rc, (lib.cr_group_last_error(None) or b"").decode())
        return bytes(buf)

    @classmethod
    def local(cls, device_ids):
        lib = load_library()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        g = C.c_void_p()
        rc = lib.cr_group_create(ids, len(device_ids), C.byref(g))
        if rc != A.CR_OK:
            raise CrucibleError(rc, (lib.cr_group_last_error(None) or b"").decode())
        return cls(g)

    @classmethod
    def rank(cls, device_id, rank, world, unique_id=None):
        lib = load_library()
        buf = (C.c_uint8 * A.CR_GROUP_ID_BYTES)(*unique_id) if unique_id is not None else None
        g = C.c_void_p()
        rc = lib.cr_group_create_rank(device_id, rank, world, buf, C.byref(g))
        if rc != A.CR_OK:
            raise CrucibleError(rc, (lib.cr_group_last_error(None) or b"").decode())
        return cls(g)

    def _check(self, rc):
        if rc != A.CR_OK:
            raise CrucibleError(rc, (self.lib.cr_group_last_error(self.g) or b"").decode())

    def close(self):
        if self.g:
            self.lib.cr_group_destroy(self.g)
            self.g = None

    def __del__(self):
        if getattr(self, "g", None):
            self.close()

    @property
    def size(self):
        return self.lib.cr_group_size(self.g)

    @property
    def local_size(self):
        return self.lib.cr_group_local_size(self.g)

    @property
    def first_rank(self):
        return self.lib.cr_group_rank(self.g)

    def upload_scene(self, flat):
        self._check(self.lib.cr_group_upload_scene(self.g, C.byref(flat.desc)))

    def render(self, cam, *, seed, real_type=A.CR_REAL_F64, sum_order=A.CR_SUM_DEFAULT):
        """Collective: returns (image (H, W, 3) on the root -- zeros elsewhere --, stats dict)."""
        import numpy as np
        cd = cam.desc()
        p = cam.params(seed, real_type, sum_order=sum_order)
        out = np.zeros((cam.image_height, cam.image_width, 3), dtype=np.float64 if real_type == A.CR_REAL_F64 else np.float32)
        st = A.CrGroupStats()
        self._check(self.lib.cr_group_render_host(self.g, C.byref(cd), C.byref(p), out.ctypes.data_as(C.c_void_p), C.byref(st)))
        d = st.render.as_dict()
        d.update(reduce_ms=st.reduce_ms, members=st.members, used_rccl=st.used_rccl)
        return out, d

    def render_device(self, cam, d_ptr, *, seed, real_type=A.CR_REAL_F64, want_stats=True, sum_order=A.CR_SUM_DEFAULT):
        """Collective: the per-pixel mean lands in device memory at d_ptr on the root member's device."""
        cd = cam.desc()
        p = cam.params(seed, real_type, sum_order=sum_order)
        st = A.CrGroupStats()
        self._check(self.lib.cr_group_render(self.g, C.byref(cd), C.byref(p), C.c_void_p(d_ptr),
                                             C.byref(st) if want_stats else None))
        if not want_stats:
            return None
        d = st.render.as_dict()
        d.update(reduce_ms=st.reduce_ms, members=st.members, used_rccl=st.used_rccl)
        return d
