"""ctypes mirror of include/crucible_hip.h (the C ABI).  Plain data only."""
import ctypes as C

CR_ABI_VERSION = 3

CR_OK, CR_ERR_INVALID_ARG, CR_ERR_NO_DEVICE, CR_ERR_HIP, CR_ERR_NO_SCENE, CR_ERR_IO, CR_ERR_NAN, CR_ERR_UNSUPPORTED, CR_ERR_PEER = range(9)
CR_REAL_F32, CR_REAL_F64 = 0, 1
CR_PRIM_SPHERE, CR_PRIM_TRIANGLE, CR_PRIM_LIST, CR_PRIM_BVH = 0, 1, 2, 3
CR_PRIM_HIDDEN, CR_PRIM_MEMBER, CR_LIST_EMPTY_BOX = 1, 2, 4
CR_MAT_LAMBERTIAN, CR_MAT_METAL, CR_MAT_DIELECTRIC = 0, 1, 2
CR_TEX_SOLID, CR_TEX_CHECKER, CR_TEX_IMAGE = 0, 1, 2
CR_SKY_DEFAULT, CR_SKY_SPHERICAL = 0, 1
CR_BVH_REFERENCE, CR_BVH_SAH, CR_BVH_SAH_ORDERED, CR_BVH_LBVH = 0, 1, 2, 3
CR_KEY_TX, CR_KEY_TY, CR_KEY_TZ, CR_KEY_RADIUS, CR_KEY_SCALE_X, CR_KEY_SCALE_Y, CR_KEY_SCALE_Z = 0, 1, 2, 3, 4, 5, 6
CR_MAX_CHECKER_DEPTH = 32
CR_KEY_NERP, CR_KEY_LERP = 0, 1
CR_SUM_DEFAULT, CR_SUM_REFERENCE_ORDER, CR_SUM_RELAXED = 0, 1, 2


class CrPrimitive(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("flags", C.c_int32), ("key_first", C.c_int32),
                ("key_count", C.c_int32), ("_pad", C.c_int32), ("v", C.c_double * 9)]


class CrMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("albedo", C.c_double * 3), ("param", C.c_double)]


class CrTexture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("even", C.c_int32), ("odd", C.c_int32), ("image", C.c_int32),
                ("color", C.c_double * 3), ("inv_scale", C.c_double)]


class CrImage(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb8", C.POINTER(C.c_uint8))]


class CrKeyframe(C.Structure):
    _fields_ = [("channel", C.c_int32), ("interp", C.c_int32), ("t0", C.c_double), ("t1", C.c_double),
                ("a", C.c_double), ("b", C.c_double)]


class CrSceneDesc(C.Structure):
    _fields_ = [("n_prims", C.c_int32), ("n_materials", C.c_int32), ("n_textures", C.c_int32),
                ("n_images", C.c_int32), ("n_keys", C.c_int32), ("sky_kind", C.c_int32), ("sky_image", C.c_int32),
                ("bvh_mode", C.c_int32), ("prims", C.POINTER(CrPrimitive)), ("materials", C.POINTER(CrMaterial)),
                ("textures", C.POINTER(CrTexture)), ("images", C.POINTER(CrImage)), ("keys", C.POINTER(CrKeyframe))]


class CrCameraDesc(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32), ("vfov_degrees", C.c_double),
                ("defocus_angle_degrees", C.c_double), ("focus_dist", C.c_double), ("look_from", C.c_double * 3),
                ("look_at", C.c_double * 3), ("vup", C.c_double * 3), ("from_key_count", C.c_int32),
                ("at_key_count", C.c_int32), ("from_keys", C.POINTER(CrKeyframe)), ("at_keys", C.POINTER(CrKeyframe))]


class CrRenderParams(C.Structure):
    _fields_ = [("samples", C.c_int32), ("sample_begin", C.c_int32), ("sample_count", C.c_int32),
                ("max_depth", C.c_int32), ("seed", C.c_uint64), ("frame", C.c_int32), ("real_type", C.c_int32),
                ("frame_rate", C.c_double), ("shutter_angle", C.c_double), ("output_sum", C.c_int32),
                ("refit_boxes", C.c_int32), ("sum_order", C.c_int32), ("_reserved", C.c_int32)]


class CrStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("node_tests", C.c_uint64),
                ("prim_tests", C.c_uint64), ("texel_fetches", C.c_uint64), ("nan_pixels", C.c_uint64),
                ("kernel_ms", C.c_double), ("upload_ms", C.c_double), ("bvh_entries", C.c_int32),
                ("scene_in_lds", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class CrGroupStats(C.Structure):
    _fields_ = [("render", CrStats), ("reduce_ms", C.c_double), ("members", C.c_int32), ("used_rccl", C.c_int32)]


CR_GROUP_ID_BYTES = 128

# every symbol include/crucible_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "cr_abi_version": (C.c_int32, []),
    "cr_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "cr_destroy": (None, [C.c_void_p]),
    "cr_upload_scene": (C.c_int32, [C.c_void_p, C.POINTER(CrSceneDesc)]),
    "cr_render_device": (C.c_int32, [C.c_void_p, C.POINTER(CrCameraDesc), C.POINTER(CrRenderParams), C.c_void_p,
                                     C.POINTER(CrStats)]),
    "cr_render_host": (C.c_int32, [C.c_void_p, C.POINTER(CrCameraDesc), C.POINTER(CrRenderParams), C.c_void_p,
                                   C.POINTER(CrStats)]),
    "cr_export_bvh": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.POINTER(C.c_int32)]),
    "cr_last_kernel_ms": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double)]),
    "cr_synchronize": (C.c_int32, [C.c_void_p]),
    "cr_stream": (C.c_void_p, [C.c_void_p]),
    "cr_write_ppm": (C.c_int32, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "cr_write_ppm_binary": (C.c_int32, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "cr_write_png": (C.c_int32, [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "cr_quantize_rgb8": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "cr_last_error": (C.c_char_p, [C.c_void_p]),
    "cr_group_shard": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cr_group_create": (C.c_int32, [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]),
    "cr_group_unique_id": (C.c_int32, [C.c_void_p]),
    "cr_group_create_rank": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cr_group_destroy": (None, [C.c_void_p]),
    "cr_group_local_size": (C.c_int32, [C.c_void_p]),
    "cr_group_size": (C.c_int32, [C.c_void_p]),
    "cr_group_rank": (C.c_int32, [C.c_void_p]),
    "cr_group_handle": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "cr_group_upload_scene": (C.c_int32, [C.c_void_p, C.POINTER(CrSceneDesc)]),
    "cr_group_render": (C.c_int32, [C.c_void_p, C.POINTER(CrCameraDesc), C.POINTER(CrRenderParams), C.c_void_p,
                                    C.POINTER(CrGroupStats)]),
    "cr_group_render_host": (C.c_int32, [C.c_void_p, C.POINTER(CrCameraDesc), C.POINTER(CrRenderParams), C.c_void_p,
                                         C.POINTER(CrGroupStats)]),
    "cr_group_last_error": (C.c_char_p, [C.c_void_p]),
}


def bind(lib, symbols=SYMBOLS):
    """Attach restype/argtypes; raises AttributeError if a declared symbol is missing."""
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
