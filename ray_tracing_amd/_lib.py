"""ctypes binding of librt_hip.so (include/rt_hip.h, include/rt_abi.h).

The library is the product; this module only declares its C ABI for Python
callers (tests, bench, the App mirror).  Loading fails loudly when the shared
library has not been built -- there is no Python or CPU stand-in for it.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_hip.so")


class Affine3A(C.Structure):
    _fields_ = [("x_axis", C.c_float * 3), ("_pad0", C.c_float),
                ("y_axis", C.c_float * 3), ("_pad1", C.c_float),
                ("z_axis", C.c_float * 3), ("_pad2", C.c_float),
                ("translation", C.c_float * 3), ("_pad3", C.c_float)]


class Vertex(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("_padding", C.c_uint8 * 4)]


class Tri(C.Structure):
    _fields_ = [("i0", C.c_uint32), ("i1", C.c_uint32), ("i2", C.c_uint32)]


class BoundingBox(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("padding_1", C.c_uint8 * 4),
                ("max", C.c_float * 3), ("padding_2", C.c_uint8 * 4)]


class Bvh(C.Structure):
    _fields_ = [("bounding_box", BoundingBox), ("child_1_or_first_tri", C.c_uint32),
                ("child_2_or_last_tri", C.c_uint32), ("mode", C.c_uint32), ("_pad", C.c_uint32)]


class Object(C.Structure):
    _fields_ = [("bvh_root", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("transform", Affine3A), ("object_id", C.c_uint32), ("_pad", C.c_uint32 * 3)]


class SceneInfo(C.Structure):
    _fields_ = [("num_instances", C.c_uint32), ("num_bvh_nodes", C.c_uint32),
                ("num_triangles", C.c_uint32), ("sun_orientation", C.c_float * 3)]


class CamData(C.Structure):
    _fields_ = [("depth", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("transform", Affine3A),
                ("canvas_width", C.c_uint32), ("canvas_height", C.c_uint32), ("fov", C.c_float),
                ("frame", C.c_uint32), ("debug_number", C.c_uint32), ("debug_information", C.c_uint8),
                ("_pad1", C.c_uint8 * 3), ("frames_without_move", C.c_float), ("_pad2", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("backface_culling", C.c_uint32), ("color", C.c_float * 3),
                ("specular", C.c_float), ("specular_roughness", C.c_float), ("roughness", C.c_float),
                ("ior", C.c_float), ("_pad", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("internal_visits", C.c_uint64),
                ("triangle_tests", C.c_uint64), ("paths", C.c_uint64)]


class SchedStats(C.Structure):
    _fields_ = [("iterations", C.c_uint64 * 3), ("lanes", C.c_uint64 * 3), ("cycles", C.c_uint64 * 3), ("wait_cycles", C.c_uint64 * 3),
                ("site_wait_cycles", C.c_uint64 * 8),
                ("shade_iterations", C.c_uint64), ("shade_slots", C.c_uint64), ("shade_cycles", C.c_uint64)]


class ValuProbe(C.Structure):
    _fields_ = [("valu_insts", C.c_uint64), ("waves", C.c_uint64), ("simds", C.c_uint64),
                ("window_us", C.c_double), ("start_spread_us", C.c_double), ("clock_mhz", C.c_double),
                ("kernel_ms", C.c_double), ("wave_share_min", C.c_double), ("wave_share_max", C.c_double),
                ("simd_cycles_per_inst", C.c_double)]


VALU_MIXES = ("fma", "mul_add", "pk_mul_add", "pk_fma", "min_max", "cndmask", "cmp", "int", "trans", "box_step",
              "min3_max3", "cndmask_vcc", "cmp_vcc", "mov", "shift_cvt", "mul_lo", "dpp", "bfi", "and_or", "salu",
              "valu_salu", "slow_salu")


class BvhParams(C.Structure):
    _fields_ = [("root_leaf_max", C.c_uint32), ("leaf_max", C.c_uint32), ("max_depth", C.c_uint32),
                ("threads", C.c_int32)]


# sizes pinned by the static asserts of rt_abi.h; checked again here so that a
# drift between the header and this file cannot go unnoticed
_SIZES = {Affine3A: 64, Vertex: 16, Tri: 12, BoundingBox: 32, Bvh: 48, Object: 4, Instance: 80,
          SceneInfo: 24, CamData: 112, Material: 40, Stats: 40, BvhParams: 16}
for _t, _n in _SIZES.items():
    assert C.sizeof(_t) == _n, (_t.__name__, C.sizeof(_t), _n)

RT_OK = 0
RT_ERR_INVALID, RT_ERR_HIP, RT_ERR_NO_SCENE, RT_ERR_LIMIT, RT_ERR_IO = -1, -2, -3, -4, -5
RT_OPT_PIPELINE, RT_OPT_PROBE_HITS, RT_OPT_COUNTERS, RT_OPT_PRESENT = 1, 2, 3, 4
RT_OPT_FRAME_STRIDE, RT_OPT_SEED_MODE = 5, 6
RT_OPT_TILE_ORDER, RT_OPT_VOTE_WEIGHTS, RT_OPT_BLOCKS_PER_CU = 7, 8, 9
RT_OPT_FORCE_BIG_OFFSETS, RT_OPT_FAST_DIV, RT_OPT_TILE_DEAL, RT_OPT_POOL_SLOTS = 10, 11, 12, 13
RT_OPT_INSTANCE_CULL = 14
RT_BVH_HOST, RT_BVH_DEVICE_LINEAR, RT_BVH_DEVICE_SAH = 0, 1, 2
RT_GROUP_TILES, RT_GROUP_SAMPLES = 0, 1
RT_MAT_GENERIC, RT_MAT_NORMAL, RT_MAT_EMISSIVE = 0, 1, 2
RT_MAT_REFRACTIVE, RT_MAT_UV, RT_MAT_DIFFUSE, RT_MAT_METAL = 3, 4, 5, 6
RT_DEBUG_NONE, RT_DEBUG_TRIANGLE_INTERSECTION, RT_DEBUG_BVH_INTERSECTION = 0, 1, 2

_vp, _sz, _u32, _i32, _f = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_float
_P = C.POINTER

# name -> (restype, argtypes).  One entry per declaration of include/rt_hip.h.
SIGNATURES = {
    "rt_version": (C.c_char_p, []),

    "rt_device_count": (_i32, [_P(C.c_int)]),
    "rt_create": (_i32, [_i32, _P(_vp)]),
    "rt_destroy": (None, [_vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_set_stream": (_i32, [_vp, _vp]),
    "rt_set_option": (_i32, [_vp, _i32, C.c_int64]),
    "rt_upload_scene": (_i32, [_vp, _P(SceneInfo), _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz]),
    "rt_set_materials": (_i32, [_vp, _vp, _sz]),
    "rt_update_instances": (_i32, [_vp, _vp, _sz]),
    "rt_default_material": (None, [_u32, _P(Material)]),
    "rt_set_rows": (_i32, [_vp, _u32, _u32, _u32]),
    "rt_render_frame": (_i32, [_vp, _P(CamData)]),
    "rt_render_frames": (_i32, [_vp, _P(CamData), _u32]),
    "rt_render_frames_noalloc": (_i32, [_vp, _P(CamData), _u32, _vp, _vp]),
    "rt_synchronize": (_i32, [_vp]),
    "rt_selftest_fastmath_device": (_i32, [C.c_int, _P(C.c_uint64)]),
    "rt_selftest_rcp_device": (_i32, [C.c_int, _P(C.c_uint32)]),
    "rt_probe_valu_issue": (_i32, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, _P(ValuProbe)]),
    "rt_save_png": (_i32, [C.c_char_p, _vp, C.c_uint32, C.c_uint32]),
    "rt_save_exr": (_i32, [C.c_char_p, _vp, C.c_uint32, C.c_uint32, C.c_float]),
    "rt_read_acc": (_i32, [_vp, _vp, _sz]),
    "rt_read_present": (_i32, [_vp, _vp, _sz]),
    "rt_read_present_rgba8": (_i32, [_vp, _vp, _sz]),
    "rt_present_begin": (_i32, [_vp]),
    "rt_present_end": (_i32, [_vp, _P(_vp), _P(_sz)]),
    "rt_write_acc": (_i32, [_vp, _vp, _sz, _u32, _u32]),
    "rt_clear_acc": (_i32, [_vp]),
    "rt_read_hits": (_i32, [_vp, _vp, _vp, _vp, _sz]),
    "rt_read_counters": (_i32, [_vp, _vp, _vp, _sz]),
    "rt_get_stats": (_i32, [_vp, _P(Stats)]),
    "rt_get_sched_stats": (_i32, [_vp, _P(SchedStats)]),
    "rt_reset_stats": (_i32, [_vp]),
    "rt_get_kernel_clock": (_i32, [_vp, _P(C.c_uint64), _P(C.c_uint64)]),
    "rt_acc_device_ptr": (_i32, [_vp, _u32, _u32, _P(_vp)]),
    "rt_last_render_ms": (_i32, [_vp, _P(_f)]),
    "rt_group_create": (_i32, [_P(C.c_int), _sz, _P(_vp)]),
    "rt_group_destroy": (None, [_vp]),
    "rt_group_last_error": (C.c_char_p, [_vp]),
    "rt_group_size": (_sz, [_vp]),
    "rt_group_context": (_vp, [_vp, _sz]),
    "rt_group_upload_scene": (_i32, [_vp, _P(SceneInfo), _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz]),
    "rt_group_upload_scene_from_builder": (_i32, [_vp, _vp]),
    "rt_group_set_materials": (_i32, [_vp, _vp, _sz]),
    "rt_group_update_instances": (_i32, [_vp, _vp, _sz]),
    "rt_group_set_option": (_i32, [_vp, _i32, C.c_int64]),
    "rt_group_render_frames": (_i32, [_vp, _P(CamData), _u32, _i32]),
    "rt_group_synchronize": (_i32, [_vp]),
    "rt_group_read_acc": (_i32, [_vp, _vp, _sz]),
    "rt_group_read_present": (_i32, [_vp, _vp, _sz]),
    "rt_group_acc_device_ptr": (_i32, [_vp, _P(_vp)]),
    "rt_bvh_params_default": (None, [_P(BvhParams)]),
    "rt_parse_obj": (_i32, [C.c_char_p, _sz, _vp, _P(_sz), _vp, _P(_sz)]),
    "rt_create_bvh": (_i32, [_vp, _sz, _vp, _sz, _P(BvhParams), _vp, _P(_sz)]),
    "rt_create_bvh_device": (_i32, [_i32, _vp, _sz, _vp, _sz, C.c_uint32, _vp, _P(_sz)]),
    "rt_create_bvh_device_sah": (_i32, [_i32, _vp, _sz, _vp, _sz, C.c_uint32, C.c_uint32, _vp, _P(_sz)]),
    "rt_create_bvh_device_sah_passes_only": (_i32, [_i32, _vp, _sz, _vp, _sz, C.c_uint32, C.c_uint32, _vp, _P(_sz)]),
    "rt_scene_builder_new": (_vp, []),
    "rt_scene_builder_free": (None, [_vp]),
    "rt_scene_builder_add_obj": (_i32, [_vp, C.c_char_p, _sz, _vp, _sz, _P(BvhParams)]),
    "rt_scene_builder_add_mesh": (_i32, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _P(BvhParams)]),
    "rt_scene_builder_use_bvh_builder": (_i32, [_vp, _i32, _i32]),
    "rt_scene_builder_add_built_mesh": (_i32, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz]),
    "rt_scene_builder_sun_orientation": (None, [_vp, _f, _f, _f]),
    "rt_scene_builder_build": (_i32, [_vp, _P(SceneInfo), _P(_vp), _P(_sz), _P(_vp), _P(_sz), _P(_vp), _P(_sz),
                                      _P(_vp), _P(_sz), _P(_vp), _P(_sz)]),
    "rt_upload_scene_from_builder": (_i32, [_vp, _vp]),
    "rt_affine_from_srt": (None, [_P(_f), _P(_f), _P(_f), _P(Affine3A)]),
    "rt_quat_from_rotation": (None, [_i32, _f, _P(_f)]),
    "rt_affine_inverse": (None, [_P(Affine3A), _P(Affine3A)]),
    "rt_cam_update_pos": (None, [_P(CamData), _P(_f)]),
    "rt_cam_update_mouse": (None, [_P(CamData), _P(_f), _f, _f]),
    "rt_selftest_div_by_rcp": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "rt_selftest_slab_filter": (C.c_uint64, [C.c_uint64, C.c_uint64, _P(C.c_uint64)]),
    "rt_selftest_tri_filter": (C.c_uint64, [C.c_uint64, C.c_uint64, _P(C.c_uint64)]),
    "rt_selftest_instance_cull": (C.c_uint64, [C.c_uint64, C.c_uint64, _P(C.c_uint64)]),
}

_lib = None


def load():
    """Load librt_hip.so and declare every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C ray_tracing_amd/csrc` "
            "(or __graft_entry__.build()).  There is no fallback implementation.")
    # PyTorch ships a HIP runtime of its own.  A process that uses both must end up with ONE
    # runtime: loaded after this library's, PyTorch's finds "no HIP GPUs".  So when PyTorch is
    # installed it goes first; this library then binds to the runtime already in the process.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code
