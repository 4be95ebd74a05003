"""MI355X-native trace path for the iKramp/Ray_tracing renderer.

`librt_hip.so` (ray_tracing_amd/csrc, C ABI in include/rt_hip.h) is the
product: hand-written HIP kernels for gfx950 behind the reference's own
scene/camera structs.  This package is the thin host side above it.
"""
from . import _lib
from ._lib import (CamData, Material, RtError, SceneInfo, RT_DEBUG_BVH_INTERSECTION, RT_DEBUG_NONE,
                   RT_DEBUG_TRIANGLE_INTERSECTION, RT_MAT_DIFFUSE, RT_MAT_EMISSIVE, RT_MAT_GENERIC,
                   RT_MAT_METAL, RT_MAT_NORMAL, RT_MAT_REFRACTIVE, RT_MAT_UV,
                   RT_OPT_COUNTERS, RT_OPT_FRAME_STRIDE, RT_OPT_PIPELINE, RT_OPT_PRESENT, RT_OPT_PROBE_HITS,
                   RT_OPT_SEED_MODE)
from .app import App, Group
from .scene import (Scene, SceneBuilder, default_scene, make_cam, sphere_field_scene, terrain_scene)

__all__ = ["App", "Group", "Scene", "SceneBuilder", "CamData", "Material", "SceneInfo", "RtError", "default_scene",
           "make_cam", "sphere_field_scene", "terrain_scene"]
