"""The C ABI boundary: the headers compile as C and C++ (their static asserts pin
every size and offset of the reference's shared structs), the library loads,
exports every symbol include/rt_hip.h declares, and refuses to work without a
GPU instead of falling back to anything.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from ray_tracing_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")


def declared_functions():
    text = open(os.path.join(INCLUDE, "rt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_headers_compile_as_c_and_cpp_and_pin_the_layout():
    probe = r'''
    #include "rt_hip.h"
    #include <stdio.h>
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rt_affine3a), sizeof(rt_vertex), sizeof(rt_tri),
               sizeof(rt_bounding_box), sizeof(rt_bvh), sizeof(rt_object), sizeof(rt_instance), sizeof(rt_scene_info),
               sizeof(rt_cam_data), sizeof(rt_material));
        printf("%zu %zu %zu %zu %zu\n", offsetof(rt_cam_data, transform), offsetof(rt_cam_data, canvas_width),
               offsetof(rt_cam_data, debug_information), offsetof(rt_cam_data, frames_without_move),
               offsetof(rt_instance, object_id));
        return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        for compiler, name, std in (("gcc", "p.c", "-std=c11"), ("g++", "p.cpp", "-std=c++17")):
            src = os.path.join(d, name)
            open(src, "w").write(probe)
            exe = os.path.join(d, name + ".out")
            subprocess.check_call([compiler, std, "-Wall", "-Werror", "-I", INCLUDE, src, "-o", exe])
            out = subprocess.check_output([exe]).decode().split("\n")
            # the sizes rustc gives the structs of shared/src/lib.rs:66-183 on x86-64
            assert out[0].split() == ["64", "16", "12", "32", "48", "4", "80", "24", "112", "40"]
            assert out[1].split() == ["16", "80", "100", "104", "64"]
    # and the ctypes mirrors agree
    assert C.sizeof(L.CamData) == 112 and L.CamData.transform.offset == 16 and L.CamData.canvas_width.offset == 80
    assert L.CamData.debug_information.offset == 100 and L.CamData.frames_without_move.offset == 104
    assert L.Instance.object_id.offset == 64 and L.Bvh.mode.offset == 40 and L.Bvh.child_1_or_first_tri.offset == 32
    assert L.SceneInfo.sun_orientation.offset == 12


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in rt_hip.h but not exported"
    # and the Python binding declares a signature for each of them, and nothing else
    assert sorted(L.SIGNATURES) == names


def test_no_torch_or_cxx_types_in_the_boundary():
    text = open(os.path.join(INCLUDE, "rt_hip.h")).read() + open(os.path.join(INCLUDE, "rt_abi.h")).read()
    assert "std::" not in text and "torch" not in text and "hip/" not in text
    assert 'extern "C"' in text


def test_create_fails_loudly_without_a_gpu():
    """There is no CPU fallback: on a machine without a HIP device rt_create returns
    RT_ERR_HIP and a message, and never a context."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        pytest.skip("a GPU is present")
    lib = L.load()
    ctx = C.c_void_p()
    rc = lib.rt_create(0, C.byref(ctx))
    assert rc == L.RT_ERR_HIP and not ctx.value
    assert b"no HIP device" in lib.rt_last_error(None)
    from ray_tracing_amd import App, default_scene
    scene, cam = default_scene(64, 36)
    with pytest.raises(L.RtError):
        App.create(cam, scene)


def test_version_and_default_material():
    lib = L.load()
    assert b"gfx950" in lib.rt_version()
    m = L.Material()
    lib.rt_default_material(0, C.byref(m))
    assert (m.kind, m.backface_culling, m.ior) == (L.RT_MAT_GENERIC, 0, 1.5)
    lib.rt_default_material(1, C.byref(m))
    assert (m.kind, m.backface_culling) == (L.RT_MAT_NORMAL, 1)
    lib.rt_default_material(7, C.byref(m))
    assert (m.kind, m.backface_culling, list(m.color)) == (L.RT_MAT_EMISSIVE, 1, [15.0, 15.0, 15.0])
    p = L.BvhParams()
    lib.rt_bvh_params_default(C.byref(p))
    assert (p.root_leaf_max, p.leaf_max, p.max_depth) == (5, 16, 32)  # bvh.rs:17, 75, 5
