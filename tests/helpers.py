"""Helpers shared by the test modules."""
import ctypes as C
import os

import numpy as np

from ray_tracing_amd import _lib as L
from ray_tracing_amd import App, RT_OPT_COUNTERS, RT_OPT_PIPELINE, RT_OPT_PROBE_HITS
from ray_tracing_amd.scene import affine_from_srt

F = np.float32


def copy_cam(cam):
    return L.CamData.from_buffer_copy(cam)


def gpu_render(scene, cam, frames, counters=True, materials=None, pipeline=1, options=()):
    app = App.create(copy_cam(cam), scene, device=0)
    try:
        app.set_option(RT_OPT_PROBE_HITS, 1)
        app.set_option(RT_OPT_COUNTERS, int(counters))
        # development hook: RT_TEST_PIPELINE=n runs every test that asks for the persistent kernel on kernel n
        if pipeline == 1 and os.environ.get("RT_TEST_PIPELINE"):
            pipeline = int(os.environ["RT_TEST_PIPELINE"])
        app.set_option(RT_OPT_PIPELINE, pipeline)
        for opt, value in options:
            app.set_option(opt, value)
        if materials is not None:
            app.set_materials(materials)
        app.render_frames(frames)
        out = {"acc": app.read_acc(), "present": app.read_present(), "hits": app.read_hits(), "stats": app.stats()}
        if counters:
            out["counters"] = app.read_counters()
        return out
    finally:
        app.destroy()


def assert_bits_equal(a, b, what, nan_equal=False):
    """Word-for-word equality.  With `nan_equal`, any NaN matches any NaN: the sign and
    payload of a NaN produced by an invalid operation differ between x86 (0xFFC00000)
    and the GPU (0x7FC00000) and carry no information."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    bad = a.view(np.uint32) != b.view(np.uint32)
    if nan_equal:
        bad &= ~(np.isnan(a.view(np.float32)) & np.isnan(b.view(np.float32)))
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} words differ, first at {np.argwhere(bad)[0]}"



def make_manual_scene(verts, idx, nodes):
    from ray_tracing_amd.scene import BVH_DT, INSTANCE_DT, VERTEX_DT, Scene

    v = np.zeros(len(verts), VERTEX_DT)
    v["pos"] = verts
    b = np.zeros(len(nodes), BVH_DT)
    for i, (mn, mx, c1, c2, mode) in enumerate(nodes):
        b[i]["min"], b[i]["max"], b[i]["c1"], b[i]["c2"], b[i]["mode"] = mn, mx, c1, c2, mode
    inst = np.zeros(1, INSTANCE_DT)
    ident = affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    inst["transform"][0] = np.frombuffer(bytes(ident), F)
    info = L.SceneInfo(1, len(nodes), len(idx), (C.c_float * 3)(1, -1, 1))
    return Scene(info, v, np.ascontiguousarray(idx, np.uint32), np.zeros(1, np.uint32), inst, b)


