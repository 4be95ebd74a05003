#!/usr/bin/env python3
"""Degenerate inputs through the library and the oracle: what each does (development probe)."""
import os, sys, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import ctypes as C
import numpy as np
import oracle_lib as O
from ray_tracing_amd import App, _lib as L, scene as S
from helpers import gpu_render, copy_cam

def case(name, scene, cam, frames=2):
    try:
        g = gpu_render(scene, cam, frames)
        ref = O.OracleRenderer(scene, copy_cam(cam), trig=O.TRIG_DET).render(frames)
        same = g["acc"].view(np.uint32) == ref.acc.view(np.uint32)
        print(name, "ok: acc equal", bool(same.all()), "stats", g["stats"], "oracle", ref.stats_dict(), "acc[0,0]", g["acc"][0, 0], flush=True)
    except Exception as e:
        print(name, "RAISED", type(e).__name__, e, flush=True)

scene, cam = S.default_scene(32, 18)
c = copy_cam(cam); c.depth = 64
case("depth 64", scene, c)
c = copy_cam(cam); c.frame = 0xFFFFFFFE
case("frame wrap", scene, c, frames=4)
c = copy_cam(cam); c.frames_without_move = 3.0e7
case("fwm 3e7", scene, c, frames=3)
for w, h in ((0, 5), (5, 0)):
    c = copy_cam(cam); c.canvas_width, c.canvas_height = w, h
    case(f"canvas {w}x{h}", scene, c)
# a singular instance transform (scale 0 along y)
sing = S.affine_from_srt((1.0, 0.0, 1.0), (0, 0, 0, 1), (0, 0, 0))
s3 = S.Scene(scene.info, scene.vertices, scene.triangles, scene.objects, scene.instances.copy(), scene.bvh)
s3.instances["transform"][1] = np.frombuffer(bytes(sing), np.float32)
case("singular transform", s3, cam)
