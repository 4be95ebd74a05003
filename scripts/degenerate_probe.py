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
for d in (0, 1, 2):
    c = copy_cam(cam); c.depth = d
    case(f"depth {d}", scene, c)
for w, h in ((1, 1), (7, 1), (1, 9), (8, 8), (9, 9)):
    s2, c2 = S.default_scene(w, h)
    case(f"canvas {w}x{h}", s2, c2)
case("0 frames", scene, cam, frames=0)
# no instances at all
info = L.SceneInfo(0, scene.info.num_bvh_nodes, scene.info.num_triangles, scene.info.sun_orientation)
empty = S.Scene(info, scene.vertices, scene.triangles, scene.objects, scene.instances[:0].copy(), scene.bvh)
case("no instances", empty, cam)
c = copy_cam(cam); c.fov = 179.0
case("fov 179", scene, c)
c = copy_cam(cam); c.fov = 1.0
case("fov 1", scene, c)
