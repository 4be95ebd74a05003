#!/usr/bin/env python3
"""Renders a scene on GPU 0 and writes the presented frame as PNG.
usage: render_png.py [default|terrain|spheres|box] out.png [spp] [width height]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ray_tracing_amd import App, default_scene, sphere_field_scene, terrain_scene  # noqa: E402
from ray_tracing_amd.image import save_png  # noqa: E402

which, out = sys.argv[1], sys.argv[2]
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
w, h = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (640, 360)
if which == "default":
    scene, cam = default_scene(w, h)
elif which == "terrain":
    scene, cam = terrain_scene(w, h, n=300, leaf_max=4)
elif which == "box":
    scene, cam = terrain_scene(w, h, n=300, leaf_max=4, closed_box=True, light=False, depth=9)
else:
    scene, cam = sphere_field_scene(w, h, n_spheres=10000, leaf_max=4)
with App.create(cam, scene) as app:
    app.render_frames(spp)
    save_png(out, app.read_present())
    print(which, app.stats(), "ms", app.last_render_ms())
