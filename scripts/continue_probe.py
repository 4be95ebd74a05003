#!/usr/bin/env python3
"""How well one launch's tile costs predict the next launch's: the same 16 frames rendered again
(what bench.py's steps do) against the NEXT 16 frames of a progressive render (frames 16..31,
32..47, ... continuing the accumulator), whose paths are different ones."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
rep = []
for _ in range(5):
    app.cam_data = L.CamData.from_buffer_copy(cam0)
    app.clear_acc()
    app.reset_stats()
    app.render_frames(16)
    app.synchronize()
    rep.append((app.last_render_ms(), app.stats()["rays"]))
print("same 16 frames again:   ", " ".join(f"{ms:.2f} ms ({r / ms / 1e3:.0f} Mrays/s)" for ms, r in rep), flush=True)
app.cam_data = L.CamData.from_buffer_copy(cam0)
app.clear_acc()
cont = []
for _ in range(6):
    app.reset_stats()
    app.render_frames(16)  # advances frame and frames_without_move: the next 16 samples
    app.synchronize()
    cont.append((app.last_render_ms(), app.stats()["rays"]))
print("16 more frames each time:", " ".join(f"{ms:.2f} ms ({r / ms / 1e3:.0f} Mrays/s)" for ms, r in cont), flush=True)
app.destroy()
