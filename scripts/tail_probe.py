#!/usr/bin/env python3
"""Kernel time of the bench scene against the number of fused frames: T(n) = a + b n.
`a` is what a launch pays once (ramp-up, the tail in which the last pixels finish alone)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
for n in (1, 2, 4, 8, 16, 32, 64):
    times = []
    for _ in range(3):
        app.cam_data = L.CamData.from_buffer_copy(cam0)
        app.clear_acc()
        app.render_frames(n)
        app.synchronize()
        times.append(app.last_render_ms())
    print(f"{n:3d} frames: {min(times):8.2f} ms  ({min(times) / n:.3f} ms/frame)", flush=True)
app.destroy()
