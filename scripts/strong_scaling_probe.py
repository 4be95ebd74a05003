#!/usr/bin/env python3
"""How long one GPU takes for its share of the bench job when the canvas is dealt to `world`
GPUs in 8-row strips (the share of rank 0), against 1/world of the one-GPU time.  One GPU only:
rehearses the per-GPU work of the N-GPU bench without N GPUs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
base = None
for world in (1, 2, 4, 8):
    app.set_rows(8, 0, world)
    times = []
    for _ in range(4):
        app.cam_data = L.CamData.from_buffer_copy(cam0)
        app.clear_acc()
        app.render_frames(16)
        app.synchronize()
        times.append(app.last_render_ms())
    t = min(times[1:])
    base = base or t
    print(f"world {world}: {t:.2f} ms for rank 0's rows; ideal {base / world:.2f} ms; efficiency {base / world / t:.2f}", flush=True)
app.destroy()
