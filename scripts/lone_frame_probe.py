#!/usr/bin/env python3
"""Time of launches of 1, 2, 4 and 16 fused 1080p frames of the bench scene (from a zeroed
accumulator, six launches each: the first one still runs in the plain tile order), under whatever
RT_* tuning variables are set."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
for n in (1, 2, 4, 8, 16):
    times = []
    for _ in range(6):
        app.cam_data = L.CamData.from_buffer_copy(cam0)
        app.clear_acc()
        app.render_frames(n)
        app.synchronize()
        times.append(app.last_render_ms())
    print(" ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RT_")), f"{n} frame(s), ms:",
          " ".join(f"{t:.2f}" for t in times), flush=True)
app.destroy()
