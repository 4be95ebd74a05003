#!/usr/bin/env python3
"""How long the bench job takes when its canvas is dealt to `world` GPUs in 8-row strips: every rank's
share is rendered alone on this one GPU and the slowest one is the job's time (the ranks of a real run
work side by side and the barrier waits for the last).  Rehearses the N-GPU bench without N GPUs;
the tree comes from the bench's default builder."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700, builder="gpu_sah")
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
base = None
for world in (1, 2, 4, 8, 16, 64):
    per_rank = []
    for rank in range(world):
        app.set_rows(8, rank, world)
        times = []
        for _ in range(3):
            app.cam_data = L.CamData.from_buffer_copy(cam0)
            app.clear_acc()
            app.render_frames(16)
            app.synchronize()
            times.append(app.last_render_ms())
        per_rank.append(min(times[1:]))
    t = max(per_rank)
    base = base or t
    print(f"world {world}: slowest rank {per_rank.index(t)} {t:.2f} ms, fastest {min(per_rank):.2f} ms, mean "
          f"{sum(per_rank) / world:.2f} ms; ideal {base / world:.2f} ms; efficiency {base / world / t:.2f}", flush=True)
app.destroy()
