#!/usr/bin/env python3
"""Does the GPU finish the bench job sooner as K concurrent launches (K contexts on K streams,
each with 1/K of the rows) than as one?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
for k in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(k)]
    apps = [App.create(L.CamData.from_buffer_copy(cam0), scene, stream=s.cuda_stream) for s in streams]
    for r, a in enumerate(apps):
        a.set_rows(8, r, k)
    best = 1e9
    for rep in range(4):
        for a in apps:
            a.cam_data = L.CamData.from_buffer_copy(cam0)
            a.clear_acc()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in apps:
            a.render_frames(16)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if rep:
            best = min(best, dt)
    rays = sum(a.stats()["rays"] for a in apps) / 4
    print(f"{k} concurrent launches: {best:.2f} ms  ({rays / best / 1e3:.0f} Mrays/s)", flush=True)
    for a in apps:
        a.destroy()
