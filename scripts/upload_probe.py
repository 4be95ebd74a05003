#!/usr/bin/env python3
"""Host-side times around the hot path: BVH build (host builder), rt_upload_scene (validation,
device layout, copy), for config 3 and config 4."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

for name in sys.argv[1:] or ["c3", "c4"]:
    t0 = time.perf_counter()
    scene, cam0, workload = bench.build_scene(name, 4, 16, 700)
    t1 = time.perf_counter()
    app = App.create(L.CamData.from_buffer_copy(cam0), scene)
    app.synchronize()
    t2 = time.perf_counter()
    app.upload_scene(scene)
    app.synchronize()
    t3 = time.perf_counter()
    print(f"{name}: {len(scene.triangles)} triangles; scene + BVH build {t1 - t0:.2f} s; create + first upload {t2 - t1:.2f} s; "
          f"second upload {t3 - t2:.2f} s", flush=True)
    app.destroy()
