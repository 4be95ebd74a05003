#!/usr/bin/env python3
"""What a displayed frame costs beyond its render: reading the image back as floats and as bytes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
for _ in range(3):
    app.render_frames(1)
app.synchronize()
for name, fn in (("render + sync", lambda: (app.render_frames(1), app.synchronize())),
                 ("read_present (RGBA32F, 33 MB)", app.read_present),
                 ("read_present_rgba8 (8 MB)", app.read_present_rgba8)):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms", flush=True)
import ctypes as C  # noqa: E402

lib, ptr, n = L.load(), C.c_void_p(), C.c_size_t()
app.render_frames(1)
lib.rt_present_begin(app._ctx)
t0 = time.perf_counter()
for _ in range(50):
    app.render_frames(1)                                      # frame k + 1, enqueued
    assert lib.rt_present_end(app._ctx, C.byref(ptr), C.byref(n)) == 0  # frame k is on the host
    assert lib.rt_present_begin(app._ctx) == 0                # frame k + 1: quantise, then copy beside frame k + 2
print(f"render + present_begin / present_end (copy beside the next render): {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms per displayed frame",
      flush=True)
app.destroy()
