#!/usr/bin/env python3
"""Per-wave start / queues-dry / end times of one launch of n frames (needs a library built with
-DRT_WAVE_TIMES: scripts/build_variant.sh wt -DRT_WAVE_TIMES with the instrumentation patch)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracing_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "scripts", "_build", "librt_wt.so")
import bench  # noqa: E402
from ray_tracing_amd import App  # noqa: E402

scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
lib = L.load()
for n in [int(a) for a in sys.argv[1:]] or [1, 16]:
    for _ in range(4):
        app.cam_data = L.CamData.from_buffer_copy(cam0)
        app.clear_acc()
        app.render_frames(n)
        app.synchronize()
    ms = app.last_render_ms()
    w = np.zeros(4096 * 4, np.uint64)
    assert lib.rt_debug_wave_times(w.ctypes.data_as(C.c_void_p), w.size) == 0
    w = w.reshape(-1, 4)
    t0 = w[:, 0].min()
    start, dry, end = (w[:, 0] - t0) / 100.0, (w[:, 1].astype(np.int64) - int(t0)) / 100.0, (w[:, 2] - t0) / 100.0  # us
    q = lambda a: " ".join(f"{np.percentile(a, p):8.0f}" for p in (0, 10, 50, 90, 100))
    print(f"{n} frame(s): launch {ms * 1e3:.0f} us")
    print("  start  (us, pct 0/10/50/90/100):", q(start))
    print("  dry                            :", q(dry))
    print("  end                            :", q(end))
    print("  end - dry                      :", q(end - dry))
    print("  iterations after dry           :", q(w[:, 3].astype(np.float64)))
    print(f"  mean idle share of the launch after a wave's end: {(end.max() - end).mean() / end.max():.3f}")
app.destroy()
