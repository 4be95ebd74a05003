#!/bin/bash
# On the GPU box: ms per lone 1080p frame (one rt_render_frames(1) per launch) and per fused step under a few
# settings of the vote weights.
mkdir -p gpurun_out/lone
for w in "$@"; do
  IFS=, read a b c <<< "$w"
  v=$((a | (b << 8) | (c << 16)))
  timeout -k 10 300 python3 - $v $w <<'PY'
import sys, time
sys.path.insert(0, '.')
import torch
from ray_tracing_amd import App, terrain_scene
from ray_tracing_amd import _lib as L
v, w = int(sys.argv[1]), sys.argv[2]
scene, cam0 = terrain_scene(1920, 1080, n=700, depth=10, leaf_max=4, builder="gpu_sah", device=0)
app = App.create(L.CamData.from_buffer_copy(cam0), scene, device=0)
app.set_option(8, v)
def lone(n=16):
    app.cam_data = L.CamData.from_buffer_copy(cam0); app.clear_acc()
    for _ in range(n): app.render_frames(1)
def fused():
    app.cam_data = L.CamData.from_buffer_copy(cam0); app.clear_acc(); app.render_frames(16)
for f, name, per in ((lone, 'lone frame', 16), (fused, 'fused frame', 16)):
    for _ in range(3): f()
    app.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    app.synchronize(); dt = (time.perf_counter() - t0) / 5 / per * 1e3
    print(f"weights {w}: {name} {dt:.3f} ms", flush=True)
app.destroy()
PY
done
