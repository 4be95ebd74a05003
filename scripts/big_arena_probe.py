#!/usr/bin/env python3
"""A scene whose node + triangle-record arena exceeds 4 GiB (about 45 M triangles, GPU-built BVH):
the persistent kernel, which then names items by record number, against the plain kernel."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_amd import App, RT_OPT_PIPELINE, RT_OPT_PROBE_HITS, scene as S  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4800
t0 = time.perf_counter()
verts, faces = S.terrain(n=grid)
faces = np.asarray(faces, np.uint32)
print(f"{len(faces)} triangles generated in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=4)
print(f"GPU BVH: {len(nodes)} nodes in {time.perf_counter() - t0:.1f} s", flush=True)
internal = int((nodes["mode"] == 0).sum())
print(f"arena: {(internal * 64 + len(tris) * 64) / 2**30:.2f} GiB", flush=True)
ident = S.affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
scene = S.single_object_scene(np.asarray(verts, np.float32), tris, nodes, [ident])
cam = S.make_cam(640, 360, S.look_at((95.0, 70.0, -120.0), (0.0, 5.0, 0.0)), fov=60.0, depth=4)
out = {}
for pipeline in (1, 0):
    t0 = time.perf_counter()
    app = App.create(L.CamData.from_buffer_copy(cam), scene)
    app.set_option(RT_OPT_PIPELINE, pipeline)
    app.set_option(RT_OPT_PROBE_HITS, 1)
    print(f"pipeline {pipeline}: upload {time.perf_counter() - t0:.1f} s", flush=True)
    app.render_frames(2)
    app.synchronize()
    out[pipeline] = (app.read_acc().copy(), [a.copy() for a in app.read_hits()], app.last_render_ms())
    print(f"pipeline {pipeline}: {out[pipeline][2]:.2f} ms for 2 frames", flush=True)
    app.destroy()
same_acc = (out[0][0].view(np.uint32) == out[1][0].view(np.uint32)).all()
same_hits = all((a == b).all() for a, b in zip(out[0][1], out[1][1]))
hit = (out[1][1][0] != 0xFFFFFFFF).mean()
print(f"accumulators identical: {bool(same_acc)}; hit buffers identical: {bool(same_hits)}; pixels that hit: {hit:.2f}")
sys.exit(0 if same_acc and same_hits and hit > 0.3 else 1)
