#!/usr/bin/env python3
"""Build time and render throughput: the reference-rule host builder against the two GPU builders."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_amd import App, scene as S  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 700
verts, faces = S.terrain(n=grid)
print(f"{len(faces)} triangles", flush=True)
ident = S.affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
cam = S.make_cam(1920, 1080, S.look_at((95.0, 70.0, -120.0), (0.0, 5.0, 0.0)), fov=60.0, depth=10)
S.create_bvh_gpu(verts[:3], np.array([[0, 1, 2]], np.uint32))  # warm the runtime up


def render(scene, tag):
    app = App.create(L.CamData.from_buffer_copy(cam), scene)
    best = 1e9
    for _ in range(3):
        app.cam_data = L.CamData.from_buffer_copy(cam)
        app.clear_acc()
        app.reset_stats()
        app.render_frames(16)
        app.synchronize()
        best = min(best, app.last_render_ms())
    rays = app.stats()["rays"]
    print(f"  {tag}: {best:.2f} ms per 16 frames, {rays / best / 1e3:.0f} Mrays/s", flush=True)
    app.destroy()


for leaf in (tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (4,)):
    t0 = time.perf_counter()
    tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=leaf)
    t_gpu = time.perf_counter() - t0
    print(f"GPU linear BVH, leaf {leaf}: {t_gpu * 1e3:.0f} ms (incl. transfers), {len(nodes)} nodes", flush=True)
    render(S.single_object_scene(verts, tris, nodes, [ident]), "render")
    S.create_bvh_gpu(verts[:3], np.array([[0, 1, 2]], np.uint32), method="sah")
    t0 = time.perf_counter()
    tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=leaf, method="sah")
    t_gpu = time.perf_counter() - t0
    print(f"GPU binned SAH, leaf {leaf}: {t_gpu * 1e3:.0f} ms (incl. transfers), {len(nodes)} nodes", flush=True)
    render(S.single_object_scene(verts, tris, nodes, [ident]), "render")
    t0 = time.perf_counter()
    b = S.SceneBuilder()
    b.add_mesh(verts, faces, [ident], S.bvh_params(leaf_max=leaf, threads=16))
    sc = b.build()
    t_host = time.perf_counter() - t0
    print(f"host builder (reference split rule), leaf {leaf}, 16 threads: {t_host * 1e3:.0f} ms, {len(sc.bvh)} nodes", flush=True)
    render(sc, "render")
