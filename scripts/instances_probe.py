#!/usr/bin/env python3
"""Scenes of many instances, with and without the cull above the instance list (RT_OPT_INSTANCE_CULL):
1000 (or argv[1]) instances of two small meshes scattered in a room, 1280x720, 8 chained frames per
launch, depth 6.  Every segment walks the whole instance list in index order (trace.rs:76-95); the
cull passes an instance by from a world-space box instead of moving the ray into its space.
Prints Mrays/s and ms per launch for both settings and checks that the accumulators are equal."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracing_amd import App, Material, RT_MAT_EMISSIVE, RT_MAT_GENERIC, RT_MAT_NORMAL  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402
from ray_tracing_amd import scene as S  # noqa: E402

F = np.float32
n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(5)


def xf(lo, hi, spread):
    axis, angle = "xyz"[int(rng.integers(0, 3))], float(rng.uniform(-3.1, 3.1))
    return S.affine_from_srt([float(x) for x in rng.uniform(lo, hi, 3)], S.quat_from_rotation(axis, angle),
                             [float(x) for x in rng.uniform(-spread, spread, 3)])


sv, sf = S.icosphere(2)
cv, cf = S.cube_mesh(inward=False)
b = S.SceneBuilder()
n_s = (n_inst - 1) * 5 // 8
b.add_mesh(np.asarray(sv, F), np.asarray(sf, np.uint32), [xf(0.3, 1.2, 16.0) for _ in range(n_s)],
           S.bvh_params(leaf_max=4, root_leaf_max=0, threads=1))
b.add_mesh(np.asarray(cv, F), np.asarray(cf, np.uint32), [xf(0.3, 1.5, 16.0) for _ in range(n_inst - 1 - n_s)],
           S.bvh_params(leaf_max=2, root_leaf_max=0, threads=1))
rv, rf = S.cube_mesh(inward=True)
b.add_mesh(np.asarray(rv, F), np.asarray(rf, np.uint32), [S.affine_from_srt([24.0, 24.0, 24.0], (0, 0, 0, 1), [0.0, 0.0, 0.0])],
           S.bvh_params(leaf_max=16, root_leaf_max=100, threads=1))
scene = b.build()
kinds = [RT_MAT_NORMAL, RT_MAT_GENERIC, RT_MAT_NORMAL, RT_MAT_EMISSIVE]
materials = []
for i in range(len(scene.instances)):
    m = Material()
    m.kind, m.backface_culling = kinds[i % 4] if i + 1 < len(scene.instances) else RT_MAT_NORMAL, 1
    m.color[:] = [0.9, 0.8, 0.7] if m.kind != RT_MAT_EMISSIVE else [4.0, 4.0, 3.0]
    m.ior, m.roughness = (1.5 if i % 8 == 1 else 0.0), 0.4
    materials.append(m)
cam = S.make_cam(1280, 720, S.look_at((0.0, 3.0, -22.0), (0.0, 0.0, 0.0)), fov=70.0, depth=6)
results = {}
for cull in (0, 1):
    with App.create(L.CamData.from_buffer_copy(cam), scene) as app:
        app.set_materials(materials)
        app.set_option(L.RT_OPT_INSTANCE_CULL, cull)
        times = []
        for rep in range(4):
            app.cam_data = L.CamData.from_buffer_copy(cam)
            app.clear_acc()
            app.reset_stats()
            app.synchronize()
            t0 = time.perf_counter()
            app.render_frames(8)
            app.synchronize()
            times.append(time.perf_counter() - t0)
        rays = app.stats()["rays"]
        results[cull] = (rays, min(times[1:]), app.read_acc())
    print(f"{len(scene.instances)} instances, cull {'on ' if cull else 'off'}: {rays} rays in {results[cull][1] * 1e3:.2f} ms "
          f"= {rays / results[cull][1] / 1e6:.0f} Mrays/s", flush=True)
same = np.array_equal(results[0][2].view(np.uint32), results[1][2].view(np.uint32))
print("accumulators equal:", same, " speed-up x%.2f" % (results[0][1] / results[1][1]))
sys.exit(0 if same else 1)
