#!/usr/bin/env python3
"""Times rt_create_bvh_device_sah on the terrain mesh (argument: grid size, 700 -> 1.0 M triangles): the
library call alone (host buffers in, host buffers out: the mesh check, device allocations, both transfers
and every kernel), then the same tree built by level-wide passes only."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_amd import _lib as L  # noqa: E402
from ray_tracing_amd import scene as S  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 700
verts, faces = S.terrain(n=grid)
lib = L.load()
S.create_bvh_gpu(verts[:3], np.array([[0, 1, 2]], np.uint32), method="sah")
v = np.zeros(len(verts), S.VERTEX_DT)
v["pos"] = np.ascontiguousarray(verts, np.float32)
nodes = np.zeros(2 * len(faces) - 1, S.BVH_DT)
for name in ("rt_create_bvh_device_sah", "rt_create_bvh_device_sah_passes_only"):
    fn = getattr(lib, name)
    for _ in range(4):
        t = np.ascontiguousarray(faces, np.uint32).copy()
        n = C.c_size_t(len(nodes))
        t0 = time.perf_counter()
        rc = fn(0, S._ptr(v), len(v), S._ptr(t), len(t), 4, 0, S._ptr(nodes), C.byref(n))
        dt = time.perf_counter() - t0
        assert rc == 0
        print(f"{name}: {len(faces)} triangles: {1e3 * dt:.1f} ms, {n.value} nodes", flush=True)
