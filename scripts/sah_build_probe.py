#!/usr/bin/env python3
"""Times rt_create_bvh_device_sah on the terrain mesh (argument: grid size, 700 -> 1.0 M triangles)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_amd import scene as S  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 700
verts, faces = S.terrain(n=grid)
S.create_bvh_gpu(verts[:3], np.array([[0, 1, 2]], np.uint32), method="sah")
for _ in range(2):
    t0 = time.perf_counter()
    tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=4, method="sah")
    print(f"{len(faces)} triangles: {1e3 * (time.perf_counter() - t0):.0f} ms, {len(nodes)} nodes", flush=True)
