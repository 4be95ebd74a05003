import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from ray_tracing_amd import scene as S, App, RT_OPT_PROBE_HITS
from ray_tracing_amd import _lib as L
import test_gpu_scenes as T
F = np.float32
for n in (60, 200):
    verts, faces = S.terrain(n=n, closed_box=True)
    pos = np.asarray(verts, F); faces = np.asarray(faces, np.uint32)
    print("mesh", len(faces), "triangles; extent", pos.min(0), pos.max(0))
    for method in ("sah", "linear"):
        tris, nodes = S.create_bvh_gpu(verts, faces, leaf_max=4, method=method)
        try:
            d = T.check_tree_vectorised(nodes, tris, pos, 4)
            print(method, "tree ok, levels", d, "nodes", len(nodes))
        except AssertionError as e:
            print(method, "TREE INVALID:", e)
    ident = S.affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    cam = S.make_cam(320, 180, S.look_at((0.0, 55.0, -95.0), (0.0, 0.0, 0.0)), depth=9)
    res = {}
    for builder in ("host", "gpu_sah"):
        sc = S.SceneBuilder().add_mesh(verts, faces, [ident], S.bvh_params(leaf_max=4, threads=4), builder=builder).build()
        with App.create(L.CamData.from_buffer_copy(cam), sc) as app:
            app.set_option(RT_OPT_PROBE_HITS, 1)
            app.render_frames(1)
            tri, inst, tb = app.read_hits()
            st = app.stats()
        res[builder] = (tb.copy(), st)
        print(builder, "rays", st["rays"], "paths", st["paths"], "missed primary", int((tri == 0xFFFFFFFF).sum()))
    print("primary t equal:", np.array_equal(res["host"][0], res["gpu_sah"][0]), "differing pixels", int((res["host"][0] != res["gpu_sah"][0]).sum()))

import oracle_lib as O
verts, faces = S.terrain(n=60, closed_box=True)
faces = np.asarray(faces, np.uint32)
cam = S.make_cam(160, 90, S.look_at((0.0, 55.0, -95.0), (0.0, 0.0, 0.0)), depth=3)
for builder in ("host", "gpu_sah", "gpu_linear"):
    sc = S.SceneBuilder().add_mesh(verts, faces, [ident], S.bvh_params(leaf_max=4, threads=4), builder=builder).build()
    ref = O.OracleRenderer(sc, L.CamData.from_buffer_copy(cam), trig=O.TRIG_DET).render(1)
    out = []
    for pipeline in (0, 1):
        with App.create(L.CamData.from_buffer_copy(cam), sc) as app:
            app.set_option(RT_OPT_PROBE_HITS, 1)
            app.set_option(1, pipeline)
            app.render_frames(1)
            tri, inst, tb = app.read_hits()
        out.append(int((tri == 0xFFFFFFFF).sum()))
        same = np.array_equal(tri, ref.hit_triangle)
        out.append(same)
    print(builder, "oracle missed", int((ref.hit_triangle == 0xFFFFFFFF).sum()), "gpu missed / equal to oracle (plain, persistent):", out,
          "depth", int(sc.bvh.shape[0]))
    # brute force: does every primary ray of the oracle's misses really miss all triangles?  use host tree result as truth
