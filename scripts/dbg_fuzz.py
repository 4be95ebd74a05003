import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib.util
import oracle_lib as O
from helpers import gpu_render
spec = importlib.util.spec_from_file_location("fz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
w, h = int(rng.integers(5, 70)), int(rng.integers(5, 50))
scene, cam, mats = fz.random_scene(rng, w, h)
ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, materials=mats).render(1)
for pipeline in (1, 0):
    g = gpu_render(scene, cam, 1, materials=mats, pipeline=pipeline)
    tri, inst, tb = g["hits"]
    bad = np.argwhere(tri != ref.hit_triangle)
    print("pipeline", pipeline, "bad", len(bad))
    for y, x in bad[:5]:
        print(" pixel", y, x, "gpu tri/inst/t", tri[y, x], inst[y, x], tb[y, x].view(np.float32), hex(tb[y, x]),
              "oracle", ref.hit_triangle[y, x], ref.hit_instance[y, x], ref.hit_t_bits[y, x].view(np.float32), hex(ref.hit_t_bits[y, x]),
              "tt gpu/or", g["counters"][1][y, x], ref.triangle_tests[y, x], "box", g["counters"][0][y, x], ref.box_tests[y, x])
        for t in (tri[y, x], ref.hit_triangle[y, x]):
            if t != 0xFFFFFFFF:
                print("   tri", t, scene.triangles[t], [scene.vertices["pos"][i] for i in scene.triangles[t]])
print("materials cull:", [m.backface_culling for m in mats])
