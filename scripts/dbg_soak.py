import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib.util
import oracle_lib as O
from helpers import gpu_render
spec = importlib.util.spec_from_file_location("fz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
F = np.float32
seed = int(sys.argv[1])
rng = np.random.default_rng(20000 + seed)
big = seed % 4 == 3
w, h = (int(rng.integers(40, 100)), int(rng.integers(30, 70))) if big else (int(rng.integers(5, 70)), int(rng.integers(5, 50)))
unit = float(F(2.0) ** int(rng.choice([0, 0, 0, -30, -12, 12, 26, 30]))) if big else 1.0
scene, cam, mats = fz.random_scene(rng, w, h, big=big, unit=unit, kinds=fz.ALL_KINDS)
print("unit", unit, "instances", len(scene.instances), "tris", len(scene.triangles), "depth", cam.depth, "env", {k: v for k, v in os.environ.items() if k.startswith("RT_")})
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, materials=mats).render(frames)
for pipeline in (1, 0):
    g = gpu_render(scene, cam, frames, materials=mats, pipeline=pipeline)
    tri, inst, tb = g["hits"]
    bad = np.argwhere((tb != ref.hit_t_bits) | (tri != ref.hit_triangle))
    accbad = (g["acc"].view(np.uint32) != ref.acc.view(np.uint32)).any(axis=-1).sum()
    print("pipeline", pipeline, "bad hits", len(bad), "acc pixels differing", int(accbad))
    for y, x in np.argwhere((g["acc"].view(np.uint32) != ref.acc.view(np.uint32)).any(axis=-1))[:4]:
        print("  acc", y, x, g["acc"][y, x], ref.acc[y, x], "first hit inst", ref.hit_instance[y, x], "kind", mats[int(ref.hit_instance[y, x])].kind if ref.hit_instance[y, x] < len(mats) else None)
    for y, x in bad[:4]:
        print(" pixel", y, x, "gpu tri/inst/t", tri[y, x], inst[y, x], tb[y, x].view(np.float32), hex(tb[y, x]),
              "oracle", ref.hit_triangle[y, x], ref.hit_instance[y, x], ref.hit_t_bits[y, x].view(np.float32), hex(ref.hit_t_bits[y, x]),
              "tt gpu/or", g["counters"][1][y, x], ref.triangle_tests[y, x], "box", g["counters"][0][y, x], ref.box_tests[y, x])
        i = int(ref.hit_instance[y, x])
        print("   instance", i, "transform", scene.instances["transform"][i], "material kind", mats[i].kind)
