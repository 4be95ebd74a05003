import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, ctypes as C
import oracle_lib as O
from ray_tracing_amd import default_scene, _lib as L, scene as S
from helpers import gpu_render
F=np.float32
def parse_obj(name):
    data = S._resource(name).encode()
    lib = L.load()
    nv, nt = C.c_size_t(0), C.c_size_t(0)
    lib.rt_parse_obj(data, len(data), None, C.byref(nv), None, C.byref(nt))
    v = np.zeros(nv.value, S.VERTEX_DT); t = np.zeros((nt.value, 3), np.uint32)
    lib.rt_parse_obj(data, len(data), S._ptr(v), C.byref(nv), S._ptr(t), C.byref(nt))
    return v["pos"].copy(), t
base, cam = default_scene(96, 54)
for k in (2.0**40, 2.0**28, 2.0**-28, 2.0**-40):
    b = S.SceneBuilder()
    for idx, name in enumerate(("default_cube.obj", "cornel_box.obj", "teapot.obj")):
        t = L.Affine3A.from_buffer_copy(base.instances["transform"][idx].tobytes())
        for axis in (t.x_axis, t.y_axis, t.z_axis):
            for i in range(3):
                axis[i] = float(F(axis[i]) / F(k))
        v, f = parse_obj(name)
        b.add_mesh(v * F(k), f, [t])
    sc = b.build()
    r = O.OracleRenderer(sc, cam).render(1)
    for p in (0, 1):
        g = gpu_render(sc, cam, 1, pipeline=p)
        tri, inst, tb = g["hits"]
        print(k, "pipeline", p, "tri diff", (tri != r.hit_triangle).sum(), "t diff", (tb != r.hit_t_bits).sum(), "box diff", (g["counters"][0] != r.box_tests).sum(), hex(tb[0,0]), hex(r.hit_t_bits[0,0]), "acc diff", (g["acc"].view(np.uint32) != r.acc.view(np.uint32)).sum())
