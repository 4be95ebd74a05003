import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import oracle_lib as O
from ray_tracing_amd import *
from ray_tracing_amd import RT_OPT_PROBE_HITS, RT_OPT_COUNTERS
from ray_tracing_amd import _lib as L
scene, cam = default_scene(640, 360)
app = App.create(L.CamData.from_buffer_copy(cam), scene)
app.set_option(RT_OPT_PROBE_HITS, 1); app.set_option(RT_OPT_COUNTERS, 1)
app.render_frames(1)
tri, inst, tb = app.read_hits()
box, tt = app.read_counters()
acc = app.read_acc()
ref = O.OracleRenderer(scene, cam, trig=O.TRIG_DET).render(1)
bad = tb != ref.hit_t_bits
print("bad", bad.sum(), "by inst", [(i, int((bad & (inst==i)).sum()), int((inst==i).sum())) for i in range(3)])
d = tb.astype(np.int64) - ref.hit_t_bits.astype(np.int64)
print("ulp diff hist", np.unique(d[bad], return_counts=True))
print("box diff", (box != ref.box_tests).sum(), "tt diff", (tt != ref.triangle_tests).sum())
print("acc diff", (acc.view(np.uint32) != ref.acc.view(np.uint32)).any(-1).sum())
ys, xs = np.nonzero(bad)
for k in range(3):
    y, x = ys[k], xs[k]
    print(y, x, tri[y,x], inst[y,x], hex(tb[y,x]), hex(ref.hit_t_bits[y,x]))
