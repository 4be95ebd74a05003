import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ray_tracing_amd import App
from ray_tracing_amd import _lib as L
scene, cam0, _ = bench.build_scene("c3", 4, 16, 700)
app = App.create(L.CamData.from_buffer_copy(cam0), scene)
for world in (2, 4, 8):
    app.set_rows(8, 0, world)
    for bpc in (4, 3, 2, 1):
        app.set_option(L.RT_OPT_BLOCKS_PER_CU, bpc)
        times = []
        for _ in range(4):
            app.cam_data = L.CamData.from_buffer_copy(cam0)
            app.clear_acc()
            app.render_frames(16)
            app.synchronize()
            times.append(app.last_render_ms())
        print(f"world {world} blocks/CU {bpc}: {min(times[1:]):.2f} ms", flush=True)
app.destroy()
