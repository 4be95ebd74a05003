#!/usr/bin/env python3
"""Two processes on one GPU, sample mode at 4K: where does the time go (render, then the gloo
all-reduce of the aliased accumulator)?  Launch with torch.distributed.run, 2 ranks."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(150, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ray_tracing_amd import App, RT_OPT_FRAME_STRIDE, RT_OPT_SEED_MODE  # noqa: E402
from ray_tracing_amd import _lib as L  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()


def say(*a):
    print(f"[rank {rank}]", *a, flush=True)


scene, cam0, _ = bench.build_scene("c4", 4, 4, 300)
app = App.create(L.CamData.from_buffer_copy(cam0), scene, device=0, stream=torch.cuda.current_stream().cuda_stream)
app.set_option(RT_OPT_FRAME_STRIDE, 2)
app.set_option(RT_OPT_SEED_MODE, 1)
acc = app.acc_tensor()
say("acc tensor", tuple(acc.shape), acc.dtype, acc.device, acc.is_contiguous())
for frames in (2, 16, 128):
    cam = L.CamData.from_buffer_copy(cam0)
    cam.frame = rank
    app.cam_data = cam
    app.clear_acc()
    t0 = time.perf_counter()
    app.render_frames(frames)
    torch.cuda.synchronize()
    say(f"{frames} frames rendered in {time.perf_counter() - t0:.2f} s")
    dist.barrier()
    t0 = time.perf_counter()
    dist.all_reduce(acc)
    torch.cuda.synchronize()
    say(f"all-reduce in {time.perf_counter() - t0:.2f} s")
app.destroy()
dist.destroy_process_group()
