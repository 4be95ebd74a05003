#!/usr/bin/env python3
"""Two processes on one GPU: sum all-reduce of CUDA tensors of growing size over gloo (what the
bench's rehearsal mode does for sample mode).  Launch with torch.distributed.run, 2 ranks."""
import faulthandler
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(100, exit=True)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
for mb in (1, 8, 32, 64, 133):
    x = torch.ones(mb * 262144, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.all_reduce(x)
    torch.cuda.synchronize()
    if dist.get_rank() == 0:
        print(f"{mb} MB: {time.perf_counter() - t0:.3f} s, value {x[0].item()}", flush=True)
dist.destroy_process_group()
