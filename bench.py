#!/usr/bin/env python3
"""Throughput of the trace path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3]

Default workload = the configuration the Mrays/s metric and its 1/2/4/8-GPU
figures are quoted on ("c3"): a seeded ~1.0 M-triangle procedural scene at
1920x1080, 16 accumulated one-sample frames per step, depth 10, scene resident
in HBM before the timed region.  With N > 1 (launched by torch.distributed.run,
one process per GPU) the canvas is dealt to the ranks in interleaved 8-row
strips; there is no collective in the data path.  By default the canvas grows
with N at a fixed aspect ratio (area x N: the same view, sampled more densely),
so every GPU keeps one 1080p's worth of pixels: "weak" scaling.  `--scaling
strong` keeps the 1920x1080 canvas for every N instead; that 25 ms job is
bounded below by its longest pixel (16 dependent frames of up to 10 segments,
about 10 ms of strictly serial steps), not by throughput.  A ray is one
trace_ray invocation (shader/src/modules/trace.rs:195), primary and secondary
alike; the count comes from the device's own counter.

The other configurations, for reference numbers (not the driver's bench
line): c2 (10k tessellated spheres, 1 spp, 5 segments), c4 (10 M triangles, 4K,
256 spp split by sample index over the GPUs with one RCCL all-reduce of the
accumulator per step), c5 (closed box, diffuse, 9 segments).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
STRIP_H = 8

CONFIGS = {
    # name: (width, height, frames per step, depth, partition)
    "c2": (1920, 1080, 1, 5, "tile"),
    "c3": (1920, 1080, 16, 10, "tile"),
    "c4": (3840, 2160, 256, 10, "sample"),
    "c5": (1920, 1080, 4, 9, "tile"),
}


def algorithmic_bytes(stats, n_instances):
    """Bytes the reference's own data structures would move for this work:
    per box test one 48-byte Bvh node (hit.rs:77), per internal visit the two
    child nodes peeked at, 2 x 32 bytes of box (hit.rs:90-91), per triangle test
    a 12-byte index triple plus three 16-byte vertices (hit.rs:57-60), per ray
    segment one 80-byte Instance per instance (trace.rs:77) and 56 bytes of ray
    state and hit record, per pixel sample a 16-byte accumulator read and write."""
    return (48 * stats["box_tests"] + 64 * stats["internal_visits"] + 60 * stats["triangle_tests"]
            + (80 * n_instances + 56) * stats["rays"] + 32 * stats["paths"])


def canvas_for(name, world, scaling):
    """Canvas of the job: the config's own, or (weak scaling of a tile-partitioned config) one of
    `world` times the area at the same aspect ratio, sides rounded to multiples of 8."""
    w, h, _, _, partition = CONFIGS[name]
    if world > 1 and scaling == "weak" and partition == "tile":
        s = math.sqrt(world)
        w, h = int(round(w * s / 8.0)) * 8, int(round(h * s / 8.0)) * 8
    return w, h


def build_scene(name, leaf_max, threads, grid, canvas=None):
    from ray_tracing_amd import sphere_field_scene, terrain_scene

    w, h, frames, depth, _ = CONFIGS[name]
    if canvas:
        w, h = canvas
    if name == "c2":
        scene, cam = sphere_field_scene(w, h, n_spheres=10000, seed=1, subdivisions=2, depth=depth, threads=threads,
                                        leaf_max=leaf_max)
        what = "10000 tessellated spheres (icosphere, 320 triangles each), diffuse instance under the sky"
    elif name == "c3":
        scene, cam = terrain_scene(w, h, n=grid, depth=depth, threads=threads, leaf_max=leaf_max)
        what = "procedural terrain, 3 instances (glass / diffuse / emissive)"
    elif name == "c4":
        scene, cam = terrain_scene(w, h, n=2236 if grid == 700 else grid, depth=depth, threads=threads,
                                   leaf_max=leaf_max)
        what = "procedural terrain, 3 instances (glass / diffuse / emissive)"
    else:
        scene, cam = terrain_scene(w, h, n=grid, depth=depth, closed_box=True, light=False, threads=threads,
                                   leaf_max=leaf_max)
        what = "procedural terrain inside a closed box, diffuse only"
    workload = (f"{what}, {scene.n_triangles} triangles, {w}x{h}, {frames} spp per step, depth {depth}, "
                f"BVH by the reference's split rule with leaves of at most {leaf_max} triangles")
    return scene, cam, workload


def cpu_baseline(scene, cam, frames_cap, budget_s=12.0):
    """The CPU oracle (a port of the reference's arithmetic: the reference itself has
    no CPU path and no Rust toolchain exists here) on the same scene and seeds,
    timed on this box's host cores on a bounded sample: whole frames of the same
    chained sequence until the budget is spent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 16)
    r = O.OracleRenderer(scene, cam, trig=O.TRIG_DET, threads=cores)
    frames, t0 = 0, time.perf_counter()
    while True:
        r.render(1)
        frames += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or frames >= frames_cap:
            break
    s = r.stats_dict()
    return {"value": s["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{frames} full {cam.canvas_width}x{cam.canvas_height} one-sample frames of the same scene, "
                      f"camera and seed chain ({s['rays']} rays in {dt:.2f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grid", type=int, default=700, help="terrain grid resolution (700 -> ~1.0 M triangles)")
    ap.add_argument("--leaf-max", type=int, default=4,
                    help="largest leaf the BVH builder leaves unsplit (the reference's builder: 16, bvh.rs:75)")
    ap.add_argument("--pipeline", type=int, default=1, help="kernel: 0 plain nested loops, 1 persistent state machine")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1, tile-partitioned configs: weak = canvas area x N (default), strong = fixed canvas")
    ap.add_argument("--present", type=int, default=1, help="0: skip the gamma-corrected present image (accumulator only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace path has no CPU fallback")
    # Rehearsal on a box with fewer GPUs than ranks: RT_BENCH_REHEARSE=1 puts every rank on GPU 0 and
    # runs the two scalar reductions over gloo.  It exercises the N > 1 code path; its timings mean nothing.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))

    from ray_tracing_amd import (App, RT_OPT_COUNTERS, RT_OPT_FRAME_STRIDE, RT_OPT_PIPELINE, RT_OPT_PRESENT,
                                 RT_OPT_SEED_MODE)
    from ray_tracing_amd import _lib as L
    from ray_tracing_amd.distributed import reduce_samples, sample_split

    _, _, frames_per_step, depth, partition = CONFIGS[args.config]
    width, height = canvas_for(args.config, world, args.scaling)
    scaling = "weak" if (args.scaling == "weak" and partition == "tile") else "strong"
    threads = max(1, min(os.cpu_count() or 1, 16) // max(1, world))
    scene, cam0, workload = build_scene(args.config, args.leaf_max, threads, args.grid, (width, height))
    n_inst = int(scene.info.num_instances)

    stream = torch.cuda.current_stream().cuda_stream
    app = App.create(L.CamData.from_buffer_copy(cam0), scene, device=device_index, stream=stream)
    app.set_option(RT_OPT_PIPELINE, args.pipeline)
    app.set_option(RT_OPT_PRESENT, args.present)
    if partition == "tile":
        app.set_rows(STRIP_H, rank, world)
        first_frame, my_frames = 0, frames_per_step
        how = f"interleaved {STRIP_H}-row strips over {world} GPU(s), no collective"
    else:
        first_frame, stride, my_frames = sample_split(frames_per_step, rank, world)
        app.set_option(RT_OPT_FRAME_STRIDE, stride)
        app.set_option(RT_OPT_SEED_MODE, 1)
        how = (f"samples split by index over {world} GPU(s) (independent-seed variant), one sum all-reduce of the "
               f"{width * height * 16 / 1e6:.0f} MB accumulator per step")
    acc_view = app.acc_tensor() if partition == "sample" else None

    def fresh_cam():
        cam = L.CamData.from_buffer_copy(cam0)
        cam.frame = first_frame
        return cam

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def launch():
        app.render_frames(my_frames)

    def one_step(events=None):
        # a step = the whole job from a zeroed accumulator
        app.cam_data = fresh_cam()
        app.clear_acc()
        if events:
            events[0].record()
        launch()
        if events:
            events[1].record()
        if acc_view is not None:
            reduce_samples(acc_view)  # RCCL over xGMI; the one exchange this mode has

    # --- counter pass (untimed): the reference's debug counters for one step --------------
    app.set_option(RT_OPT_COUNTERS, 1)
    app.reset_stats()
    one_step()
    app.synchronize()
    counted = app.stats()
    sched = app.sched_stats()
    app.set_option(RT_OPT_COUNTERS, 0)

    for _ in range(args.warmup):
        one_step()
    sync()
    app.reset_stats()
    # HIP events on the stream the kernel is launched on (torch's current stream was handed
    # to the library), one pair around every launch of the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(ev[k])
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)  # average kernel time per launch
    stats = app.stats()

    red_dev = "cpu" if rehearse else "cuda"
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    r = torch.tensor([float(stats["rays"])], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(r[0].item())
    ms_per_step = elapsed / args.steps * 1e3

    def measured_traffic():
        """HBM-side bytes per launch from the committed rocprofv3 PMC passes of this very
        workload (profiles/traffic_bench.json); None when no matching profile exists."""
        try:
            with open(os.path.join(ROOT, "profiles", "traffic_bench.json")) as f:
                tr = json.load(f)
            if tr["workload"] == workload and tr["n_gpus"] == world:
                return tr["traffic_bytes_per_launch"], tr.get("valu_pipe_busy")
        except (OSError, KeyError, ValueError):
            pass
        return None, None

    out = None
    if rank == 0:
        # roofline of the one kernel of the step, on this rank's share of the work
        rank_bytes = float(algorithmic_bytes(counted, n_inst))
        achieved = rank_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, valu_busy = measured_traffic()
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_frame": ms_per_step / frames_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "partition": how,
                       "rays_per_step": total_rays / args.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # frac > 1: the reference's byte model counts every node and triangle a ray touches as
                         # a memory read, the caches serve most of them (compare `traffic`); what the committed
                         # profile shows to be the binding resource is VALU issue
                         "valu_pipe_busy": valu_busy,
                         "kernel": "path_kernel_persistent" if args.pipeline else "path_kernel",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": rank_bytes,
                         "counters_per_launch": counted, "schedule_per_launch": sched},
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cam = L.CamData.from_buffer_copy(cam0)
        out["cpu_baseline"] = cpu_baseline(scene, cam, frames_cap=max(frames_per_step, 64))
    app.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
