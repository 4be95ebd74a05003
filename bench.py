#!/usr/bin/env python3
"""Throughput of the trace path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3]

Default workload = the configuration the Mrays/s metric and its 1/2/4/8-GPU
figures are quoted on ("c3"): a seeded ~1.0 M-triangle procedural scene at
1920x1080, 16 accumulated one-sample frames per step, depth 10, scene resident
in HBM before the timed region.  The terrain's BVH comes from the library's own
binned surface-area builder on the GPU (--builder gpu_sah, the default); the line
also carries the same job on the tree of the reference's split rule built on the
host (`reference_rule_tree`; --builder host makes that the headline) and on the tree the reference's
builder makes with its own leaf size of 16 (`reference_builder_tree`).  With N > 1 (launched by torch.distributed.run,
one process per GPU) the canvas is dealt to the ranks in interleaved 8-row
strips; there is no collective in the data path.  The job is the same for every
N (1920x1080, 16 spp, the same scene): "strong" scaling.  That 17 ms job is
bounded below by its longest pixel (16 dependent frames of up to 10 segments:
`strong_scaling_floor` on the N = 1 line is the step time of the slowest of the 64
shares of a 64-way split, about 10 ms, which no number of GPUs gets below).  For N > 1 the line also carries
`weak_scaling`: the same view on a canvas of N times the area, so that every GPU
keeps one 1080p's worth of pixels (`--scaling weak` makes that the headline).
A ray is one trace_ray invocation (shader/src/modules/trace.rs:195), primary and
secondary alike; the count comes from the device's own counter.

The other configurations, for reference numbers (not the driver's bench
line): c2 (10k tessellated spheres, 1 spp, 5 segments), c4 (10 M triangles, 4K,
256 spp split by sample index over the GPUs with one RCCL all-reduce of the
accumulator per step), c5 (closed box, diffuse, 9 segments).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
STRIP_H = 8

# what the hardware counters of a committed profile depend on: the trace kernel and everything it
# includes, and the code that makes the tree and the device records the kernel walks
PROFILED_SOURCES = ("kernels.hip", "traverse_steps.inc", "instance_walk.inc", "trace_device.hpp", "fastdiv.hpp",
                    "trifilter.hpp", "fastexact.hpp", "wave_scan.hpp", "device_types.hpp", "kernels.hpp", "Makefile",
                    "bvh_sah_gpu.hip", "bvh_pad.hpp", "gpu_layout.cpp", "host_scene.cpp", "instance_cull.hpp", "rt_ctx.hpp")


def profiled_source_hash():
    """SHA-256 over the sources a committed hardware-counter profile is only valid for."""
    import hashlib

    h = hashlib.sha256()
    for name in PROFILED_SOURCES:
        with open(os.path.join(ROOT, "ray_tracing_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def launch_ranks(n):
    """`python bench.py --gpus N` started plainly: start the N ranks as children (one process per
    GPU, torch.distributed.run on 127.0.0.1) before anything in this process touches a GPU, pass rank
    0's JSON line through as the only line on stdout, and return the children's worst exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(os.cpu_count() or 1, 16) // n)))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = [l for l in proc.stdout.splitlines() if l.lstrip().startswith("{")]
    if lines:
        print(lines[-1])
    else:
        sys.stderr.write(proc.stdout)
    return proc.returncode if proc.returncode else (0 if lines else 1)

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


BUILDERS = {"host": "BVH by the reference's split rule",
            "gpu_sah": "BVH by binned surface-area splits on the GPU",
            "gpu_linear": "linear BVH built on the GPU"}


def build_scene(name, leaf_max, threads, grid, canvas=None, builder="host", device=0):
    from ray_tracing_amd import sphere_field_scene, terrain_scene

    w, h, frames, depth, _ = CONFIGS[name]
    if canvas:
        w, h = canvas
    if name == "c2":
        scene, cam = sphere_field_scene(w, h, n_spheres=10000, seed=1, subdivisions=2, depth=depth, threads=threads,
                                        leaf_max=leaf_max, builder=builder, device=device)
        what = "10000 tessellated spheres (icosphere, 320 triangles each), diffuse instance under the sky"
    elif name == "c3":
        scene, cam = terrain_scene(w, h, n=grid, depth=depth, threads=threads, leaf_max=leaf_max, builder=builder, device=device)
        what = "procedural terrain, 3 instances (glass / diffuse / emissive)"
    elif name == "c4":
        scene, cam = terrain_scene(w, h, n=2236 if grid == 700 else grid, depth=depth, threads=threads,
                                   leaf_max=leaf_max, builder=builder, device=device)
        what = "procedural terrain, 3 instances (glass / diffuse / emissive)"
    else:
        scene, cam = terrain_scene(w, h, n=grid, depth=depth, closed_box=True, light=False, threads=threads,
                                   leaf_max=leaf_max, builder=builder, device=device)
        what = "procedural terrain inside a closed box, diffuse only"
    workload = (f"{what}, {scene.n_triangles} triangles, {w}x{h}, {frames} spp per step, depth {depth}, "
                f"{BUILDERS[builder]} with leaves of at most {leaf_max} triangles")
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
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of host time the cpu_baseline leg may take")
    ap.add_argument("--grid", type=int, default=700, help="terrain grid resolution (700 -> ~1.0 M triangles)")
    ap.add_argument("--leaf-max", type=int, default=4,
                    help="largest leaf the BVH builder leaves unsplit (the reference's builder: 16, bvh.rs:75)")
    ap.add_argument("--builder", choices=sorted(BUILDERS), default="gpu_sah",
                    help="who builds the big mesh's BVH: the library's binned surface-area builder on the GPU "
                         "(default), its linear builder, or the reference's split rule on the host")
    ap.add_argument("--pipeline", type=int, default=1, help="kernel: 0 plain nested loops, 1 persistent state machine")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1, tile-partitioned configs: strong = the config's own canvas (default), "
                         "weak = canvas area x N")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline measurement (for profiler runs: every launch of the kernel is then a step)")
    ap.add_argument("--opt", action="append", default=[], metavar="ID=VALUE",
                    help="rt_set_option(ID, VALUE) before rendering (tuning switches; none changes a pixel)")
    ap.add_argument("--present", type=int, default=1, help="0: skip the gamma-corrected present image (accumulator only)")
    args = ap.parse_args()

    if os.environ.get("RT_BENCH_WATCHDOG_S"):
        # debugging aid: every rank dumps its Python stack to stderr and exits if it is still running then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG_S"]), exit=True)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))  # no launcher around us: be the launcher

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    scene, cam0, workload = build_scene(args.config, args.leaf_max, threads, args.grid, (width, height), args.builder,
                                        device_index)
    n_inst = int(scene.info.num_instances)

    stream = torch.cuda.current_stream().cuda_stream
    app = App.create(L.CamData.from_buffer_copy(cam0), scene, device=device_index, stream=stream)
    app.set_option(RT_OPT_PIPELINE, args.pipeline)
    app.set_option(RT_OPT_PRESENT, args.present)
    for kv in args.opt:
        k, v = kv.split("=")
        app.set_option(int(k), int(v))
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
            reduce_samples(acc_view, dst=0)  # RCCL over xGMI; the one exchange this mode has (rank 0 holds the image)

    # --- counter pass (untimed): the reference's debug counters for one step --------------
    app.set_option(RT_OPT_COUNTERS, 1)
    app.reset_stats()
    one_step()
    app.synchronize()
    counted = app.stats()
    sched = app.sched_stats()
    app.set_option(RT_OPT_COUNTERS, 0)

    last_clock_mhz = None

    def timed(n_steps, step):
        """n_steps of `step` between two barriers; wall seconds (max over ranks), rays of all
        ranks, and this rank's average HIP-event time per launch.  Leaves the shader clock the
        launches ran at (from the kernel's own counters) in `last_clock_mhz`."""
        nonlocal last_clock_mhz
        sync()
        app.reset_stats()
        # HIP events on the stream the kernel is launched on (torch's current stream was handed
        # to the library), one pair around every launch of the timed region
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]
        t0 = time.perf_counter()
        for k in range(n_steps):
            step(ev[k])
        sync()
        dt = time.perf_counter() - t0
        ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, n_steps)
        last_clock_mhz = app.kernel_clock_mhz()
        red_dev = "cpu" if rehearse else "cuda"
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        r = torch.tensor([float(app.stats()["rays"])], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return float(t.item()), float(r.item()), ms

    # the first launch of a view: no tile order learnt yet (what follows every camera move)
    first_launch = None
    if world == 1 and partition == "tile" and not args.no_extras:
        app.set_option(L.RT_OPT_TILE_ORDER, 1)  # forgets the order the counter pass left behind
        dt, rays, ms = timed(1, one_step)
        first_launch = {"kernel_ms": ms, "Mrays_per_s": rays / (ms * 1e-3) / 1e6,
                        "what": "one step launched with no tile order learnt from an earlier launch of the view"}

    for _ in range(args.warmup):
        one_step()
    elapsed, total_rays, kernel_ms = timed(args.steps, one_step)
    ms_per_step = elapsed / args.steps * 1e3
    kernel_clock_mhz = last_clock_mhz  # of the timed region's launches

    def strong_scaling_floor():
        """What no number of GPUs gets below: the slowest of the 64 shares of a 64-way split, each rendered
        alone on this GPU (the share that holds the dearest pixels: a pixel's frames and segments are one
        chain of dependent steps however few pixels keep it company)."""
        shares = []
        for r in range(64):
            app.set_rows(STRIP_H, r, 64)
            one_step()  # learns this share's tile order
            dt, _, _ = timed(1, one_step)
            shares.append(dt * 1e3)
        app.set_rows(STRIP_H, rank, world)
        return {"share_of_canvas": 1.0 / 64, "ms_per_step": max(shares), "slowest_rank": shares.index(max(shares)),
                "mean_ms_per_step": sum(shares) / len(shares),
                "what": "step time of the slowest of the 64 shares of a 64-way split, each rendered alone on one GPU"}

    extras = {}
    if first_launch:
        extras["first_launch_of_a_view"] = first_launch

    def extra_lone_frames():
        # the reference's own mode: one render call per frame (main.rs:129-149)
        def lone_frames(events=None):
            app.cam_data = fresh_cam()
            app.clear_acc()
            if events:
                events[0].record()
            for _ in range(frames_per_step):
                app.render_frames(1)
            if events:
                events[1].record()

        for _ in range(2):
            lone_frames()
        dt, _, _ = timed(3, lone_frames)
        extras["ms_per_lone_frame"] = dt / 3 / frames_per_step * 1e3
        extras["strong_scaling_floor"] = strong_scaling_floor()

    def attempt(what, fn):
        """An extra must not cost the line itself: what goes wrong in one (on every rank alike, or the ranks
        would part ways) is reported in its place."""
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            extras[what + "_error"] = repr(e)

    def extra_weak_scaling():
        # weak scaling beside the headline: the same view on N times the area
        ww, wh = canvas_for(args.config, world, "weak")
        cam_weak = L.CamData.from_buffer_copy(cam0)
        cam_weak.canvas_width, cam_weak.canvas_height = ww, wh

        def weak_step(events=None):
            app.cam_data = L.CamData.from_buffer_copy(cam_weak)
            app.clear_acc()
            if events:
                events[0].record()
            launch()
            if events:
                events[1].record()

        for _ in range(2):
            weak_step()
        dt, rays, ms = timed(max(2, args.steps // 2), weak_step)
        extras["weak_scaling"] = {"canvas": [ww, wh], "Mrays_per_s": rays / dt / 1e6,
                                  "ms_per_step": dt / max(2, args.steps // 2) * 1e3}
        # why the headline does not scale: the job's longest chain of dependent steps, on one GPU
        extras["strong_scaling_floor"] = strong_scaling_floor()

    def extra_gather():
        # the frame brought to one place: every rank's strips gathered on rank 0, device to device (one
        # gather over RCCL; host memory and gloo in a one-GPU rehearsal).  Outside `value`: the
        # metric counts rays traced, and a displayed frame needs this once, not once per sample.
        from ray_tracing_amd.distributed import gather_strips
        app.cam_data = fresh_cam()
        app.clear_acc()
        launch()
        local = torch.from_numpy(app.read_acc()) if rehearse else app.acc_tensor()
        sync()
        t0 = time.perf_counter()
        full = gather_strips(local, STRIP_H, dst=0)
        sync()
        extras["gather_ms"] = (time.perf_counter() - t0) * 1e3
        extras["gathered_image_on_rank0"] = list(full.shape) if full is not None else None

    if world == 1 and partition == "tile" and not args.no_extras:
        attempt("lone_frames", extra_lone_frames)
    if world > 1 and partition == "tile" and scaling == "strong" and not args.no_extras:
        attempt("weak_scaling", extra_weak_scaling)
    if world > 1 and partition == "tile" and not args.no_extras:
        attempt("gather", extra_gather)

    def same_job_on_another_tree(builder, leaf_max):
        """The headline's scene, camera and frames with the terrain's tree from another builder."""
        nonlocal app
        t0 = time.perf_counter()
        scene2, _, _ = build_scene(args.config, leaf_max, threads, args.grid, (width, height), builder, device_index)
        build_s = time.perf_counter() - t0
        app2 = App.create(L.CamData.from_buffer_copy(cam0), scene2, device=device_index, stream=stream)
        app2.set_option(RT_OPT_PIPELINE, args.pipeline)
        app2.set_option(RT_OPT_PRESENT, args.present)
        main_app, app = app, app2
        try:
            for _ in range(2):
                one_step()
            dt, rays, ms = timed(5, one_step)
        finally:
            app = main_app
            app2.destroy()
        return {"builder": f"{BUILDERS[builder]}, leaves of at most {leaf_max} triangles", "Mrays_per_s": rays / dt / 1e6,
                "ms_per_step": dt / 5 * 1e3, "kernel_ms": ms, "scene_build_s": build_s,
                "what": "same scene, camera and frames, the terrain's tree from another builder "
                        "(scene_build_s covers mesh generation, the build and scene assembly)"}

    if world == 1 and partition == "tile" and args.config == "c3" and args.builder != "gpu_linear" and not args.no_extras:
        # the same job on the other tree of the same mesh: the reference's split rule on the host when the
        # headline is on the GPU-built tree, and the other way round
        other = "host" if args.builder == "gpu_sah" else "gpu_sah"
        name = "reference_rule_tree" if other == "host" else "gpu_built_tree"
        attempt(name, lambda: extras.__setitem__(name, same_job_on_another_tree(other, args.leaf_max)))
        if args.leaf_max != 16:
            # and on the tree the reference's builder makes when left alone: its rule with its own leaf size (bvh.rs:75)
            attempt("reference_builder_tree",
                    lambda: extras.__setitem__("reference_builder_tree", same_job_on_another_tree("host", 16)))

    def committed_profile():
        """Per-launch hardware counters of this very workload from the committed rocprofv3 PMC passes
        (profiles/counters_bench.json), and why they cannot be used when they cannot: the counters
        are only valid for the sources they were taken on (a hash of the kernel, tree-builder and
        layout sources rides in the file) and for the very traversal this run counted."""
        try:
            with open(os.path.join(ROOT, "profiles", "counters_bench.json")) as f:
                tr = json.load(f)
        except (OSError, ValueError):
            return {}, "no committed profile (profiles/counters_bench.json)"
        if tr.get("workload") != workload or tr.get("n_gpus") != 1:
            return {}, "the committed profile is of another workload"
        if tr.get("source_hash") != profiled_source_hash():
            return {}, "the kernel or builder sources changed after the committed profile was taken"
        want = tr.get("counters_per_launch") or {}
        if world == 1 and any(int(want.get(k, -1)) != int(counted[k]) for k in ("rays", "box_tests", "internal_visits", "triangle_tests")):
            return {}, "this run's traversal counters differ from those of the committed profile"
        return tr, None

    out = None
    if rank == 0:
        # The one kernel of the step, on this rank's share of the work.  It is bound by vector
        # instruction issue, and the fraction is formed in one clock domain, the kernel's own:
        # a SIMD issues one full-rate vector instruction (v_mul_f32 / v_add_f32 / v_fma_f32) per
        # 2 cycles, so `peak` = SIMDs x f / 2 with f the shader clock the timed launches ran at, read
        # from the kernel's own cycle and 100 MHz counters; `achieved` = the launch's vector
        # instructions (hardware counter of the committed profile of these very sources and this very
        # traversal) over its HIP-event time.  frac = 2 x instructions / (SIMDs x kernel cycles).
        # The device micro-probe, run now, shows the 2 cycles (and the lower clock a pure issue loop
        # holds, which is why its clock stays out of `peak`).
        prof, stale = committed_profile()
        probe = L.ValuProbe()
        have_probe = L.load().rt_probe_valu_issue(device_index, L.VALU_MIXES.index("mul_add"), 4, 2000, C.byref(probe)) == 0
        simds = int(probe.simds) if have_probe else 1024
        peak = simds * kernel_clock_mhz * 1e6 / 2.0 / 1e9 if kernel_clock_mhz else None
        insts = prof.get("valu_insts_per_launch")
        traffic = prof.get("traffic_bytes_per_launch")
        if insts and world > 1:
            # this rank's share of a launch profiled on one GPU: instructions and bytes go with the rays
            share = counted["rays"] / prof["rays_per_launch"]
            insts, traffic = insts * share, traffic * share
        achieved = insts / (kernel_ms * 1e-3) / 1e9 if insts else None
        rank_bytes = float(algorithmic_bytes(counted, n_inst))
        out = {
            "metric": "Mrays/s (primary+secondary)", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_frame": ms_per_step / frames_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "partition": how,
                       "rays_per_step": total_rays / args.steps},
            "roofline": {"bound": "valu_issue", "achieved": achieved, "peak": peak, "unit": "Ginst/s",
                         "frac": achieved / peak if achieved and peak else None,
                         "profile_stale": stale,
                         "traffic": traffic,
                         "kernel_clock_mhz": kernel_clock_mhz,
                         "kernel_simd_cycles": (simds * kernel_ms * 1e-3 * kernel_clock_mhz * 1e6
                                                if kernel_clock_mhz else None),
                         "profile": prof.get("source"),
                         "kernel": "path_kernel_persistent" if args.pipeline else "path_kernel",
                         "kernel_ms": kernel_ms,
                         "valu_insts_per_launch": insts,
                         # the same launch's vector instructions priced by class (2 / 4 / 8 SIMD cycles, the
                         # probe's table) over its SIMD cycles; two bounds, see scripts/summarize_profile.py
                         "valu_pipe_time_frac_bounds": prof.get("valu_pipe_time_frac_bounds") if world == 1 else None,
                         "probe": {"class": "v_mul_f32 / v_add_f32", "waves_per_simd": 4,
                                   "simd_cycles_per_inst": probe.simd_cycles_per_inst,
                                   "clock_mhz": probe.clock_mhz} if have_probe else None,
                         # the byte model of the reference's own data structures, and what reaches the
                         # fabric: the working set is cache-resident, HBM is not the bound
                         "hbm": {"algorithmic_bytes_per_launch": rank_bytes,
                                 "algorithmic_GBps": rank_bytes / (kernel_ms * 1e-3) / 1e9,
                                 "traffic_GBps": traffic / (kernel_ms * 1e-3) / 1e9 if traffic else None,
                                 "peak_GBps": HBM_PEAK_GBS,
                                 "traffic_frac_of_peak": traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None},
                         "counters_per_launch": counted, "schedule_per_launch": sched},
        }
        out.update(extras)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cam = L.CamData.from_buffer_copy(cam0)
        out["cpu_baseline"] = cpu_baseline(scene, cam, frames_cap=max(frames_per_step, 64), budget_s=args.cpu_budget)
    app.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
