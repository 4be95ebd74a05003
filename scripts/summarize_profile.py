#!/usr/bin/env python3
"""Condenses a gpurun_out/<tag> directory written by scripts/profile_gpu.sh into
profiles/<tag>_summary.json (+ the kernel stats CSV): per-kernel average duration from
the kernel trace and per-launch PMC values for the dominant kernel."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import profiled_source_hash  # noqa: E402

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
out = {"tag": tag, "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"}

stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join("profiles", f"{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
    out["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                            "pct": float(r["Percentage"])} for r in rows[:6]]

KERNEL = "path_kernel_persistent<32, false, false>"
pmc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(files[0])):
        if KERNEL in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    n = len(per)
    if not n:
        continue
    tot = collections.defaultdict(float)
    for v in per.values():
        for k, x in v.items():
            tot[k] += x
    for k, x in tot.items():
        pmc[k] = x / n
out["pmc_per_launch"] = pmc
out["launches_averaged"] = "all dispatches of " + KERNEL

d = {}
if "SQ_THREAD_CYCLES_VALU" in pmc:
    d["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"] / (64 * pmc["SQ_ACTIVE_INST_VALU"])
    d["wave_time_waiting_on_memory"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
    d["wave_time_issuing_valu"] = pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"]
if "SQ_INSTS_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
    # SIMD cycles of the launch: GRBM_GUI_ACTIVE sums the 8 XCDs; 256 CUs x 4 SIMDs.  A vector
    # instruction takes the issue port of its SIMD for 2 cycles (full-rate classes) and many take
    # their unit for 4 (the device micro-probe's table), so 2 x instructions is the floor of the
    # pipe time and 4 x instructions its ceiling.  SQ_ACTIVE_INST_VALU is NOT pipe time: the probe
    # shows it reads one quad-cycle per instruction whatever the instruction costs.
    simd_cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    d["simd_cycles"] = simd_cycles
    d["valu_issue_floor_frac"] = 2.0 * pmc["SQ_INSTS_VALU"] / simd_cycles
    d["valu_issue_ceiling_frac"] = 4.0 * pmc["SQ_INSTS_VALU"] / simd_cycles
    if "SQ_INSTS_VALU_ADD_F32" in pmc:
        f32 = pmc["SQ_INSTS_VALU_ADD_F32"] + pmc["SQ_INSTS_VALU_MUL_F32"] + pmc["SQ_INSTS_VALU_FMA_F32"]
        d["valu_f32_add_mul_fma_share"] = f32 / pmc["SQ_INSTS_VALU"]
        d["valu_trans_share"] = pmc["SQ_INSTS_VALU_TRANS_F32"] / pmc["SQ_INSTS_VALU"]
        d["valu_int_logic_share"] = pmc["SQ_INSTS_VALU_IOPS"] / pmc["SQ_INSTS_VALU"]
        d["valu_cvt_share"] = pmc["SQ_INSTS_VALU_CVT"] / pmc["SQ_INSTS_VALU"]
        # what no class counter sees: min / max / med3, compares, selects, moves, lane reads
        d["valu_uncounted_share"] = 1.0 - (f32 + pmc["SQ_INSTS_VALU_TRANS_F32"] + pmc["SQ_INSTS_VALU_IOPS"]
                                           + pmc["SQ_INSTS_VALU_CVT"]) / pmc["SQ_INSTS_VALU"]
        # Pipe time priced by class with the micro-probe's table (2 cycles: f32 add / mul / fma, integer
        # add and logic; 8: transcendentals; 4: the rest).  The integer counters do not say which of
        # their instructions are the 2-cycle ones, hence two bounds: all of them, or none.
        trans = pmc["SQ_INSTS_VALU_TRANS_F32"]
        ints = min(pmc["SQ_INSTS_VALU_INT32"] + pmc["SQ_INSTS_VALU_IOPS"], pmc["SQ_INSTS_VALU"] - f32 - trans)
        d["valu_pipe_time_frac_low"] = (2 * (f32 + ints) + 8 * trans + 4 * (pmc["SQ_INSTS_VALU"] - f32 - ints - trans)) / simd_cycles
        d["valu_pipe_time_frac_high"] = (2 * f32 + 8 * trans + 4 * (pmc["SQ_INSTS_VALU"] - f32 - trans)) / simd_cycles
if "TCC_EA0_RDREQ_sum" in pmc:
    n32, n128 = pmc.get("TCC_EA0_RDREQ_32B_sum", 0), pmc.get("TCC_EA0_RDREQ_128B_sum", 0)
    n64 = pmc["TCC_EA0_RDREQ_sum"] - n32 - n128
    d["l2_fabric_read_bytes"] = 32 * n32 + 64 * n64 + 128 * n128
    d["l2_fabric_read_bytes_from_dram_requests"] = pmc.get("TCC_EA0_RDREQ_DRAM_sum", 0) * 64
if "TCC_EA0_WRREQ_sum" in pmc:
    n64 = pmc.get("TCC_EA0_WRREQ_64B_sum", 0)
    d["l2_fabric_write_bytes"] = 64 * n64 + 32 * (pmc["TCC_EA0_WRREQ_sum"] - n64)
if "TCC_HIT_sum" in pmc:
    d["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
if "FETCH_SIZE" in pmc:
    d["FETCH_SIZE_bytes_raw"] = pmc["FETCH_SIZE"] * 1024
    d["FETCH_SIZE_bytes_x2_gfx950_streaming_correction"] = pmc["FETCH_SIZE"] * 2048
if "WRITE_SIZE" in pmc:
    d["WRITE_SIZE_bytes"] = pmc["WRITE_SIZE"] * 1024
if "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc and "TCP_TCC_READ_REQ_sum" in pmc:
    d["l1_requests_to_l2_per_l1_access"] = pmc["TCP_TCC_READ_REQ_sum"] / pmc["TCP_TOTAL_CACHE_ACCESSES_sum"]
out["derived"] = d
for name in ("bench_trace.json",):
    p = os.path.join(src, name)
    if os.path.exists(p):
        try:
            out["bench_line_under_trace"] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            pass
if "bench_line_under_trace" in out and "l2_fabric_read_bytes" in d:
    line = out["bench_line_under_trace"]
    traffic = {"workload": line["config"]["workload"], "n_gpus": line["n_gpus"],
               "traffic_bytes_per_launch": d["l2_fabric_read_bytes"] + d.get("l2_fabric_write_bytes", 0.0),
               "method": "TCC_EA0_RDREQ (32B/64B/128B split) and TCC_EA0_WRREQ (64B split) per dispatch of the "
                         "persistent kernel, separate rocprofv3 --pmc passes; equals 2 x FETCH_SIZE + WRITE_SIZE, "
                         "the gfx950 correction for 128-byte read requests tallied at 64 bytes",
               "valu_insts_per_launch": pmc.get("SQ_INSTS_VALU"),
               "rays_per_launch": line["config"]["rays_per_step"],
               "simd_cycles_per_launch": d.get("simd_cycles"),
               "valu_pipe_time_frac_bounds": [d.get("valu_pipe_time_frac_low"), d.get("valu_pipe_time_frac_high")],
               "valu_issue_floor_frac": d.get("valu_issue_floor_frac"),
               # what the counters are valid for: these sources (hash of the kernel, builder and layout
               # sources at the time of the run) and this traversal (the reference's debug counters of a launch)
               "source_hash": profiled_source_hash(),
               "counters_per_launch": {k: line["roofline"]["counters_per_launch"][k]
                                       for k in ("rays", "box_tests", "internal_visits", "triangle_tests", "paths")},
               "source": f"profiles/{tag}_summary.json"}
    json.dump(traffic, open(os.path.join("profiles", "counters_bench.json"), "w"), indent=1)
path = os.path.join("profiles", f"{tag}_summary.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path)
print(json.dumps(d, indent=1))
if "kernel_stats" in out:
    for k in out["kernel_stats"]:
        print(k)
