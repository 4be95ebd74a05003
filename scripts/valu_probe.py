#!/usr/bin/env python3
"""Runs the device micro-probe of vector-instruction issue (rt_probe_valu_issue) for every
instruction class at 1, 2, 4 and 8 waves per SIMD and writes the table as JSON.

    python3 scripts/valu_probe.py [out.json] [--waves 4] [--us 3000]

Under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES ...` the same run
calibrates what those counters read for a stream that issues nothing but vector instructions
(every class is a kernel of its own, so the counter rows can be told apart by kernel name)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?", default=None)
ap.add_argument("--waves", type=int, nargs="*", default=[1, 2, 4])
ap.add_argument("--us", type=int, default=3000, help="length of the timed window in microseconds")
args = ap.parse_args()

lib = L.load()
rows = []
for w in args.waves:
    for mix, name in enumerate(L.VALU_MIXES):
        r = L.ValuProbe()
        rc = lib.rt_probe_valu_issue(0, mix, w, args.us, C.byref(r))
        if rc != 0:
            raise SystemExit(f"rt_probe_valu_issue({name}, {w}) failed: {rc}")
        row = {"mix": name, "waves_per_simd": w, "simd_cycles_per_inst": r.simd_cycles_per_inst,
               "clock_mhz": r.clock_mhz, "kernel_ms": r.kernel_ms, "window_us": r.window_us,
               "start_spread_us": r.start_spread_us, "wave_share_min": r.wave_share_min,
               "wave_share_max": r.wave_share_max, "valu_insts": r.valu_insts, "waves": r.waves, "simds": r.simds,
               "chip_ginst_per_s": r.simds * r.clock_mhz * 1e6 / r.simd_cycles_per_inst / 1e9}
        rows.append(row)
        print(f"{name:11s} W={w}: {r.simd_cycles_per_inst:6.3f} SIMD cycles/inst, {r.clock_mhz:6.0f} MHz, "
              f"{row['chip_ginst_per_s']:7.1f} Ginst/s, window {r.window_us:.0f} us (starts within "
              f"{r.start_spread_us:.2f} us), wave share {r.wave_share_min:.2f}..{r.wave_share_max:.2f}", flush=True)
if args.out:
    json.dump({"window_us": args.us, "rows": rows}, open(args.out, "w"), indent=1)
