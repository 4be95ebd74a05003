#!/usr/bin/env python3
"""Condense bench.py's JSON line (stdin) to one short line for tuning sweeps."""
import json
import sys

d = json.loads(sys.stdin.read())
r = d["roofline"]
s = r["schedule_per_launch"]
it, la, cy = s["iterations"], s["lanes"], s.get("cycles", [0] * len(s["iterations"]))
tot = max(1, sum(cy))
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), "Mrays/s", round(r["kernel_ms"], 1), "ms",
      "iters(M)", [round(x / 1e6, 1) for x in it], "lanes/iter", [round(a / max(b, 1), 1) for a, b in zip(la, it)],
      "cyc/iter", [round(a / max(b, 1)) for a, b in zip(cy, it)], "wait/iter",
      [round(a / max(b, 1)) for a, b in zip(s.get("wait_cycles", [0] * len(it)), it)], "share",
      [round(x / tot, 3) for x in cy])
