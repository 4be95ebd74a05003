"""The bench line the driver reads: one JSON object on the last line of stdout, with the metric of
the 1080p / 16-spp / 1 M-triangle configuration, an issue-bound roofline whose fraction is a
fraction, and the CPU oracle timed beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_has_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--cpu-budget", "1.5"], check=True, capture_output=True, text=True, timeout=280, cwd=ROOT).stdout
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    assert d["metric"].startswith("Mrays/s") and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "1920x1080" in d["config"]["workload"] and "16 spp" in d["config"]["workload"]
    assert d["value"] > 1000.0  # the target the metric is quoted with: 1 Grays/s per GPU
    assert abs(d["value"] - d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["unit"] == "Ginst/s" and r["kernel"] == "path_kernel_persistent"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.5 * d["ms_per_step"] < r["kernel_ms"] <= d["ms_per_step"] * 1.01  # the step is that one kernel (slack for a busy host)
    lo, hi = r["valu_pipe_time_frac_bounds"]
    assert 0.0 < lo <= hi <= 1.0
    assert 0.0 < r["hbm"]["traffic_frac_of_peak"] < 0.5 and r["traffic"] > 0  # not the bound
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1 and 0.0 < c["value"] < d["value"]
    assert d["ms_per_lone_frame"] > d["ms_per_frame"] > 0.0
    assert d["first_launch_of_a_view"]["kernel_ms"] > 0.0 and d["strong_scaling_floor"]["ms_per_step"] < d["ms_per_step"]
    assert "binned surface-area splits on the GPU" in d["config"]["workload"]
    g = d["reference_rule_tree"]  # the same job on the tree of the reference's own split rule
    assert g["Mrays_per_s"] > 1000.0 and "reference's split rule" in g["builder"]
    g16 = d["reference_builder_tree"]  # ... and with the reference builder's own leaf size
    assert g16["Mrays_per_s"] > 1000.0 and "at most 16 triangles" in g16["builder"]
