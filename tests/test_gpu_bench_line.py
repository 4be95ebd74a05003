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
    assert 0.5 * d["ms_per_step"] < r["kernel_ms"] <= d["ms_per_step"] * 1.01  # the step is that one kernel (slack for a busy host)
    # the clock the kernel ran at comes from the kernel's own counters; the peak is SIMDs x clock / 2
    assert 1000.0 < r["kernel_clock_mhz"] < 2600.0
    assert abs(r["peak"] - 1024 * r["kernel_clock_mhz"] * 1e6 / 2 / 1e9) < 1e-6 * r["peak"]
    with open(os.path.join(ROOT, "profiles", "counters_bench.json")) as f:
        committed = json.load(f)
    if r["profile_stale"] is None:
        # the committed hardware counters are of these very sources and this very traversal: the fraction
        # is then one formed in the kernel's clock domain and agrees with the profile's own cycle-domain
        # figure (vector instructions x 2 over the SIMD cycles GRBM_GUI_ACTIVE gives) within 2 %
        assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert abs(r["frac"] - 2.0 * r["valu_insts_per_launch"] / r["kernel_simd_cycles"]) < 1e-9
        assert abs(r["frac"] / committed["valu_issue_floor_frac"] - 1.0) < 0.02
        lo, hi = r["valu_pipe_time_frac_bounds"]
        assert 0.0 < lo <= hi <= 1.0
        assert 0.0 < r["hbm"]["traffic_frac_of_peak"] < 0.5 and r["traffic"] > 0  # not the bound
    else:
        # sources or tree changed since the profile was taken: no fraction rather than a stale one
        assert r["frac"] is None and r["achieved"] is None and isinstance(r["profile_stale"], str)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1 and 0.0 < c["value"] < d["value"]
    assert d["ms_per_lone_frame"] > d["ms_per_frame"] > 0.0
    assert d["first_launch_of_a_view"]["kernel_ms"] > 0.0 and d["strong_scaling_floor"]["ms_per_step"] < d["ms_per_step"]
    assert "binned surface-area splits on the GPU" in d["config"]["workload"]
    g = d["reference_rule_tree"]  # the same job on the tree of the reference's own split rule
    assert g["Mrays_per_s"] > 1000.0 and "reference's split rule" in g["builder"]
    g16 = d["reference_builder_tree"]  # ... and with the reference builder's own leaf size
    assert g16["Mrays_per_s"] > 1000.0 and "at most 16 triangles" in g16["builder"]


def test_bench_started_plainly_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (both on this
    box's one GPU under RT_BENCH_REHEARSE: the timings mean nothing, the N > 1 code path is what runs)."""
    env = dict(os.environ, RT_BENCH_REHEARSE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert "1920x1080" in d["config"]["workload"] and "2 GPU(s)" in d["config"]["partition"]
    assert d["value"] > 0.0 and d["weak_scaling"]["Mrays_per_s"] > 0.0
    assert d["strong_scaling_floor"]["ms_per_step"] > 0.0
    assert d["gather_ms"] > 0.0  # the strips brought together on rank 0, timed beside the step
