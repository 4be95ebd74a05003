"""A C++ host that uses nothing but the C ABI (examples/render_default.cpp, the shape of the
reference's main.rs) must produce, bit for bit, what the Python mirror produces."""
import os
import subprocess

import numpy as np
import pytest

from ray_tracing_amd import App, default_scene
from ray_tracing_amd import _lib as L

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "_build", "render_default")


def test_cpp_host_renders_the_start_up_scene_like_the_python_mirror(tmp_path):
    if not os.path.exists(EXE):  # normally built by __graft_entry__.build()
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    w, h, frames = 320, 180, 3
    png, acc_file = tmp_path / "out.png", tmp_path / "acc.bin"
    res = os.path.join(ROOT, "ray_tracing_amd", "resources")
    out = subprocess.run([EXE, res, str(png), str(frames), str(w), str(h), str(acc_file)], check=True,
                         capture_output=True, text=True, timeout=120).stdout
    assert "merged: " in out and "BVH nodes" in out and f"{frames} frames of {w}x{h}" in out
    got = np.fromfile(acc_file, np.float32).reshape(h, w, 4)
    scene, cam = default_scene(w, h)
    with App.create(cam, scene) as app:
        for _ in range(frames):  # one render per event-loop turn, main.rs:129-149
            app.render_frames(1)
        want = app.read_acc()
    assert np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).reshape(h, w, 4).view(np.uint32))
    assert (got[..., 3] == frames).all() or got[..., :3].max() > 0
    data = png.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and len(data) > 1000


def test_cpp_host_with_a_group_of_contexts_renders_the_same_image(tmp_path):
    """`--gpus 3`: the event loop owns a group (three contexts on this box's one GPU) and presents from
    the first context, which holds the assembled image: the same accumulator bit for bit."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    w, h, frames = 320, 180, 3
    res = os.path.join(ROOT, "ray_tracing_amd", "resources")
    accs = []
    for extra in ([], ["--gpus", "3"]):
        png, acc_file = tmp_path / f"out{len(extra)}.png", tmp_path / f"acc{len(extra)}.bin"
        out = subprocess.run([EXE, res, str(png), str(frames), str(w), str(h), str(acc_file)] + extra, check=True,
                             capture_output=True, text=True, timeout=120).stdout
        assert ("group of 3 contexts" in out) == bool(extra)
        accs.append(np.fromfile(acc_file, np.uint32))
        assert png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    assert np.array_equal(accs[0], accs[1])
