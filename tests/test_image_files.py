"""The image writers that stand in for the reference's swapchain copy (vulkan.rs:1045-1101),
read back with a reader written here from the PNG and OpenEXR file layouts."""
import ctypes as C
import struct
import zlib

import numpy as np

from ray_tracing_amd import _lib as L

F = np.float32


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body)
        chunks.append((tag, body))
        pos += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"gAMA", b"IDAT", b"IEND"]
    w, h, depth, colour, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, filt, lace) == (8, 2, 0, 0, 0)
    assert struct.unpack(">I", chunks[1][1]) == (45455,)
    raw = np.frombuffer(zlib.decompress(chunks[2][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def read_exr(path):
    data = open(path, "rb").read()
    assert struct.unpack("<II", data[:8]) == (20000630, 2)
    pos, attrs = 8, {}
    while data[pos] != 0:
        e = data.index(b"\0", pos)
        name = data[pos:e].decode()
        e2 = data.index(b"\0", e + 1)
        kind = data[e + 1:e2].decode()
        (n,) = struct.unpack("<I", data[e2 + 1:e2 + 5])
        attrs[name] = (kind, data[e2 + 5:e2 + 5 + n])
        pos = e2 + 5 + n
    pos += 1
    for required in ("channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio",
                     "screenWindowCenter", "screenWindowWidth"):
        assert required in attrs
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, cp, ch = [], 0, attrs["channels"][1]
    while ch[cp] != 0:
        e = ch.index(b"\0", cp)
        names.append(ch[cp:e].decode())
        assert struct.unpack("<iB3xii", ch[e + 1:e + 17]) == (2, 0, 1, 1)
        cp = e + 17
    assert names == ["B", "G", "R"]
    offsets = struct.unpack(f"<{h}Q", data[pos:pos + 8 * h])
    img = np.empty((h, w, 3), F)
    for y, off in enumerate(offsets):
        yy, n = struct.unpack("<iI", data[off:off + 8])
        assert yy == y and n == 12 * w
        planes = np.frombuffer(data[off + 8:off + 8 + n], F).reshape(3, w)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = planes[0], planes[1], planes[2]
    assert offsets[-1] + 8 + 12 * w == len(data)
    return img


def test_png_holds_the_displayed_image_rounded_to_bytes(tmp_path):
    lib = L.load()
    rng = np.random.default_rng(3)
    w, h = 301, 233  # more than one 65535-byte stored block
    img = rng.uniform(-0.2, 1.2, (h, w, 4)).astype(F)
    img[0, 0, :3] = [np.nan, 0.5, 1.0]
    img[0, 1, :3] = [np.inf, -np.inf, 0.0]
    path = tmp_path / "a.png"
    assert lib.rt_save_png(str(path).encode(), _vp(img), w, h) == 0
    got = read_png(path)
    want = np.rint(np.clip(np.nan_to_num(img[..., :3], nan=0.0, posinf=1.0, neginf=0.0), 0, 1) * F(255)).astype(np.uint8)
    assert (got == want).all()
    assert list(got[0, 0]) == [0, 128, 255] and list(got[0, 1]) == [255, 0, 0]


def test_exr_holds_the_scaled_floats_exactly(tmp_path):
    lib = L.load()
    rng = np.random.default_rng(4)
    w, h = 37, 19
    acc = rng.uniform(0, 50, (h, w, 4)).astype(F)
    path = tmp_path / "a.exr"
    assert lib.rt_save_exr(str(path).encode(), _vp(acc), w, h, F(0.25)) == 0
    got = read_exr(path)
    assert (got.view(np.uint32) == (acc[..., :3] * F(0.25)).view(np.uint32)).all()


def test_writers_report_bad_arguments_and_unwritable_paths(tmp_path):
    lib = L.load()
    img = np.zeros((2, 2, 4), F)
    assert lib.rt_save_png(None, _vp(img), 2, 2) == L.RT_ERR_INVALID
    assert lib.rt_save_png(b"x.png", None, 2, 2) == L.RT_ERR_INVALID
    assert lib.rt_save_exr(str(tmp_path / "a.exr").encode(), _vp(img), 0, 2, F(1)) == L.RT_ERR_INVALID
    assert lib.rt_save_png(str(tmp_path / "no_such_dir" / "a.png").encode(), _vp(img), 2, 2) == L.RT_ERR_IO


import pytest


@pytest.mark.gpu
def test_app_saves_what_it_renders(tmp_path):
    from ray_tracing_amd import App, default_scene
    scene, cam = default_scene(96, 54)
    app = App.create(cam, scene)
    app.render_frames(4)
    app.synchronize()
    app.save_png(tmp_path / "f.png")
    app.save_exr(tmp_path / "f.exr")
    present, acc = app.read_present(), app.read_acc()
    want = np.rint(np.clip(np.nan_to_num(present[..., :3]), 0, 1) * F(255)).astype(np.uint8)
    assert (read_png(tmp_path / "f.png") == want).all()
    assert (read_exr(tmp_path / "f.exr").view(np.uint32) == (acc[..., :3] * F(0.25)).view(np.uint32)).all()
    app.destroy()
