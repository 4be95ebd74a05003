"""Writes a presented frame (binding 8 of the reference's kernel: gamma-corrected
RGBA32F, shader/src/lib.rs:101-108) to an 8-bit PNG.  Stands in for the copy of
the storage image into the swapchain (program/src/modules/vulkan.rs:1045-1101)
on machines without a display.  Pure Python (zlib), no image library."""
import struct
import zlib

import numpy as np


def save_png(path, rgba):
    """rgba: (H, W, 4) float array of the present buffer; values are clipped to [0, 1]."""
    img = np.asarray(rgba, np.float32)
    h, w = img.shape[:2]
    rgb = np.nan_to_num(img[..., :3], nan=0.0, posinf=1.0, neginf=0.0)
    rgb8 = (np.clip(rgb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))
