"""Minimal PNG writer (zlib only) + the reference's tonemapper (Application.cpp:2253-2297 semantics, simplified
to gamma) for eyeballing renders against reference assets/img/*.png. Not part of the hot path."""
import struct
import zlib

import numpy as np


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def tonemap(rgba, gamma=2.2, exposure=1.0):
    rgb = np.clip(rgba[..., :3] * exposure, 0.0, None)
    rgb = rgb / (1.0 + rgb * 0.25)
    rgb = np.clip(rgb, 0, 1) ** (1.0 / gamma)
    return (rgb[::-1] * 255.0 + 0.5).astype(np.uint8)  # row 0 of the buffer is the bottom of the image
