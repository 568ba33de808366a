#!/usr/bin/env python3
"""The time view of a scene (≙ USE_TIME_VIEW of the reference, twk_set_time_view): renders `iterations` samples per pixel with
the view on and writes the alpha channel — mean shader-clock cycles of a pixel's samples x clockFactor x 1e-9 — through a
black-blue-green-yellow-red ramp (the reference's rasteriser uses a ramp texture, Rasterizer.cpp:300-361) next to the
tonemapped image. usage (GPU box): python tools/time_view.py [out prefix] [width height] [iterations]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import tweeker_raytracer_amd as twk  # noqa: E402
from png import tonemap, write_png  # noqa: E402

prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/time_view"
width, height = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 360)
iterations = int(sys.argv[4]) if len(sys.argv) > 4 else 32
app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
app.setResolution(width, height)
dev = twk.Device(ordinal=0, miss=app.info.miss)
app.initDevice(dev)
dev.setTimeView(True)
for it in range(iterations):
    dev.render(it)
out = dev.getOutputBufferHost()
dev.close()
alpha = out[..., 3]
x = np.clip(alpha / np.percentile(alpha, 99.5), 0.0, 1.0)[::-1]
stops = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [1, 1, 0], [1, 0, 0]], np.float32)
pos = x * (len(stops) - 1)
i0 = np.minimum(pos.astype(np.int32), len(stops) - 2)
f = (pos - i0)[..., None]
ramp = stops[i0] * (1.0 - f) + stops[i0 + 1] * f
os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
write_png(prefix + "_ramp.png", (ramp * 255.0 + 0.5).astype(np.uint8))
write_png(prefix + "_image.png", tonemap(out))
print(f"alpha (cycles x clockFactor x 1e-9): min {alpha.min():.4g} median {np.median(alpha):.4g} max {alpha.max():.4g}; wrote {prefix}_ramp.png, {prefix}_image.png")
