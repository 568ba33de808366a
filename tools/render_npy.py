#!/usr/bin/env python3
"""Renders `iterations` iterations of a scene on GPU 0 with the library TWK_LIB names (default: the in-tree exact build) and
saves the accumulation buffer as .npy — the way tests/test_gpu_native_math.py gets an image out of the approximate build
(libtweeker_hip_fast.so) without loading two builds of the library into one process.
usage: [TWK_LIB=...] python tools/render_npy.py system.txt scene.txt width height iterations out.npy [--procedural-textures]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tweeker_raytracer_amd as twk  # noqa: E402

system, scene, width, height, iterations, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
app = twk.Application(system, scene)
app.setResolution(width, height)
dev = twk.Device(ordinal=0, miss=app.info.miss)
if "--procedural-textures" in sys.argv:
    from procedural import albedo_checker, cutout_slots, environment_hdr
    for slot, img in ((0, albedo_checker()), (1, cutout_slots()), (2, environment_hdr())):
        dev.initTexture(slot, img)
app.initDevice(dev)
for it in range(iterations):
    dev.render(it)
np.save(out, dev.getOutputBufferHost())
print("library", twk.LIB_PATH)
dev.close()
