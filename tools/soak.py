#!/usr/bin/env python3
"""Soak run: every shipped scene at several resolutions, batch depths and flatten policies for a few hundred iterations each;
checks that every launch returns, the image is finite and the NaN-free running mean moves. usage (GPU box): python tools/soak.py [seconds]"""
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tweeker_raytracer_amd as twk  # noqa: E402
from procedural import albedo_checker, cutout_slots, environment_hdr  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
scenes = [("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", False), ("system_intro_07.txt", "scene_intro_07.txt", True),
          ("system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", False), ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", False)]
sizes = [(1920, 1080), (333, 77), (64, 40), (1280, 720)]
batches = [64, 1, 7, 16]
policies = [(4, 2), (0, 0), (1 << 30, 1 << 30)]
t_end = time.time() + budget
runs = 0
for (system, scene, textured), size, batch, policy in itertools.cycle(itertools.product(scenes, sizes, batches, policies)):
    if time.time() > t_end:
        break
    text = open(os.path.join(ROOT, "scenes", system)).read()
    text = "\n".join(l for l in text.splitlines() if not l.startswith("resolution")) + f"\nresolution {size[0]} {size[1]}\n"
    app = twk.Application(system_text=text, scene_text=open(os.path.join(ROOT, "scenes", scene)).read())
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    if textured:
        for slot, img in ((0, albedo_checker()), (1, cutout_slots()), (2, environment_hdr())):
            dev.initTexture(slot, img)
    dev.setFlattenPolicy(*policy)
    app.initDevice(dev)
    dev.setLaunchBatch(batch)
    n = 96 if size[0] * size[1] > 500000 else 256
    for it in range(n):
        dev.render(it)
    img = dev.getOutputBufferHost()
    assert np.isfinite(img).all() and img[..., :3].max() > 0.0 and np.all(img[..., 3] == 1.0), (system, size, batch, policy)
    dev.close()
    runs += 1
    print(json.dumps({"scene": scene, "size": size, "batch": batch, "policy": policy, "iterations": n, "mean": float(img[..., :3].mean())}), flush=True)
print("soak ok:", runs, "configurations")
