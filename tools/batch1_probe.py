#!/usr/bin/env python3
"""Rate at ONE wavefront pass per iteration (what a caller that synchronises after every launch gets), with and without
the tail kernel of tools/experiments/r04_tail_kernel.patch, TWK_TAIL_DEPTH, where that patch is applied). usage (GPU box): python tools/batch1_probe.py [batch ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
for batch in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.setLaunchBatch(batch)
    n = 32
    for it in range(n):
        dev.render(it)
    dev.synchronizeStream()
    t0 = time.perf_counter()
    for it in range(n, 3 * n):
        dev.render(it)
        if batch == 1:
            dev.synchronizeStream()  # the reference's per-iteration cuStreamSynchronize (DeviceSingleGPU.cpp:147)
    dev.synchronizeStream()
    dt = time.perf_counter() - t0
    print(json.dumps({"tail_depth": os.environ.get("TWK_TAIL_DEPTH", "0"), "batch": batch, "Msamples_per_s": round(1920 * 1080 * 2 * n / dt / 1e6, 1),
                      "ms_per_iteration": round(dt * 1e3 / (2 * n), 3)}), flush=True)
    dev.close()
