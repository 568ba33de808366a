#!/usr/bin/env python3
"""C4 instances (100 instances of boxes / spheres / tori + ground) under flatten policies: which geometries become world-space
trees (twk_set_flatten_policy(maxTriangles, maxReferences)). usage (GPU box): python tools/probes/flatten_policy_probe.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_instances.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_instances.txt"))
for policy in [(4, 2), (16, 2), (16, 1 << 30), (40000, 2)]:
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    dev.setFlattenPolicy(*policy)
    app.initDevice(dev)
    info = dev.buildInfo()
    n = 64
    for it in range(n):
        dev.render(it)
    dev.synchronizeStream()
    t0 = time.perf_counter()
    for it in range(n, 2 * n):
        dev.render(it)
    dev.synchronizeStream()
    dt = time.perf_counter() - t0
    print(json.dumps({"policy": policy, "flattened": int(info["flattenedInstances"]), "direct_leaves": int(info["directLeafInstances"]), "nodes": int(info["nodes"]),
                      "Msamples_per_s": round(1920 * 1080 * n / dt / 1e6, 1)}), flush=True)
    dev.close()
