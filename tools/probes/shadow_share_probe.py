#!/usr/bin/env python3
"""What the shadow rays cost: C2 with and without next-event estimation (light 1 / light 0; the paths differ in their draws but
not in kind): rays, visits and traversal time per step. usage (GPU box): python tools/probes/shadow_share_probe.py"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

base = open(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt")).read()
scene = open(os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt")).read()
for light in (1, 0):
    text = re.sub(r"(?m)^light .*$", f"light {light}", base)
    app = twk.Application(system_text=text, scene_text=scene)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    iters = 64
    for it in range(iters):
        dev.render(it)
    dev.synchronizeStream()
    dev.profileReset(); dev.profileEnable(True)
    for it in range(iters, 2 * iters):
        dev.render(it)
    dev.synchronizeStream()
    prof = dev.profileGet(); dev.profileEnable(False)
    dev.statsEnable(True); dev.statsGet(True)
    for it in range(2 * iters, 3 * iters):
        dev.render(it)
    dev.synchronizeStream()
    s = dev.statsGet(True)
    rays = s["radianceRays"] + s["shadowRays"]
    print(json.dumps({"light": light, "radiance_Mrays_per_step": round(s["radianceRays"] / iters / 1e6, 3), "shadow_Mrays_per_step": round(s["shadowRays"] / iters / 1e6, 3),
                      "nodes_per_ray": round(s["nodesVisited"] / rays, 3), "tris_per_ray": round(s["trianglesTested"] / rays, 3),
                      "Mnodes_per_step": round(s["nodesVisited"] / iters / 1e6, 2), "Mtris_per_step": round(s["trianglesTested"] / iters / 1e6, 2),
                      "trace_ms_per_step": round(prof["trace"]["ms"] / iters, 4), "shade_ms_per_step": round(prof["shade"]["ms"] / iters, 4)}))
    dev.close()
