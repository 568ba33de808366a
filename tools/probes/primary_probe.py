#!/usr/bin/env python3
"""The traversal launch of the primary rays alone: C2 with pathLengths 1 1 and no light (no shadow rays): rays, visits,
kernel time per step. usage (GPU box): [TWK_TILE_ENTRIES=0] python tools/probes/primary_probe.py [scene system]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "scene_rtigo3_cornell_box.txt"
system = sys.argv[2] if len(sys.argv) > 2 else "system_rtigo3_cornell_box.txt"
text = open(os.path.join(ROOT, "scenes", system)).read()
text = re.sub(r"(?m)^pathLengths .*$", "pathLengths 1 1", text)
text = re.sub(r"(?m)^light .*$", "light 0", text)
app = twk.Application(system_text=text, scene_text=open(os.path.join(ROOT, "scenes", scene)).read())
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
print(json.dumps({"scene": scene, "tile_entries": os.environ.get("TWK_TILE_ENTRIES", "1"), "Mrays_per_step": round(rays / iters / 1e6, 3), "nodes_per_ray": round(s["nodesVisited"] / rays, 3),
                  "tris_per_ray": round(s["trianglesTested"] / rays, 3), "trace_ms_per_step": round(prof["trace"]["ms"] / iters, 4), "shade_ms_per_step": round(prof["shade"]["ms"] / iters, 4)}))
dev.close()
