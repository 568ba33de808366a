#!/usr/bin/env python3
"""Where the waves of the persistent traversal kernel spend their time (TwkLaunchStats.waveCycles): renders a few
iterations with the counting kernel variant and prints the share of wave time per phase of the outer loop next to the
work counters. usage (GPU box): python tools/phase_profile.py [system.txt scene.txt] [iterations]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

args = sys.argv[1:]
system = args[0] if len(args) >= 2 else os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt")
scene = args[1] if len(args) >= 2 else os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt")
iterations = int(args[2]) if len(args) >= 3 else (int(args[0]) if len(args) == 1 else 64)
app = twk.Application(system, scene)
dev = twk.Device(ordinal=0, miss=app.info.miss)
app.initDevice(dev)
for it in range(iterations):  # warm-up pass, not counted
    dev.render(it)
dev.synchronizeStream()
dev.statsEnable(True)
dev.statsGet(reset=True)
for it in range(iterations, 2 * iterations):
    dev.render(it)
dev.synchronizeStream()
st = dev.statsGet(reset=True)
names = ["refill (ray fetch)", "node loop", "leaf / instance step", "triangle loop", "pop + result write"]
total = max(1, st["waveCycles"][5])
rays = st["radianceRays"] + st["shadowRays"]
out = {"rays": rays, "nodes_per_ray": st["nodesVisited"] / max(1, rays), "triangles_per_ray": st["trianglesTested"] / max(1, rays),
       "node_step_lane_occupancy": st["nodesVisited"] / max(1, 64 * st["nodeWaveSteps"]),
       "triangle_lane_occupancy": st["trianglesTested"] / max(1, 64 * st["triangleWaveSteps"]),
       "wave_cycles_per_node_wave_step": st["waveCycles"][1] / max(1, st["nodeWaveSteps"]),
       "wave_cycles_per_triangle_wave_step": st["waveCycles"][3] / max(1, st["triangleWaveSteps"]),
       "wave_cycles_per_leaf_wave_step": st["waveCycles"][2] / max(1, st["leafWaveSteps"]),
       "share_of_wave_time": {n: round(st["waveCycles"][k] / total, 4) for k, n in enumerate(names)}}
print(json.dumps(out, indent=1))
dev.close()
