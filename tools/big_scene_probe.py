#!/usr/bin/env python3
"""Scale check of builder and traversal on a scene far larger than the reference's: spheres of `tess` x `tess/2`
quads (two triangles each) in a Cornell-like room. Prints build time, throughput, visit counts and how many rays
overflowed the LDS traversal stack. usage: python tools/big_scene_probe.py [tessU ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

SYSTEM = open(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt")).read()
SCENE = open(os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt")).read()

for tess in [int(a) for a in sys.argv[1:]] or [180, 1000, 2800]:
    scene = SCENE.replace("sphere 180 90", f"sphere {tess} {tess // 2}")
    assert tess == 180 or scene != SCENE
    app = twk.Application(system_text=SYSTEM, scene_text=scene)
    tris = sum(app.geometry(g)[1].shape[0] // 3 for g, *_ in app.instances)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    t0 = time.perf_counter()
    app.initDevice(dev)
    dev.synchronizeStream()
    build_s = time.perf_counter() - t0
    steps = 32
    dev.reserveLaunchBatch(steps)
    for it in range(steps):
        dev.render(it)
    dev.synchronizeStream()
    t0 = time.perf_counter()
    for it in range(steps, 2 * steps):
        dev.render(it)
    dev.synchronizeStream()
    dt = time.perf_counter() - t0
    dev.statsEnable(True)
    dev.statsGet(True)
    for it in range(2 * steps, 3 * steps):
        dev.render(it)
    st = dev.statsGet(True)
    rays = st["radianceRays"] + st["shadowRays"]
    px = app.info.resolution[0] * app.info.resolution[1]
    print(json.dumps({"instanced_triangles": tris, "nodes": int(dev.buildInfo()["nodes"]), "upload_and_build_s": round(build_s, 3), "Msamples_per_s": round(px * steps / dt / 1e6, 1),
                      "nodes_per_ray": round(st["nodesVisited"] / rays, 2), "triangles_per_ray": round(st["trianglesTested"] / rays, 2),
                      "max_nodes_per_ray": st["maxNodesPerRay"], "overflow_rays_per_M": round(1e6 * st["overflowRays"] / rays, 2)}), flush=True)
    dev.close()
