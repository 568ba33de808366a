#!/usr/bin/env python3
"""LBVH vs binned-SAH acceleration structures (twk_set_build_quality) on every configuration and on the scale probes:
SAH cost terms (twk_get_build_info), build time, wide-node visits / triangle tests per ray, traversal and shade ms per
step, Msamples/s. One JSON line per (scene, quality). usage (GPU box): python tools/bvh_quality.py [--steps 32] [--big]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tweeker_raytracer_amd as twk  # noqa: E402
from procedural import albedo_checker, cutout_slots, environment_hdr  # noqa: E402


def run(name, app, steps, quality, textures=False):
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    dev.setBuildQuality(quality)
    if textures:
        for slot, img in ((0, albedo_checker()), (1, cutout_slots()), (2, environment_hdr())):
            dev.initTexture(slot, img)
    app.initDevice(dev)
    if textures:
        mats = app.materials
        mats[1].useAlbedoTexture = 1
        mats[4].useCutoutTexture = 1
        dev.initMaterials(mats)
    info = dev.buildInfo()
    dev.reserveLaunchBatch(min(64, steps))
    for it in range(steps):
        dev.render(it)
    dev.synchronizeStream()
    t0 = time.perf_counter()
    for it in range(steps, 2 * steps):
        dev.render(it)
    dev.synchronizeStream()
    dt = time.perf_counter() - t0
    dev.profileEnable(True)
    dev.profileReset()
    for it in range(2 * steps, 3 * steps):
        dev.render(it)
    dev.synchronizeStream()
    prof = dev.profileGet()
    dev.profileEnable(False)
    dev.statsEnable(True)
    dev.statsGet(True)
    for it in range(3 * steps, 4 * steps):
        dev.render(it)
    st = dev.statsGet(True)
    rays = max(1, st["radianceRays"] + st["shadowRays"])
    px = dev.launchWidth * app.info.resolution[1]
    print(json.dumps({"scene": name, "quality": "SAH" if quality else "LBVH", "triangle_slots": info["triangleSlots"], "trees": info["trees"],
                      "instances": info["instances"], "flattened": info["flattenedInstances"],
                      "sah_inner": round(info["sahInnerCost"], 2), "sah_leaf": round(info["sahLeafCost"], 2),
                      "build_ms": round(info["buildMilliseconds"], 1), "Msamples_per_s": round(px * steps / dt / 1e6, 1),
                      "nodes_per_ray": round(st["nodesVisited"] / rays, 2), "cached_nodes_per_ray": round(st["cachedNodesVisited"] / rays, 2),
                      "triangles_per_ray": round(st["trianglesTested"] / rays, 2), "instance_entries_per_ray": round(st["instancesEntered"] / rays, 2),
                      "overflow_rays_per_M": round(1e6 * st["overflowRays"] / rays, 2),
                      "trace_ms_per_step": round(prof["trace"]["ms"] / steps, 4), "shade_ms_per_step": round(prof["shade"]["ms"] / steps, 4)}), flush=True)
    dev.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--big", action="store_true", help="also the 2.0 M and 15.7 M triangle probes (C2 room with finer spheres)")
    a = ap.parse_args()
    sc = lambda f: os.path.join(ROOT, "scenes", f)
    scenes = [("C2 cornell", twk.Application(sc("system_rtigo3_cornell_box.txt"), sc("scene_rtigo3_cornell_box.txt")), False),
              ("C3 intro_07", twk.Application(sc("system_intro_07.txt"), sc("scene_intro_07.txt")), True),
              ("C4 geometry", twk.Application(sc("system_rtigo3_geometry.txt"), sc("scene_rtigo3_geometry.txt")), False),
              ("C4 instances", twk.Application(sc("system_rtigo3_instances.txt"), sc("scene_rtigo3_instances.txt")), False)]
    if a.big:
        system = open(sc("system_rtigo3_cornell_box.txt")).read()
        scene = open(sc("scene_rtigo3_cornell_box.txt")).read()
        for tess in (1000, 2800):
            scenes.append((f"C2 room, spheres {tess}x{tess // 2}", twk.Application(system_text=system, scene_text=scene.replace("sphere 180 90", f"sphere {tess} {tess // 2}")), False))
    for name, app, tex in scenes:
        for quality in (0, 1):
            run(name, app, a.steps, quality, tex)


if __name__ == "__main__":
    main()
