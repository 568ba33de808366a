#!/usr/bin/env python3
"""Throughput of the other configurations of BASELINE.json at their full size (parity-test cases, not bench lines):
C3 intro_07 (spherical HDR environment, albedo texture, cutout opacity — procedural pictures as in the tests),
C4 geometry and instances scenes, C5's per-rank share (device 3 of 8 of the 3840x2160 frame).
usage: python tools/scene_perf.py [--steps 64]"""
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


def run(name, system, scene, steps, textures=False, index=0, count=1):
    app = twk.Application(os.path.join(ROOT, "scenes", system), os.path.join(ROOT, "scenes", scene))
    dev = twk.Device(ordinal=0, index=index, count=count, miss=app.info.miss)
    if textures:
        for slot, img in ((0, albedo_checker()), (1, cutout_slots()), (2, environment_hdr())):
            dev.initTexture(slot, img)
    app.initDevice(dev, distribution=1 if count > 1 else None)
    if textures:
        mats = app.materials
        mats[1].useAlbedoTexture = 1
        mats[4].useCutoutTexture = 1
        dev.initMaterials(mats)
    dev.reserveLaunchBatch(min(64, steps))
    for it in range(steps):
        dev.render(it)
    dev.synchronizeStream()
    dev.profileEnable(True)
    dev.profileReset()
    t0 = time.perf_counter()
    for it in range(steps, 2 * steps):
        dev.render(it)
    dev.synchronizeStream()
    dt = time.perf_counter() - t0
    prof = dev.profileGet()
    pixels = dev.launchWidth * app.info.resolution[1]
    out = {"scene": name, "pixels": pixels, "steps": steps, "Msamples_per_s_profiled": pixels * steps / dt / 1e6,
           "kernel_ms_per_step": {k: round(v["ms"] / steps, 4) for k, v in prof.items()}}
    dev.profileEnable(False)
    t0 = time.perf_counter()
    for it in range(2 * steps, 3 * steps):
        dev.render(it)
    dev.synchronizeStream()
    out["Msamples_per_s"] = round(pixels * steps / (time.perf_counter() - t0) / 1e6, 1)
    out["Msamples_per_s_profiled"] = round(out["Msamples_per_s_profiled"], 1)
    dev.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=64)
    a = ap.parse_args()
    run("C2 cornell", "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", a.steps)
    run("C3 intro_07", "system_intro_07.txt", "scene_intro_07.txt", a.steps, textures=True)
    run("C4 geometry", "system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", a.steps)
    run("C4 instances", "system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", a.steps)
    run("C5 share 3/8", "system_rtigo3_cornell_box_c5.txt", "scene_rtigo3_cornell_box.txt", a.steps, index=3, count=8)
