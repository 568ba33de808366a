#!/usr/bin/env python3
"""Per-bounce picture of the traversal kernel on C2: rays, node / triangle visits, lane occupancies and kernel time of the
launches of depth 0 .. d, from runs with pathLengths.y = 1, 2, ... (differences of cumulative counters).
usage (GPU box): python tools/depth_profile.py [iterations per pass]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 64
app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
prev = None
for maxd in range(1, 11):
    st = app.state
    st.pathLengths[1] = maxd
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.setState(st)
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
    cur = {"rays": s["radianceRays"] + s["shadowRays"], "nodes": s["nodesVisited"], "tris": s["trianglesTested"], "nodeSteps": s["nodeWaveSteps"],
           "triSteps": s["triangleWaveSteps"], "trace_ms": prof["trace"]["ms"], "shade_ms": prof["shade"]["ms"]}
    d = {k: cur[k] - (prev[k] if prev else 0) for k in cur}
    print(json.dumps({"depth_launches_added": maxd, "Mrays_per_step": round(d["rays"] / iters / 1e6, 3), "nodes_per_ray": round(d["nodes"] / max(1, d["rays"]), 2),
                      "tris_per_ray": round(d["tris"] / max(1, d["rays"]), 2), "node_occupancy": round(d["nodes"] / max(1, 64 * d["nodeSteps"]), 3),
                      "tri_occupancy": round(d["tris"] / max(1, 64 * d["triSteps"]), 3), "trace_ms_per_step": round(d["trace_ms"] / iters, 4),
                      "shade_ms_per_step": round(d["shade_ms"] / iters, 4), "Grays_per_s": round(d["rays"] / max(1e-9, d["trace_ms"]) / 1e6, 2)}), flush=True)
    prev = cur
    dev.close()
