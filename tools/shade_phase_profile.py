#!/usr/bin/env python3
"""Where the waves of shadeKernel spend their instructions and their time (TwkLaunchStats.shadePhase*: the measurement build
of the kernel tallies, per phase of the shading of a path segment, wave executions, lanes and shader-clock cycles).
Prints one table: lane occupancy of each phase, its share of the kernel's wave time, cycles per wave execution.
usage (GPU box): python tools/shade_phase_profile.py [system.txt scene.txt] [iterations] [--json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

NAMES = ["path (whole shadePath)", "volume stack fetch", "miss program", "hit: records + normals + front face", "GGX tangent", "texcoord",
         "implicit light hit", "sample: Lambert", "sample: mirror", "sample: glass", "sample: GGX brdf", "sample: GGX bsdf",
         "NEE: draws + light sampler", "NEE: BSDF eval + contribution", "radiance read-modify-write", "integrator tail (whole)",
         "volume stack push / pop", "AOV writes", "kernel: wait for the slot's streams", "kernel: append (ballots, barriers, atomic, stores)",
         "kernel: block iteration (whole)", "append: to behind barrier 1 (slowest wave)", "append: the atomic's round trip (issuing lane)",
         "append: barrier 1 to behind barrier 2"]

args = [a for a in sys.argv[1:] if not a.startswith("--")]
as_json = "--json" in sys.argv
system = args[0] if len(args) >= 2 else os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt")
scene = args[1] if len(args) >= 2 else os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt")
iterations = int(args[2]) if len(args) >= 3 else (int(args[0]) if len(args) == 1 else 20)
app = twk.Application(system, scene)
dev = twk.Device(ordinal=0, miss=app.info.miss)
app.initDevice(dev)
dev.setLaunchBatch(min(64, iterations)) if hasattr(dev, "setLaunchBatch") else None
for it in range(iterations):  # warm-up pass, not counted
    dev.render(it)
dev.synchronizeStream()
dev.statsEnable(True)
dev.statsGet(reset=True)
for it in range(iterations, 2 * iterations):
    dev.render(it)
dev.synchronizeStream()
st = dev.statsGet(reset=True)
ws, ln, cy = st["shadePhaseWaveSteps"], st["shadePhaseLanes"], st["shadePhaseCycles"]
ITER = 20
total_cycles = max(1, cy[ITER])
rows = []
for k, name in enumerate(NAMES):
    if ws[k] == 0:
        continue
    rows.append({"phase": name, "wave_executions": ws[k], "lanes": ln[k], "lane_occupancy": round(ln[k] / (64.0 * ws[k]), 4),
                 "share_of_kernel_wave_time": round(cy[k] / total_cycles, 4), "cycles_per_wave_execution": round(cy[k] / ws[k], 1),
                 "executions_per_block_iteration_wave": round(ws[k] / max(1, ws[ITER]), 4)})
out = {"scene": os.path.basename(scene), "iterations": iterations, "shaded_hits": st["shadedHits"], "missed": st["missed"], "phases": rows}
if as_json:
    print(json.dumps(out))
else:
    print(f"# shadeKernel phases, {os.path.basename(scene)}, {iterations} iterations per pass; segments shaded {st['shadedHits'] + st['missed']}")
    print(f"{'phase':52s} {'wave exec':>12s} {'per iter':>9s} {'lanes/64':>9s} {'time share':>11s} {'cycles/exec':>12s}")
    for r in rows:
        print(f"{r['phase']:52s} {r['wave_executions']:12d} {r['executions_per_block_iteration_wave']:9.3f} {r['lane_occupancy']:9.3f} {r['share_of_kernel_wave_time']:11.3f} {r['cycles_per_wave_execution']:12.1f}")
dev.close()
