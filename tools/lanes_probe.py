#!/usr/bin/env python3
"""Pass lanes (device_api.hip renderPass): rate of C2 at several pass sizes with the pass cut into 1 / 2 / 3 lanes
(TWK_PASS_LANES), the image checked against the one-lane image bit for bit.
usage (GPU box): python tools/lanes_probe.py [batch ...]"""
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import tweeker_raytracer_amd as twk
    batch, res = int(sys.argv[2]), (int(sys.argv[3]), int(sys.argv[4]))
    app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
    app.setResolution(*res)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.setLaunchBatch(batch)
    n = max(batch, 16)
    n = (n + batch - 1) // batch * batch
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
    img = dev.getOutputBufferHost()
    print(json.dumps({"lanes": os.environ.get("TWK_PASS_LANES", "auto"), "batch": batch, "resolution": list(res), "Msamples_per_s": round(res[0] * res[1] * 2 * n / dt / 1e6, 1),
                      "ms_per_iteration": round(dt * 1e3 / (2 * n), 4), "crc32": "%08x" % (zlib.crc32(img.tobytes()) & 0xffffffff)}), flush=True)
    dev.close()
    sys.exit(0)

res = (1920, 1080)
for batch in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 20, 64]:
    for lanes in os.environ.get("TWK_PROBE_LANES", "1 2 3 4").split():
        env = dict(os.environ, TWK_PASS_LANES=lanes)
        subprocess.run([sys.executable, __file__, "--child", str(batch), str(res[0]), str(res[1])], env=env, check=False)
