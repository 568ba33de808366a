#!/usr/bin/env python3
"""Large-footprint check: ONE handle renders the whole 3840x2160 frame of config C5, 64 iterations in a single wavefront
pass (531 M paths, 183 GB of path streams, stream offsets beyond 2^32 bytes), and a crop is compared with the oracle
after the same 64 iterations. usage: python tools/stress_4k_single_pass.py"""
import sys, time, numpy as np, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk
from oracle import orc
app = twk.Application(os.path.join(ROOT, "scenes/system_rtigo3_cornell_box_c5.txt"), os.path.join(ROOT, "scenes/scene_rtigo3_cornell_box.txt"))
print("resolution", list(app.info.resolution), "strategy", app.info.strategy, flush=True)
dev = twk.Device(ordinal=0, miss=app.info.miss)        # ONE device renders the whole 3840x2160 frame
app.initDevice(dev, distribution=0)
steps = 64
t0 = time.perf_counter()
dev.reserveLaunchBatch(steps)
print("reserve %.2f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
for it in range(steps): dev.render(it)
dev.synchronizeStream()
dt = time.perf_counter() - t0
px = app.info.resolution[0] * app.info.resolution[1]
print("64 iterations of %d px in one pass: %.3f s = %.1f Msamples/s" % (px, dt, px * steps / dt / 1e6), flush=True)
img = dev.getOutputBufferHost()
x0, y0, x1, y1 = 1900, 1000, 1964, 1032
ref = orc.Oracle(miss=app.info.miss)
st = app.state; st.distribution = 0
ref.loadApplication(app, state=st)
for it in range(steps): ref.render(it, rect=(x0, y0, x1, y1), threads=8)
cpu = ref.getOutputBufferHost()
same = np.array_equal(img[y0:y1, x0:x1].view(np.uint32), cpu[y0:y1, x0:x1].view(np.uint32))
print("crop bit-identical to the oracle after 64 iterations:", same, flush=True)
dev.close()
sys.exit(0 if same else 1)
