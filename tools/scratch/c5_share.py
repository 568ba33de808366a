import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box_c5.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
dev = twk.Device(ordinal=0, index=3, count=8, miss=app.info.miss)
app.initDevice(dev, distribution=1)
dev.reserveLaunchBatch(min(64, steps))
for rep in range(4):
    t0 = time.perf_counter()
    for it in range(rep * steps, (rep + 1) * steps):
        dev.render(it)
    dev.synchronizeStream()
    dt = time.perf_counter() - t0
    print("rep", rep, "%.1f Msamples/s  %.3f ms" % (dev.launchWidth * app.info.resolution[1] * steps / dt / 1e6, dt * 1e3), flush=True)
dev.close()
