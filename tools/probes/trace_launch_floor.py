"""Duration of ONE persistent trace launch against the number of rays in its queue (twk_debug_trace_queue), under rocprofv3 --kernel-trace:
rays that miss the scene at once (launch overhead) and random rays inside the Cornell box (longest-ray latency).
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d out -o kt -- python3 tools/probes/trace_launch_floor.py
                 python3 tools/probes/trace_launch_floor_report.py out/.../kt_kernel_trace.csv"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tools/probes/ -> repository root
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk
app = twk.Application(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt"), os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
dev = twk.Device(ordinal=0, miss=app.info.miss)
app.initDevice(dev)
dev.render(0); dev.synchronizeStream()
rng = np.random.default_rng(5)
for n in (64, 1024, 16384, 65536, 262144, 1048576, 2073600):
    o = np.stack([rng.uniform(-0.9, 0.9, n), rng.uniform(0.1, 1.9, n), rng.uniform(-0.9, 0.9, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    inside = np.concatenate([o, np.full((n, 1), 5e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], 1).astype(np.float32)
    away = inside.copy(); away[:, 0:3] = (50.0, 50.0, 50.0); away[:, 4:7] = (0.0, 1.0, 0.0)
    for rays in (away, inside):
        for rep in range(3):
            dev.debugTraceQueue(closest=rays)
dev.close()
