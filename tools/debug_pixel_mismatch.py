#!/usr/bin/env python3
"""Debug aid (GPU box): render a scene on the HIP device and the oracle one iteration at a time, locate the first
differing pixel, dump the rays the oracle's path traces for it and compare every one of them through
twk_trace_rays (device BVH, single-ray traversal) with the oracle's BVH and brute-force answers.
usage: python tools/debug_pixel_mismatch.py system.txt scene.txt W H iterations"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402
from oracle import orc  # noqa: E402


def main():
    system, scene, w, h, iters = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    app = twk.Application(os.path.join(ROOT, "scenes", system), os.path.join(ROOT, "scenes", scene))
    app.setResolution(w, h)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.setLaunchBatch(1)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    brute = orc.Oracle(miss=app.info.miss)
    brute.loadApplication(app)
    brute.setTraceMode(False)
    for it in range(iters):
        # iteration `it` alone (a fresh accumulation would need iteration 0; compare the running means instead)
        dev.render(it)
        ref.render(it, threads=16)
        g, c = dev.getOutputBufferHost(), ref.getOutputBufferHost()
        diff = (g.view(np.uint32) != c.view(np.uint32)).any(axis=2)
        print(f"iteration {it}: {diff.sum()} differing pixels", flush=True)
        if not diff.any():
            continue
        for (y, x) in np.argwhere(diff)[:4]:
            print(f" pixel x {x} y {y}: gpu {g[y, x]} oracle {c[y, x]}")
            rays = ref.debugPath(it, int(x), int(y))
            for k, r in enumerate(rays):
                any_hit = r[8] != 0
                q = r[None, :8].copy()
                gt, gi = dev.traceRays(q, anyHit=any_hit)
                ot, oi = ref.traceRays(q, anyHit=any_hit)
                bt, bi = brute.traceRays(q, anyHit=any_hit)
                same = np.array_equal(gi, oi) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32))
                sameb = np.array_equal(bi, oi) and np.array_equal(bt.view(np.uint32), ot.view(np.uint32))
                print(f"  ray {k} kind {int(r[8])} o {r[:3]} tmin {r[3]} d {r[4:7]} tmax {r[7]}\n     device {gt[0]} {gi[0]} | oracle bvh {ot[0]} {oi[0]} | oracle brute {bt[0]} {bi[0]} | device==oracle {same} brute==bvh {sameb}")
        break
    dev.close()


if __name__ == "__main__":
    main()
