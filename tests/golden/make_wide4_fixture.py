#!/usr/bin/env python3
"""Generates tests/golden/wide4_small_room.npz ON THE GPU BOX: the quantised 4-ary nodes and the triangle slots twk_build
produced for a small Cornell room (spheres 24 x 12), a fixed set of rays, and the hit records of the persistent kernel for
them. The CPU suite walks this tree with oracle/same_bvh_walk.cpp (orc_walk_same_bvh) and compares with the device's records
and with the oracle's brute force over the same scene (tests/test_oracle_golden.py). Data only: arrays.
usage (GPU box): python tests/golden/make_wide4_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
WIDTH = 4
if __name__ == "__main__":
    os.environ["TWK_TILE_ENTRIES"] = "0"
import tweeker_raytracer_amd as twk  # noqa: E402

SYSTEM = open(os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt")).read()
SCENE = open(os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt")).read().replace("sphere 180 90", "sphere 24 12")


def fixture_rays():
    rng = np.random.default_rng(8)
    n = 4096
    o = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(0.05, 1.9, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:64, 0] = 0.0
    d[64:128, 1] = 0.0
    d[128:160, 1:] = 0.0
    d[:160] /= np.linalg.norm(d[:160] + 1e-30, axis=1, keepdims=True)
    return np.concatenate([o, np.full((n, 1), 5e-5, np.float32), d.astype(np.float32), np.full((n, 1), 1e27, np.float32)], 1).astype(np.float32)


if __name__ == "__main__":
    app = twk.Application(system_text=SYSTEM, scene_text=SCENE)
    app.setResolution(32, 32)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    info, nodes, tris, inst = dev.readAcceleration()
    assert info["nodeFloats"] == 16
    rays = fixture_rays()
    dev.statsEnable(True)
    dev.statsGet(True)
    rec, instance, _ = dev.debugTraceQueue(rays, None)
    st = dev.statsGet(True)
    prim = np.where(instance >= 0, tris[:, 3].view(np.int32)[np.maximum(rec[:, 3].view(np.int32), 0)], -1).astype(np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "wide%d_small_room.npz" % WIDTH), nodes=nodes, triangles=tris, rays=rays,
                        root=np.int32(info["root"]), root2=np.int32(info["root2"]),
                        device_tbg=rec[:, :3].copy(), device_instance=instance.astype(np.int32), device_primitive=prim,
                        device_nodes_visited=np.int64(st["nodesVisited"]), device_triangles_tested=np.int64(st["trianglesTested"]))
    print("nodes", nodes.shape, "triangles", tris.shape, "hits", int((instance >= 0).sum()), "of", rays.shape[0], "visits", st["nodesVisited"], st["trianglesTested"])
    dev.close()
