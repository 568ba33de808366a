#!/usr/bin/env python3
"""Prints the top of the acceleration structure the device built for a scene: every wide node down to `levels` levels below
the root with the boxes (decoded from the quantised node) and the kind of each entry. usage: python tools/print_tree_top.py [scene] [levels]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tweeker_raytracer_amd as twk  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "scene_rtigo3_cornell_box.txt"
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
system = open(os.path.join(ROOT, "scenes", sys.argv[3] if len(sys.argv) > 3 else "system_rtigo3_cornell_box.txt")).read()
app = twk.Application(system_text=system, scene_text=open(os.path.join(ROOT, "scenes", scene)).read())
dev = twk.Device(ordinal=0, miss=app.info.miss)
app.initDevice(dev)
info, nodes, tris, inst = dev.readAcceleration()
print(info)
SENT, WORLD = 0x7fffffff, 0x40000000


def show(index, depth):
    w = nodes[index]
    org, cell = w[0:3], w[3:6]
    q = w[6:12].view(np.uint32)
    refs = w[12:16].view(np.int32)
    print("  " * depth + f"node {index}")
    for k in range(4):
        lo = [int((q[c] >> (8 * k)) & 0xff) for c in range(3)]
        hi = [int((q[3 + c] >> (8 * k)) & 0xff) for c in range(3)]
        if lo[0] > hi[0]:
            print("  " * depth + f"   [{k}] unused")
            continue
        blo = [float(org[c] + lo[c] * cell[c]) for c in range(3)]
        bhi = [float(org[c] + hi[c] * cell[c]) for c in range(3)]
        r = int(refs[k])
        box = "(" + ", ".join(f"{a:.2f}..{b:.2f}" for a, b in zip(blo, bhi)) + ")"
        if r >= 0:
            print("  " * depth + f"   [{k}] inner {r} {box}")
            if depth + 1 < levels:
                show(r, depth + 1)
        else:
            p = ~r
            if p & WORLD:
                first, n = p & 0x0fffffff, ((p >> 28) & 3) + 1
                print("  " * depth + f"   [{k}] world leaf: slots {first}..{first + n - 1} (instance {int(tris[first, 7:8].view(np.int32)[0])}) {box}")
            else:
                print("  " * depth + f"   [{k}] leaf payload {p} {box}")


show(info["root"], 0)
if info["root2"] >= 0:
    print("second node of the 8-wide root:")
    show(info["root2"], 0)
dev.close()
