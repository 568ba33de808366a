#!/usr/bin/env python3
"""Prints the kernels of one wavefront pass in launch order — start, duration, gap to the previous kernel — from a rocprofv3
kernel-trace CSV of `python bench.py`. A pass ends with accumulateKernel (there is no generateKernel any more on the fused
path). usage: tools/step_timeline.py <kernel_trace.csv> [which pass: index among the passes of at least `min ms`, default the
last but one] [min ms of kernel time of a pass, default 5]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
floor_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
names = ("traceKernel<", "shadeKernel", "generateKernel", "accumulateKernel", "traceOverflowKernel", "tileEntryKernel", "tailKernel")
passes, cur = [], []
for r in rows:
    if any(k in r["Kernel_Name"] for k in names):
        cur.append(r)
        if "accumulateKernel" in r["Kernel_Name"]:
            passes.append(cur)
            cur = []
big = [p for p in passes if sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in p) >= floor_ms * 1.0e6]
p = big[int(sys.argv[2])] if len(sys.argv) > 2 else big[-2 if len(big) > 1 else -1]
t0 = int(p[0]["Start_Timestamp"])
prev = t0
busy = 0
for r in p:
    n, s, e = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = n[n.index("twk::") + 5:] if "twk::" in n else n
    short = short[:short.index("(")] if "(" in short else short
    print("%-52s start %9.1f us  dur %8.1f us  gap %5.1f" % (short[:52], (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
    busy += e - s
print("pass total %.1f us, kernels %.1f us, %d launches" % ((prev - t0) / 1e3, busy / 1e3, len(p)))
