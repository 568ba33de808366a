#!/usr/bin/env python3
"""Average launch duration of the traversal kernel per wavefront pass, from a rocprofv3 --kernel-trace CSV of
`python bench.py`. The --stats table averages over every launch of the process (warm-up pass of 4 iterations, the
one-iteration passes behind config.batch1_Msamples_per_s, the composite check); bench.py's roofline.avg_launch_ms is the
average over the launches of the TIMED pass only. This groups the launches by pass (a pass ends with accumulateKernel)
and prints the passes of the bench's batch depth, so the two can be compared.
usage: tools/trace_launch_average.py <kernel_trace.csv> [min total ms of a pass to print]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
floor_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
passes, cur = [], {"variants": {}, "launches": []}
for r in rows:
    name = r["Kernel_Name"]
    if "traceKernel<" in name:
        variant = name[name.index("traceKernel"):name.index(">") + 1]
        cur["variants"][variant] = cur["variants"].get(variant, 0) + 1
        cur["launches"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1.0e6)
    elif "accumulateKernel" in name:
        if cur["launches"]:
            cur["variant"] = max(cur["variants"], key=cur["variants"].get)  # the first launch of a pass is the PRIMARY build
            passes.append(cur)
        cur = {"variants": {}, "launches": []}
print("| pass | kernel | launches | total ms | average launch ms |\n|---|---|---|---|---|")
for i, p in enumerate(passes):
    total = sum(p["launches"])
    if total >= floor_ms:
        print(f"| {i} | `{p['variant']}` | {len(p['launches'])} | {total:.3f} | {total / len(p['launches']):.3f} |")
