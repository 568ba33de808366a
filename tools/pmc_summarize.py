#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one row per dispatch and counter) per kernel: sum of each counter, dispatch count.
usage: tools/pmc_summarize.py gpurun_out/pmc [out.md]"""
import collections
import csv
import glob
import os
import sys


def short(name):
    if "traceKernel<" in name:
        return name[name.index("traceKernel<"):name.index(">", name.index("traceKernel<")) + 1]
    for k in ("shadeKernel", "generateKernel", "accumulateKernel"):
        if k in name:
            return k
    return None


def main(root, out=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r.get("Kernel_Name", ""))
            if k is None:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    lines = []
    for k in sorted(agg):
        lines.append(f"### {k}")
        lines.append("| counter | sum over dispatches | dispatches | per dispatch |")
        lines.append("|---|---|---|---|")
        for c in sorted(agg[k]):
            n = len(disp[(k, c)])
            lines.append(f"| {c} | {agg[k][c]:.6g} | {n} | {agg[k][c] / max(1, n):.6g} |")
        lines.append("")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
