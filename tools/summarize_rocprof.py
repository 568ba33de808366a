#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats run (…_kernel_stats.csv) into a short table for profiles/."""
import csv
import sys


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            name = r["Name"]
            if len(name) > 70:
                name = name[:67] + "..."
            f.write(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                    f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {100 * float(r['TotalDurationNs']) / total:.2f} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
