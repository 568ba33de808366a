#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel of one HIP source, as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py <file.hip> [name filter] [-- extra hipcc flags]"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
src = args[0]
flt = args[1] if len(args) > 1 else ""
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950", "-Wno-unused-function",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPR Spill", "SGPR Spill"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key.split(" [")[0]] = int(m.group(1))
for r in rows:
    if flt in r["name"]:
        print(f'{r["name"]:90s} vgpr {r.get("VGPRs", -1):4d} scratch {r.get("ScratchSize", -1):4d} occ {r.get("Occupancy", -1):2d} lds {r.get("LDS Size", -1):6d}')
