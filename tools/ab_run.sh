#!/bin/bash
# Runs bench.py (no CPU baseline) for each "name[:ENV=VALUE]" argument: name = build/lib_<name>.so variant ("base" = the in-tree library).
# name wt_<x> = an older tree under build/wt_<x> (git worktree add + make there).
# usage (GPU box): [STEPS=20 WARMUP=5] bash tools/ab_run.sh base base:TWK_MAX_LEAF=3 k32 ...
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  [[ "$spec" == *:* ]] && envs=${spec#*:}
  lib=""; [[ "$name" != "base" ]] && lib="TWK_LIB=build/lib_$name.so"
  out=gpurun_out/ab_${spec//[:=]/_}_s${STEPS:-64}.json
  if [[ "$name" == wt_* ]]; then   # a whole older tree built under build/<name> (git worktree): its own bench.py and library
    ( cd build/$name && env ${envs//,/ } timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-64} --warmup ${WARMUP:-4} ) > $out 2>/dev/null
  else
  env $lib ${envs//,/ } timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${STEPS:-64} --warmup ${WARMUP:-4} > $out 2>/dev/null
  fi
  python - "$spec" "$out" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); rf = r["roofline"]; k = rf["kernel_ms_per_step"]
    print("AB %-28s %7.1f Msamples/s  trace %.4f shade %.4f  nodes/ray %.2f tris/ray %.2f  occ %.3f %.3f" % (
        sys.argv[1], r["value"], k["trace"], k["shade"], rf["nodes_per_ray"], rf["triangles_per_ray"],
        rf["lane_occupancy"]["node_step"], rf["lane_occupancy"]["triangle_test"]), flush=True)
except Exception as e:
    print("AB", sys.argv[1], "failed", e, flush=True)
PY
done
