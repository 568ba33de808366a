#!/bin/bash
# Like ab_run.sh, at the driver's launch size too: for each "name[:ENV=VALUE,...]" prints the 20-step and the 64-step line.
# usage (GPU box): bash tools/ab_run2.sh base sq8 base:TWK_PACKED_QUEUE=0 ...
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  [[ "$spec" == *:* ]] && envs=${spec#*:}
  lib=""; [[ "$name" != "base" ]] && lib="TWK_LIB=build/lib_$name.so"
  for steps in "20 5" "64 4"; do
    set -- $steps
    out=gpurun_out/ab2_${spec//[:=,\/]/_}_s$1.json
    env $lib ${envs//,/ } timeout -k 10 300 python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline > $out 2>/dev/null
    python3 - "$spec" "$out" $1 <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[2])); rf = r["roofline"]; k = rf["kernel_ms_per_step"]
    print("AB2 %-26s s%-3s %7.1f Msamples/s  trace %.4f shade %.4f  b1 %.1f  occ %.3f %.3f" % (
        sys.argv[1], sys.argv[3], r["value"], k["trace"], k["shade"], r["config"].get("batch1_Msamples_per_s", 0.0),
        rf["lane_occupancy"]["node_step"], rf["lane_occupancy"]["triangle_test"]), flush=True)
except Exception as e:
    print("AB2", sys.argv[1], "failed", e, flush=True)
PY
  done
done
