#!/bin/bash
# The HBM roofline of the traversal kernel on scenes the caches do not hold (profiles/r03_big_scene_pmc.md): PMC passes, traffic
# records and bench lines of `python bench.py --sphere-tess N --steps 32 --warmup 32` for N = 1000 (2.0 M triangles) and 2800
# (15.7 M). usage (GPU box): bash tools/big_scene_profiles.sh <tag>   -> gpurun_out/<tag>/..., profiles/r05_trace_hbm_traffic_tessN_s32.json
set -u
TAG=${1:-big}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
for N in 1000 2800; do
  bash tools/pmc_collect.sh $TAG/pmc_tess$N 32 32 --sphere-tess $N > $OUT/pmc_tess$N.log 2>&1
  python3 tools/pmc_traffic.py $OUT/pmc_tess$N profiles/r05_trace_hbm_traffic_tess${N}_s32.json --sphere-tess $N > /dev/null && cp profiles/r05_trace_hbm_traffic_tess${N}_s32.json $OUT/
  python3 tools/pmc_summarize.py $OUT/pmc_tess$N $OUT/pmc_counters_tess$N.md > /dev/null
  echo "pmc tess $N done"
  python3 bench.py --sphere-tess $N --steps 32 --warmup 32 --no-cpu-baseline > $OUT/bench_tess$N.json 2> $OUT/bench_tess$N.err; echo "bench tess $N rc $?"
done
