#!/bin/bash
# Copies what tools/final_profiles.sh <tag> left under gpurun_out/<tag>/ into profiles/<tag>_* (the tracked summaries).
# usage (here, after the gpurun call): bash tools/copy_final_profiles.sh <tag>
set -e
TAG=${1:-r05z}; SRC=gpurun_out/$TAG; DST=profiles
for n in s20 default; do
  cp $SRC/bench_$n.json $DST/${TAG}_bench_$n.json
  cp $SRC/bench_${n}_profiled.json $DST/${TAG}_bench_${n}_profiled.json
  cp $SRC/kernel_stats_$n.csv $DST/${TAG}_kernel_stats_$n.csv
  { cat $SRC/kernel_stats_$n.md; echo; echo "Per-pass launch averages of the traversal kernel (tools/trace_launch_average.py):"; echo; cat $SRC/trace_launch_average_$n.md; } > $DST/${TAG}_kernel_stats_$n.md
done
cp $SRC/pmc_summary_s20.md $DST/${TAG}_pmc_summary_s20.md
cp $SRC/pmc_summary_s64.md $DST/${TAG}_pmc_summary_s64.md
cp $SRC/scene_perf.jsonl $DST/${TAG}_scene_perf.jsonl
cp $SRC/depth_profile_b64.jsonl $DST/${TAG}_depth_profile_b64.jsonl
cp $SRC/depth_profile_b20.jsonl $DST/${TAG}_depth_profile_b20.jsonl
cp $SRC/big_scene_probe.jsonl $DST/${TAG}_big_scene_probe.jsonl
for n in s20 default; do t=$(find $SRC/stats_$n -name "*kernel_trace.csv" | head -1); python3 tools/step_timeline.py $t > $DST/${TAG}_pass_timeline_$n.txt; done
cp $SRC/r05_trace_hbm_traffic_s20.json $SRC/r05_trace_hbm_traffic_s64.json $SRC/r05_shade_hbm_traffic_s20.json $SRC/r05_shade_hbm_traffic_s64.json $DST/
ls $DST | grep ${TAG}_
