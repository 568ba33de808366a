#!/bin/bash
# Everything profiles/ keeps for one state of the code, in ONE gpurun call: the bench lines (driver's invocation and default),
# the same two commands under rocprofv3 --kernel-trace --stats, the --pmc passes at both launch sizes (the records bench.py
# carries as roofline.traffic), the other configurations at full size, the per-bounce profile and the big-scene probe.
# usage (GPU box): bash tools/final_profiles.sh <tag>          -> gpurun_out/<tag>/...
set -u
TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p $OUT
# PMC passes first: the bench lines below then carry the traffic of THESE kernels
bash tools/pmc_collect.sh $TAG/pmc_s20 20 5 > $OUT/pmc_s20.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_s20 profiles/r05_trace_hbm_traffic_s20.json > /dev/null && cp profiles/r05_trace_hbm_traffic_s20.json $OUT/
python3 tools/pmc_summarize.py $OUT/pmc_s20 $OUT/pmc_summary_s20.md > /dev/null
python3 tools/pmc_traffic.py $OUT/pmc_s20 profiles/r05_shade_hbm_traffic_s20.json --kernel shadeKernel > /dev/null && cp profiles/r05_shade_hbm_traffic_s20.json $OUT/
bash tools/pmc_collect.sh $TAG/pmc_s64 64 64 > $OUT/pmc_s64.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_s64 profiles/r05_trace_hbm_traffic_s64.json > /dev/null && cp profiles/r05_trace_hbm_traffic_s64.json $OUT/
python3 tools/pmc_summarize.py $OUT/pmc_s64 $OUT/pmc_summary_s64.md > /dev/null
python3 tools/pmc_traffic.py $OUT/pmc_s64 profiles/r05_shade_hbm_traffic_s64.json --kernel shadeKernel > /dev/null && cp profiles/r05_shade_hbm_traffic_s64.json $OUT/
echo "pmc done"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_s20.json 2> $OUT/bench_s20.err; echo "bench s20 rc $?"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc $?"
for cfg in "s20 --steps 20 --warmup 5" "default"; do
  set -- $cfg; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o stats -- python3 bench.py "$@" --no-cpu-baseline > $OUT/bench_${name}_profiled.json 2> $OUT/stats_$name.err
  f=$(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1); t=$(find $OUT/stats_$name -name "*kernel_trace.csv" | head -1)
  python3 tools/summarize_rocprof.py $f $OUT/kernel_stats_$name.md; cp $f $OUT/kernel_stats_$name.csv
  python3 tools/trace_launch_average.py $t 5 > $OUT/trace_launch_average_$name.md
done
echo "stats done"
python3 tools/scene_perf.py > $OUT/scene_perf.jsonl 2> /dev/null; echo "scenes done"
python3 tools/depth_profile.py 64 > $OUT/depth_profile_b64.jsonl 2> /dev/null
python3 tools/depth_profile.py 20 > $OUT/depth_profile_b20.jsonl 2> /dev/null
python3 tools/big_scene_probe.py > $OUT/big_scene_probe.jsonl 2> /dev/null; echo "probes done"
tail -c 400 $OUT/bench_s20.json
