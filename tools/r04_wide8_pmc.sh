#!/bin/bash
# Round 4: the counters behind "the compressed 8-ary nodes are slower": PMC passes of the same command line with the 4-ary and
# the 8-ary nodes, on C2 (driver's launch size) and on the 15.7 M-triangle room.   -> gpurun_out/r04w/...
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04w; mkdir -p $OUT
for w in 0 1; do
  export TWK_WIDE8=$w
  bash tools/pmc_collect.sh r04w/pmc_c2_w$w 20 5 > $OUT/pmc_c2_w$w.log 2>&1
  python3 tools/pmc_traffic.py $OUT/pmc_c2_w$w $OUT/traffic_c2_w$w.json > /dev/null
  python3 tools/pmc_summarize.py $OUT/pmc_c2_w$w $OUT/pmc_summary_c2_w$w.md > /dev/null
  echo "c2 wide8=$w done"
  bash tools/pmc_collect.sh r04w/pmc_tess2800_w$w 32 32 --sphere-tess 2800 > $OUT/pmc_tess2800_w$w.log 2>&1
  python3 tools/pmc_traffic.py $OUT/pmc_tess2800_w$w $OUT/traffic_tess2800_w$w.json --sphere-tess 2800 > /dev/null
  python3 tools/pmc_summarize.py $OUT/pmc_tess2800_w$w $OUT/pmc_summary_tess2800_w$w.md > /dev/null
  python3 bench.py --sphere-tess 2800 --steps 32 --warmup 32 --no-cpu-baseline > $OUT/bench_tess2800_w$w.json 2> /dev/null
  echo "tess 2800 wide8=$w done"
done
unset TWK_WIDE8
for f in $OUT/traffic_*.json; do echo $f; cat $f; echo; done
find $OUT -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT
