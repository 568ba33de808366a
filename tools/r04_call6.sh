#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04f; mkdir -p $OUT
python3 tools/depth_profile.py 20 > $OUT/depth_profile_b20.jsonl 2>/dev/null; cat $OUT/depth_profile_b20.jsonl | head -40
for td in 0 5 6 7 8; do
  TWK_TAIL_DEPTH=$td python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_s20_tail$td.json 2>/dev/null
  python3 -c "
import json
r=json.load(open('$OUT/bench_s20_tail$td.json')); print('tail depth $td s20', round(r['value'],1))"
done
