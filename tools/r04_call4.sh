#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04d; mkdir -p $OUT
./build/launch_gap_probe > $OUT/launch_gap_probe.jsonl 2>&1; cat $OUT/launch_gap_probe.jsonl
TWK_LIB=build/lib_presetup.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_bounds.py -x -q -m gpu -k "image_bit_identical or full_sample_count or launch_batching" > $OUT/pytest_presetup.log 2>&1; echo "presetup parity rc $?"; tail -2 $OUT/pytest_presetup.log
bash tools/ab_run.sh base presetup base presetup 2>&1 | grep AB
for spec in base presetup; do
  lib=""; [[ "$spec" != "base" ]] && export TWK_LIB=build/lib_$spec.so || unset TWK_LIB
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_s20_$spec.json 2>/dev/null
  python3 -c "
import json; r=json.load(open('$OUT/bench_s20_$spec.json')); k=r['roofline']['kernel_ms_per_step']; print('S20 $spec', round(r['value'],1), round(k['trace'],4), round(k['shade'],4), 'b1', round(r['config']['batch1_Msamples_per_s'],1))"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$spec -o stats -- python3 bench.py --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/stats_$spec.err
  f=$(find $OUT/stats_$spec -name "*kernel_stats.csv" | head -1); python3 tools/summarize_rocprof.py $f $OUT/kernel_stats_$spec.md > /dev/null; head -12 $OUT/kernel_stats_$spec.md
done
