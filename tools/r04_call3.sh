#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_time_view.py "tests/test_gpu_big_scenes.py" -x -q -m gpu -k "not 1000" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_s20.json 2> $OUT/bench_s20.err; echo "bench s20 rc $?"
python3 bench.py --no-cpu-baseline > $OUT/bench_s64.json 2> $OUT/bench_s64.err; echo "bench s64 rc $?"
python3 tools/big_scene_probe.py > $OUT/big_scene_probe.jsonl 2> /dev/null; cat $OUT/big_scene_probe.jsonl
python3 -c "
import json
for n in ('s20','s64'):
    r=json.load(open('$OUT/bench_'+n+'.json')); rf=r['roofline']
    print(n, round(r['value'],1), {k:round(v,4) for k,v in rf['kernel_ms_per_step'].items()}, 'b1', round(r['config'].get('batch1_Msamples_per_s',0),1))
"
