#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_pass_variants.py tests/test_gpu_time_view.py tests/test_gpu_parity_bounds.py tests/test_gpu_host_kernels.py tests/test_gpu_edge_cases.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
for pq in 1 0 1 0; do
  TWK_PACKED_QUEUE=$pq python3 bench.py --no-cpu-baseline > $OUT/bench_s64_pq$pq.json 2>/dev/null
  TWK_PACKED_QUEUE=$pq python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_s20_pq$pq.json 2>/dev/null
  python3 -c "
import json
for n in ('s20','s64'):
    r=json.load(open('$OUT/bench_'+n+'_pq$pq.json')); rf=r['roofline']
    print('packed $pq', n, round(r['value'],1), {k:round(v,4) for k,v in rf['kernel_ms_per_step'].items()}, 'b1', round(r['config'].get('batch1_Msamples_per_s',0),1))
"
done
TWK_PACKED_QUEUE=1 python3 tools/scene_perf.py > $OUT/scene_perf_pq1.jsonl 2>/dev/null; cat $OUT/scene_perf_pq1.jsonl
TWK_PACKED_QUEUE=0 python3 tools/scene_perf.py > $OUT/scene_perf_pq0.jsonl 2>/dev/null; cat $OUT/scene_perf_pq0.jsonl
