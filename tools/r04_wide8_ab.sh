#!/bin/bash
# round 4: the compressed 8-ary nodes — structure check first (no traversal launch), then parity, then the A/B numbers.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04b; mkdir -p $OUT
timeout -k 10 300 python3 -m pytest tests/test_gpu_scenes.py -x -q -m gpu -k "compressed_8ary" > $OUT/pytest_structure.log 2>&1 || { echo "structure test failed"; tail -30 $OUT/pytest_structure.log; exit 1; }
echo "structure ok"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pass_variants.py -x -q -m gpu -k "host_walker or shortcuts_change_no_bit" > $OUT/pytest_parity.log 2>&1 || { echo "parity failed"; tail -40 $OUT/pytest_parity.log; exit 1; }
echo "walker + knob parity ok"
TWK_WIDE8=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_scenes.py tests/test_gpu_parity_bounds.py -q -m gpu > $OUT/pytest_wide8_all.log 2>&1; echo "suite under TWK_WIDE8=1 rc $?"; tail -15 $OUT/pytest_wide8_all.log
for w in 0 1; do
  TWK_WIDE8=$w python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_s20_w$w.json 2> $OUT/bench_s20_w$w.err; echo "bench s20 wide8=$w rc $?"
  TWK_WIDE8=$w python3 bench.py --no-cpu-baseline > $OUT/bench_s64_w$w.json 2> $OUT/bench_s64_w$w.err; echo "bench s64 wide8=$w rc $?"
  TWK_WIDE8=$w python3 tools/big_scene_probe.py > $OUT/big_scene_probe_w$w.jsonl 2> /dev/null; echo "probe wide8=$w rc $?"
  cat $OUT/big_scene_probe_w$w.jsonl
done
python3 -c "
import json
for w in (0,1):
  for n in ('s20','s64'):
    r=json.load(open('$OUT/bench_%s_w%d.json'%(n,w))); rf=r['roofline']
    print(w, n, round(r['value'],1), {k:round(v,4) for k,v in rf['kernel_ms_per_step'].items()}, 'b1', round(r['config'].get('batch1_Msamples_per_s',0),1), 'nodes/ray', round(rf['nodes_per_ray'],2), 'cached', round(rf['nodes_per_ray_from_lds_cache'],2), 'tris/ray', round(rf['triangles_per_ray'],2), 'occ', {k:round(v,3) for k,v in list(rf['lane_occupancy'].items())[:2]}, rf['wave_time_shares'])
"
