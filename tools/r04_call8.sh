#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04i; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_edge_cases.py tests/test_gpu_pass_variants.py tests/test_gpu_time_view.py tests/test_gpu_parity_bounds.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
bash tools/ab_run2.sh base base 2>&1 | grep AB2
python3 tools/scene_perf.py > $OUT/scene_perf.jsonl 2>/dev/null; cut -c1-210 $OUT/scene_perf.jsonl
python3 tools/big_scene_probe.py 1000 2800 > $OUT/big_scene_probe.jsonl 2>/dev/null; cat $OUT/big_scene_probe.jsonl
