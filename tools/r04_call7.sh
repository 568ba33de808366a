#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04g; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_scenes.py tests/test_gpu_edge_cases.py tests/test_gpu_parity_bounds.py tests/test_gpu_pass_variants.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/pytest.log
python3 tools/scene_perf.py > $OUT/scene_perf.jsonl 2>/dev/null; cat $OUT/scene_perf.jsonl
