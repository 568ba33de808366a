#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04a; mkdir -p $OUT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_big_scenes.py tests/test_gpu_pass_variants.py tests/test_gpu_time_view.py tests/test_gpu_edge_cases.py -q -m gpu -s > $OUT/pytest.log 2>&1; echo "pytest rc $?"
grep -n "overflow rays\|passed\|failed\|Error" $OUT/pytest.log | head -30
