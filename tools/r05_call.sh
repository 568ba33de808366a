set -e
mkdir -p gpurun_out
TWK_LANE_STAGGER=3 python -m pytest tests/test_gpu_pass_variants.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05j_pytest.log 2>&1 || { tail -40 gpurun_out/r05j_pytest.log; exit 1; }
tail -2 gpurun_out/r05j_pytest.log
STEPS=20 WARMUP=5 bash tools/ab_run.sh base base:TWK_PASS_LANES=2 base:TWK_LANE_STAGGER=2 base:TWK_LANE_STAGGER=3 base:TWK_LANE_STAGGER=5 base:TWK_LANE_STAGGER=3,TWK_LANE_TRACE_WAVES=6 base:TWK_LANE_STAGGER=5,TWK_LANE_TRACE_WAVES=6 base base:TWK_LANE_STAGGER=3 base:TWK_LANE_STAGGER=3,TWK_LANE_TRACE_WAVES=6 | tee gpurun_out/r05j_stagger_s20.txt
STEPS=64 WARMUP=4 bash tools/ab_run.sh base base:TWK_LANE_STAGGER=3 base:TWK_LANE_STAGGER=3,TWK_LANE_TRACE_WAVES=6 base:TWK_LANE_STAGGER=5,TWK_LANE_TRACE_WAVES=6 | tee gpurun_out/r05j_stagger_s64.txt
STEPS=10 WARMUP=5 bash tools/ab_run.sh base base:TWK_LANE_STAGGER=3 base:TWK_LANE_STAGGER=3,TWK_LANE_TRACE_WAVES=6 base:TWK_LANE_STAGGER=5,TWK_LANE_TRACE_WAVES=6 | tee gpurun_out/r05j_stagger_s10.txt
