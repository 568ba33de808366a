set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pass_variants.py tests/test_gpu_parity.py tests/test_gpu_time_view.py -m gpu -x -q > gpurun_out/r05p_pytest.log 2>&1 || { tail -40 gpurun_out/r05p_pytest.log; exit 1; }
tail -2 gpurun_out/r05p_pytest.log
STEPS=20 WARMUP=5 bash tools/ab_run.sh base before base before base before | tee gpurun_out/r05p_early_exit_s20.txt
STEPS=64 WARMUP=4 bash tools/ab_run.sh base before base before | tee gpurun_out/r05p_early_exit_s64.txt
STEPS=4 WARMUP=4 bash tools/ab_run.sh base before base before | tee gpurun_out/r05p_early_exit_s4.txt
