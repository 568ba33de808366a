set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pass_variants.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05h_pytest.log 2>&1 || { tail -40 gpurun_out/r05h_pytest.log; exit 1; }
tail -3 gpurun_out/r05h_pytest.log
STEPS=20 WARMUP=5 bash tools/ab_run.sh wt_nee base base:TWK_SHADE_SORT=0 wt_nee base base:TWK_SHADE_SORT=0 | tee gpurun_out/r05h_ab_s20.txt
STEPS=64 WARMUP=4 bash tools/ab_run.sh wt_nee base base:TWK_SHADE_SORT=0 wt_nee base base:TWK_SHADE_SORT=0 | tee gpurun_out/r05h_ab_s64.txt
