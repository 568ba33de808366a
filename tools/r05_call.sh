set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pass_variants.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05r_pytest.log 2>&1 || { tail -40 gpurun_out/r05r_pytest.log; exit 1; }
tail -2 gpurun_out/r05r_pytest.log
TWK_LIB=build/lib_g16.so python -m pytest tests/test_gpu_pass_variants.py -m gpu -x -q 2>&1 | tail -1
STEPS=20 WARMUP=5 bash tools/ab_run.sh base:TWK_SHADE_SORT=0 base g8 g16 g32 base:TWK_SHADE_SORT=0 base g8 g16 g32 | tee gpurun_out/r05r_granule_s20.txt
STEPS=64 WARMUP=4 bash tools/ab_run.sh base:TWK_SHADE_SORT=0 base g8 g16 g32 base:TWK_SHADE_SORT=0 base g8 g16 g32 | tee gpurun_out/r05r_granule_s64.txt
