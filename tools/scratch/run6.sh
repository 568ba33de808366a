timeout -k 10 180 python3 -m pytest tests/test_gpu_pass_variants.py -x -q -m gpu -k shortcuts > gpurun_out/rg_variants.log 2>&1; rc=$?
tail -15 gpurun_out/rg_variants.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_run2.sh base base:TWK_TRACE_REGROUP=1
