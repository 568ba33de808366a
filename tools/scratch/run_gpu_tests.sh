timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r05s_gpu_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r05s_gpu_tests.log
exit $rc
