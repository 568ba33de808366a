timeout -k 10 500 python3 -m pytest tests/test_gpu_pass_variants.py -x -q -m gpu > gpurun_out/r05s_variants.log 2>&1 || { tail -30 gpurun_out/r05s_variants.log; exit 1; }
tail -3 gpurun_out/r05s_variants.log
F=tweeker_raytracer_amd/libtweeker_hip_fast.so
bash tools/ab_run2.sh base:TWK_SHADE_SORT=0 base:TWK_SHADE_SORT=1 base:TWK_LIB=$F,TWK_SHADE_SORT=0 base:TWK_LIB=$F,TWK_SHADE_SORT=1 base:TWK_SHADE_SORT=0 base:TWK_SHADE_SORT=1
