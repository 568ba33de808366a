set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05a_pytest.log 2>&1 || { tail -30 gpurun_out/r05a_pytest.log; exit 1; }
tail -3 gpurun_out/r05a_pytest.log
STEPS=20 WARMUP=5 bash tools/ab_run.sh base fma256 fma1024 base fma256 fma1024 | tee gpurun_out/r05a_fma_s20.txt
STEPS=64 WARMUP=4 bash tools/ab_run.sh base fma256 fma1024 base fma256 fma1024 | tee gpurun_out/r05a_fma_s64.txt
python tools/shade_phase_profile.py 20 > gpurun_out/r05a_shade_phases_c2_b20.txt
python tools/shade_phase_profile.py 64 > gpurun_out/r05a_shade_phases_c2_b64.txt
python tools/shade_phase_profile.py scenes/system_rtigo3_geometry.txt scenes/scene_rtigo3_geometry.txt 20 > gpurun_out/r05a_shade_phases_c4g_b20.txt || true
cat gpurun_out/r05a_shade_phases_c2_b20.txt
