set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05z_pytest.log 2>&1 || { tail -40 gpurun_out/r05z_pytest.log; exit 1; }
tail -2 gpurun_out/r05z_pytest.log
bash tools/final_profiles.sh r05z
bash tools/big_scene_profiles.sh r05t
