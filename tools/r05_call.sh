set -e
mkdir -p gpurun_out
python tools/shade_phase_profile.py 20 > gpurun_out/r05n_phases_exact.txt
TWK_LIB=build/lib_fast.so python tools/shade_phase_profile.py 20 > gpurun_out/r05n_phases_native.txt
paste -d'\n' gpurun_out/r05n_phases_exact.txt gpurun_out/r05n_phases_native.txt | cut -c1-120
