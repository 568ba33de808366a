set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_rehearsal.py -m gpu -x -q > gpurun_out/r05q_pytest.log 2>&1 || { tail -60 gpurun_out/r05q_pytest.log; exit 1; }
tail -3 gpurun_out/r05q_pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --rehearse-rccl > gpurun_out/r05q_bench_rehearse_rccl.json 2> gpurun_out/r05q_bench_rehearse_rccl.err
python -c "
import json; r=json.load(open('gpurun_out/r05q_bench_rehearse_rccl.json')); print(r['value'], r['config']['collective'], r['config']['gather_plus_compositor_ms'], r['config']['closing_barrier_ms'], r['config']['composite_bit_identical_to_single_device'], r.get('single_device_same_frame_Msamples_per_s'))"
