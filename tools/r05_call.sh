set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05final_pytest.log 2>&1 || { tail -40 gpurun_out/r05final_pytest.log; exit 1; }
tail -2 gpurun_out/r05final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05final_bench_s20.json
python -c "
import json; r=json.load(open('gpurun_out/r05final_bench_s20.json')); print(r['value'], r['roofline']['bound'], round(r['roofline']['frac'],3), r['roofline']['kernel_ms_per_step'], r['cpu_baseline']['sample_bit_identical_to_gpu'])"
