python3 bench.py --native-math --steps 20 --warmup 5 > gpurun_out/r05z_bench_native_math_s20.json 2> gpurun_out/nm20.err
python3 bench.py --native-math > gpurun_out/r05z_bench_native_math_default.json 2> gpurun_out/nm64.err
python3 -c "
import json
for n in ['s20','default']:
    d=json.load(open('gpurun_out/r05z_bench_native_math_%s.json'%n)); print(n, d['value'], d['roofline']['kernel_ms_per_step'], d['config'].get('arithmetic','')[:60])"
