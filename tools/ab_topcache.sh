set -x
for c in 1 0; do TWK_TOP_CACHE=$c timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r02e_bench_cache$c.json 2>/dev/null; python - <<PY
import json
r=json.load(open("gpurun_out/r02e_bench_cache$c.json"))
rf=r["roofline"]
print("cache $c", r["value"], rf["kernel_ms_per_step"], rf["nodes_per_ray"], rf["lane_occupancy"]["node_step"], rf["frac"])
PY
done
