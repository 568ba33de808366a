set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_native_math.py -m gpu -x -q -s > gpurun_out/r05i_pytest.log 2>&1 || { tail -40 gpurun_out/r05i_pytest.log; exit 1; }
grep -E "native-math|passed|failed" gpurun_out/r05i_pytest.log
for s in "20 5" "64 4"; do set -- $s
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline > gpurun_out/r05i_bench_exact_s$1.json
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --native-math > gpurun_out/r05i_bench_native_s$1.json
  python - $1 <<'PY'
import json, sys
for k in ("exact", "native"):
    r = json.load(open(f"gpurun_out/r05i_bench_{k}_s{sys.argv[1]}.json")); q = r["roofline"]["kernel_ms_per_step"]
    print(f"{k:7s} steps {sys.argv[1]}: {r['value']:.1f} Msamples/s trace {q['trace']:.4f} shade {q['shade']:.4f}")
PY
done
TWK_LIB=tweeker_raytracer_amd/libtweeker_hip_fast.so python tools/scene_perf.py > gpurun_out/r05i_scene_perf_native.jsonl 2>&1 || true
python tools/scene_perf.py > gpurun_out/r05i_scene_perf_exact.jsonl 2>&1 || true
tail -5 gpurun_out/r05i_scene_perf_native.jsonl; tail -5 gpurun_out/r05i_scene_perf_exact.jsonl
