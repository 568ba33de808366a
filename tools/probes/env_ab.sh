# usage (GPU box): bash tools/probes/env_ab.sh VAR valueA valueB [repeats] — tools/scene_perf.py with an environment knob of the library at two values
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do for v in $A $B; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 200 python tools/scene_perf.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l); print('  %-14s %7.1f Msamples/s trace %.4f' % (r['scene'], r['Msamples_per_s'], r['kernel_ms_per_step']['trace']))
    except Exception: pass"
done; done
