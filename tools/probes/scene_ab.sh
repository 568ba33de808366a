# usage (GPU box): bash tools/probes/scene_ab.sh <variant> ...   — tools/scene_perf.py per build/lib_<variant>.so ("base" = in-tree)
for v in "$@"; do
  if [ $v = base ]; then unset TWK_LIB; else export TWK_LIB=build/lib_$v.so; fi
  echo "== $v"; timeout -k 10 200 python tools/scene_perf.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l); print('  %-14s %7.1f Msamples/s trace %.4f' % (r['scene'], r['Msamples_per_s'], r['kernel_ms_per_step']['trace']))
    except Exception: pass"
done
