# usage (GPU box): bash tools/probes/waves_ab.sh   — six against seven trace blocks per CU (TWK_TRACE_WAVES_RUNTIME) on every scene
for w in 6 7; do
  export TWK_TRACE_WAVES_RUNTIME=$w
  echo "== trace blocks per CU: $w (where the scene allows)"
  timeout -k 10 200 python tools/scene_perf.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l); print('  %-14s %7.1f Msamples/s trace %.4f' % (r['scene'], r['Msamples_per_s'], r['kernel_ms_per_step']['trace']))
    except Exception: pass"
  for s in 20 64; do timeout -k 10 200 python bench.py --steps $s --warmup 5 2>/dev/null | python -c "
import sys, json
r=json.loads(sys.stdin.readline()); print('  C2 steps %d: %.1f Msamples/s' % (r['steps'], r['value']))"; done
done
unset TWK_TRACE_WAVES_RUNTIME
