# usage (GPU box): bash tools/probes/batch_ab.sh <variant> ...  — bench.py at launch batches 1, 4, 16, 64 per build/lib_<variant>.so ("base" = in-tree)
for v in "$@"; do
  if [ $v = base ]; then unset TWK_LIB; else export TWK_LIB=build/lib_$v.so; fi
  for b in 1 4 16 64; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --batch $b 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$v batch $b: %.1f Msamples/s' % r['value'])"
  done
done
