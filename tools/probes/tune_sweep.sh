for v in base c256 c384 c768 r44 r58 n38 n58 t3 base; do
  if [ $v = base ]; then unset TWK_LIB; else export TWK_LIB=build/lib_$v.so; fi
  a=$(timeout -k 10 100 python bench.py --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('%.1f trace %.4f' % (r['value'], r['roofline']['kernel_ms_per_step']['trace']))")
  b=$(timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('%.1f trace %.4f' % (r['value'], r['roofline']['kernel_ms_per_step']['trace']))")
  echo "$v: s64 $a | s20 $b"
done
