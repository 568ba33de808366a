cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in base ss1; do
  if [ $v = base ]; then unset TWK_LIB; else export TWK_LIB=build/lib_$v.so; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmcs_$v -o p -- python3 bench.py --steps 64 --warmup 64 --no-cpu-baseline --no-roofline > gpurun_out/pmcs_$v.log 2>&1
done
