#!/bin/bash
# SQ counters of the traversal kernel for library variants side by side (build/lib_<name>.so, "base" = in-tree):
# usage (GPU box): bash tools/pmc_ab.sh <tag> <variant> ...   -> gpurun_out/<tag>/<variant>.md
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ $v = base ]; then unset TWK_LIB; else export TWK_LIB=build/lib_$v.so; fi
  OUT=gpurun_out/$TAG/$v; mkdir -p $OUT
  ARGS="--steps 64 --warmup 64 --no-cpu-baseline --no-roofline"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/sq1 -o sq1 -- python3 bench.py $ARGS > $OUT/sq1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- python3 bench.py $ARGS > $OUT/sq2.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT --kernel-trace --output-format csv -d $OUT/sq3 -o sq3 -- python3 bench.py $ARGS > $OUT/sq3.log 2>&1
  python3 tools/pmc_summarize.py $OUT gpurun_out/$TAG/$v.md > /dev/null
  grep -A 30 "traceKernel<false, false, false" gpurun_out/$TAG/$v.md | head -34
done
