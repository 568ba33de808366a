#!/bin/bash
# One rocprofv3 --pmc pass (the SQ issue counters) of bench.py per "name[:ENV=VALUE,...]" argument, then the per-kernel sums.
# name = base | build/lib_<name>.so variant. usage (GPU box): [STEPS=20 WARMUP=5] bash tools/pmc_ab.sh <tag> base base:TWK_SHADE_SORT=0 ...
# Output: gpurun_out/<tag>/<spec>/..._counter_collection.csv and gpurun_out/<tag>/<spec>.md
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TWK_PASS_LANES=1
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  [[ "$spec" == *:* ]] && envs=${spec#*:}
  tag=${spec//[:=,]/_}
  OUT=gpurun_out/$TAG/$tag; mkdir -p $OUT
  ( [[ "$name" != "base" ]] && export TWK_LIB=build/lib_$name.so
    for kv in ${envs//,/ }; do export "$kv"; done
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT -o sq1 -- python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace --output-format csv -d $OUT -o sq2 -- python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-roofline > $OUT/bench2.log 2>&1 ) || echo "pass $spec failed"
  python3 tools/pmc_summarize.py $OUT gpurun_out/$TAG/$tag.md > /dev/null
  echo "== $spec"; grep -A 22 "### shadeKernel" gpurun_out/$TAG/$tag.md | grep -E "SQ_INSTS_VALU |SQ_ACTIVE_INST_VALU|SQ_THREAD_CYCLES_VALU|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_BUSY|GRBM|SQ_INSTS_LDS|SQ_INSTS_SALU"
done
