#!/bin/bash
# PMC collection on the GPU box: separate rocprofv3 passes (SQ has 8 slots, TCC 4; FETCH_SIZE takes 3, WRITE_SIZE 2),
# counters only combined with --kernel-trace. Output: gpurun_out/<tag>/<pass>/..._counter_collection.csv
# usage: tools/pmc_collect.sh <tag> [steps [warmup [further bench args]]]      (default 64 64: every launch the same size)
# The driver's invocation is `--steps 20 --warmup 5`: tools/pmc_traffic.py then averages over the launches of the TIMED pass only.
set -u
TAG=${1:-pmc}; shift || true
STEPS=${1:-64}; shift || true
WARMUP=${1:-64}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
ARGS="--steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-roofline $*"
# one lane for every pass: a warm-up pass of <= 21 M paths would run as two lanes (twice the dispatches of the timed pass), and
# tools/pmc_traffic.py finds the timed pass's dispatches by position; the timed passes of both launch sizes are one lane anyway
export TWK_PASS_LANES=1
echo "$STEPS $WARMUP" > $OUT/steps_warmup.txt
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_VALU_TRANS_F32
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# TA_* counters: two round-1 runs that added a TA pass were killed by gpurun's silence guard; see profiles/README.md
# ("The two killed TA_* counter runs") for what is and is not known about them. They are not added back.
find $OUT -name "*.csv" | head -20
