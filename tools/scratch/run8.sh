cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/scratch/c5_share.py 20
TWK_PASS_LANES=1 python3 tools/scratch/c5_share.py 20
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c5trace -o c5 -- python3 tools/scratch/c5_share.py 20 > /dev/null 2>&1
t=$(find gpurun_out/c5trace -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $t -1 3 > gpurun_out/c5_share_timeline_s20.txt
tail -50 gpurun_out/c5_share_timeline_s20.txt
rm -rf gpurun_out/c5trace
