# Share of the traversal kernel's duration during which its waves are alive (SQ_WAVE_CYCLES x 4 / (waves x kernel clocks)):
# what an uneven finish of the persistent waves costs. usage (GPU box): bash tools/probes/pmc_wave_residency.sh [variant]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
v=${1:-base}
if [ $v != base ]; then export TWK_LIB=build/lib_$v.so; fi
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcw_$v -o p -- python3 bench.py --steps 64 --warmup 64 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python3 - "$v" <<'PY'
import csv, sys, glob, collections
v = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/pmcw_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "traceKernel<false" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.values(), key=lambda a: -a["GRBM_GUI_ACTIVE"])[:11]
for a in rows:
    clocks = a["GRBM_GUI_ACTIVE"] / 8.0
    print(v, "launch %.2f M clocks: waves alive %.3f of it" % (clocks / 1e6, a["SQ_WAVE_CYCLES"] * 4.0 / max(1.0, a["SQ_WAVES"] * clocks)))
PY
