# Duration of the depth-0 shadeKernel launch of one 64-iteration pass per library variant (rocprofv3 kernel trace).
# usage (GPU box): bash tools/probes/shade_depth0_time.sh base <variant> ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/one_pass.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import tweeker_raytracer_amd as twk
R = os.environ["GRAFT_REPO_ROOT"]
app = twk.Application(os.path.join(R, "scenes", "system_rtigo3_cornell_box.txt"), os.path.join(R, "scenes", "scene_rtigo3_cornell_box.txt"))
dev = twk.Device(ordinal=0, miss=app.info.miss)
app.initDevice(dev)
for it in range(64): dev.render(it)
dev.synchronizeStream()
dev.close()
PY
for v in "$@"; do
  if [ $v = base ]; then unset TWK_LIB; else export TWK_LIB=build/lib_$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sd0_$v -o t -- python3 /tmp/one_pass.py > /dev/null 2>&1
  python3 - "$v" <<'PY'
import csv, sys, glob
v = sys.argv[1]
rows = [r for f in glob.glob(f"gpurun_out/sd0_{v}/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "shadeKernel" in r["Kernel_Name"]), reverse=True)
print(v, "shade launches (ms), largest first:", [round(x, 3) for x in d[:4]])
PY
done
