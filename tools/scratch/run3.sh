python3 tools/scene_perf.py > gpurun_out/sp_base.jsonl 2>/dev/null
TWK_LIB=build/lib_st14.so python3 tools/scene_perf.py > gpurun_out/sp_st14.jsonl 2>/dev/null
TWK_SHADE_SORT=0 python3 tools/scene_perf.py > gpurun_out/sp_nosort.jsonl 2>/dev/null
python3 - <<'PY'
import json
for n in ["base","st14","nosort"]:
    for l in open("gpurun_out/sp_%s.jsonl"%n):
        l=l.strip()
        if l:
            d=json.loads(l); k=d["kernel_ms_per_step"]; print("%-7s %-14s %7.1f trace %.4f shade %.4f"%(n,d["scene"],d["Msamples_per_s"],k["trace"],k["shade"]))
PY
