python3 tools/scene_perf.py --steps 20 > gpurun_out/sp20.jsonl 2>/dev/null
python3 - <<'PY'
import json
for l in open("gpurun_out/sp20.jsonl"):
    l=l.strip()
    if l:
        d=json.loads(l); k=d["kernel_ms_per_step"]; print("%-14s %7.1f trace %.4f shade %.4f"%(d["scene"],d["Msamples_per_s"],k["trace"],k["shade"]))
PY
python3 bench.py --gpus 1 --rehearse-rccl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/rehearse_rccl_s20.json 2> gpurun_out/rehearse_rccl_s20.err
python3 -c "
import json; d=json.load(open('gpurun_out/rehearse_rccl_s20.json')); print(d['value'], d['ms_per_step'], {k:v for k,v in d['config'].items() if 'same' in k or 'exchange' in k or 'barrier' in k})"
