set -e
mkdir -p gpurun_out
for v in 1 0; do
  export TWK_TREELET_LAYOUT=$v
  bash tools/pmc_collect.sh r05l/pmc_tess2800_layout$v 32 32 --sphere-tess 2800 > gpurun_out/r05l_pmc_layout$v.log 2>&1
  python3 tools/pmc_traffic.py gpurun_out/r05l/pmc_tess2800_layout$v gpurun_out/r05l_traffic_tess2800_layout$v.json --sphere-tess 2800 > /dev/null
  python3 tools/pmc_summarize.py gpurun_out/r05l/pmc_tess2800_layout$v gpurun_out/r05l_counters_tess2800_layout$v.md > /dev/null
  python3 -c "
import json; r=json.load(open('gpurun_out/r05l_traffic_tess2800_layout$v.json'))
print('layout $v', {k: r[k] for k in ('hbm_bytes_per_launch','l2_hit_rate','l2_hits_per_launch','l2_misses_per_launch','wave_cycles_waiting_on_memory','valu_issue_ratio_uncapped_4_clock_model','valu_lane_utilisation')})"
done
