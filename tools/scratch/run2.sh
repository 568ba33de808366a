cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TWK_SHADE_SORT=1 python3 tools/shade_phase_profile.py 20 > gpurun_out/r05t_shade_phases_sorted_b20.txt 2>&1
TWK_SHADE_SORT=0 python3 tools/shade_phase_profile.py 20 > gpurun_out/r05t_shade_phases_slot_order_b20.txt 2>&1
bash tools/pmc_collect.sh r05t_pmc_s20 20 5 > gpurun_out/r05t_pmc_s20.log 2>&1
python3 tools/pmc_summarize.py gpurun_out/r05t_pmc_s20 gpurun_out/r05t_pmc_summary_s20.md > /dev/null
python3 tools/pmc_traffic.py gpurun_out/r05t_pmc_s20 gpurun_out/r05t_shade_traffic_s20.json --kernel shadeKernel
rm -rf gpurun_out/r05t_pmc_s20/*/ 2>/dev/null
bash tools/ab_run2.sh base seg8 w6 w6t
