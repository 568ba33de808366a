#!/usr/bin/env python3
"""HBM-side traffic of the traversal kernel from rocprofv3 --pmc passes (tools/pmc_collect.sh):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of 16-B-per-lane reads
(MI355X_MICROARCH.md §HBM), so it is doubled; WRITE_SIZE is exact for 16-B stores. Averages per dispatch of
twk::traceKernel<false, ...>; steps / batch depth / resolution of the profiled bench run are recorded so that bench.py
reports the figure only for a matching run. usage: tools/pmc_traffic.py gpurun_out/<tag> profiles/r02_trace_hbm_traffic.json"""
import csv, glob, json, os, sys

root, out = sys.argv[1], sys.argv[2]
acc = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "traceKernel<false" not in r["Kernel_Name"]:
            continue
        c = r["Counter_Name"]
        if c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                 "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU"):
            a = acc.setdefault(c, [0.0, set()])
            a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
per = {c: v[0] / max(1, len(v[1])) for c, v in acc.items()}
res = {
    "kernel": "twk::traceKernel<false, false, *>",
    "steps": int(os.environ.get("TWK_PMC_STEPS", "64")), "batch_depth": int(os.environ.get("TWK_PMC_STEPS", "64")), "resolution": [1920, 1080],
    "dispatches": len(acc["FETCH_SIZE"][1]),
    "fetch_size_kib_per_launch": per["FETCH_SIZE"],
    "write_size_kib_per_launch": per["WRITE_SIZE"],
    "fetch_correction": 2.0,
    "hbm_bytes_per_launch": (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0,
    "l2_hit_rate": per.get("TCC_HIT_sum", 0) / max(1.0, per.get("TCC_HIT_sum", 0) + per.get("TCC_MISS_sum", 0)),
    # per wave: share of its cycles parked on memory / waiting for an issue slot / issuing
    "wave_cycles_waiting_on_memory": per.get("SQ_WAIT_ANY", 0) / max(1.0, per.get("SQ_WAVE_CYCLES", 0)),
    "wave_cycles_waiting_for_issue": per.get("SQ_WAIT_INST_ANY", 0) / max(1.0, per.get("SQ_WAVE_CYCLES", 0)),
    "wave_cycles_issuing": per.get("SQ_ACTIVE_INST_ANY", 0) / max(1.0, per.get("SQ_WAVE_CYCLES", 0)),
    # SIMD-level: GRBM_GUI_ACTIVE is summed over the 8 XCDs, the chip has 1024 SIMDs; SQ_ACTIVE_INST_VALU counts 4-clock units.
    # valu_issue_utilisation = share of the SIMD clocks of the dispatch taken by vector instructions at 4 clocks each (what
    # every vector instruction but fma / mul / add costs on this chip, tools/probes/valu_issue_probe.hip; capped at 1)
    "simd_clocks_per_vector_instruction": (per.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0) / max(1.0, per.get("SQ_INSTS_VALU", 0)),
    "valu_issue_utilisation": min(1.0, 4.0 * per.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, per.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0)),
    "valu_lane_utilisation": per.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64.0 * per.get("SQ_ACTIVE_INST_VALU", 0)),
    "note": "launches of 64 iterations (default batch); separate --pmc passes for FETCH_SIZE and WRITE_SIZE; counters include Infinity-Cache hits (memory-side of L2)",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
