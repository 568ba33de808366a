#!/usr/bin/env python3
"""HBM-side traffic of the traversal kernel from rocprofv3 --pmc passes (tools/pmc_collect.sh):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of 16-B-per-lane reads
(MI355X_MICROARCH.md §HBM), so it is doubled; WRITE_SIZE is exact for 16-B stores. Averages per dispatch of
twk::traceKernel<false, ...> over the launches of the TIMED pass(es) only (the last `steps` iterations: the warm-up pass
of a `--steps 20 --warmup 5` run launches a quarter of the rays and must not enter the average); steps / batch depth /
resolution of the profiled bench run are recorded so that bench.py reports the figure only for a matching run.
usage: tools/pmc_traffic.py gpurun_out/<tag> profiles/r03_trace_hbm_traffic_s<steps>.json [--steps N --warmup W --batch B --resolution W H]"""
import argparse, csv, glob, json, os

ap = argparse.ArgumentParser()
ap.add_argument("root"); ap.add_argument("out")
ap.add_argument("--steps", type=int, default=None); ap.add_argument("--warmup", type=int, default=None)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--resolution", type=int, nargs=2, default=[1920, 1080])
ap.add_argument("--kernel", default="traceKernel<false")
ap.add_argument("--sphere-tess", type=int, default=180)
a = ap.parse_args()
if a.steps is None:
    sw = os.path.join(a.root, "steps_warmup.txt")
    a.steps, a.warmup = (int(x) for x in open(sw).read().split()) if os.path.exists(sw) else (64, 64)
batch = min(a.batch, a.steps)
passes_timed = -(-a.steps // batch)
passes_warm = -(-a.warmup // min(a.batch, max(1, a.warmup))) if a.warmup > 0 else 0

WANTED = ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
          "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES")
per, used = {}, {}
for path in glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True):
    rows = {}  # counter -> {dispatch id: value}   (one file = one rocprofv3 pass = one process)
    for r in csv.DictReader(open(path)):
        if a.kernel not in r["Kernel_Name"] or r["Counter_Name"] not in WANTED:
            continue
        d = rows.setdefault(r["Counter_Name"], {})
        d[int(r["Dispatch_Id"])] = d.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for c, d in rows.items():
        ids = sorted(d)
        per_pass = len(ids) // max(1, passes_timed + passes_warm)
        timed = ids[len(ids) - per_pass * passes_timed:] if per_pass else ids
        per[c] = sum(d[i] for i in timed) / max(1, len(timed))
        used[c] = [len(timed), len(ids)]
g = lambda c: per.get(c, 0.0)
res = {
    "kernel": a.kernel + "*",
    "steps": a.steps, "warmup": a.warmup, "batch_depth": batch, "resolution": list(a.resolution), "sphere_tess": a.sphere_tess,
    "dispatches_averaged_of_all": used.get("FETCH_SIZE"),
    "fetch_size_kib_per_launch": g("FETCH_SIZE"),
    "write_size_kib_per_launch": g("WRITE_SIZE"),
    "fetch_correction": 2.0,
    "hbm_bytes_per_launch": (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0,
    "l2_hit_rate": g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum")),
    "l2_hits_per_launch": g("TCC_HIT_sum"), "l2_misses_per_launch": g("TCC_MISS_sum"),
    # per wave: share of its cycles parked on memory / waiting for an issue slot / issuing
    "wave_cycles_waiting_on_memory": g("SQ_WAIT_ANY") / max(1.0, g("SQ_WAVE_CYCLES")),
    "wave_cycles_waiting_for_issue": g("SQ_WAIT_INST_ANY") / max(1.0, g("SQ_WAVE_CYCLES")),
    "wave_cycles_issuing": g("SQ_ACTIVE_INST_ANY") / max(1.0, g("SQ_WAVE_CYCLES")),
    # SIMD-level: GRBM_GUI_ACTIVE is summed over the 8 XCDs, the chip has 1024 SIMDs; SQ_ACTIVE_INST_VALU counts 4-clock units.
    # valu_issue_utilisation = share of the SIMD clocks of the dispatch taken by vector instructions at 4 clocks each (what
    # every vector instruction but fma / mul / add costs on this chip, tools/probes/valu_issue_probe.hip; capped at 1)
    "simd_clocks_per_vector_instruction": (g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0) / max(1.0, g("SQ_INSTS_VALU")),
    "valu_issue_utilisation": min(1.0, 4.0 * g("SQ_ACTIVE_INST_VALU") / max(1.0, g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0)),
    # the same without the cap. An ESTIMATE: it prices every SQ_ACTIVE_INST_VALU unit at 4 SIMD clocks; on this chip the plain
    # fma / mul / add issue in fewer (tools/probes/valu_issue_probe.hip: everything else costs 1.6 x an fma), so the 4-clock model
    # over-counts and the ratio passes 1 on kernels whose SIMDs issue back to back.
    "valu_issue_ratio_uncapped_4_clock_model": 4.0 * g("SQ_ACTIVE_INST_VALU") / max(1.0, g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0),
    "valu_lane_utilisation": g("SQ_THREAD_CYCLES_VALU") / max(1.0, 64.0 * g("SQ_ACTIVE_INST_VALU")),
    "lds_bank_conflict_cycles_per_lds_instruction": g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_INSTS_LDS")),
    "note": f"launches of {batch} iterations (the timed pass of `bench.py --steps {a.steps} --warmup {a.warmup}`); separate --pmc passes for FETCH_SIZE and WRITE_SIZE; counters include Infinity-Cache hits (memory-side of L2)",
}
json.dump(res, open(a.out, "w"), indent=1)
print(json.dumps(res))
