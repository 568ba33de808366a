#!/usr/bin/env python3
"""bench.py — Msamples/s of the HIP wavefront path tracer on the Cornell box (BASELINE.json metric).

A step is one launch of the hot path (≙ one optixLaunch, reference DeviceSingleGPU.cpp:164) = one sample per pixel
over the whole frame; K steps = K progressive iterations (reference Application::benchmark, Application.cpp:491-513:
the timer brackets the launch loop and the final device sync; scene/BVH build and image write-out are outside).
Default K = 64 = the 64 spp of config C2.

N = 1: config C2, Cornell box 1920x1080, full BSDF set (scenes/system_rtigo3_cornell_box.txt).
N > 1 (torch.distributed.run, one rank per GPU): config C5 — the FIXED 3840x2160 Cornell frame
(scenes/system_rtigo3_cornell_box_c5.txt) tile-interleaved over the ranks exactly like the reference's distribute()
(raygeneration.cu:152-164; launchWidth = roundup(ceil(W / N), 8), DeviceMultiGPULocalCopy.cpp:84-97): STRONG scaling.
Every rank accumulates its launchWidth x H share locally for all K steps with no communication; then ONE gather to
rank 0 over RCCL and one compositor kernel assemble the image, inside the timed region. After the timed region rank
0 renders the same iterations of the whole frame on its own GPU and the composed image must equal it bit for bit
(`composite_bit_identical_to_single_device`, CRC-32 of both images in the line). `--weak` keeps round 1's mode
(16:9 frame grown to N x 2,073,600 pixels).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_SPEC_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; the denominator used below is the copy rate MEASURED on the box

# Algorithmic bytes of the traversal kernel, SURVEY.md §8(d)'s record table (cache hits do not reduce them):
B_RAY_FIXED = 48      # 32 B ray record in + 16 B hit record out (shadow rays: 32 B ray + 16 B pending contribution)
B_NODE = 64           # one 4-ary wide-node visit: two levels of the binary tree in ONE 64-byte quantised record (device_types.h), the size of §8(d)'s binary node
B_TRIANGLE = 48       # one Woop triangle slot (three float4)
B_INSTANCE = 64       # instance entry (two-level scenes only): world-to-object rows + BVH root, what an IAS leaf hands an OptiX traversal
# 16-byte lane loads the kernel issues per unit (what the divergent-gather ceiling prices)
L_RAY, L_NODE, L_TRIANGLE, L_INSTANCE = 2, 4, 3, 4


def weak_frame_for(n_gpus, base=(1920, 1080)):
    if n_gpus == 1:
        return base
    s = math.sqrt(n_gpus)
    return (int(round(base[0] * s / 8.0)) * 8, int(round(base[1] * s / 8.0)) * 8)


def crc(img):
    return "%08x" % (zlib.crc32(img.tobytes()) & 0xFFFFFFFF)


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` without an external launcher: one child `python -m torch.distributed.run` with one rank
    per GPU (rendezvous on 127.0.0.1, a free port), whose output is relayed line by line; returns its exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def pick_pmc_record(steps, batch_depth, resolution, sphere_tess=180, kernel="trace"):
    """HBM-side bytes come from rocprofv3 --pmc passes (tools/pmc_collect.sh + tools/pmc_traffic.py), which cannot run
    inside this process. A committed record is carried ONLY when it was taken at this run's launch size: same steps,
    same batch depth, same resolution. Newest record (by name) wins. Returns (record, file name) or (None, reason)."""
    import glob
    seen = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{kernel}_hbm_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except Exception:
            continue
        if (rec.get("steps") == steps and rec.get("batch_depth") == batch_depth and rec.get("resolution") == list(resolution)
                and rec.get("sphere_tess", 180) == sphere_tess):
            return rec, os.path.basename(path)
        seen.append(f"{os.path.basename(path)}: steps {rec.get('steps')} batch {rec.get('batch_depth')} {rec.get('resolution')} tess {rec.get('sphere_tess', 180)}")
    return None, "no PMC record at this launch size (have: " + "; ".join(seen) + ")" if seen else "no PMC record under profiles/"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--system", default=None, help="system description (default: C2 at N = 1, C5 at N > 1)")
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
    ap.add_argument("--batch", type=int, default=64, help="iterations rendered together per wavefront pass (twk_set_launch_batch, 1..64)")
    ap.add_argument("--weak", action="store_true", help="N > 1: round 1's weak-scaling frame (N x 2,073,600 pixels) instead of the fixed C5 frame")
    ap.add_argument("--c5", action="store_true", help="N = 1: render the C5 frame (3840x2160) instead of C2 — the one-GPU point of the strong-scaling curve")
    ap.add_argument("--sphere-tess", type=int, default=180,
                    help="N = 1: tessellation of the two spheres of the Cornell scene (`model sphere U U/2`): 180 is the scene file as it stands (64 k triangles, "
                         "cache-resident); 1000 gives 2.0 M and 2800 gives 15.7 M triangles — a scene the caches do not hold, where the HBM roofline applies")
    ap.add_argument("--native-math", action="store_true",
                    help="the opt-in approximate build (libtweeker_hip_fast.so: native sin / cos / exp / rcp / sqrt in the shading kernels, the reference's --use_fast_math mode); not bit-identical to the oracle, tests/test_gpu_native_math.py bounds it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-composite-check", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the oracle sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (a 1-GPU box has a share of 16 cores)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="with --gpus 1: go through the N > 1 path — process group on RCCL, tile distribution, gather, compositor, MAX all-reduce, composite check — with a world of ONE rank (the only way to execute the RCCL branch on a one-GPU box)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N > 1 path on ONE GPU: all ranks share device 0, the gather goes over gloo through host memory")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks from here — a child process, started BEFORE anything in this
        # process touches the GPU (no torch / HIP import yet), never a re-exec — relay rank 0's JSON line, pass on the exit code
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != n_gpus:
        n_gpus = world

    if args.native_math:  # before the package is imported: it binds the library it finds in TWK_LIB
        os.environ["TWK_LIB"] = os.path.join(ROOT, "tweeker_raytracer_amd", "libtweeker_hip_fast.so")
    import numpy as np
    import torch
    import tweeker_raytracer_amd as twk

    dist = None
    multi = n_gpus > 1 or args.rehearse_rccl  # the exchange path (a world of one rank with --rehearse-rccl)
    if args.rehearse_rccl and "WORLD_SIZE" not in os.environ:
        os.environ.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": os.environ.get("MASTER_PORT", "29531")})
    if args.rehearse_gloo:
        local_rank = 0
    # ONE JSON line on stdout: RCCL prints a version banner to file descriptor 1 when its communicator comes up (found by the
    # one-rank rehearsal, round 5). In the N > 1 path everything this process or its libraries print goes to stderr; the
    # result line is written to the saved descriptor at the end.
    result_fd = None
    if multi:
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
    if multi:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    device = torch.device("cuda", local_rank)
    c5 = (multi and not args.weak) or (n_gpus == 1 and args.c5)
    system = args.system or os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box_c5.txt" if c5 else "system_rtigo3_cornell_box.txt")
    if args.sphere_tess != 180:
        scene_text = open(args.scene).read().replace("sphere 180 90", f"sphere {args.sphere_tess} {args.sphere_tess // 2}")
        assert "sphere 180 90" in open(args.scene).read(), "--sphere-tess expects the Cornell scene file"
        app = twk.Application(system_text=open(system).read(), scene_text=scene_text)
    else:
        app = twk.Application(system, args.scene)
    if n_gpus > 1 and args.weak:
        app.setResolution(*weak_frame_for(n_gpus))
    info = app.info
    width, height = info.resolution[0], info.resolution[1]

    dev = twk.Device(ordinal=local_rank, index=rank, count=n_gpus, miss=info.miss)
    app.initDevice(dev, distribution=1 if multi else 0)
    batch = max(1, min(64, args.batch))
    dev.setLaunchBatch(batch)
    # path streams for the largest pass the timed loop will issue: allocated here, not inside the timed region
    dev.reserveLaunchBatch(min(batch, max(1, args.steps)))
    lw = dev.launchWidth

    # The accumulation buffer is a torch tensor so RCCL can send it without a copy.
    accum = torch.zeros((height, lw, 4), dtype=torch.float32, device=device)
    dev.setOutputDevicePointer(accum.data_ptr(), accum.numel() * 4)
    gathered = composed = None
    if multi and rank == 0:
        gathered = torch.empty((n_gpus, height, lw, 4), dtype=torch.float32, device=device)
        composed = torch.zeros((height, width, 4), dtype=torch.float32, device=device)

    def barrier():
        dev.synchronizeStream()
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()

    exchange = {"ms": 0.0, "render_ms": 0.0, "mode": "gather", "calls": 0, "all": None, "note": None}

    def run_steps(first, count, finish=True):
        tr = time.perf_counter()
        for it in range(first, first + count):
            dev.render(it)
        dev.synchronizeStream()
        exchange["render_ms"] = (time.perf_counter() - tr) * 1.0e3  # this rank's rendering alone (launches + device sync)
        if dist is not None and finish:
            # the one exchange step of the path: gather the packed tile buffers, scatter them into the image
            tx = time.perf_counter()
            if args.rehearse_gloo:
                host = accum.cpu()
                parts = [torch.empty_like(host) for _ in range(n_gpus)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    gathered.copy_(torch.stack(parts))
            elif exchange["mode"] == "gather":
                try:
                    dist.gather(accum, list(gathered.unbind(0)) if rank == 0 else None, dst=0)
                except (RuntimeError, NotImplementedError) as e:  # a backend build without gather: every rank falls back alike
                    if exchange["calls"] > 0:
                        raise
                    exchange["mode"] = "all_gather"
                    exchange["note"] = f"dist.gather unavailable ({type(e).__name__}), all_gather_into_tensor instead"
            if exchange["mode"] == "all_gather" and not args.rehearse_gloo:
                if exchange["all"] is None:
                    exchange["all"] = torch.empty((n_gpus, height, lw, 4), dtype=torch.float32, device=device)
                dist.all_gather_into_tensor(exchange["all"], accum)
                if rank == 0:
                    gathered.copy_(exchange["all"])
            exchange["calls"] += 1
            if rank == 0:
                torch.cuda.synchronize(device)
                dev.compositor(gathered.data_ptr(), composed.data_ptr())
                dev.synchronizeStream()
            exchange["ms"] = (time.perf_counter() - tx) * 1.0e3

    run_steps(0, args.warmup)
    if dist is not None:
        # the warm-up pass went through the same gather: the communicator exists before the timed region starts
        assert exchange["calls"] >= 1, "warm-up must exercise the exchange step (communicator set-up is not rendering)"
    barrier()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    # The clock of a rank stops when ITS work is done and synchronised (rank 0: the gather and the compositor launch included);
    # the job's time is the MAX over ranks, taken after the closing barrier. The barrier itself brackets the region as the
    # contract asks but is not rendering: with N ranks its own latency (reported as closing_barrier_ms) would otherwise sit
    # serially behind rank 0's synchronised compositor in a ~9 ms region ("weak" 7).
    dev.synchronizeStream()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    tb = time.perf_counter()
    barrier()
    closing_barrier_ms = (time.perf_counter() - tb) * 1.0e3
    render_ms_max = exchange["render_ms"]
    if dist is not None:
        tmax = torch.tensor([elapsed, exchange["render_ms"], closing_barrier_ms * 1.0e-3], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, render_ms_max, closing_barrier_ms = float(tmax[0].item()), float(tmax[1].item()), float(tmax[2].item()) * 1.0e3

    # ---- N > 1: the composed image against a single-device render of the same iterations (outside the timed region)
    composite = None
    if dist is not None and rank == 0 and not args.no_composite_check and os.environ.get("TWK_BENCH_CHECK_COMPOSITE", "1") == "1":
        single = twk.Device(ordinal=local_rank, miss=info.miss)
        app.initDevice(single)
        single_batch = max(1, batch // n_gpus)           # the frame is n_gpus times a rank's share: same stream memory as a rank
        single.setLaunchBatch(single_batch)
        single.reserveLaunchBatch(min(single_batch, max(1, args.steps)))
        for it in range(0, args.warmup):                # the accumulator keeps the warm-up iterations, like the ranks' buffers
            single.render(it)
        single.synchronizeStream()
        # the SAME frame and steps on ONE device, timed like the N = 1 line: the denominator of this line's scaling factor
        ts = time.perf_counter()
        for it in range(args.warmup, args.warmup + args.steps):
            single.render(it)
        single.synchronizeStream()
        single_s = time.perf_counter() - ts
        ref_img = single.getOutputBufferHost()
        single.close()
        comp_img = composed.cpu().numpy()
        composite = {"ok": bool(np.array_equal(ref_img.view(np.uint32), comp_img.view(np.uint32))),
                     "crc_composed": crc(comp_img), "crc_single": crc(ref_img)}

    samples = float(width) * float(height) * float(args.steps)
    if c5:
        workload = "C5: Cornell box (scene_rtigo3_cornell_box.txt, full BSDF set, 64,096 triangles, 8 instances), fixed frame tiled over the ranks"
    elif n_gpus > 1:
        workload = "C2 grown for weak scaling: Cornell box, 16:9 frame of N x 2,073,600 pixels"
    else:
        workload = "C2: Cornell box (scene_rtigo3_cornell_box.txt: Lambert + GGX wall + mirror + glass spheres, 64,096 triangles, 8 instances)"
        if args.sphere_tess != 180:
            workload = f"C2 room with the two spheres tessellated {args.sphere_tess} x {args.sphere_tess // 2} (NOT the baseline configuration: a scene the caches do not hold, for the HBM roofline)"
    result = {
        "metric": f"Msamples/s (paths x spp x res / s), Cornell box {width}x{height}",
        "value": samples / elapsed / 1.0e6,
        "unit": "Msamples/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1.0e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak" if (n_gpus > 1 and args.weak) else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload + f", {width}x{height}, {args.steps} spp progressive (1 spp per step), pathLengths {info.pathLengths[0]} {info.pathLengths[1]}, light 1, miss 0",
            "resolution": [width, height],
            "pixels_per_gpu_per_step": lw * height,
            "launch_width": lw,
            "batch_depth": min(batch, max(1, args.steps)),
            "batch_note": "twk_launch is deferred: up to batch_depth consecutive iterations are rendered as ONE wavefront pass (bit-identical image); batch1_Msamples_per_s is the rate at one pass per iteration",
            "arithmetic": ("approximate (--native-math: libtweeker_hip_fast.so, the reference's --use_fast_math mode; NOT the bit-identical build, bounded by tests/test_gpu_native_math.py)"
                           if args.native_math else "exact (bit-identical to the CPU oracle: correctly rounded division / square root, fixed-algorithm sin / cos / exp / atan)"),
            "parallelism": "single GPU" if n_gpus == 1 else f"tile-interleaved pixels over {n_gpus} ranks (8x8 tiles, distribute()), local accumulation, one RCCL gather + one compositor launch at the end",
        },
    }
    if dist is not None:
        result["config"]["gather_bytes_per_rank"] = lw * height * 16
        result["config"]["gather_bytes_into_root"] = lw * height * 16 * (n_gpus - 1)
        result["config"]["gather_plus_compositor_ms"] = exchange["ms"]  # rank 0, inside the timed region
        result["config"]["closing_barrier_ms"] = closing_barrier_ms  # after every rank's clock has stopped: not in `value`
        # where the timed region went: the slowest rank's rendering, the one exchange step on rank 0, and the rate the
        # rendering alone would give (value stays the honest whole: rendering + exchange + barriers)
        result["render_ms_max_rank"] = render_ms_max
        result["exchange_ms"] = exchange["ms"]
        result["value_render_only"] = samples / max(render_ms_max * 1.0e-3, 1e-12) / 1.0e6
        result["config"]["collective"] = exchange["note"] or "one gather to rank 0"
    if composite is not None:
        # measured on rank 0's GPU right after the timed region (the other ranks idle): the whole frame, same steps, one device
        result["single_device_same_frame_Msamples_per_s"] = samples / max(single_s, 1e-12) / 1.0e6
        result["strong_scaling_vs_same_frame"] = (samples / elapsed) / (samples / max(single_s, 1e-12))
        result["single_device_same_frame_note"] = (f"rank 0 alone renders the {width}x{height} frame, {args.steps} steps after {args.warmup} warm-up steps, passes of {single_batch} iterations "
                                                   "(a pass of the whole frame is n_gpus times a rank's); " + ("NOT a scaling measurement: all ranks share GPU 0" if args.rehearse_gloo else "measured on this node, after the timed region"))
        result["config"]["composite_bit_identical_to_single_device"] = composite["ok"]
        result["config"]["crc32_composed"] = composite["crc_composed"]
        result["config"]["crc32_single_device"] = composite["crc_single"]
    if args.rehearse_gloo:
        result["config"]["rehearsal"] = "gloo, all ranks on GPU 0 — NOT a scaling measurement"
    if args.rehearse_rccl:
        result["config"]["rehearsal"] = "RCCL with a world of one rank: the N > 1 path's calls on the real backend — NOT a scaling measurement"

    # ---- one pass per iteration (what a caller that synchronises after every launch gets, DeviceSingleGPU.cpp:147)
    if rank == 0 and n_gpus == 1 and not multi and not args.no_roofline:
        dev.setLaunchBatch(1)
        k1 = max(2, min(args.steps, 16))
        for it in range(2):
            dev.render(it)
        dev.synchronizeStream()
        tb = time.perf_counter()
        for it in range(args.warmup, args.warmup + k1):
            dev.render(it)
        dev.synchronizeStream()
        result["config"]["batch1_Msamples_per_s"] = width * height * k1 / (time.perf_counter() - tb) / 1.0e6
        dev.setLaunchBatch(batch)

    # ---- roofline of the dominant kernel (traversal), measured on the same steps --------------------
    if not args.no_roofline and rank == 0:
        dev.profileEnable(True)
        dev.profileReset()
        run_steps(args.warmup, args.steps, finish=False)
        prof = dev.profileGet()
        dev.profileEnable(False)
        dev.statsEnable(True)
        dev.statsGet(reset=True)
        run_steps(args.warmup, args.steps, finish=False)
        st = dev.statsGet(reset=True)
        dev.statsEnable(False)
        stream_peak = dev.streamPeakGBps(1 << 30, 10)       # float4 copy, read + write bytes, measured on this box
        gather_peak = dev.gatherPeak(32 << 20)              # G lane-loads/s (16 B each) from a scene-sized, cache-resident table
        rays = st["radianceRays"] + st["shadowRays"]
        bi = dev.buildInfo()
        has_cutout = any(bool(m.useCutoutTexture) for m in app.materials)  # the CUTOUT builds of the traversal kernel run the any-hit candidate loop
        b_node, l_node = B_NODE, L_NODE
        algo_bytes = B_RAY_FIXED * rays + b_node * st["nodesVisited"] + B_TRIANGLE * st["trianglesTested"] + B_INSTANCE * st["instancesEntered"]
        memory_nodes = st["nodesVisited"] - st["cachedNodesVisited"]  # the top of the tree is served from LDS (TWK_NODE_CACHED)
        lane_loads = L_RAY * rays + l_node * memory_nodes + L_TRIANGLE * st["trianglesTested"] + L_INSTANCE * st["instancesEntered"]
        trace_ms, trace_launches = prof["trace"]["ms"], max(1, prof["trace"]["launches"])
        trace_s = max(trace_ms * 1.0e-3, 1e-12)
        algo_gbps = algo_bytes / trace_s / 1.0e9
        gather_gbps = lane_loads * 16 / trace_s / 1.0e9
        avg_launch_s = trace_s / trace_launches
        pmc, pmc_source = (None, "N > 1") if n_gpus != 1 else pick_pmc_record(args.steps, result["config"]["batch_depth"], (width, height), args.sphere_tess)
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        hbm_side_gbps = (traffic / avg_launch_s / 1.0e9) if traffic else None
        scene_bytes = int(bi["nodes"]) * 64 + int(bi["triangleSlots"]) * 48  # what the traversal kernel reads of the scene: its nodes + triangle slots
        cache_resident = scene_bytes < (256 << 20)                           # MI355X_MICROARCH: Infinity Cache 256 MiB (L2 32 MiB aggregate)
        algo_frac_spec = algo_gbps / HBM_SPEC_GBPS
        # What binds the kernel, decided from what was MEASURED where a PMC record of this launch size exists: the
        # HBM-side rate against the stream peak of this run, and the issue ratio of the record (an estimate: 4 SIMD clocks per
        # SQ_ACTIVE_INST_VALU unit; plain fma / mul / add issue in fewer, so it passes 1 on a kernel that issues back to back).
        issue_ratio = (pmc.get("valu_issue_ratio_uncapped_4_clock_model") or pmc.get("valu_issue_utilisation")) if pmc else None
        hbm_frac_measured = (hbm_side_gbps / stream_peak) if hbm_side_gbps else None
        if pmc and hbm_frac_measured is not None and issue_ratio is not None:
            # vector issue where the SIMDs issue back to back; else HBM where it carries at least half of what a stream copy moves (the
            # gathers of a scene the caches do not hold: bandwidth AND latency of HBM); else the latency of cached gathers
            bound = "valu-issue" if issue_ratio >= 0.85 else ("hbm" if hbm_frac_measured >= 0.5 else "memory-latency")
            bound_source = f"measured: HBM-side {hbm_frac_measured:.2f} of the stream peak, vector issue ratio {issue_ratio:.2f} ({pmc_source})"
        else:
            bound = "valu-issue" if cache_resident else "hbm"
            bound_source = "assumed from the size of the scene (" + ("cache-resident" if cache_resident else "larger than the Infinity Cache") + "): no PMC record at this launch size"
        # the shade kernel beside it (housekeeping): time, lanes per phase from this run's measurement build, and
        # its own PMC record when one exists at this launch size
        spmc, spmc_source = (None, "N > 1") if n_gpus != 1 else pick_pmc_record(args.steps, result["config"]["batch_depth"], (width, height), args.sphere_tess, kernel="shade")
        sp_ws, sp_ln, sp_cy = st["shadePhaseWaveSteps"], st["shadePhaseLanes"], st["shadePhaseCycles"]
        phase_names = ["path", "volume_fetch", "miss", "hit_record", "tangent", "texcoord", "light_hit", "sample_lambert", "sample_mirror", "sample_glass", "sample_ggx",
                       "sample_ggx_glass", "nee_sample", "nee_eval", "radiance_rmw", "tail", "volume_push", "aov", "kernel_load", "kernel_append", "kernel_iteration"]
        shade_ms, shade_launches = prof["shade"]["ms"], max(1, prof["shade"]["launches"])
        shade_block = {
            "kernel": "twk::shadeKernel<ENV, TEX, PRIMARY, LDS_TABLES, MEASURE, SORT>",
            "ms_per_step": shade_ms / args.steps, "launches": shade_launches,
            "segments_per_step": (st["shadedHits"] + st["missed"]) / args.steps,
            "bound": "the latency of a block iteration at 20 resident waves per CU (profiles/r05_shade_diagnosis.md 7-8): its slowest wave (the GGX class, exact arithmetic), two barriers and a returning atomic; 1.2 x above the 0.16 ms per step its compulsory streams take at 4.2 TB/s. Until round 5: the returning atomic on ONE counter word (87.8 per microsecond)",
            "valu_issue_ratio_4_clock_model_pmc": spmc.get("valu_issue_ratio_uncapped_4_clock_model") if spmc else None,
            "valu_lane_utilisation_pmc": spmc.get("valu_lane_utilisation") if spmc else None,
            "hbm_side_gbps_pmc": (spmc["hbm_bytes_per_launch"] / (shade_ms * 1.0e-3 / shade_launches) / 1.0e9) if spmc else None,
            "pmc_source": spmc_source if spmc else f"null: {spmc_source}",
            # lane occupancy and share of the kernel's wave time per phase of the shading of a segment (measurement build, this run)
            "phases": {n: {"lanes_of_64": round(sp_ln[k] / (64.0 * sp_ws[k]), 3), "wave_executions_per_iteration": round(sp_ws[k] / max(1, sp_ws[20]), 3),
                           "share_of_wave_time": round(sp_cy[k] / max(1, sp_cy[20]), 3)} for k, n in enumerate(phase_names) if sp_ws[k]},
        }
        # SURVEY 8(d) / task contract: achieved = ALGORITHMIC bytes per launch / average launch duration of the dominant kernel,
        # peak = HBM spec. On a scene that lives in the caches these bytes are served by LDS / L1 / L2, so `frac` can pass 1:
        # it then says "not HBM-bound", nothing more. What does bound the kernel, and the HBM-side bytes, are stated beside it.
        result["roofline"] = {
            "kernel": "twk::traceKernel<false, %s, %s, %s, false>" % ("true" if has_cutout else "false", "true" if st["instancesEntered"] else "false", "true" if bi["traceBlocksPerCU"] == 7 else "false"),
            "kernel_note": "template arguments: COUNT, CUTOUT, TWO_LEVEL, W7 (seven resident blocks per CU), PRIMARY; the first of a pass's launches is the <.., true> PRIMARY build, which computes the primary rays instead of fetching them",
            "node_width": 4,
            "trace_blocks_per_cu": int(bi["traceBlocksPerCU"]),
            # what binds the kernel (VERDICT / ADVICE round 3): on a scene the caches hold it is not HBM — `frac` below stays the
            # contract's algorithmic-bytes fraction and may exceed 1 there (frac_valid says so); frac_of_binding_limit is the
            # fraction of the limit that does bind (vector-issue slots x lanes active per instruction, from the PMC record)
            "bound": bound,
            "bound_source": bound_source,
            "frac_valid": bool(not cache_resident or algo_frac_spec <= 1.0),
            # of the limit that binds: HBM-side rate / stream peak where HBM binds; where vector issue binds, the lanes doing work per
            # issued vector instruction (the issue slots themselves are full: see issue.valu_issue_ratio_4_clock_model_pmc)
            "frac_of_binding_limit": ((hbm_side_gbps / stream_peak) if (bound == "hbm" and hbm_side_gbps) else ((pmc.get("valu_lane_utilisation") if pmc else None))),
            "achieved": algo_gbps,
            "peak": HBM_SPEC_GBPS,
            "unit": "GB/s",
            "frac": algo_frac_spec,
            "frac_definition": "SURVEY 8(d) algorithmic bytes (48 B per ray + 64 B per 4-ary node visit + 48 B per triangle test + 64 B per instance entry, cache hits NOT deducted) / average launch duration of this kernel (hipEvents on the handle's stream, this run) / 8 TB/s HBM spec",
            "traffic": traffic,
            "traffic_source": (f"{pmc_source}: rocprofv3 --pmc passes of this command line (FETCH_SIZE x 2 + WRITE_SIZE per launch of this kernel, tools/pmc_collect.sh, tools/pmc_traffic.py), same steps / batch / resolution; carried, not measured in this run"
                               if pmc else f"null: {pmc_source}"),
            "scene_traversal_bytes": scene_bytes,
            "scene_cache_resident": cache_resident,
            "binding_limit": ("vector-instruction issue at partial lane occupancy (SIMDs issue back to back with about half the lanes active: `issue` block; sensitivity runs in DESIGN.md 4.1) - not HBM: the scene is cache-resident"
                              if cache_resident else "memory: node / triangle gathers that miss the caches (see hbm_side)"),
            "north_star_hbm_target": {
                "target": 0.70,
                "hbm_side_frac_of_spec": (hbm_side_gbps / HBM_SPEC_GBPS) if hbm_side_gbps else None,
                "hbm_side_frac_of_measured_stream_peak": (hbm_side_gbps / stream_peak) if hbm_side_gbps else None,
                "algorithmic_frac_of_spec": algo_frac_spec,
                "algorithmic_frac_of_measured_stream_peak": algo_gbps / stream_peak,
                # two readings of "70 % of the measured HBM roofline": SURVEY 8(d) prices ALGORITHMIC bytes (cache hits not deducted);
                # the HBM-side reading asks what the memory actually carried. On a cache-resident scene the first is no bound
                # and the second cannot be reached; on a scene the caches do not hold both mean something.
                "met_by_algorithmic_bytes_vs_measured_stream_peak": bool(algo_gbps / stream_peak >= 0.70),
                "met_by_hbm_side_bytes_vs_measured_stream_peak": bool(hbm_side_gbps and hbm_side_gbps / stream_peak >= 0.70),
                "met": bool((not cache_resident) and algo_gbps / stream_peak >= 0.70),
                "scene_cache_resident": cache_resident,
                "note": ("the Cornell scene's nodes and triangles sit in L2 / Infinity Cache: HBM carries only the ray / hit streams, so the HBM-side fraction cannot reach 0.70 on this scene whatever the kernel does, "
                         "while the algorithmic fraction passes it without being a bound; the target is testable only on a scene that is not cache-resident (tools/big_scene_probe.py, profiles/r03_big_scene_pmc.md)"
                         if cache_resident else "scene larger than the caches (nodes + triangles exceed the 256 MiB Infinity Cache): both fractions are meaningful here; `met` = the SURVEY 8(d) reading on this scene"),
            },
            "hbm_side": {"bytes_per_launch": traffic, "achieved_gbps": hbm_side_gbps,
                         "l2_hit_rate": pmc.get("l2_hit_rate") if pmc else None,
                         "traffic_over_algorithmic_bytes": (traffic / (algo_bytes / trace_launches)) if traffic else None,
                         "compulsory_stream_bytes_per_launch": (36 + 20) * rays / trace_launches,  # ray record in (32 B + 4 B launch index), hit record out (16 B + 4 B instance)
                         },
            # memory-side ceiling nearest to the kernel on a cache-resident scene: 16-byte lane loads issued to the vector memory
            # path (2 per ray, 4 per wide node NOT served by the LDS top-of-tree cache, 3 per triangle, 4 per instance entry)
            # against the chip's divergent-gather ceiling measured in this run (twk_gather_peak, 32 MB table)
            "gather": {"achieved_gbps": gather_gbps, "peak_gbps": gather_peak * 16.0, "frac": gather_gbps / (gather_peak * 16.0),
                       "gather_peak_glaneloads_per_s_measured": gather_peak},
            # vector-instruction issue, from the same PMC passes as `traffic` (null without a matching record)
            "issue": {"valu_issue_utilisation_pmc": pmc.get("valu_issue_utilisation") if pmc else None,
                      "valu_issue_ratio_4_clock_model_pmc": issue_ratio,
                      "issue_ratio_note": "SQ_ACTIVE_INST_VALU x 4 clocks / SIMD clocks of the dispatch, uncapped: an estimate (fma / mul / add issue in fewer than 4 clocks, so a kernel issuing back to back reads above 1)",
                      "simd_clocks_per_vector_instruction_pmc": pmc.get("simd_clocks_per_vector_instruction") if pmc else None,
                      "valu_lane_utilisation_pmc": pmc.get("valu_lane_utilisation") if pmc else None,
                      "wave_cycles_waiting_on_memory_pmc": pmc.get("wave_cycles_waiting_on_memory") if pmc else None},
            "stream_peak_gbps_measured": stream_peak,
            "hbm_spec_gbps": HBM_SPEC_GBPS,
            "algorithmic_bytes_per_launch": algo_bytes / trace_launches,
            "algorithmic_record_table": {"ray_in_hit_out": B_RAY_FIXED, "wide_node_visit(quantised 4-ary node, two binary levels)": b_node, "triangle": B_TRIANGLE, "instance_entry": B_INSTANCE},
            "avg_launch_ms": trace_ms / trace_launches,
            "launches": trace_launches,
            "rays_per_step": rays / args.steps,
            "nodes_per_ray": st["nodesVisited"] / max(1, rays),
            "nodes_per_ray_from_lds_cache": st["cachedNodesVisited"] / max(1, rays),
            "triangles_per_ray": st["trianglesTested"] / max(1, rays),
            "instance_entries_per_ray": st["instancesEntered"] / max(1, rays),
            "lane_occupancy": {"node_step": st["nodesVisited"] / max(1, 64 * st["nodeWaveSteps"]),
                               "triangle_test": st["trianglesTested"] / max(1, 64 * st["triangleWaveSteps"]),
                               "node_wave_steps_per_launch": st["nodeWaveSteps"] / trace_launches,
                               "triangle_wave_steps_per_launch": st["triangleWaveSteps"] / trace_launches,
                               "leaf_wave_steps_per_launch": st["leafWaveSteps"] / trace_launches},
            # where the waves of the kernel spend their time (shader-clock cycles summed over waves, waits included; counting variant of the kernel)
            "wave_time_shares": {n: st["waveCycles"][k] / max(1, st["waveCycles"][5]) for k, n in
                                 enumerate(["refill_ray_fetch", "node_loop", "leaf_or_instance_step", "triangle_loop", "pop_and_result_write"])},
            "Mrays_per_s": rays / trace_s / 1.0e6,
            "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items()},
            "shade": shade_block,
        }

    # ---- CPU baseline: the oracle on the host cores, a bounded sample of the same workload -------------
    if not args.no_cpu_baseline and rank == 0 and n_gpus == 1 and not multi:
        from oracle import orc
        ref = orc.Oracle(miss=info.miss)
        ref.loadApplication(app)
        try:
            allowed = len(os.sched_getaffinity(0))
        except AttributeError:
            allowed = os.cpu_count() or 1
        threads = max(1, min(args.cpu_threads, allowed))
        # first call builds the oracle's BVH and calibrates: 16 rows of iteration 0
        y0 = height // 2 - 8
        ref.render(0, rect=(0, y0, width, y0 + 16), threads=threads)
        tc = time.perf_counter()
        ref.render(0, rect=(0, y0, width, y0 + 16), threads=threads)
        per_row = (time.perf_counter() - tc) / 16.0
        # whole frames, iterations 0 .. n-1, as many as fit the time budget (at least one)
        iterations = int(max(1, min(args.steps, args.cpu_seconds / max(per_row * height, 1e-9))))
        tc = time.perf_counter()
        for it in range(iterations):
            ref.render(it, threads=threads)
        cpu_s = time.perf_counter() - tc
        cpu_img = ref.getOutputBufferHost()
        # the same iterations on the GPU: a free parity check of the benchmarked path on the full-size frame
        dev.setOutputDevicePointer(0, 0)
        for it in range(iterations):
            dev.render(it)
        dev.synchronizeStream()
        gpu_img = dev.getOutputBufferHost()
        result["cpu_baseline"] = {
            "value": width * height * iterations / cpu_s / 1.0e6,
            "unit": "Msamples/s",
            "cores": threads,
            "kind": "port",
            "sample": f"oracle (oracle/liboracle.so, {threads} host threads over rows, own BVH): iterations 0..{iterations - 1} of the {width}x{height} frame = {width * height * iterations} samples in {cpu_s:.1f} s",
            "host_cores_available": allowed,
            "sample_bit_identical_to_gpu": bool(np.array_equal(cpu_img.view(np.uint32), gpu_img.view(np.uint32))),
        }

        # ---- the same BVH on one host core: the device-built tree walked by the host walker of the test tooling on the
        # rays real paths trace (closest-hit and shadow rays of 4,096 pixels of iteration 0, taken from the oracle)
        acc = dev.readAcceleration()
        rng = np.random.default_rng(11)
        pix = rng.integers(0, width * height, 4096)
        path_rays = np.concatenate([ref.debugPath(0, int(i % width), int(i // width)) for i in pix])
        closest, shadow = path_rays[path_rays[:, 8] == 0][:, :8], path_rays[path_rays[:, 8] != 0][:, :8]
        tw = time.perf_counter()
        repeats = 0
        counts = {"nodesVisited": 0, "trianglesTested": 0}
        while time.perf_counter() - tw < 2.0:
            for rays_, any_hit in ((closest, False), (shadow, True)):
                if len(rays_):
                    _, _, c = orc.walk_same_bvh(acc, rays_, anyHit=any_hit)
                    counts = {k: counts[k] + c[k] for k in counts}
            repeats += 1
        walk_s = time.perf_counter() - tw
        n_walk = (len(closest) + len(shadow)) * repeats
        result["cpu_baseline"]["same_bvh_traversal"] = {
            "value": n_walk / walk_s / 1.0e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"oracle/same_bvh_walk.cpp on the device-built tree: {len(closest)} closest-hit + {len(shadow)} shadow rays of 4096 pixels' paths, {repeats} repeats in {walk_s:.1f} s",
            "nodes_per_ray": counts["nodesVisited"] / max(1, n_walk), "triangles_per_ray": counts["trianglesTested"] / max(1, n_walk),
        }

        # ---- the same kernels on one host core: the product's kernel headers compiled by g++ (oracle/host_kernels.cpp), run as
        # the same wavefront on the device-built scene — north_star's "single-threaded C++ CPU fallback of the same kernels"
        try:
            host = orc.HostKernels(dev)
            host_iterations = 0
            while host.seconds < args.cpu_seconds * 0.5 and host_iterations < iterations:
                host.render(host_iterations)
                host_iterations += 1
            dev.setLaunchBatch(batch)
            dev.setOutputDevicePointer(0, 0)
            probe = twk.Device(ordinal=local_rank, miss=info.miss)   # a fresh accumulation buffer for the same iterations
            app.initDevice(probe)
            for it in range(host_iterations):
                probe.render(it)
            gpu_same = probe.getOutputBufferHost()
            probe.close()
            hc = host.counts
            result["cpu_baseline"]["same_kernels_one_core"] = {
                "value": width * height * host_iterations / max(host.seconds, 1e-9) / 1.0e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                "sample": f"oracle/host_kernels.cpp (shade_device.h / trace_device.h / device_math.h compiled by g++ -O2 -ffp-contract=off, the wavefront stages on one thread, the device-built BVH): iterations 0..{host_iterations - 1} of the {width}x{height} frame in {host.seconds:.1f} s",
                "Mrays_per_s": (hc["radianceRays"] + hc["shadowRays"]) / max(host.seconds, 1e-9) / 1.0e6,
                "nodes_per_ray_binary_traversal": hc["nodesVisited"] / max(1, hc["radianceRays"] + hc["shadowRays"]),
                "sample_bit_identical_to_gpu": bool(np.array_equal(host.getOutputBufferHost().view(np.uint32), gpu_same.view(np.uint32))),
            }
        except Exception as e:  # the checker must not take the measured line down with it
            result["cpu_baseline"]["same_kernels_one_core"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        if result_fd is not None:
            os.write(result_fd, (json.dumps(result) + "\n").encode())
        else:
            print(json.dumps(result), flush=True)
    dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
