#!/usr/bin/env python3
"""bench.py — Msamples/s of the HIP wavefront path tracer on the Cornell box (BASELINE.json metric).

A step is one launch of the hot path (≙ one optixLaunch, reference DeviceSingleGPU.cpp:164) = one sample
per pixel over the whole frame; K steps = K progressive iterations (reference Application::benchmark,
Application.cpp:491-513: the timer brackets the launch loop and the final device sync; scene/BVH build and
image write-out are outside). Default K = 64 = the 64 spp of config C2.

N = 1: config C2, Cornell box 1920x1080, full BSDF set, scenes/*cornell_box.txt.
N > 1 (torchrun, one rank per GPU): weak scaling — the frame keeps the 16:9 camera but grows to N x 2,073,600
pixels, tile-interleaved over the ranks exactly like the reference's distribute() (raygeneration.cu:152-164);
every rank accumulates its launchWidth x H share locally for all K steps, then ONE gather to rank 0 over RCCL
and one compositor kernel assemble the image inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)

# Algorithmic bytes of the traversal kernel (DESIGN.md "Roofline model"): fixed record sizes, cache hits do not reduce them.
B_RAY_FIXED = 48      # 32 B ray record in + 16 B hit record out (shadow rays: 32 B ray + 16 B pending contribution)
B_NODE = 128          # one 4-ary wide node (four child boxes + references, two 64-byte halves)
B_TRIANGLE = 48       # one triangle slot (three float4)
B_INSTANCE = 64       # world-to-object rows + BVH root of the instance record


def frame_for(n_gpus, base=(1920, 1080)):
    if n_gpus == 1:
        return base
    s = math.sqrt(n_gpus)
    w = int(round(base[0] * s / 8.0)) * 8
    h = int(round(base[1] * s / 8.0)) * 8
    return (w, h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--system", default=os.path.join(ROOT, "scenes", "system_rtigo3_cornell_box.txt"))
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "scene_rtigo3_cornell_box.txt"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the oracle sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (a 1-GPU box has a share of 16 cores)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N > 1 path on ONE GPU: all ranks share device 0, the gather goes over gloo through host memory")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != n_gpus:
        if world == 1 and n_gpus > 1:
            raise SystemExit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        n_gpus = world

    import numpy as np
    import torch
    import tweeker_raytracer_amd as twk

    dist = None
    if args.rehearse_gloo:
        local_rank = 0
    if n_gpus > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.rehearse_gloo:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    device = torch.device("cuda", local_rank)
    width, height = frame_for(n_gpus)

    app = twk.Application(args.system, args.scene)
    app.setResolution(width, height)
    info = app.info
    dev = twk.Device(ordinal=local_rank, index=rank, count=n_gpus, miss=info.miss)
    app.initDevice(dev, distribution=1 if n_gpus > 1 else 0)
    # path streams for the largest pass the timed loop will issue (twk_launch batches up to 64 iterations per pass):
    # allocated here, not inside the timed region
    dev.reserveLaunchBatch(min(64, max(1, args.steps)))
    lw = dev.launchWidth

    # The accumulation buffer is a torch tensor so RCCL can send it without a copy.
    accum = torch.zeros((height, lw, 4), dtype=torch.float32, device=device)
    dev.setOutputDevicePointer(accum.data_ptr(), accum.numel() * 4)
    gathered = composed = None
    if n_gpus > 1 and rank == 0:
        gathered = torch.empty((n_gpus, height, lw, 4), dtype=torch.float32, device=device)
        composed = torch.zeros((height, width, 4), dtype=torch.float32, device=device)

    def barrier():
        dev.synchronizeStream()
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()

    def check_composite():
        """N > 1 only, after the timed region: the composed image must equal a single-device render of the frame."""
        single = twk.Device(ordinal=local_rank, miss=info.miss)
        app.initDevice(single)
        single.setLaunchBatch(max(1, 64 // n_gpus))  # the frame is n_gpus times a rank's share: same stream memory as a rank
        for it in range(0, args.warmup + args.steps):  # the accumulator keeps the warm-up iterations, like the ranks' buffers
            single.render(it)
        ref_img = single.getOutputBufferHost()
        single.close()
        return bool(np.array_equal(ref_img.view(np.uint32), composed.cpu().numpy().view(np.uint32)))

    def run_steps(first, count, finish=True):
        for it in range(first, first + count):
            dev.render(it)
        dev.synchronizeStream()
        if dist is not None and finish:
            # the one exchange step of the path: gather the packed tile buffers, scatter into the image
            if args.rehearse_gloo:
                host = accum.cpu()
                parts = [torch.empty_like(host) for _ in range(n_gpus)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    gathered.copy_(torch.stack(parts))
            else:
                dist.gather(accum, list(gathered.unbind(0)) if rank == 0 else None, dst=0)
            if rank == 0:
                torch.cuda.synchronize(device)
                dev.compositor(gathered.data_ptr(), composed.data_ptr())
                dev.synchronizeStream()

    run_steps(0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    composite_ok = None
    if dist is not None and rank == 0 and os.environ.get("TWK_BENCH_CHECK_COMPOSITE", "1" if args.rehearse_gloo else "0") == "1":
        composite_ok = check_composite()

    samples = float(width) * float(height) * float(args.steps)
    result = {
        "metric": "Msamples/s (paths x spp x res / s), Cornell box 1920x1080",
        "value": samples / elapsed / 1.0e6,
        "unit": "Msamples/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1.0e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "C2: Cornell box (scene_rtigo3_cornell_box.txt: Lambert + GGX wall + mirror + glass spheres, 64,096 triangles, 8 instances), "
                        f"{width}x{height}, {args.steps} spp progressive (1 spp per step), pathLengths {info.pathLengths[0]} {info.pathLengths[1]}, light 1, miss 0",
            "resolution": [width, height],
            "pixels_per_gpu_per_step": lw * height,
            "parallelism": "single GPU" if n_gpus == 1 else f"tile-interleaved pixels over {n_gpus} ranks (8x8 tiles), one RCCL gather + compositor at the end",
        },
    }
    if composite_ok is not None:
        result["config"]["composite_bit_identical_to_single_device"] = composite_ok
    if args.rehearse_gloo:
        result["config"]["rehearsal"] = "gloo, all ranks on GPU 0 — NOT a scaling measurement"

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
        rays = st["radianceRays"] + st["shadowRays"]
        algo_bytes = (B_RAY_FIXED * rays + B_NODE * st["nodesVisited"] + B_TRIANGLE * st["trianglesTested"] + B_INSTANCE * st["instancesEntered"])
        trace_ms, trace_launches = prof["trace"]["ms"], prof["trace"]["launches"]
        achieved = algo_bytes / (trace_ms * 1.0e-3) / 1.0e9 if trace_ms > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_trace_hbm_traffic.json")
        if os.path.exists(pmc_path):
            try:
                with open(pmc_path) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {
            "kernel": "twk::traceKernel<false, false>",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "note": "achieved counts every node / triangle / instance fetch at full record size (SURVEY 8d: cache hits do not reduce them); the 19 MB scene is served from L2 / Infinity Cache, so frac can exceed 1 - the HBM-side bytes are `traffic` (PMC); what bounds the kernel: DESIGN.md 4.1",
            "algorithmic_bytes_per_launch": algo_bytes / max(1, trace_launches),
            "avg_launch_ms": trace_ms / max(1, trace_launches),
            "launches": trace_launches,
            "rays_per_step": rays / args.steps,
            "nodes_per_ray": st["nodesVisited"] / max(1, rays),
            "triangles_per_ray": st["trianglesTested"] / max(1, rays),
            "lane_occupancy": {"node_step": st["nodesVisited"] / max(1, 64 * st["nodeWaveSteps"]),
                               "triangle_test": st["trianglesTested"] / max(1, 64 * st["triangleWaveSteps"]),
                               "node_wave_steps_per_launch": st["nodeWaveSteps"] / max(1, trace_launches),
                               "triangle_wave_steps_per_launch": st["triangleWaveSteps"] / max(1, trace_launches),
                               "leaf_wave_steps_per_launch": st["leafWaveSteps"] / max(1, trace_launches)},
            "Mrays_per_s": rays / (trace_ms * 1.0e-3) / 1.0e6 if trace_ms > 0 else 0.0,
            "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items()},
        }

    # ---- CPU baseline: the oracle on the host cores, a bounded sample of the same workload -------------
    if not args.no_cpu_baseline and rank == 0 and n_gpus == 1:
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

    if rank == 0:
        print(json.dumps(result), flush=True)
    dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
