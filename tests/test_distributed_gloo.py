"""CPU, world_size 2 over gloo: the N > 1 path of bench.py — tile-interleaved shares (distribution 1), local packed
accumulation, ONE gather to rank 0, compositor scatter — executed with the oracle standing in for the GPU renderer.
The assembled image must equal the single-device render bit for bit (seeding by absolute pixel)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

RES = (72, 40)  # not a multiple of tile * world: exercises the out-of-image tile columns
ITERS = 2


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import tweeker_raytracer_amd as twk
    from oracle import orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scenes = os.path.join(ROOT, "scenes")
    app = twk.Application(os.path.join(scenes, "system_rtigo3_cornell_box.txt"), os.path.join(scenes, "scene_rtigo3_cornell_box.txt"))
    app.setResolution(*RES)
    ren = orc.Oracle(index=rank, count=world, miss=app.info.miss)
    ren.loadApplication(app, distribution=1)
    lw = ren.launchWidth
    assert lw == twk.launch_width(RES[0], 8, world)
    for it in range(ITERS):
        ren.render(it)
    local = torch.from_numpy(ren.getOutputBufferHost().copy())
    gathered = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, gathered, dst=0)
    if rank == 0:
        tiles = torch.stack(gathered).numpy()  # [world][H][launchWidth][4] == the layout twk_compositor consumes
        out = np.zeros((RES[1], RES[0], 4), np.float32)
        for d in range(world):
            for y in range(RES[1]):
                for x in range(lw):
                    px = twk.tile_column(x, y, (8, 8), world, d)  # compositor.cu:45-54
                    if px < RES[0]:
                        out[y, px] = tiles[d, y, x]
        assert np.array_equal(out, orc.oracle_compositor(tiles, RES[0], (8, 8)))  # compositor.cu:38-64 restated in the oracle
        np.save(os.path.join(outdir, "composed.npy"), out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_composite_equals_single(tmp_path, twk, orc):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    composed = np.load(os.path.join(str(tmp_path), "composed.npy"))

    scenes = os.path.join(ROOT, "scenes")
    app = twk.Application(os.path.join(scenes, "system_rtigo3_cornell_box.txt"), os.path.join(scenes, "scene_rtigo3_cornell_box.txt"))
    app.setResolution(*RES)
    single = orc.Oracle(miss=app.info.miss)
    single.loadApplication(app)
    for it in range(ITERS):
        single.render(it)
    full = single.getOutputBufferHost()
    assert np.array_equal(composed.view(np.uint32), full.view(np.uint32))


def test_bench_self_launch_relays_failure_without_a_gpu(twk):
    """`python bench.py --gpus 2` with no external launcher starts its own ranks (child torch.distributed.run on 127.0.0.1)
    and passes their exit code on: with no GPU every rank fails loudly in twk_device_create (there is no CPU path), so the
    plain command must exit non-zero and print no result line. (On a GPU box: tests/test_gpu_bench_rehearsal.py.)"""
    import subprocess
    try:
        if twk.device_count() > 0:
            pytest.skip("a GPU is present")
    except twk.TwkError:
        pass
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rehearse-gloo", "--no-roofline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "no CPU path" in out.stderr or "TwkError" in out.stderr or "ChildFailedError" in out.stderr, out.stderr[-1500:]
