"""-m gpu: bench.py's N > 1 path rehearsed on ONE GPU (2 ranks share device 0, gather over gloo through host memory):
tile-interleaved shares, local accumulation, gather, compositor kernel — the composed image must equal the
single-device render of the same frame, bit for bit. (The real 8-GPU run over RCCL is the driver's.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _rehearse(ranks, extra=()):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("WORLD_SIZE", None)
    # the PLAIN command, no external launcher: bench.py starts its ranks itself (torch.distributed.run as a child process,
    # before this process touches the GPU) and relays rank 0's line — what `python bench.py --gpus N` does on the 8-GPU node
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "8", "--warmup", "3",
           "--batch", "8", "--rehearse-gloo", "--no-roofline", *extra]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_compose_the_single_device_image_of_c5(ranks):
    """Default N > 1 mode = C5: the fixed 3840x2160 Cornell frame, strong scaling, launchWidth = roundup(ceil(W / N), 8)
    (DeviceMultiGPULocalCopy.cpp:84-97); the composite check runs by default and its CRCs are in the line."""
    res = _rehearse(ranks)
    assert res["n_gpus"] == ranks and res["steps"] == 8 and res["scaling"] == "strong"
    cfg = res["config"]
    assert cfg["resolution"] == [3840, 2160]
    lw = -(-3840 // ranks)
    lw = (lw + 7) & ~7
    assert cfg["launch_width"] == lw and cfg["pixels_per_gpu_per_step"] == lw * 2160
    assert cfg["gather_bytes_per_rank"] == lw * 2160 * 16 and cfg["gather_plus_compositor_ms"] > 0
    assert cfg["composite_bit_identical_to_single_device"] is True
    assert cfg["crc32_composed"] == cfg["crc32_single_device"]
    # where the timed region went: slowest rank's rendering + the one exchange step <= the whole
    assert 0 < res["render_ms_max_rank"] <= res["ms_per_step"] * res["steps"] * 1.001
    assert res["exchange_ms"] == cfg["gather_plus_compositor_ms"] > 0
    assert res["value_render_only"] >= res["value"] > 0
    assert res["metric"].endswith("3840x2160")


def test_external_launcher_still_works():
    """The driver's other form: python -m torch.distributed.run ... bench.py --gpus N (WORLD_SIZE set: no self-launch)."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--batch", "4", "--rehearse-gloo", "--no-roofline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["composite_bit_identical_to_single_device"] is True
    # the N > 1 line carries its own denominator: the same frame and steps timed on one device
    assert res["single_device_same_frame_Msamples_per_s"] > 0 and res["strong_scaling_vs_same_frame"] > 0
    assert abs(res["strong_scaling_vs_same_frame"] - res["value"] / res["single_device_same_frame_Msamples_per_s"]) < 1e-6 * res["strong_scaling_vs_same_frame"]


def test_two_ranks_weak_mode():
    res = _rehearse(2, ("--weak",))
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    assert res["config"]["composite_bit_identical_to_single_device"] is True
    w, h = res["config"]["resolution"]
    assert abs(w * h - 2 * 1920 * 1080) < 0.01 * 2 * 1920 * 1080 and w % 8 == 0 and h % 8 == 0


def test_single_gpu_line_carries_the_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` — the driver's N = 1 command with a short CPU sample: ONE JSON line with
    the contract's fields, the roofline block of the dominant kernel (algorithmic bytes / measured launch duration / spec
    peak) and the CPU baseline; value = pixels x steps / the timed region."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--cpu-seconds", "2"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in res, key
    assert res["n_gpus"] == 1 and res["steps"] == 6 and res["warmup"] == 2 and res["unit"] == "Msamples/s" and res["higher_is_better"] is True
    assert res["dtype"] == "f32" and res["data"] == "synthetic" and res["vs_baseline"] is None and "workload" in res["config"]
    assert abs(res["value"] - 1920 * 1080 / (res["ms_per_step"] * 1.0e3)) < 1.0e-6 * res["value"]
    roof = res["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in roof, key
    # the Cornell scene is cache-resident: what binds the traversal kernel there is vector-instruction issue, and the line says so
    # beside the contract's algorithmic-bytes fraction (which may pass 1 on such a scene: frac_valid)
    # (this run's launch size — 6 steps — has no PMC record under profiles/: the bound is then the one assumed from the scene's size, and says so)
    assert roof["bound"] == "valu-issue" and roof["bound_source"].startswith("assumed") and roof["scene_cache_resident"] is True and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    shade = roof["shade"]
    assert shade["ms_per_step"] > 0 and 0.2 < shade["phases"]["nee_eval"]["lanes_of_64"] <= 1.0 and shade["phases"]["kernel_iteration"]["share_of_wave_time"] == 1.0
    assert roof["frac_valid"] == (roof["frac"] <= 1.0)
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1.0e-9 and roof["achieved"] > 0
    assert roof["kernel"].startswith("twk::traceKernel<") and roof["avg_launch_ms"] > 0
    cpu = res["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["sample_bit_identical_to_gpu"] is True


def test_rccl_branch_with_a_world_of_one_rank():
    """The RCCL branch of the exchange step — process group on the nccl backend, dist.gather of the device tensor into rank 0's
    list of views, the compositor launch, the MAX all-reduce and the barriers — cannot run with two ranks on one GPU (RCCL
    refuses a device twice). `--rehearse-rccl` executes it with a world of ONE rank: every call of the N > 1 path on the real
    backend, the composed frame equal to the plain single-device render."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(key, None)
    env["MASTER_PORT"] = str(29700 + (os.getpid() % 200))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "4", "--rehearse-rccl"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"ONE JSON line on stdout (RCCL's version banner belongs on stderr): {lines[:6]}"
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == 1 and cfg["resolution"] == [3840, 2160] and cfg["launch_width"] == 3840
    assert cfg["composite_bit_identical_to_single_device"] is True and cfg["crc32_composed"] == cfg["crc32_single_device"]
    assert cfg["gather_plus_compositor_ms"] > 0 and cfg["closing_barrier_ms"] >= 0
    assert cfg["collective"] == "one gather to rank 0" and cfg["rehearsal"].startswith("RCCL with a world of one rank")
