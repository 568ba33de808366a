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


def test_two_ranks_compose_the_single_device_image():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "3",
           "--rehearse-gloo", "--no-roofline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 8 and res["scaling"] == "weak"
    assert res["config"]["composite_bit_identical_to_single_device"] is True
    w, h = res["config"]["resolution"]
    assert abs(w * h - 2 * 1920 * 1080) < 0.01 * 2 * 1920 * 1080 and w % 8 == 0 and h % 8 == 0
