"""-m gpu: the screenshot path on the device against the oracle, bytes exact:
  * device log / pow (the tonemapper's only new elementary functions) == oracle,
  * tonemapKernel (twk_tonemap; Application.cpp:2259-2297 as the device kernel its authors ask for) == oracle tonemapper
    on a rendered frame and on synthetic HDR values incl. NaN / inf / negative / huge,
  * rtigo3_hip -s system -d scene -m 1 end to end: benchmark line, screenshot PNG == tonemapped oracle render, and the
    multi-device strategy (N handles sharing this GPU) produces the same picture.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_app, scene_path
from test_screenshot_files import CLI, read_png_rgb8

pytestmark = pytest.mark.gpu

TM = (2.2, 1.0, 1.0, 0.95, 0.9, 0.8, 0.2, 1.2, 0.8)  # gamma, whitePoint, colorBalance, burn, crush, saturation, brightness


def _tm(twk, v):
    return twk.Tonemapper(v[0], v[1], tuple(v[2:5]), v[5], v[6], v[7], v[8])


def test_device_log_pow_bit_exact(twk, orc):
    rng = np.random.default_rng(77)
    dev = twk.Device(ordinal=0)
    n = 1 << 20
    x = np.exp(rng.uniform(-87, 88, n)).astype(np.float32)
    x[:6] = [0.0, 1.0, 0.5, 0.70710678, 1.1754944e-38, 3.4028235e38]
    assert np.array_equal(dev.debugMath(8, x).view(np.uint32), orc.oracle_math(8, x).view(np.uint32))
    b = rng.uniform(0, 8, n).astype(np.float32)
    e = rng.uniform(0.1, 4.0, n).astype(np.float32)
    b[:4] = [0, 0, 3, 1]
    e[:4] = [2, 0, 1, 7]
    assert np.array_equal(dev.debugMath(9, b, e).view(np.uint32), orc.oracle_math(9, b, e).view(np.uint32))
    dev.close()


def test_tonemap_kernel_matches_oracle(twk, orc):
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for i in range(4):
        dev.render(i)
    hdr = dev.getOutputBufferHost()
    for tm in (TM, (1, 1, 1, 1, 1, 1, 0, 1, 1), (1.8, 2.0, 0.7, 1.0, 1.1, 0.0, 1.0, 0.0, 1.5)):
        got = dev.tonemap(_tm(twk, tm))
        assert got.shape == (90, 160, 3) and got.dtype == np.uint8
        assert np.array_equal(got, orc.oracle_tonemap(hdr, tm)), tm
    assert got.max() > 100  # not a black frame

    # synthetic values through a caller-owned device buffer
    hip = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(9)
    vals = np.concatenate([rng.gamma(1.0, 1.0, (1 << 18, 4)), 10.0 ** rng.uniform(-12, 12, (1 << 16, 4)), -rng.random((64, 4))]).astype(np.float32)
    vals[:5, :3] = [[np.nan, 1, 1], [np.inf, 0.5, 0.5], [0, 0, 0], [1, 1, 1], [-np.inf, 2, 3]]
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), C.c_size_t(vals.nbytes)) == 0
    assert hip.hipMemcpy(d, vals.ctypes.data_as(C.c_void_p), C.c_size_t(vals.nbytes), 1) == 0
    got = dev.tonemap(_tm(twk, TM), rgbaDevicePointer=d.value, shape=(vals.shape[0], 1))
    hip.hipFree(d)
    assert np.array_equal(got.reshape(-1, 3), orc.oracle_tonemap(vals, TM))
    with pytest.raises(twk.TwkError):
        dev.tonemap(twk.Tonemapper(gamma=0.0))
    dev.close()


def _run_cli(tmp_path, strategy, env=None, base="system_rtigo3_cornell_box.txt", scene="scene_rtigo3_cornell_box.txt", extra=""):
    system = tmp_path / f"system_{strategy}.txt"
    text = open(scene_path(base)).read()
    text = re.sub(r"(?m)^resolution .*$", "resolution 96 64", text)
    text = re.sub(r"(?m)^samplesSqrt .*$", "samplesSqrt 2", text)
    text = re.sub(r"(?m)^strategy .*$", f"strategy {strategy}", text)
    text += f"\nprefixScreenshot {tmp_path}/shot{strategy}\ngamma 2.2\ncolorBalance 1.0 0.95 0.9\nburnHighlights 0.8\ncrushBlacks 0.2\nsaturation 1.2\nbrightness 0.8\n" + extra
    system.write_text(text)
    r = subprocess.run([CLI, "-s", str(system), "-d", scene_path(scene), "-m", "1"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=300, env={**os.environ, **(env or {})})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    m = re.fullmatch(r"4 / (\d+\.\d{3}) = (\d+\.\d{3}) fps", lines[0])  # Application.cpp:508-511
    assert m and float(m.group(1)) >= 0.0
    assert re.fullmatch(rf"{re.escape(str(tmp_path))}/shot{strategy}_4spp_\d{{7}}_\d{{6}}_000\.png", lines[1])
    return str(system), read_png_rgb8(lines[1])


def test_command_line_benchmark_and_screenshot(twk, orc, tmp_path):
    system, png = _run_cli(tmp_path, 0)
    app = twk.Application(system, scene_path("scene_rtigo3_cornell_box.txt"))
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    for i in range(4):
        ref.render(i)
    expect = orc.oracle_tonemap(ref.getOutputBufferHost(), TM)[::-1]  # the file stores the top row first
    assert png.shape == (64, 96, 3) and np.array_equal(png, expect)
    # The reference's three multi-GPU buffer strategies, three handles sharing this GPU: 3 = local copy (packed tile
    # buffers → peer copies → one compositor launch), 1 = zero copy (every handle accumulates into ONE pinned host frame,
    # DeviceMultiGPUZeroCopy.cpp:106-118), 2 = peer access (ONE frame in the first device's memory,
    # DeviceMultiGPUPeerAccess.cpp:110-158) → the same picture each time
    for strategy in (3, 1, 2):
        _, png3 = _run_cli(tmp_path, strategy, env={"TWK_CLI_VIRTUAL_DEVICES": "3"})
        assert np.array_equal(png3, expect), f"strategy {strategy}: {(png3 != expect).any(-1).sum()} of {png3.shape[0] * png3.shape[1]} pixels differ, columns {np.unique(np.nonzero((png3 != expect).any(-1))[1])[:40]}"


def test_command_line_environment_map_from_file(twk, orc, tmp_path):
    """C3's scene through the command line with the spherical environment read from picture files ("envMap", miss 2):
    a PFM (exact floats) and a Radiance .hdr (RGBE), decoded by the library's own readers; the oracle gets the same
    decoded texels. Exercises createPictures → initTextures → calculateSphericalCDF → miss / light programs."""
    from procedural import environment_hdr
    env = environment_hdr(128, 64)
    (tmp_path / "sky.pfm").write_bytes(b"PF\n128 64\n-1.0\n" + env[..., :3].astype("<f4").tobytes())
    twk.write_hdr(str(tmp_path / "sky.hdr"), env, bottomUp=True)
    assert np.array_equal(twk.load_image(str(tmp_path / "sky.pfm")), env)
    pictures = {}
    for k, name in enumerate(("sky.pfm", "sky.hdr")):
        system, png = _run_cli(tmp_path, 0, base="system_intro_07.txt", scene="scene_intro_07.txt", extra=f"envMap {tmp_path}/{name}\n")
        app = twk.Application(system, scene_path("scene_intro_07.txt"))
        assert app.environment.endswith(name) and app.info.miss == 2
        ref = orc.Oracle(miss=2)
        ref.initTexture(2, twk.load_image(app.environment))
        ref.loadApplication(app)
        for i in range(4):
            ref.render(i)
        assert np.array_equal(png, orc.oracle_tonemap(ref.getOutputBufferHost(), TM)[::-1]), name
        pictures[name] = png
    assert pictures["sky.pfm"].max() > 150 and not np.array_equal(pictures["sky.pfm"], pictures["sky.hdr"])  # RGBE quantisation is visible in the bytes
    # without the file the run stops with the library's message instead of rendering a black sky
    system = tmp_path / "system_missing.txt"
    system.write_text(open(scene_path("system_intro_07.txt")).read() + f"\nresolution 32 32\nenvMap {tmp_path}/nothing.hdr\n")
    r = subprocess.run([CLI, "-s", str(system), "-d", scene_path("scene_intro_07.txt"), "-m", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "not loaded" in r.stderr and "environment texture" in r.stderr
