"""CPU: the screenshot path around the renderer — tonemapper restatement, PNG / Radiance HDR
writers, screenshot file name, tonemapper options of the system description, command-line option handling of the
rtigo3_hip front end (Options.cpp:44-156). Reference: Application::screenshot (Application.cpp:2231-2335).
No GPU calls; the device tonemap kernel is checked against the same oracle in tests/test_gpu_screenshot.py."""
import os
import re
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, scene_path

CLI = os.path.join(ROOT, "tweeker_raytracer_amd", "rtigo3_hip")


def read_png_rgb8(path):
    """Strict reader for the PNG subset the writer emits (8-bit RGB, filter 0); checks every chunk CRC."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        kind, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(kind + body) == crc, kind
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks][0] == b"IHDR" and chunks[-1] == (b"IEND", b"")
    w, h, depth, colour, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, flt, lace) == (8, 2, 0, 0, 0)
    raw = zlib.decompress(b"".join(b for k, b in chunks if k == b"IDAT"))
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def read_hdr(path):
    data = open(path, "rb").read()
    head, _, rest = data.partition(b"\n\n")
    assert head.startswith(b"#?RADIANCE") and b"FORMAT=32-bit_rle_rgbe" in head
    line, _, pixels = rest.partition(b"\n")
    m = re.fullmatch(rb"-Y (\d+) \+X (\d+)", line)
    h, w = int(m.group(1)), int(m.group(2))
    rgbe = np.frombuffer(pixels, np.uint8).reshape(h, w, 4)
    scale = np.where(rgbe[..., 3:] == 0, 0.0, np.ldexp(1.0, rgbe[..., 3:].astype(np.int32) - 136))
    return rgbe[..., :3].astype(np.float64) * scale, rgbe


def numpy_tonemap(rgba, tm):
    """float32 numpy restatement with numpy's pow (not bit-comparable: libm pow), for a ±1 LSB sanity bound."""
    gamma, white, br, bg, bb, burn, crush, sat, bright = [np.float32(v) for v in tm]
    f = np.float32
    c = (bright / white) * np.array([br, bg, bb], f) * rgba[..., :3].astype(f)
    c = c * ((c * burn + f(1)) / (c + f(1)))
    lum = (c * np.array([0.3, 0.59, 0.11], f)).sum(-1, keepdims=True, dtype=f)
    c = np.maximum(f(0), lum + sat * (c - lum))
    lum = (c * np.array([0.3, 0.59, 0.11], f)).sum(-1, keepdims=True, dtype=f)
    crushed = np.power(c, crush + crush + f(1), dtype=f)
    c = np.where(lum < 1, np.maximum(f(0), crushed + np.sqrt(lum) * (c - crushed)), c)
    c = np.clip(np.power(c, f(1) / gamma, dtype=f), 0, 1)
    return (c * f(255)).astype(np.uint8)


def test_portable_log_and_pow_are_close_to_libm(orc):
    rng = np.random.default_rng(11)
    x = np.concatenate([np.exp(rng.uniform(-80, 80, 200000)).astype(np.float32),
                        np.float32([1.0, 0.5, 2.0, 0.70710678, 1.1754944e-38, 3.4028235e38])])
    got, ref = orc.oracle_math(8, x), orc.oracle_math(8, x, libm=True)
    assert np.abs(got - ref).max() <= 2e-7 * np.maximum(1.0, np.abs(ref)).max() and np.abs(got - ref).max() < 1e-5
    assert orc.oracle_math(8, np.float32([0.0]))[0] == -np.inf and np.isnan(orc.oracle_math(8, np.float32([-1.0]))[0])
    b = rng.uniform(0, 4, 200000).astype(np.float32)
    e = rng.uniform(0.2, 3.0, 200000).astype(np.float32)
    got, ref = orc.oracle_math(9, b, e), orc.oracle_math(9, b, e, libm=True)
    assert np.abs(got - ref).max() <= 4e-6 * max(1.0, ref.max())
    # exact cases the tonemapper relies on: neutral settings are the identity
    assert np.array_equal(orc.oracle_math(9, b, np.ones_like(b)), b)
    assert np.array_equal(orc.oracle_math(9, np.float32([0, 0, 5]), np.float32([2, 0, 0])), np.float32([0, 1, 1]))


def test_oracle_tonemapper(orc):
    rng = np.random.default_rng(5)
    hdr = np.concatenate([rng.gamma(1.0, 0.7, (5000, 4)), rng.uniform(0, 1, (5000, 4)), np.zeros((8, 4)),
                          np.full((8, 4), 1e9)]).astype(np.float32)
    hdr[:, 3] = 1.0
    # neutral tonemapper (Application.cpp:111-120): ldr = clamp(hdr), truncated to a byte
    neutral = (1, 1, 1, 1, 1, 1, 0, 1, 1)
    got = orc.oracle_tonemap(hdr, neutral)
    assert np.array_equal(got, (np.clip(hdr[:, :3], 0, 1) * np.float32(255)).astype(np.uint8))
    # a typical display setting of the reference's system files: gamma 2.2, burn 0.8, crush 0.2, saturation 1.2
    tm = (2.2, 1.0, 1.0, 0.95, 0.9, 0.8, 0.2, 1.2, 0.8)
    got, approx = orc.oracle_tonemap(hdr, tm).astype(int), numpy_tonemap(hdr, tm).astype(int)
    assert np.abs(got - approx).max() <= 1 and (got != approx).mean() < 0.01
    assert np.abs(orc.oracle_tonemap(hdr, tm, libm=True).astype(int) - got).max() <= 1
    # NaN, infinite and negative radiance: the burn term turns inf into NaN (inf/inf), NaN poisons the luminance, and
    # fmaxf(0, .) maps every NaN to 0 before the gamma curve (Application.cpp:2276-2280) — such pixels come out black
    bad = np.float32([[np.nan, -1.0, 0.25, 1.0], [np.inf, 0.5, -np.inf, 1.0], [-1.0, 0.5, 0.25, 1.0]])
    out = orc.oracle_tonemap(bad, tm)
    assert (out[:2] == 0).all() and out[2, 0] == 0 and out[2, 1] > 0


def test_png_writer(twk, tmp_path):
    rng = np.random.default_rng(2)
    for (h, w) in [(1, 1), (5, 7), (200, 150)]:  # the last one needs more than one stored deflate block
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"t_{w}x{h}.png")
        twk.write_png(p, img, bottomUp=True)
        assert np.array_equal(read_png_rgb8(p), img[::-1])
        twk.write_png(p, img, bottomUp=False)
        assert np.array_equal(read_png_rgb8(p), img)
    with pytest.raises(twk.TwkError):
        twk.write_png(str(tmp_path / "no_such_dir" / "x.png"), np.zeros((2, 2, 3), np.uint8))


def test_hdr_writer(twk, tmp_path):
    rng = np.random.default_rng(3)
    img = (rng.gamma(1.0, 2.0, (33, 20, 4)) * 10.0 ** rng.integers(-6, 6, (33, 20, 1))).astype(np.float32)
    img[0, 0, :3] = 0.0
    img[0, 1, :3] = [np.nan, 1.0, 1.0]
    img[0, 2, :3] = [2.0, 2.0, 200.0]  # mantissas (2, 2, >=128): must not read as a run-length header
    p = str(tmp_path / "t.hdr")
    twk.write_hdr(p, img, bottomUp=True)
    rgb, rgbe = read_hdr(p)
    assert rgb.shape == (33, 20, 3)
    src = img[::-1, :, :3].astype(np.float64)
    ok = np.isfinite(src).all(-1)
    peak = src.max(-1, keepdims=True)
    assert (np.abs(rgb - src)[ok] <= (peak / 128.0 + 1e-38).repeat(3, -1)[ok]).all()  # 8-bit mantissa under the shared exponent
    assert (rgbe[-1, 0] == 0).all() and (rgbe[-1, 1] == 0).all()                   # black and NaN pixels → 0 0 0 0
    assert rgbe[-1, 2, 2] >= 128 and tuple(rgbe[-1, 2, :2]) == (2, 2)


def test_tonemapper_options_and_screenshot_name(twk):
    sys_text = ("resolution 32 16\nsamplesSqrt 3\nlight 1\nprefixScreenshot /tmp/shots/cornell\n"
                "gamma 2.2\ncolorBalance 1 0.9 0.8\nwhitePoint 1.5\nburnHighlights 0.8\ncrushBlacks 0.2\nsaturation 1.2\nbrightness 0.7\n")
    app = twk.Application(system_text=sys_text, scene_text=open(scene_path("scene_rtigo3_cornell_box_c1.txt")).read())
    tm = app.tonemapper
    f = np.float32
    assert (tm.gamma, tm.whitePoint, tm.burnHighlights, tm.crushBlacks, tm.saturation, tm.brightness) == (f(2.2), f(1.5), f(0.8), f(0.2), f(1.2), f(0.7))
    assert list(tm.colorBalance) == [f(1), f(0.9), f(0.8)]
    # <prefix>_<spp>spp_<tm_year><mm><dd>_<HHMMSS>_000.<ext> (getDateTime prints tm_year and the 0-based month as they are)
    assert re.fullmatch(r"/tmp/shots/cornell_9spp_\d{3}\d{2}\d{2}_\d{6}_000\.png", app.screenshotPath(True))
    assert app.screenshotPath(False).endswith("_000.hdr")
    scene = open(scene_path("scene_rtigo3_cornell_box_c1.txt")).read()
    neutral = twk.Application(system_text="resolution 8 8\n", scene_text=scene).tonemapper
    assert (neutral.gamma, neutral.whitePoint, list(neutral.colorBalance), neutral.burnHighlights, neutral.crushBlacks,
            neutral.saturation, neutral.brightness) == (1, 1, [1, 1, 1], 1, 0, 1, 1)
    # strategy > 0 means tiled distribution across devices (Application.cpp:223-245)
    assert twk.Application(system_text="strategy 0\n", scene_text=scene).state.distribution == 0
    assert twk.Application(system_text="strategy 3\n", scene_text=scene).state.distribution == 1


def test_command_line_options():
    """Options.cpp:44-156: unknown option, missing argument, help → usage text and a non-zero exit, before any device
    is touched; the interactive mode is refused with a clear message."""
    def run(*args):
        r = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=60)
        return r.returncode, r.stdout + r.stderr
    rc, out = run("--bogus")
    assert rc != 0 and "Unknown option '--bogus'" in out and "Usage:" in out
    rc, out = run("-s")
    assert rc != 0 and "Option '-s' requires additional argument." in out
    rc, out = run("help")
    assert rc != 0 and "-d | --desc   <filename>" in out
    rc, out = run("-w", "64", "-h", "64")
    assert rc != 0 and "system (-s) and scene (-d)" in out
    rc, out = run("-s", scene_path("system_rtigo3_cornell_box_c1.txt"), "-d", scene_path("scene_rtigo3_cornell_box_c1.txt"))
    assert rc != 0 and "-m 1" in out  # default mode 0 = interactive
    rc, out = run("-s", "/nonexistent/system.txt", "-d", "/nonexistent/scene.txt", "-m", "1")
    assert rc != 0 and "ERROR" in out
