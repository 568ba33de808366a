"""CPU: picture files → RGBA32F as Picture::load + Texture::create* deliver them (Picture.cpp:231-560,
Texture.cpp:933-1042): PNG of every colour type / bit depth / scanline filter (encoded here with zlib), Radiance .hdr
with run-length and flat scanlines, PFM; luminance → (L, L, L, 1), RGB → (R, G, B, 1), integers as value / max,
row 0 of the result = bottom row of the picture (IL_ORIGIN_LOWER_LEFT). DevIL itself is not available (third party,
SURVEY §8c): these are format-conformance tests of the replacement readers, not a comparison against DevIL."""
import struct
import zlib

import numpy as np
import pytest

from test_screenshot_files import read_hdr


def _chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def encode_png(samples, depth, colour, filters, palette=None, trns=None, split_idat=False):
    """samples: int array [H, W, channels] of `depth`-bit values. Applies the given filter type per row (cycled)."""
    h, w, ch = samples.shape
    rows = []
    bpp = max(1, ch * depth // 8)
    prev = bytearray((w * ch * depth + 7) // 8)
    for y in range(h):
        vals = samples[y].reshape(-1)
        if depth == 16:
            raw = bytearray(b"".join(struct.pack(">H", int(v)) for v in vals))
        elif depth == 8:
            raw = bytearray(int(v) for v in vals)
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in vals)
            bits += "0" * (-len(bits) % 8)
            raw = bytearray(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
        f = filters[y % len(filters)]
        out = bytearray(len(raw))
        for i in range(len(raw)):
            a = raw[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) // 2, _paeth(a, b, c)][f]
            out[i] = (raw[i] - pred) & 0xff
        rows.append(bytes([f]) + bytes(out))
        prev = raw
    z = zlib.compress(b"".join(rows), 6)
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 0))
    if palette is not None:
        data += _chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    if trns is not None:
        data += _chunk(b"tRNS", bytes(trns))
    data += _chunk(b"tEXt", b"Comment\x00made by the test")
    if split_idat:
        data += _chunk(b"IDAT", z[:len(z) // 2]) + _chunk(b"IDAT", z[len(z) // 2:])
    else:
        data += _chunk(b"IDAT", z)
    return data + _chunk(b"IEND", b"")


@pytest.mark.parametrize("colour,channels,depths", [(0, 1, (1, 2, 4, 8, 16)), (2, 3, (8, 16)), (4, 2, (8, 16)), (6, 4, (8, 16))])
def test_png_colour_types_bit_depths_filters(twk, tmp_path, colour, channels, depths):
    rng = np.random.default_rng(colour)
    for depth in depths:
        h, w = 13, 11
        s = rng.integers(0, 1 << depth, (h, w, channels))
        s[0, 0] = 0
        s[0, 1] = (1 << depth) - 1
        p = tmp_path / f"c{colour}_d{depth}.png"
        p.write_bytes(encode_png(s, depth, colour, filters=[0, 1, 2, 3, 4], split_idat=(depth == 8)))
        got = twk.load_image(str(p))
        v = (s.astype(np.float32) / np.float32((1 << depth) - 1))[::-1]  # normalised floats, bottom row first
        one = np.ones((h, w, 1), np.float32)
        expect = {0: np.concatenate([v, v, v, one], -1), 2: np.concatenate([v, one], -1),
                  4: np.concatenate([v[..., :1]] * 3 + [v[..., 1:]], -1), 6: v}[colour]
        assert got.dtype == np.float32 and got.shape == (h, w, 4)
        assert np.array_equal(got, expect), (colour, depth)


def test_png_palette_and_errors(twk, tmp_path):
    rng = np.random.default_rng(8)
    palette = rng.integers(0, 256, (16, 3))
    trns = [255, 128, 0]  # alpha of the first three palette entries, the rest opaque
    for depth in (1, 2, 4, 8):
        n = min(16, 1 << depth)
        idx = rng.integers(0, n, (9, 10, 1))
        p = tmp_path / f"pal{depth}.png"
        p.write_bytes(encode_png(idx, depth, 3, filters=[0, 2], palette=palette[:n], trns=trns[:min(3, n)]))
        got = twk.load_image(str(p))
        rgb = (palette[idx[..., 0]].astype(np.float32) / np.float32(255))[::-1]
        alpha = np.where(idx[..., 0] < min(3, n), np.float32(trns + [255] * 13)[idx[..., 0]] / np.float32(255), np.float32(1))[::-1]
        assert np.array_equal(got[..., :3], rgb) and np.array_equal(got[..., 3], alpha.astype(np.float32))
    good = encode_png(rng.integers(0, 256, (4, 4, 3)), 8, 2, [1])
    bad = bytearray(good)
    bad[40] ^= 0xff  # flips a byte inside a chunk: the checksum no longer matches
    (tmp_path / "bad.png").write_bytes(bytes(bad))
    (tmp_path / "photo.jpg").write_bytes(b"\xff\xd8\xff\xe0\x00\x10" + bytes(14) + b"\xff\xd9")
    (tmp_path / "junk.bin").write_bytes(b"hello world, not a picture")
    for name, text in (("bad.png", "checksum"), ("photo.jpg", "JPEG: no scan"), ("junk.bin", "unknown image format"), ("missing.png", "cannot read")):
        with pytest.raises(twk.TwkError, match=text):
            twk.load_image(str(tmp_path / name))


def _rle_channel(values):
    """Radiance new-style run-length encoding of one channel of a scanline."""
    out, i, n = bytearray(), 0, len(values)
    while i < n:
        run = 1
        while i + run < n and run < 127 and values[i + run] == values[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, values[i]])
            i += run
        else:
            j = i
            while j < n and j - i < 128:
                r = 1
                while j + r < n and r < 4 and values[j + r] == values[j]:
                    r += 1
                if r >= 4:
                    break
                j += 1
            out += bytes([j - i]) + bytes(values[i:j])
            i = j
    return bytes(out)


def test_hdr_reader_rle_flat_and_round_trip(twk, tmp_path):
    rng = np.random.default_rng(4)
    h, w = 12, 40
    rgbe = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
    rgbe[:, 5:25, :] = rgbe[:, 5:6, :]       # long runs
    rgbe[2, :, 3] = 0                        # exponent 0 → black
    rgbe[..., 3] = np.clip(rgbe[..., 3], 100, 150)
    rgbe[2, :, 3] = 0
    head = b"#?RADIANCE\n# written by the test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w)
    body = b"".join(bytes([2, 2, w >> 8, w & 255]) + b"".join(_rle_channel(list(row[:, c])) for c in range(4)) for row in rgbe)
    (tmp_path / "rle.hdr").write_bytes(head + body)
    (tmp_path / "flat.hdr").write_bytes(head + rgbe.tobytes())
    scale = np.where(rgbe[..., 3:] == 0, 0.0, np.ldexp(1.0, rgbe[..., 3:].astype(np.int32) - 136)).astype(np.float32)
    expect = np.concatenate([rgbe[..., :3].astype(np.float32) * scale, np.ones((h, w, 1), np.float32)], -1)[::-1]
    for name in ("rle.hdr", "flat.hdr"):
        got = twk.load_image(str(tmp_path / name))
        assert got.shape == (h, w, 4) and np.array_equal(got, expect), name
    # writer → reader: decoding what twk_write_hdr stored gives the RGBE-quantised picture, rows in place
    img = (rng.gamma(1.0, 2.0, (7, 9, 4))).astype(np.float32)
    twk.write_hdr(str(tmp_path / "w.hdr"), img, bottomUp=True)
    back = twk.load_image(str(tmp_path / "w.hdr"))
    rgb, _ = read_hdr(str(tmp_path / "w.hdr"))
    assert np.array_equal(back[..., :3], rgb[::-1].astype(np.float32))
    assert (np.abs(back[..., :3] - img[..., :3]) <= img[..., :3].max(-1, keepdims=True) / 128.0).all()
    (tmp_path / "rot.hdr").write_bytes(b"#?RADIANCE\n\n+X 4 -Y 4\n" + bytes(64))
    with pytest.raises(twk.TwkError, match="orientation"):
        twk.load_image(str(tmp_path / "rot.hdr"))


def test_pfm_reader(twk, tmp_path):
    rng = np.random.default_rng(6)
    rgb = rng.normal(0, 3, (5, 8, 3)).astype(np.float32)
    (tmp_path / "c.pfm").write_bytes(b"PF\n8 5\n-1.0\n" + rgb.astype("<f4").tobytes())
    (tmp_path / "b.pfm").write_bytes(b"PF\n8 5\n1.0\n" + rgb.astype(">f4").tobytes())
    grey = rng.random((5, 8)).astype(np.float32)
    (tmp_path / "g.pfm").write_bytes(b"Pf\n8 5\n-1.0\n" + grey.astype("<f4").tobytes())
    one = np.ones((5, 8, 1), np.float32)
    assert np.array_equal(twk.load_image(str(tmp_path / "c.pfm")), np.concatenate([rgb, one], -1))
    assert np.array_equal(twk.load_image(str(tmp_path / "b.pfm")), np.concatenate([rgb, one], -1))
    assert np.array_equal(twk.load_image(str(tmp_path / "g.pfm")), np.concatenate([grey[..., None]] * 3 + [one], -1))


def test_env_map_name_of_the_system_description(twk):
    from conftest import scene_path
    scene = open(scene_path("scene_rtigo3_cornell_box_c1.txt")).read()
    assert twk.Application(system_text="miss 2\nenvMap /data/sky_latlong.hdr\n", scene_text=scene).environment == "/data/sky_latlong.hdr"
    assert twk.Application(system_text="miss 1\n", scene_text=scene).environment == ""


def test_jpeg_reader_matches_libjpeg(twk, tmp_path):
    """Baseline JPEG through twk_load_image == Pillow's libjpeg decode, byte for byte: greyscale, 4:4:4, 4:2:2, 4:2:0,
    sizes that are not multiples of the MCU, optimised Huffman tables, restart intervals, several qualities.
    Progressive files are refused."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(12)

    def picture(h, w):
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos((x + y) / 9.0), (x * 255 / max(1, w - 1))], -1)
        return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)

    cases = [((16, 16), dict(quality=90, subsampling=0)), ((37, 29), dict(quality=75, subsampling=0)),
             ((33, 50), dict(quality=85, subsampling=1)), ((64, 48), dict(quality=60, subsampling=2)),
             ((45, 71), dict(quality=92, subsampling=2, optimize=True)), ((9, 5), dict(quality=80, subsampling=2)),
             ((1, 1), dict(quality=80, subsampling=2)), ((120, 200), dict(quality=70, subsampling=1, restart_marker_blocks=3)),
             ((88, 64), dict(quality=50, subsampling=2, restart_marker_rows=1))]
    for k, ((h, w), opts) in enumerate(cases):
        p = str(tmp_path / f"c{k}.jpg")
        Image.fromarray(picture(h, w)).save(p, **opts)
        expect = np.asarray(Image.open(p).convert("RGB"))
        got = twk.load_image(p)
        assert got.shape == (h, w, 4) and (got[..., 3] == 1).all()
        assert np.array_equal(got[::-1, :, :3], expect.astype(np.float32) / np.float32(255)), (k, opts)
    p = str(tmp_path / "grey.jpg")
    Image.fromarray(picture(40, 31)[..., 0]).save(p, quality=88)
    expect = np.asarray(Image.open(p))
    got = twk.load_image(p)
    assert np.array_equal(got[::-1, :, 0], expect.astype(np.float32) / np.float32(255)) and np.array_equal(got[..., 0], got[..., 2])
    p = str(tmp_path / "prog.jpg")
    Image.fromarray(picture(32, 32)).save(p, progressive=True)
    with pytest.raises(twk.TwkError, match="progressive"):
        twk.load_image(p)


def test_png_header_that_lies_about_its_size_is_refused_without_allocating(twk, tmp_path):
    """IHDR is untrusted input: 60000 x 60000 RGBA16 (28.8 GB raw) over a 20-byte IDAT must be an error return, not an
    allocation of the header's size (or a std::bad_alloc escaping through the C ABI)."""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 60000, 60000, 16, 6, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(b"\x00" * 9)) + chunk(b"IEND", b"")
    path = tmp_path / "liar.png"
    path.write_bytes(png)
    with pytest.raises(twk.TwkError) as e:
        twk.load_image(str(path))
    assert "dimensions" in str(e.value) or "PNG" in str(e.value)
