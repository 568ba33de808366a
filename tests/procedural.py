"""Deterministic procedural image inputs (no RNG): the reference's texture files are not part of its repository
(SURVEY.md §0.5), so tests use formula-generated stand-ins of the same shape and value range."""
import numpy as np


def environment_hdr(width=256, height=128):
    """Analytic lat-long HDR: vertical sky gradient, a warm sun lobe and a dark ground half. Row 0 = v 0 = south pole."""
    v = (np.arange(height, dtype=np.float32) + np.float32(0.5)) / np.float32(height)
    u = (np.arange(width, dtype=np.float32) + np.float32(0.5)) / np.float32(width)
    uu, vv = np.meshgrid(u, v)
    theta = vv * np.float32(np.pi)
    phi = uu * np.float32(2 * np.pi)
    d = np.stack([-np.sin(phi) * np.sin(theta), -np.cos(theta), np.cos(phi) * np.sin(theta)], axis=-1).astype(np.float32)
    sun = np.array([0.4, 0.7, 0.59], np.float32)
    sun /= np.linalg.norm(sun)
    c = np.clip((d * sun).sum(-1), 0, 1)
    sky = np.where(d[..., 1:2] > 0, np.float32(0.3) + np.float32(0.7) * d[..., 1:2] * np.array([0.5, 0.7, 1.0], np.float32),
                   np.array([0.08, 0.07, 0.06], np.float32)).astype(np.float32)
    img = sky + (np.float32(40.0) * c[..., None] ** np.float32(200.0)) * np.array([1.0, 0.9, 0.7], np.float32)
    out = np.ones((height, width, 4), np.float32)
    out[..., :3] = img.astype(np.float32)
    return out


def albedo_checker(size=64):
    """Coloured checker (14-colour cycle in the spirit of Picture::generateRGBA8, apps/Optix7Gui/src/Picture.cpp:658-739)."""
    colours = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1], [1, 1, 1],
                        [.5, 0, 0], [0, .5, 0], [0, 0, .5], [.5, .5, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, .5]], np.float32)
    y, x = np.mgrid[0:size, 0:size]
    idx = ((x // 8) + (y // 8) * 3) % 14
    out = np.ones((size, size, 4), np.float32)
    out[..., :3] = colours[idx]
    return out


def cutout_slots(size=64):
    """Slot pattern for cutout opacity: opaque bars (1), half-transparent bars (0.5, exercises the stochastic draw)
    and holes (0) — the role of slots_alpha.png, which the reference repository does not contain."""
    y, x = np.mgrid[0:size, 0:size]
    band = (x // 4) % 4
    op = np.where(band == 0, 0.0, np.where(band == 1, 0.5, 1.0)).astype(np.float32)
    op = np.where((y // 16) % 2 == 0, op, 1.0).astype(np.float32)
    out = np.ones((size, size, 4), np.float32)
    out[..., 0] = out[..., 1] = out[..., 2] = op
    return out
