"""-m gpu: parity at the scale where the HBM-roofline fractions are quoted ("What's missing" 3).

`profiles/r0*_big_scene_pmc.md` quotes the traversal kernel's fraction of the HBM roofline on the Cornell room with its two
spheres tessellated 1000 x 500 (2.0 M triangles) and 2800 x 1400 (15.7 M triangles, 1.75 GB of nodes and triangle slots —
not cache-resident). Those scenes take code paths no shipped scene takes: rays outgrow the LDS traversal stack of the
persistent kernel and are continued in HBM, and the six-blocks-per-CU build of the kernel is the default (the seven-block
build is the default below a million nodes). Stands in for optixAccelBuild + optixTrace at that scale
(apps/rtigo3/src/Device.cpp:1362-1407,1456-1486, shaders/raygeneration.cu:84-89).

Per scene and per build of the kernel (six / seven resident blocks per CU):
  * the full 1920x1080 frame, 8 iterations: four 64x48 windows (mirror sphere, glass sphere, their silhouettes, the floor under the glass) equal the
    oracle's (its own median-split BVH over the same 15.7 M triangles) bit for bit;
  * >= 20 k rays real paths trace (closest-hit and shadow rays of random pixels, taken from the oracle) plus rays built to
    outgrow the LDS stack (along the spheres' axes through the pole fans, where thousands of sliver boxes overlap) go through
    ONE launch of the persistent kernel (twk_debug_trace_queue): hit records and occlusion flags equal the oracle's,
    and `overflowRays > 0` is asserted — the HBM-continued path is the one being checked;
  * 20 k rays of that set whose hit lies in a subset of the triangles — all walls and the light, and of each sphere the
    north pole fan + a band of rows at the equator (~50 k triangles per sphere; rays aimed at those bands are part of the
    set) — are BRUTE-FORCED by the oracle over that subset: the same instance, primitive and t / beta / gamma bits. The
    hit is an intersection in the oracle's arithmetic, independent of any BVH.
"""
import os

import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu

WINDOWS = [(640, 300, 704, 348), (1100, 300, 1164, 348), (930, 400, 994, 448), (1100, 40, 1164, 88)]  # left rim of the mirror sphere (its centre mirrors the open front of the room: black), glass sphere, the silhouette of one against the other, the floor under the glass sphere
ITERATIONS = 8


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _app(twk, tess):
    system = open(os.path.join(SCENES, "system_rtigo3_cornell_box.txt")).read()
    scene = open(os.path.join(SCENES, "scene_rtigo3_cornell_box.txt")).read()
    assert "sphere 180 90" in scene
    return twk.Application(system_text=system, scene_text=scene.replace("sphere 180 90", f"sphere {tess} {tess // 2}"))


def _unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _pole_rays(app, rng, n_per_sphere):
    """Rays along the axis of each sphere, through both pole fans: every sliver triangle of a fan has the pole in its box."""
    rays = []
    for (g, t, m, l) in app.instances:
        attr, idx = app.geometry(g)
        if idx.shape[0] // 3 < 1000:
            continue
        t = np.asarray(t, np.float32).reshape(3, 4)
        centre = t[:, 3]
        n = n_per_sphere
        off = (rng.normal(size=(n, 3)) * rng.choice([1e-5, 1e-4, 1e-3, 1e-2], size=(n, 1))).astype(np.float32)
        off[:, 1] = 0.0
        o = (centre + np.array([0.0, 1.35, 0.0], np.float32) + off).astype(np.float32)
        d = _unit(np.array([0.0, -1.0, 0.0], np.float32) + (rng.normal(size=(n, 3)) * 1e-3).astype(np.float32))
        rays.append(np.concatenate([o, np.full((n, 1), 5e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], 1))
    return np.concatenate(rays).astype(np.float32)


def _band_rows(tess, rows):
    """Rows of a sphere's triangle grid kept in the brute-force subset: the north pole fan and a band at the equator, about
    50 k triangles together (the oracle brute-forces ~70 M triangle tests per second)."""
    n = max(6, 50_000 // (2 * tess))
    fan = n // 3
    return np.concatenate([np.arange(rows - fan, rows), np.arange(rows // 2 - (n - fan) // 2, rows // 2 - (n - fan) // 2 + (n - fan))])


def _aimed_rays(app, tess, rng, n_per_sphere):
    """Rays from outside a sphere towards points of its equator band (triangle_meshes.cpp makeSphere: latitude theta from the
    south pole, position = radius * (cos phi sin theta, -cos theta, -sin phi sin theta))."""
    rays = []
    for (g, t, m, l) in app.instances:
        attr, idx = app.geometry(g)
        if idx.shape[0] // 3 < 1000:
            continue
        t = np.asarray(t, np.float32).reshape(3, 4)
        centre, radius = t[:, 3], float(t[0, 0])
        rows = idx.shape[0] // 3 // (2 * tess)
        band = _band_rows(tess, rows)
        equator = band[band < rows * 3 // 4]
        n = n_per_sphere
        theta = rng.uniform(equator.min() + 0.2, equator.max() + 0.8, n) * (np.pi / rows)
        phi = rng.uniform(0.0, 2.0 * np.pi, n)
        normal = np.stack([np.cos(phi) * np.sin(theta), -np.cos(theta), -np.sin(phi) * np.sin(theta)], 1)
        p = centre + radius * normal
        o = (p + normal * rng.uniform(0.05, 0.3, (n, 1)) + rng.normal(size=(n, 3)) * 0.02).astype(np.float32)
        d = _unit((p - o).astype(np.float32))
        rays.append(np.concatenate([o, np.full((n, 1), 5e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], 1))
    return np.concatenate(rays).astype(np.float32)


def _path_rays(ref, width, height, rng, pixels):
    pix = rng.integers(0, width * height, pixels)
    rays = np.concatenate([ref.debugPath(0, int(i % width), int(i // width)) for i in pix])
    return rays[rays[:, 8] == 0][:, :8].copy(), rays[rays[:, 8] != 0][:, :8].copy()


def _subset_oracle(orc, app, tess):
    """An oracle over the small instances + a band of each sphere's triangle rows; returns it and, per geometry, the map
    from original primitive index to the subset's (-1: not in the subset)."""
    sub = orc.Oracle(miss=app.info.miss)
    sub.setShaderVariant(getattr(app.info, "shaderVariant", 0))
    sub.setState(app.state)
    sub.initCameras(app.cameras)
    sub.initLights(app.lights)
    sub.initMaterials(app.materials)
    sub.clearScene()
    maps = []
    for g in range(app.info.numGeometries):
        attr, idx = app.geometry(g)
        tris = idx.shape[0] // 3
        if tris < 1000:
            assert sub.addGeometry(attr, idx) == g
            maps.append(np.arange(tris, dtype=np.int64))
            continue
        per_row = 2 * tess                     # gridIndices: two triangles per quad, tess quads per row of latitude
        rows = tris // per_row
        assert rows * per_row == tris
        keep_rows = _band_rows(tess, rows)
        keep = (keep_rows[:, None] * per_row + np.arange(per_row)[None, :]).reshape(-1)
        new_index = np.full(tris, -1, np.int64)
        new_index[keep] = np.arange(keep.shape[0])
        assert sub.addGeometry(attr, idx.reshape(-1, 3)[keep].reshape(-1)) == g
        maps.append(new_index)
    for (g, t, m, l) in app.instances:
        sub.addInstance(g, t, m, l)
    sub.setTraceMode(False)  # brute force over every triangle of the subset
    return sub, maps


@pytest.mark.parametrize("tess,blocks", [(1000, 6), (1000, 7), (2800, 6), (2800, 7)])
def test_big_room_windows_path_rays_and_overflow_against_the_oracle(twk, orc, monkeypatch, tess, blocks):
    monkeypatch.setenv("TWK_TRACE_WAVES_RUNTIME", str(blocks))  # read at twk_device_create: which build of the persistent kernel runs
    app = _app(twk, tess)
    width, height = app.info.resolution[0], app.info.resolution[1]
    assert (width, height) == (1920, 1080)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    info = dev.buildInfo()
    assert info["traceBlocksPerCU"] == blocks and info["triangleSlots"] > (1_900_000 if tess == 1000 else 15_000_000)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)

    # ---- the frame: windows of the full-size image after 8 iterations, statistics of the whole frame
    dev.statsEnable(True)
    dev.statsGet(True)
    for it in range(ITERATIONS):
        dev.render(it)
    dev.synchronizeStream()
    frame_stats = dev.statsGet(True)
    gpu = dev.getOutputBufferHost()
    for window in WINDOWS:
        for it in range(ITERATIONS):
            ref.render(it, rect=window, threads=8)
    cpu = ref.getOutputBufferHost()
    for (x0, y0, x1, y1) in WINDOWS:
        g, c = gpu[y0:y1, x0:x1], cpu[y0:y1, x0:x1]
        assert c[..., :3].std() > 1e-3 and np.isfinite(g).all(), f"window {(x0, y0, x1, y1)} shows something"
        mism = (_bits(g) != _bits(c)).any(axis=2).sum()
        assert mism == 0, f"tess {tess}, {blocks} blocks: {mism} pixels of window {(x0, y0, x1, y1)} differ, max |diff| {np.abs(g - c).max()}"

    # ---- one launch of the persistent kernel over real path rays + rays built to outgrow the LDS stack
    rng = np.random.default_rng(4 + tess)
    closest, shadow = _path_rays(ref, width, height, rng, 8192)
    assert closest.shape[0] + shadow.shape[0] >= 20_000
    poles = _pole_rays(app, rng, 3000)
    closest = np.concatenate([closest, poles, _aimed_rays(app, tess, rng, 4500)]).astype(np.float32)
    pole_shadow = poles.copy()
    pole_shadow[:, 7] = rng.uniform(0.3, 1.3, pole_shadow.shape[0]).astype(np.float32)
    shadow = np.concatenate([shadow, pole_shadow]).astype(np.float32)
    rec, inst, occ = dev.debugTraceQueue(closest, shadow)
    query_stats = dev.statsGet(True)
    dev.statsEnable(False)
    print(f"tess {tess}, {blocks} blocks per CU: frame overflow rays {frame_stats['overflowRays']} of {frame_stats['radianceRays'] + frame_stats['shadowRays']}, "
          f"query overflow rays {query_stats['overflowRays']} of {closest.shape[0] + shadow.shape[0]}, deepest ray {query_stats['maxNodesPerRay']} node steps, dropped pushes {query_stats['droppedStackPushes']}")
    assert query_stats["overflowRays"] > 0, "the pole rays are built to outgrow the LDS stack: the HBM-continued traversal is what is checked here"
    assert query_stats["droppedStackPushes"] == 0 and frame_stats["droppedStackPushes"] == 0
    _, _, tris, _ = dev.readAcceleration()
    slot_primitive = tris[:, 3].view(np.int32)
    o_tbg, o_ids = ref.traceRays(closest)  # the oracle's own BVH
    assert np.array_equal(inst, o_ids[:, 0]), f"{(inst != o_ids[:, 0]).sum()} instance ids differ"
    hit = o_ids[:, 0] >= 0
    assert hit.mean() > 0.7  # the room is open towards the camera
    prim = slot_primitive[rec[hit, 3].view(np.int32)]
    assert np.array_equal(prim, o_ids[hit, 1]), f"{(prim != o_ids[hit, 1]).sum()} primitive ids differ"
    assert np.array_equal(_bits(rec[hit, :3]), _bits(o_tbg[hit])), "t / beta / gamma differ"
    _, s_ids = ref.traceRays(shadow, anyHit=True)
    assert np.array_equal(occ, s_ids[:, 0]), f"{(occ != s_ids[:, 0]).sum()} occlusion flags differ"

    # ---- brute force, independent of any BVH: the rays whose hit lies in the subset, over the subset
    sub, maps = _subset_oracle(orc, app, tess)
    geometry_of_instance = np.array([g for (g, t, m, l) in app.instances])
    sub_prim = np.full(closest.shape[0], -1, np.int64)
    sub_prim[hit] = np.array([maps[geometry_of_instance[i]][p] for i, p in zip(o_ids[hit, 0], o_ids[hit, 1])])
    chosen = np.nonzero(sub_prim >= 0)[0]
    big = np.array([maps[geometry_of_instance[i]].shape[0] >= 1000 for i in o_ids[chosen, 0]])
    assert chosen.shape[0] >= 20000 and big.sum() >= 2000, f"{chosen.shape[0]} rays hit the subset, {big.sum()} of them a sphere band"
    chosen = np.concatenate([chosen[big], chosen[~big]])[:20000]  # every sphere-band hit, the rest walls and light
    b_tbg, b_ids = sub.traceRays(closest[chosen])
    assert np.array_equal(b_ids[:, 0], inst[chosen]) and np.array_equal(b_ids[:, 1], sub_prim[chosen])
    assert np.array_equal(_bits(b_tbg), _bits(rec[chosen, :3])), "brute force over the triangle subset: t / beta / gamma differ from the device's"
    dev.close()
