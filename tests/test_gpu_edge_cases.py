"""-m gpu: edge cases of the optixTrace contract through the device BVH — tiny geometries (1, 2, 3, 5 triangles:
single-leaf node, top-level inline test, just above the inline threshold), degenerate triangles, sheared / mirrored
instance transforms, coincident geometry (ties → smallest instance, primitive) and a scene built to overflow the
LDS traversal stack. All against the oracle's brute force, bit for bit."""
import numpy as np
import pytest

from conftest import load_app

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _attrs(verts):
    """TriangleAttributes from positions: tangent x, normal z, texcoord = position.xy."""
    verts = np.asarray(verts, np.float32).reshape(-1, 3)
    a = np.zeros((verts.shape[0], 12), np.float32)
    a[:, 0:3] = verts
    a[:, 3] = 1.0
    a[:, 8] = 1.0
    a[:, 9:11] = verts[:, 0:2]
    return a


def _state(twk, w, h, depth=2):
    st = twk.DeviceState()
    st.resolution[0], st.resolution[1] = w, h
    st.tileSize[0], st.tileSize[1] = 8, 8
    st.pathLengths[0], st.pathLengths[1] = 1, depth
    st.distribution, st.samplesSqrt, st.lensShader = 0, 1, 0
    st.epsilonFactor, st.envRotation, st.clockFactor = 500.0, 0.0, 1000.0
    return st


def _material(twk, bsdf=0, albedo=(0.7, 0.6, 0.5)):
    m = twk.MaterialGUI()
    m.indexBSDF = bsdf
    m.albedo[0], m.albedo[1], m.albedo[2] = albedo
    m.absorptionColor[0] = m.absorptionColor[1] = m.absorptionColor[2] = 1.0
    m.absorptionScale, m.ior, m.thinwalled = 0.0, 1.5, 0
    m.useAlbedoTexture = m.useCutoutTexture = 0
    m.roughness[0] = m.roughness[1] = 0.1
    return m


def _pair(twk, orc, geometries, instances, w=48, h=32, depth=2, miss=1):
    """geometries: list of (attrs, indices); instances: list of (geometry, 3x4 transform, material)."""
    cam = twk.camera_frustum((0.0, 0.0, 0.0), 0.75, 0.5, 50.0, 6.0, w / h)
    light = twk.LightDefinition()
    light.type = 0
    light.area = 12.566371
    light.emission[0] = light.emission[1] = light.emission[2] = 1.0
    mats = [_material(twk), _material(twk, 1, (0.9, 0.9, 0.9)), _material(twk, 3, (0.8, 0.7, 0.3))]
    out = []
    for make in (lambda: twk.Device(ordinal=0, miss=miss), lambda: orc.Oracle(miss=miss)):
        r = make()
        r.setState(_state(twk, w, h, depth))
        r.initCameras([cam])
        r.initLights([light])
        r.initMaterials(mats)
        for a, i in geometries:
            r.addGeometry(a, i)
        for g, t, m in instances:
            r.addInstance(g, t, m)
        r.build()
        out.append(r)
    return out


IDENT = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]


def _rays(n, seed, spread=2.5):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(2.0, 5.0, n)
    d = (rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32) - o * np.float32(0.6)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, np.full((n, 1), 1e-4, np.float32), d.astype(np.float32), np.full((n, 1), 1e27, np.float32)], 1).astype(np.float32)


def _compare(dev, ref, rays):
    ref.setTraceMode(False)  # brute force
    g, o = dev.traceRays(rays), ref.traceRays(rays)
    assert np.array_equal(g[1], o[1]), "ids differ"
    hit = o[1][:, 0] >= 0
    assert np.array_equal(_bits(g[0][hit]), _bits(o[0][hit]))
    ga, oa = dev.traceRays(rays, anyHit=True), ref.traceRays(rays, anyHit=True)
    assert np.array_equal(ga[1][:, 0], oa[1][:, 0])
    return hit


def test_tiny_and_degenerate_geometries(twk, orc):
    rng = np.random.default_rng(5)
    geos = []
    for ntri in (1, 2, 3, 5, 9):
        v = rng.uniform(-1, 1, (ntri * 3, 3)).astype(np.float32)
        geos.append((_attrs(v), np.arange(ntri * 3, dtype=np.uint32)))
    # degenerate: zero-area triangle, repeated vertex, plus one proper triangle sharing an edge with a sliver
    v = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0],  [0, 0, 0], [0, 0, 0], [1, 1, 0],  [-1, -1, 0.5], [1, -1, 0.5], [0, 1, 0.5],
                  [-1, -1, 0.5], [0, 1, 0.5], [-1, -1 + 1e-7, 0.5]], np.float32)
    geos.append((_attrs(v), np.arange(12, dtype=np.uint32)))
    inst = []
    for g in range(len(geos)):
        t = np.array(IDENT, np.float32).reshape(3, 4)
        t[:, 3] = [(g % 3 - 1) * 2.2, (g // 3 - 0.5) * 2.2, 0]
        inst.append((g, t.reshape(-1), g % 3))
    # sheared, non-uniformly scaled and mirrored instances of the 5-triangle geometry
    inst.append((3, [1.5, 0.4, 0, 0.3, 0, 0.7, 0.2, -0.2, 0.1, 0, -1.2, 0.5], 0))
    inst.append((3, [-1, 0, 0, -0.5, 0, 1, 0, 0.4, 0, 0, 1, -0.8], 2))
    dev, ref = _pair(twk, orc, geos, inst)
    hit = _compare(dev, ref, _rays(6000, 1))
    assert 0.05 < hit.mean() < 0.95
    for it in range(2):
        dev.render(it)
        ref.render(it)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(ref.getOutputBufferHost()))
    dev.close()


def test_coincident_geometry_ties_and_lds_stack_overflow(twk, orc):
    """64 coincident instances of 4096 coincident triangles: every box of every level is hit, the wide-node step
    pushes three entries per level and the 24-entry LDS stack overflows; the rays are re-traced by
    traceOverflowKernel. All candidates tie in t: the smallest (instance, primitive) must win on both sides."""
    tri = np.array([[-1.5, -1.2, 0], [1.5, -1.2, 0], [0, 1.6, 0]], np.float32)
    ntri = 4096
    geo = (_attrs(np.tile(tri, (ntri, 1))), np.arange(ntri * 3, dtype=np.uint32))
    inst = [(0, IDENT, i % 3) for i in range(64)]
    dev, ref = _pair(twk, orc, [geo], inst, w=8, h=8, depth=2)
    rays = _rays(64, 3, spread=0.6)
    hit = _compare(dev, ref, rays)
    assert hit.any()
    g = dev.traceRays(rays)
    assert (g[1][hit] == 0).all(), "ties must resolve to instance 0, primitive 0"
    dev.statsEnable(True)
    dev.statsGet(True)
    for it in range(2):
        dev.render(it)
        ref.render(it)
    st = dev.statsGet(True)
    assert st["overflowRays"] > 0, "the scene is built to overflow the LDS stack"
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(ref.getOutputBufferHost()))
    dev.close()


def test_lds_stack_overflow_with_cutout_opacity_on_primary_rays(twk, orc):
    """The same stack-overflowing pile of coincident triangles, every one with stochastic cutout opacity: a primary ray
    ignores candidates (drawing from the seed the first traversal launch stored in queue 0), overflows the LDS stack and is
    handed to traceOverflowKernel, which continues behind the candidates ignored so far with the same seed. The fused
    first launch (no generateKernel) must give the oracle's image, and the same image as with the kernel."""
    from procedural import cutout_slots
    import os
    tri = np.array([[-1.5, -1.2, 0], [1.5, -1.2, 0], [0, 1.6, 0]], np.float32)
    ntri = 2048
    geo = (_attrs(np.tile(tri, (ntri, 1))), np.arange(ntri * 3, dtype=np.uint32))
    inst = [(0, IDENT, 0) for _ in range(48)]
    cam = twk.camera_frustum((0.0, 0.0, 0.0), 0.75, 0.5, 50.0, 6.0, 48 / 32)
    light = twk.LightDefinition()
    light.type = 0
    light.area = 12.566371
    light.emission[0] = light.emission[1] = light.emission[2] = 1.0
    mat = _material(twk)
    mat.useCutoutTexture = 1
    images = []
    for fused in ("1", "0", None):
        if fused is not None:
            os.environ["TWK_FUSED_PRIMARY"] = fused
        try:
            r = twk.Device(ordinal=0, miss=1) if fused is not None else orc.Oracle(miss=1)
        finally:
            os.environ.pop("TWK_FUSED_PRIMARY", None)
        r.initTexture(1, cutout_slots())
        r.setState(_state(twk, 48, 32, 3))
        r.initCameras([cam])
        r.initLights([light])
        r.initMaterials([mat])
        r.addGeometry(*geo)
        for g, t, m in inst:
            r.addInstance(g, t, m)
        r.build()
        if fused is not None:
            r.statsEnable(True)
            r.statsGet(True)
        for it in range(3):
            r.render(it)
        if fused is not None:
            assert r.statsGet(True)["overflowRays"] > 0, "the scene is built to overflow the LDS stack"
        images.append(np.array(r.getOutputBufferHost()))
        if fused is not None:
            r.close()
    assert np.isfinite(images[2]).all() and images[2][..., :3].max() > 0.1
    assert np.array_equal(_bits(images[0]), _bits(images[2])), f"fused: {(_bits(images[0]) != _bits(images[2])).any(axis=2).sum()} pixels differ from the oracle"
    assert np.array_equal(_bits(images[1]), _bits(images[2]))


def _random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("seed,offset", [(1, (0.0, 0.0, 0.0)), (2, (0.0, 0.0, 0.0)), (3, (900.0, -350.0, 1700.0)), (4, (-64.0, 8192.0, 33.0))])
def test_random_scenes_far_from_the_origin(twk, orc, seed, offset):
    """Randomised scenes through the PERSISTENT wide-node kernel (twk_launch; twk_trace_rays walks the binary tree):
    triangle soups with triangle sizes over three decades, instances with random rotation / non-uniform scale / shear,
    the whole scene optionally translated far from the origin, where box culling has the least slack relative to
    the rounding of o/d. Images must equal the oracle's brute-force traversal bit for bit: a culled hit shows up as a
    differing pixel."""
    rng = np.random.default_rng(seed)
    geos = []
    for ntri in (7, 60, 300, 1500):
        centre = rng.uniform(-1, 1, (ntri, 1, 3))
        size = 10.0 ** rng.uniform(-2.5, -0.2, (ntri, 1, 1))
        v = (centre + size * rng.normal(size=(ntri, 3, 3))).astype(np.float32).reshape(-1, 3)
        geos.append((_attrs(v), np.arange(ntri * 3, dtype=np.uint32)))
    # a floor under everything so that most paths bounce
    floor = np.array([[-6, -1.3, -6], [6, -1.3, -6], [6, -1.3, 6], [-6, -1.3, -6], [6, -1.3, 6], [-6, -1.3, 6]], np.float32)
    geos.append((_attrs(floor), np.arange(6, dtype=np.uint32)))
    off = np.array(offset)
    inst = []
    for k in range(14):
        m = _random_rotation(rng) @ np.diag(rng.uniform(0.3, 1.4, 3)) @ (np.eye(3) + np.triu(rng.uniform(-0.3, 0.3, (3, 3)), 1))
        t = np.concatenate([m, (rng.uniform(-2.2, 2.2, 3) + off)[:, None]], 1).astype(np.float32)
        inst.append((k % 4, t.reshape(-1), k % 3))
    t = np.concatenate([np.eye(3), off[:, None]], 1).astype(np.float32)
    inst.append((4, t.reshape(-1), 0))

    w, h = 40, 30
    cam = twk.camera_frustum(tuple(float(c) for c in off), 0.75, 0.55, 55.0, 7.0, w / h)
    light = twk.LightDefinition()
    light.type = 0
    light.area = 12.566371
    light.emission[0] = light.emission[1] = light.emission[2] = 1.0
    mats = [_material(twk), _material(twk, 1, (0.9, 0.9, 0.9)), _material(twk, 3, (0.8, 0.7, 0.3))]
    imgs = []
    for r in (twk.Device(ordinal=0, miss=1), orc.Oracle(miss=1)):
        r.setState(_state(twk, w, h, depth=4))
        r.initCameras([cam])
        r.initLights([light])
        r.initMaterials(mats)
        for a, i in geos:
            r.addGeometry(a, i)
        for g, tr, m in inst:
            r.addInstance(g, tr, m)
        r.build()
        if hasattr(r, "setTraceMode"):
            r.setTraceMode(False)  # brute force over all triangles of all instances
        for it in range(3):
            r.render(it)
        imgs.append(r.getOutputBufferHost())
    gpu, cpu = imgs
    assert np.isfinite(cpu).all() and (cpu[..., :3] > 0).mean() > 0.5
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{mism} of {w * h} pixels differ"


@pytest.mark.parametrize("system,scene,lo,hi", [
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (-0.99, 0.01, -0.99), (0.99, 1.99, 0.99)),
    ("system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", (-7.9, -0.5, -7.9), (7.9, 3.0, 7.9)),
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (-12.0, -0.5, -12.0), (12.0, 3.5, 12.0)),
])
def test_persistent_kernel_against_oracle_on_adversarial_rays(twk, orc, system, scene, lo, hi):
    """The hot traversal kernel itself (twk_debug_trace_queue: one launch of the persistent kernel, radiance and shadow
    queue) on 400 k rays chosen to be hard: random rays, rays leaving the surfaces they start on at grazing angles with
    the scene epsilon and with tmin 0, axis-parallel rays, rays from vertices along edges. Hit records (t, beta, gamma,
    instance, primitive) equal the oracle's bit for bit, occlusion flags equal. (This is how round 2 would have found
    the flat-floor box-padding miss at once instead of through one pixel of one image.)"""
    from conftest import load_app
    app = load_app(twk, system, scene, (32, 32))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    _, _, tris, _ = dev.readAcceleration()
    slot_primitive = tris[:, 3].view(np.int32)
    rng = np.random.default_rng(2026)
    lo, hi = np.array(lo, np.float32), np.array(hi, np.float32)

    def unit(v):
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    n = 100_000
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = unit(rng.normal(size=(n, 3)))
    d[:2000, 0] = 0.0
    d[2000:4000, 1] = 0.0
    d[4000:5000, 1:] = 0.0  # along +-x
    d[:5000] = unit(d[:5000] + 1e-30)
    base = np.concatenate([o, np.full((n, 1), 5e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], 1)
    o_tbg, o_ids = ref.traceRays(base)
    hit = o_ids[:, 0] >= 0
    # second generation: leave the hit points at grazing angles (origin ON a surface, as every bounce ray is)
    p = (o[hit] + d[hit] * o_tbg[hit, :1]).astype(np.float32)
    m = p.shape[0]
    g = unit(rng.normal(size=(m, 3)))
    graze = unit(d[hit] + 1e-3 * g)  # continue almost along the incoming direction = almost tangent for grazing incidences
    surf_a = np.concatenate([p, np.full((m, 1), 5e-5, np.float32), g, np.full((m, 1), 1e27, np.float32)], 1)
    surf_b = np.concatenate([p, np.zeros((m, 1), np.float32), graze, np.full((m, 1), 1e27, np.float32)], 1)
    attr, _ = app.geometry(0)
    closest = np.concatenate([base, surf_a, surf_b]).astype(np.float32)
    shadow = closest.copy()
    shadow[:, 7] = rng.uniform(0.05, 6.0, shadow.shape[0]).astype(np.float32)

    rec, inst, occ = dev.debugTraceQueue(closest, shadow)
    o_tbg, o_ids = ref.traceRays(closest)
    assert np.array_equal(inst, o_ids[:, 0]), f"{(inst != o_ids[:, 0]).sum()} instance ids differ"
    hit = o_ids[:, 0] >= 0
    assert hit.mean() > 0.2  # the open scenes let half of the rays escape
    prim = slot_primitive[rec[hit, 3].view(np.int32)]
    assert np.array_equal(prim, o_ids[hit, 1]), f"{(prim != o_ids[hit, 1]).sum()} primitive ids differ"
    assert np.array_equal(_bits(rec[hit, :3]), _bits(o_tbg[hit])), "t / beta / gamma differ"
    _, s_ids = ref.traceRays(shadow, anyHit=True)
    assert np.array_equal(occ, s_ids[:, 0]), f"{(occ != s_ids[:, 0]).sum()} occlusion flags differ"
    assert 0.02 < occ.mean() < 0.98
    dev.close()


def test_tree_height_is_measured_and_a_scene_beyond_the_stacks_is_refused(twk, monkeypatch):
    """the SAH builder may peel one primitive per level; a tree deeper than the traversal stacks
    (20 LDS + 72 HBM entries, trace_device.h) would drop subtrees without a word. twk_build measures the binary-tree
    height in its refit pass, reports it (TwkBuildInfo.maxTraversalDepth) and refuses a scene beyond the stacks; the
    limit can only be lowered (TWK_MAX_TRAVERSAL_DEPTH), which is how the refusal is exercised here."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (32, 32))
    for quality in (0, 1):
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        dev.setBuildQuality(quality)
        app.initDevice(dev)
        info = dev.buildInfo()
        # 32,040 triangles per sphere, leaves of <= 2: at least log2(16,020) = 14 levels below a top level of >= 3
        assert 17 <= info["maxTraversalDepth"] <= 64, info
        dev.close()
    monkeypatch.setenv("TWK_MAX_TRAVERSAL_DEPTH", "8")
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    with pytest.raises(twk.TwkError) as e:
        app.initDevice(dev)
    assert "levels deep" in str(e.value) and "traversal stacks hold 92" in str(e.value)
    with pytest.raises(twk.TwkError):
        dev.render(0)
        dev.synchronizeStream()
    dev.close()


@pytest.mark.parametrize("system,scene", [
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt"),
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt"),
])
def test_sah_build_is_reproducible(twk, system, scene):
    """the binned-SAH builder ranked primitives and numbered nodes with atomics, so topology, slot order and
    the LDS top-of-tree cache differed from run to run (hit records never did). Round 3: stable partition by prefix sums,
    node indices by per-level scans — the same input gives the same acceleration structure, byte for byte."""
    app = load_app(twk, system, scene, (32, 32))
    builds = []
    for _ in range(3):
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        app.initDevice(dev)
        info, nodes, tris, inst = dev.readAcceleration()
        builds.append((info, nodes.copy(), tris.copy(), inst.copy(), dev.buildInfo()))
        dev.close()
    for other in builds[1:]:
        assert other[0] == builds[0][0]
        for a, b in zip(other[1:4], builds[0][1:4]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert other[4]["sahInnerCost"] == builds[0][4]["sahInnerCost"] and other[4]["maxTraversalDepth"] == builds[0][4]["maxTraversalDepth"]
