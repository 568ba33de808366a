"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: integer work (RNG, ids, counters) bit-exact. Floating point: the kernels and the oracle evaluate the same
IEEE expressions (no FMA contraction, portable Cephes sin/cos/exp/atan/acos, correctly rounded sqrt and
division), so the stated tolerance is ZERO ulp: images and hit records must be bit-identical. A relative-L2
budget of 1e-4 per image is kept as a documented fallback only for cutout scenes (implementation-defined
any-hit order) — none of the scenes below uses it.
"""
import numpy as np
import pytest

from conftest import load_app

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def dev0(twk):
    if twk.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests must run on the GPU box")
    return 0


def test_device_math_bit_exact(twk, orc, dev0):
    """Device sin/cos/exp/atan2/acos/atan/sqrt/div == oracle, bit for bit, on the argument ranges the shaders use."""
    rng = np.random.default_rng(1234)
    dev = twk.Device(ordinal=dev0)
    n = 1 << 20
    cases = {
        0: rng.uniform(-7.0, 13.0, n), 1: rng.uniform(-7.0, 13.0, n), 2: -rng.exponential(8.0, n),
        4: rng.uniform(-1.0, 1.0, n), 5: rng.uniform(-50.0, 50.0, n), 6: rng.uniform(0.0, 1e6, n), 7: rng.uniform(-1e3, 1e3, n),
    }
    for op, x in cases.items():
        x = x.astype(np.float32)
        assert np.array_equal(_bits(dev.debugMath(op, x)), _bits(orc.oracle_math(op, x))), f"math op {op} differs"
    x = rng.uniform(-2, 2, n).astype(np.float32)
    y = rng.uniform(-2, 2, n).astype(np.float32)
    x[:4] = [0, 0, 1, -1]
    y[:4] = [0, 1, 0, 0]
    assert np.array_equal(_bits(dev.debugMath(3, x, y)), _bits(orc.oracle_math(3, x, y)))
    dev.close()


def _render_both(twk, orc, system, scene, res, iterations, capture=False):
    app = load_app(twk, system, scene, res)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    if capture:
        dev.debugCapture(True)
    for it in range(iterations):
        dev.render(it)
    dev.synchronizeStream()
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    if capture:
        ref.captureFirstHits(True)
    for it in range(iterations):
        ref.render(it)
    return app, dev, ref


def test_first_hits_match_oracle_c1(twk, orc):
    """Stage tap: first-bounce hit records (t, beta, gamma, instance, primitive) of C1 at 128x128."""
    app, dev, ref = _render_both(twk, orc, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (128, 128), 1, capture=True)
    g_tbg, g_ids = dev.debugReadFirstHits()
    o_tbg, o_ids = ref.readFirstHits()
    assert np.array_equal(g_ids[:, 0], o_ids[:, 0]), "instance ids differ"
    hit = o_ids[:, 0] >= 0
    assert hit.any() and (~hit).any() is not None
    assert np.array_equal(g_ids[hit, 1], o_ids[hit, 1]), "primitive ids differ"
    assert np.array_equal(_bits(g_tbg[hit]), _bits(o_tbg[hit])), "t/beta/gamma differ"
    dev.close()


@pytest.mark.parametrize("system,scene,res,iters", [
    ("system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (128, 128), 2),   # C1 Lambert only
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90), 3),          # C2 full BSDF set
])
def test_image_bit_identical(twk, orc, system, scene, res, iters):
    app, dev, ref = _render_both(twk, orc, system, scene, res, iters)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert gpu.shape == cpu.shape == (res[1], res[0], 4)
    assert np.isfinite(cpu).all() and cpu[..., :3].max() > 0
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{mism} of {res[0] * res[1]} pixels differ; max |diff| {np.abs(gpu - cpu).max()}"
    dev.close()


def test_trace_rays_vs_oracle_brute_force(twk, orc):
    """optixTrace contract: random rays through the device BVH == brute force over every triangle (closest + any hit)."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (32, 32))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    ref.setTraceMode(False)
    rng = np.random.default_rng(7)
    n = 3000
    o = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(0.05, 1.9, n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50, 0] = 0.0  # axis-parallel components
    d[50:100, 1] = 0.0
    rays = np.concatenate([o, np.full((n, 1), 5e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], axis=1).astype(np.float32)
    g_tbg, g_ids = dev.traceRays(rays)
    o_tbg, o_ids = ref.traceRays(rays)
    assert np.array_equal(g_ids, o_ids)
    hit = o_ids[:, 0] >= 0
    assert hit.mean() > 0.6
    assert np.array_equal(_bits(g_tbg[hit]), _bits(o_tbg[hit]))
    rays[:, 7] = rng.uniform(0.1, 3.0, n).astype(np.float32)
    g_tbg, g_ids = dev.traceRays(rays, anyHit=True)
    o_tbg, o_ids = ref.traceRays(rays, anyHit=True)
    assert np.array_equal(g_ids[:, 0], o_ids[:, 0])
    dev.close()


def test_crop_parity_at_full_size(twk, orc):
    """BASELINE config C2 at its full 1920x1080: the oracle renders only a 96x64 window (pixels are independent),
    the GPU renders the whole frame; the window must be bit-identical."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt")
    assert list(app.info.resolution) == [1920, 1080]
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for it in range(2):
        dev.render(it)
    dev.synchronizeStream()
    gpu = dev.getOutputBufferHost()
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    x0, y0, x1, y1 = 1000, 300, 1096, 364
    for it in range(2):
        ref.render(it, rect=(x0, y0, x1, y1))
    cpu = ref.getOutputBufferHost()
    assert np.array_equal(_bits(gpu[y0:y1, x0:x1]), _bits(cpu[y0:y1, x0:x1]))
    assert np.isfinite(gpu).all()
    dev.close()


def test_launch_is_deterministic_and_restartable(twk):
    """Idempotence: rendering iterations 0..3 twice gives the same bits; iteration 0 overwrites the accumulator
    (raygeneration.cu:246-253), so a restart needs no clear."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (256, 144))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    imgs = []
    for _ in range(2):
        for it in range(4):
            dev.render(it)
        dev.synchronizeStream()
        imgs.append(dev.getOutputBufferHost())
    assert np.array_equal(_bits(imgs[0]), _bits(imgs[1]))
    dev.close()


def test_launch_batching_does_not_change_the_image(twk):
    """twk_launch is deferred: 1, 3 or 4 iterations per wavefront pass, a non-consecutive iteration index in between
    and a restart must give the same bits as one pass per iteration."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (192, 108))
    imgs = []
    for batch in (1, 3, 4):
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        app.initDevice(dev)
        dev.setLaunchBatch(batch)
        for it in (0, 1, 2, 3, 4, 5, 6):
            dev.render(it)
        dev.render(0)  # restart: iteration 0 overwrites
        dev.render(1)
        dev.render(7)  # gap in the iteration index: folded with weight 1/8 exactly as the reference would
        imgs.append(dev.getOutputBufferHost())
        dev.close()
    assert np.array_equal(_bits(imgs[0]), _bits(imgs[1])) and np.array_equal(_bits(imgs[0]), _bits(imgs[2]))


def test_tiled_equals_single_device(twk):
    """Tile-interleaved distribution (raygeneration.cu:152-164): N device handles on one GPU, each renders its
    checkerboard share into a packed launchWidth x H buffer; compositing them equals the single-device image."""
    res = (200, 120)  # not a multiple of tile * N: exercises the out-of-image tile columns
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", res)
    single = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(single)
    for it in range(2):
        single.render(it)
    single.synchronizeStream()
    full = single.getOutputBufferHost()
    single.close()
    for n in (2, 3):
        out = np.zeros_like(full)
        for i in range(n):
            d = twk.Device(ordinal=0, index=i, count=n, miss=app.info.miss)
            app.initDevice(d, distribution=1)
            for it in range(2):
                d.render(it)
            d.synchronizeStream()
            tile = d.getOutputBufferHost()
            lw = d.launchWidth
            assert lw == twk.launch_width(res[0], 8, n)
            for y in range(res[1]):
                for x in range(lw):
                    px = twk.tile_column(x, y, (8, 8), n, i)
                    if px < res[0]:
                        out[y, px] = tile[y, x]
            d.close()
        assert np.array_equal(_bits(out), _bits(full)), f"tiled N={n} differs from single device"


def test_shared_frame_strategies_equal_single_device(twk):
    """twk_set_shared_frame (≙ the ZeroCopy / PeerAccess strategies): N handles accumulate straight into ONE W x H frame
    at the pixels their launch indices map to; the frame equals the single-device image bit for bit. The frame here is
    device memory of the one GPU all handles share."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    res = (200, 120)  # not a multiple of 8 * N: the last tile column is partly outside the image
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", res)
    single = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(single)
    for it in range(3):
        single.render(it)
    full = single.getOutputBufferHost()
    single.close()
    for n in (2, 3):
        nbytes = res[0] * res[1] * 16
        frame = C.c_void_p()
        assert hip.hipMalloc(C.byref(frame), C.c_size_t(nbytes)) == 0
        assert hip.hipMemset(frame, 0, C.c_size_t(nbytes)) == 0
        handles = []
        for i in range(n):
            d = twk.Device(ordinal=0, index=i, count=n, miss=app.info.miss)
            app.initDevice(d, distribution=1)
            d.setSharedFrame(frame.value, nbytes)
            handles.append(d)
        for it in range(3):
            for d in handles:
                d.render(it)
        for d in handles:
            d.synchronizeStream()
        host = np.empty((res[1], res[0], 4), np.float32)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), frame, C.c_size_t(nbytes), 2) == 0  # hipMemcpyDeviceToHost
        assert np.array_equal(_bits(host), _bits(full)), f"shared frame of {n} handles differs from the single-device image"
        assert np.array_equal(_bits(handles[0].getOutputBufferHost()), _bits(full))
        for d in handles:
            d.close()
        assert hip.hipFree(frame) == 0


def test_error_paths(twk):
    """Error behaviour of the C ABI: invalid calls return codes + message, never crash."""
    dev = twk.Device(ordinal=0)
    with pytest.raises(twk.TwkError):
        dev.render(0)  # no state, no scene
    st = twk.DeviceState()
    st.resolution[0], st.resolution[1] = 8, 8
    st.tileSize[0], st.tileSize[1] = 6, 8
    with pytest.raises(twk.TwkError):
        dev.setState(st)  # tile size not a power of two
    with pytest.raises(twk.TwkError):
        dev.addGeometry(np.zeros((3, 12), np.float32), np.array([0, 1, 5], np.uint32))  # index out of range
    with pytest.raises(twk.TwkError):
        dev.build()  # empty scene
    with pytest.raises(twk.TwkError):
        twk.Device(ordinal=99)
    dev.close()


def test_fewer_materials_or_lights_than_the_built_scene_uses_is_refused(twk):
    """A built scene's instances hold material / light indices; re-initialising either table with fewer entries would
    leave them dangling (an out-of-bounds read in the shade kernel). Refused with INVALID_STATE, state unchanged."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (32, 32))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.render(0)
    before = dev.getOutputBufferHost().copy()
    mats, lights = app.materials, app.lights
    assert len(mats) > 1 and len(lights) == 1
    with pytest.raises(twk.TwkError) as e:
        dev.initMaterials(mats[:1])
    assert e.value.code == 4  # TWK_ERROR_INVALID_STATE
    with pytest.raises(twk.TwkError) as e:
        dev.initLights([])
    assert e.value.code == 4  # TWK_ERROR_INVALID_STATE
    dev.initMaterials(mats)      # same count: fine
    dev.initLights(lights)
    dev.render(0)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(before))
    dev.clearScene()             # after the scene is gone the tables may shrink
    dev.initLights([])
    dev.initMaterials(mats[:1])
    dev.close()


def test_pass_is_split_when_its_streams_do_not_fit(twk, monkeypatch):
    """A pass whose path streams exceed the memory it may take (here a 6 MiB budget: the real limit is the device's
    HBM) is cut in halves until it fits; the image is the one of the undivided pass, and a frame that does not fit
    even for one iteration reports out-of-memory instead of crashing."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (96, 54))
    imgs = []
    for budget in (None, "6"):  # 96*54 px * 280 B = 1.4 MiB per iteration: 16 iterations need 22 MiB, 4 fit in 6 MiB
        if budget is None:
            monkeypatch.delenv("TWK_STREAM_BUDGET_MB", raising=False)
        else:
            monkeypatch.setenv("TWK_STREAM_BUDGET_MB", budget)
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        app.initDevice(dev)
        for it in range(16):
            dev.render(it)
        imgs.append(dev.getOutputBufferHost())
        dev.close()
    assert np.array_equal(_bits(imgs[0]), _bits(imgs[1]))
    monkeypatch.setenv("TWK_STREAM_BUDGET_MB", "1")
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    with pytest.raises(twk.TwkError, match="TWK_STREAM_BUDGET_MB"):
        dev.render(0)                # recorded; reported by the call that runs the pass
        dev.synchronizeStream()
    dev.close()


@pytest.mark.parametrize("scene_file,two_level", [("scene_rtigo3_cornell_box.txt", False), ("scene_rtigo3_instances.txt", True)])
def test_same_bvh_host_walker_hits_and_visit_counts(twk, orc, monkeypatch, scene_file, two_level):
    """(Primary rays start at their tile's entry points since round 3 and visit fewer nodes than a walk from the root: the
    walker's counts are compared with the kernel's with that shortcut off, TWK_TILE_ENTRIES=0; the hit records with it on
    are compared in tests/test_gpu_pass_variants.py.)
    The single-threaded host walker of the test tooling (oracle/same_bvh_walk.cpp) walks the tree the DEVICE built
    (twk_debug_read_acceleration) with the persistent kernel's per-ray algorithm: its hit records equal the device's
    bit for bit and its visit counts equal the counting kernel's (counts from the CPU running the same
    BVH on the same rays). Primary rays only: no lights, black miss, one segment."""
    from conftest import scene_path
    system = "\n".join(["resolution 64 40", "tileSize 8 8", "samplesSqrt 1", "miss 0", "light 0", "pathLengths 1 1", "epsilonFactor 500",
                        "lensShader 0", "center 0 1 0", "camera 0.75 0.5 45 3.41" if not two_level else "camera 0.75 0.55 50 14"]) + "\n"
    app = twk.Application(system_text=system, scene_text=open(scene_path(scene_file)).read())
    monkeypatch.setenv("TWK_TILE_ENTRIES", "0")
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    acc = dev.readAcceleration()
    assert bool(acc[0]["twoLevel"]) == two_level and acc[0]["nodeFloats"] == 16
    dev.debugCapture(True)
    dev.statsEnable(True)
    dev.statsGet(reset=True)
    dev.render(0)
    st = dev.statsGet(reset=True)
    g_tbg, g_ids = dev.debugReadFirstHits()
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    w, h = 64, 40
    rays = np.stack([ref.debugPath(0, x, y)[0][:8] for y in range(h) for x in range(w)])
    assert st["radianceRays"] == w * h and st["shadowRays"] == 0
    c_tbg, c_ids, counts = orc.walk_same_bvh(acc, rays)
    assert np.array_equal(c_ids, g_ids)
    hit = c_ids[:, 0] >= 0
    assert 0.3 < hit.mean() <= 1.0
    assert np.array_equal(_bits(c_tbg[hit]), _bits(g_tbg[hit]))
    # v_rcp_f32 (device) vs 1.0f / d (host) in the culling test may move a borderline box decision: a few visits in a million
    assert abs(int(counts["nodesVisited"]) - int(st["nodesVisited"])) <= max(2, st["nodesVisited"] // 2000), (counts, st["nodesVisited"])
    assert abs(int(counts["trianglesTested"]) - int(st["trianglesTested"])) <= max(2, st["trianglesTested"] // 2000), (counts, st["trianglesTested"])
    assert int(counts["instancesEntered"]) == int(st["instancesEntered"]) or abs(int(counts["instancesEntered"]) - int(st["instancesEntered"])) <= 2
    if two_level:
        assert counts["instancesEntered"] > 0
    # and against brute force through the oracle
    o_tbg, o_ids = ref.traceRays(rays)
    assert np.array_equal(o_ids, c_ids) and np.array_equal(_bits(o_tbg[hit]), _bits(c_tbg[hit]))
    dev.close()
