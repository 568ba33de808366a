"""-m gpu: the build- and pass-level shortcuts of round 3 change no bit of any image.

  * small flattened instances as direct leaves of the top level (device_api.hip twk_build)
  * wide-node cuts chosen by expected node visits (bvh_build.hip refitKernel)
  * seven against six resident blocks per CU of the traversal kernel (device_types.h TWK_TRACE_WAVES7)
  * primary rays computed by the first traversal / shade launch instead of written by generateKernel
    (shade_kernels.hip "primary rays"), with and without a tile distribution that leaves launch indices inactive
  * the instance / material / light records read from LDS copies by shadeKernel, and from the scene's arrays when
    the tables do not fit (a scene of 200 instances)
  * the root as two wide nodes of up to eight entries (bvh_build.hip wideRootKernel)
  * primary rays starting at their tile's entry points instead of at the root (trace_kernels.hip tileEntryKernel):
    first-hit records and images, frames that are no multiple of the tile, cameras close to and inside geometry,
    a camera moved between launches

Each knob is an environment variable read by twk_device_create; the default (everything on) is what every other
parity test runs against the oracle, so "equal to the default" here means "equal to the oracle" there."""
import numpy as np
import pytest

from conftest import load_app, scene_path

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _render(twk, app, iterations, index=0, count=1, batch=None):
    dev = twk.Device(ordinal=0, index=index, count=count, miss=app.info.miss)
    app.initDevice(dev, distribution=1 if count > 1 else None)
    if batch is not None:
        dev.setLaunchBatch(batch)
    for it in range(iterations):
        dev.render(it)
    out = dev.getOutputBufferHost().copy()
    info = dev.buildInfo()
    dev.close()
    return out, info


KNOBS = [("TWK_DIRECT_SMALL_LEAVES", "0"), ("TWK_COSTED_CUTS", "0"), ("TWK_TRACE_WAVES_RUNTIME", "6"), ("TWK_FUSED_PRIMARY", "0"), ("TWK_TILE_ENTRIES", "0"), ("TWK_WIDE_ROOT", "0"), ("TWK_SHADE_SORT", "0"), ("TWK_SHADE_SORT", "2")]


@pytest.mark.parametrize("system,scene,res", [
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90)),
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (128, 72)),
    ("system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", (128, 72)),
])
def test_shortcuts_change_no_bit(twk, monkeypatch, system, scene, res):
    app = load_app(twk, system, scene, res)
    base, info = _render(twk, app, 3)
    assert np.isfinite(base).all() and base[..., :3].max() > 0.2
    for name, value in KNOBS:
        monkeypatch.setenv(name, value)
        other, other_info = _render(twk, app, 3)
        monkeypatch.delenv(name)
        assert np.array_equal(_bits(other), _bits(base)), f"{name}={value}: {(_bits(other) != _bits(base)).any(axis=2).sum()} pixels differ"
        if name == "TWK_DIRECT_SMALL_LEAVES":
            assert other_info["directLeafInstances"] == 0 and info["directLeafInstances"] > 0
        if name == "TWK_TRACE_WAVES_RUNTIME":
            assert other_info["traceBlocksPerCU"] == 6


def test_wide_root_is_built_where_it_pays(twk, monkeypatch):
    """Cornell box: the root's two large children ({floor + spheres}, {three walls}) are entered by nearly every ray — the
    root becomes two nodes of eight entries together; with the switch off, or a scene of a single tree, one node."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (32, 32))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    info, nodes, tris, inst = dev.readAcceleration()
    dev.close()
    assert info["root2"] == info["root"] + 1 and info["numNodes"] == info["root2"] + 1
    used = 0
    for index in (info["root"], info["root2"]):
        q = nodes[index][6:12].view(np.uint32)
        used += sum(1 for k in range(4) if ((q[0] >> (8 * k)) & 0xff) <= ((q[3] >> (8 * k)) & 0xff))
    assert used == 8
    monkeypatch.setenv("TWK_WIDE_ROOT", "0")
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    info0 = dev.readAcceleration()[0]
    dev.close()
    assert info0["root2"] == -1 and info0["numNodes"] == info["numNodes"] - 2


def test_build_info_names_the_direct_leaves_and_the_kernel_build(twk):
    """Cornell box: five walls + the area light are two-triangle instances, all flattened -> the seven-block build;
    the instances scene enters instances -> six blocks (intro_07, flattened with cutout opacity: seven since round 4)."""
    for system, scene, leaves, blocks in [("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", 6, 7),
                                          ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", 1, 6)]:
        app = load_app(twk, system, scene, (32, 32))
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        app.initDevice(dev)
        info = dev.buildInfo()
        assert info["directLeafInstances"] == leaves, info
        assert info["traceBlocksPerCU"] == blocks, info
        dev.close()


@pytest.mark.parametrize("count,index,tile", [(3, 1, 8), (8, 5, 8), (2, 1, 16), (3, 0, 4)])
def test_fused_primary_rays_with_inactive_launch_indices(twk, monkeypatch, count, index, tile):
    """A device of a tile distribution whose launch width reaches beyond the image: the launch indices without a pixel
    are skipped by the fused path exactly as generateKernel marks them (weight 0 in the running mean), over several
    iterations and one pass per iteration as well as one pass for all. With distribution tiles of 8 and 16 the primary
    rays also start at their entry points (an entry tile is one pixel square of a distribution tile); with tiles of 4 the
    lists are off."""
    import re
    system = re.sub(r"(?m)^tileSize .*$", f"tileSize {tile} {tile}", open(scene_path("system_rtigo3_cornell_box.txt")).read())
    system = re.sub(r"(?m)^resolution .*$", "resolution 200 64", system)  # 25 tiles of 8 over 3 / 8 devices: ragged
    app = twk.Application(system_text=system, scene_text=open(scene_path("scene_rtigo3_cornell_box.txt")).read())
    results = []
    for fused in ("1", "0"):
        for batch in (1, 4):
            monkeypatch.setenv("TWK_FUSED_PRIMARY", fused)
            out, _ = _render(twk, app, 4, index=index, count=count, batch=batch)
            results.append(out)
    monkeypatch.delenv("TWK_FUSED_PRIMARY")
    assert np.isfinite(results[0]).all() and results[0][..., :3].max() > 0.2
    for other in results[1:]:
        assert np.array_equal(_bits(other), _bits(results[0]))


def _many_instances_scene(n_side):
    lines = ["albedo 0.7 0.7 0.7", "material default brdf_diffuse", "albedo 0.5 0.5 0.5", "material floor brdf_diffuse"]
    for k in range(6):
        lines += [f"albedo {0.3 + 0.1 * k:.2f} {0.9 - 0.1 * k:.2f} 0.4", f"material m{k} " + ("brdf_diffuse" if k % 2 == 0 else "brdf_specular")]
    lines += ["push", "scale 12 1 12", "model plane 1 1 1 floor", "pop"]
    for j in range(n_side):
        for i in range(n_side):
            lines += ["push", "scale 0.3 0.3 0.3", f"translate {1.2 * i - 0.6 * (n_side - 1):.3f} 0.3 {1.2 * j - 0.6 * (n_side - 1):.3f}",
                      ("model sphere 24 12 1.0 " if (i + j) % 2 else "model box ") + f"m{(i * 7 + j) % 6}", "pop"]
    return "\n".join(lines) + "\n"


def test_tables_beyond_the_lds_budget_use_the_scene_arrays(twk, orc):
    """197 instances x 128 B + materials + lights > 20 KiB: shadeKernel reads the records from the scene's arrays (the
    <.., LDS_TABLES = false> builds); 26 instances: from LDS. Both bit-identical to the oracle."""
    system = "\n".join(["resolution 96 54", "tileSize 8 8", "samplesSqrt 1", "miss 1", "light 0", "pathLengths 2 4", "epsilonFactor 500",
                        "lensShader 0", "center 0 0.5 0", "camera 0.75 0.6 50 16"]) + "\n"
    for n_side, fits in ((14, False), (5, True)):
        app = twk.Application(system_text=system, scene_text=_many_instances_scene(n_side))
        assert (len(app.instances) * 128 + 64 * 16 <= 20480) == fits
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        ref = orc.Oracle(miss=app.info.miss)
        app.initDevice(dev)
        ref.loadApplication(app)
        for it in range(2):
            dev.render(it)
            ref.render(it)
        gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
        assert cpu[..., :3].max() > 0.3 and np.isfinite(cpu).all()
        mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
        assert mism == 0, f"{n_side}x{n_side}: {mism} pixels differ"
        dev.close()


def _first_hits(twk, app, iterations=1):
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.debugCapture(True)
    for it in range(iterations):
        dev.render(it)
    tbg, ids = dev.debugReadFirstHits()
    out = dev.getOutputBufferHost().copy()
    dev.close()
    return tbg.copy(), ids.copy(), out


@pytest.mark.parametrize("scene,camera,res", [
    ("scene_rtigo3_cornell_box.txt", "camera 0.75 0.5 45 3.41", (203, 117)),    # no multiple of 8 either way
    ("scene_rtigo3_cornell_box.txt", "camera 0.1 0.5 100 0.9", (160, 96)),      # wide angle from inside the room, between the spheres
    ("scene_rtigo3_cornell_box.txt", "camera 0.75 0.93 60 1.5", (96, 160)),     # looking down, portrait frame
    ("scene_rtigo3_geometry.txt", "camera 0.8 0.45 60 9", (192, 108)),          # open scene: tiles of sky only
    ("scene_rtigo3_instances.txt", "camera 0.75 0.55 50 14", (192, 108)),       # two-level: instance leaves in the lists
    ("scene_rtigo3_instances.txt", "camera 0.3 0.52 90 1.2", (128, 128)),       # inside the grid of instances
])
def test_tile_entry_points_change_no_first_hit(twk, monkeypatch, scene, camera, res):
    """The first hit of every pixel (distance, barycentrics, primitive, instance) and the image with the primary rays
    starting at their tile's entry points equal those with every ray starting at the root, bit for bit."""
    system = "\n".join([f"resolution {res[0]} {res[1]}", "tileSize 8 8", "samplesSqrt 1", "miss 1", "light 0", "pathLengths 2 3", "epsilonFactor 500",
                        "lensShader 0", "center 0 1 0", camera]) + "\n"
    app = twk.Application(system_text=system, scene_text=open(scene_path(scene)).read())
    on = _first_hits(twk, app, 2)
    monkeypatch.setenv("TWK_TILE_ENTRIES", "0")
    off = _first_hits(twk, app, 2)
    monkeypatch.delenv("TWK_TILE_ENTRIES")
    assert np.array_equal(on[1], off[1]), f"{(on[1] != off[1]).any(axis=1).sum()} first hits differ"
    assert np.array_equal(_bits(on[0]), _bits(off[0]))
    assert np.array_equal(_bits(on[2]), _bits(off[2]))
    assert (on[1][:, 0] >= 0).mean() > 0.3


def test_tile_entry_points_follow_the_camera(twk, monkeypatch):
    """The lists are rebuilt when the camera moves between launches (and only then): every view equals the one rendered
    with the lists off."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90))
    views = [(0.75, 0.5, 45.0, 3.41), (0.6, 0.4, 60.0, 2.5), (0.9, 0.7, 30.0, 3.0)]

    def run():
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        app.initDevice(dev)
        outs = []
        for phi, theta, fov, distance in views:
            dev.updateCamera(0, twk.camera_frustum((0.0, 1.0, 0.0), phi, theta, fov, distance, 160 / 90))
            for it in range(2):
                dev.render(it)
            outs.append(dev.getOutputBufferHost().copy())
        dev.close()
        return outs

    on = run()
    monkeypatch.setenv("TWK_TILE_ENTRIES", "0")
    off = run()
    monkeypatch.delenv("TWK_TILE_ENTRIES")
    for a, b in zip(on, off):
        assert np.array_equal(_bits(a), _bits(b))
    assert not np.array_equal(_bits(on[0]), _bits(on[1]))


def test_profiled_kernel_times_do_not_exceed_the_wall_time_of_a_two_lane_pass(twk):
    """passes of at most TWK_LANES2_MAX_PATHS paths run as two lanes whose launches overlap in time; the
    per-kind sums of twk_profile_get are sums of launch durations, so with overlapping lanes they came out at about twice
    the wall time (bench.py feeds them into the roofline). A profiled pass runs as one lane: the sum of all kinds stays
    within the wall time of the same passes, and the image does not depend on it."""
    import time
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (960, 540))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.setLaunchBatch(8)           # 4.1 M paths per pass: two lanes by default
    for it in range(8):
        dev.render(it)
    dev.synchronizeStream()
    plain = dev.getOutputBufferHost().copy()
    dev.profileEnable(True)
    dev.profileReset()
    t0 = time.perf_counter()
    for it in range(8, 24):
        dev.render(it)
    dev.synchronizeStream()
    wall_ms = (time.perf_counter() - t0) * 1.0e3
    prof = dev.profileGet()
    dev.profileEnable(False)
    total = sum(v["ms"] for v in prof.values())
    assert prof["trace"]["launches"] == 2 * 11 and prof["shade"]["launches"] == 2 * 10, "one lane: one launch per kind and depth and pass"
    assert 0.0 < total <= wall_ms, f"kernel times {total:.3f} ms of two passes exceed their wall time {wall_ms:.3f} ms"
    # the same 24 iterations without the profile (two lanes): the same image
    other = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(other)
    other.setLaunchBatch(8)
    for it in range(24):
        other.render(it)
    assert np.array_equal(_bits(other.getOutputBufferHost()), _bits(dev.getOutputBufferHost()))
    assert not np.array_equal(_bits(plain), _bits(dev.getOutputBufferHost()))
    other.close()
    dev.close()
