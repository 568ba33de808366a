"""-m gpu: parity of the remaining reference programs and configurations against the oracle, all bit-identical:
  C4   geometry scene (constant environment miss 1 + env light sampling, GGX BRDF, rough-glass BSDF, torus, box,
       open half sphere) and the instanced scene (101 instances of 5 geometries, two-level BVH)
  C3   intro_07 scene: spherical HDR environment (miss 2: importance sampling through the CDFs of
       Texture::calculateSphericalCDF, miss_env_sphere MIS), albedo texture (tex2D bilinear), nested dielectrics,
       stochastic cutout opacity on radiance and shadow rays
  lens fisheye and sphere lens shaders
  C5   the 3840x2160 frame tiled over 8 device indices (launchWidth 480): crop parity of one tile set
  compositor kernel
"""
import sys

import numpy as np
import pytest

from conftest import load_app, scene_path
from procedural import albedo_checker, cutout_slots, environment_hdr

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _both(twk, orc, app, iterations, textures=(), material_edit=None, state_edit=None, index=0, count=1, rect=None):
    dev = twk.Device(ordinal=0, index=index, count=count, miss=app.info.miss)
    ref = orc.Oracle(index=index, count=count, miss=app.info.miss)
    for slot, img in textures:
        dev.initTexture(slot, img)
        ref.initTexture(slot, img)
    app.initDevice(dev, distribution=1 if count > 1 else None)
    st = app.state
    if count > 1:
        st.distribution = 1
    if state_edit:
        state_edit(st)
        dev.setState(st)
    ref.loadApplication(app, state=st)
    if material_edit:
        mats = app.materials
        material_edit(mats)
        dev.initMaterials(mats)
        ref.initMaterials(mats)
    for it in range(iterations):
        dev.render(it)
        ref.render(it, rect=rect)
    return dev, ref


@pytest.mark.parametrize("system,scene,res,iters", [
    ("system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", (160, 90), 3),
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (128, 72), 2),
])
def test_c4_scenes_bit_identical(twk, orc, system, scene, res, iters):
    app = load_app(twk, system, scene, res)
    dev, ref = _both(twk, orc, app, iters)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert cpu[..., :3].max() > 0.5 and np.isfinite(cpu).all()
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{mism} pixels differ, max |diff| {np.abs(gpu - cpu).max()}"
    dev.close()


def test_instanced_scene_trace_rays_vs_brute_force(twk, orc):
    app = load_app(twk, "system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (16, 16))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)  # oracle BVH mode (brute force over 13 M triangle instances is too slow); BVH == brute is tested on CPU
    rng = np.random.default_rng(21)
    n = 4000
    o = np.stack([rng.uniform(-11, 11, n), rng.uniform(0.05, 3.0, n), rng.uniform(-11, 11, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, np.full((n, 1), 5e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], 1).astype(np.float32)
    g, o_ = dev.traceRays(rays), ref.traceRays(rays)
    assert np.array_equal(g[1], o_[1])
    hit = o_[1][:, 0] >= 0
    assert hit.mean() > 0.4 and np.array_equal(_bits(g[0][hit]), _bits(o_[0][hit]))
    dev.close()


def test_environment_map_and_albedo_texture(twk, orc):
    """miss 2: lat-long HDR environment as light 0 (CDF importance sampling + MIS on implicit hits), albedo texture on
    the floor, no area light."""
    system = open(scene_path("system_rtigo3_geometry.txt")).read().replace("miss 1", "miss 2").replace("resolution 1920 1080", "resolution 160 90")
    system += "\nenvRotation 0.15\n"
    app = twk.Application(system_text=system, scene_text=open(scene_path("scene_rtigo3_geometry.txt")).read())
    assert app.info.miss == 2 and app.info.numLights == 1

    def use_texture(mats):
        mats[1].useAlbedoTexture = 1  # 'floor'
        mats[2].useAlbedoTexture = 1  # 'redbox'

    dev, ref = _both(twk, orc, app, 3, textures=((twk_slot(0), albedo_checker()), (twk_slot(2), environment_hdr())), material_edit=use_texture)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert cpu[..., :3].max() > 1.0 and np.isfinite(cpu).all()
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{mism} pixels differ, max |diff| {np.abs(gpu - cpu).max()}"
    dev.close()


def twk_slot(i):
    return i  # TWK_TEXTURE_ALBEDO = 0, TWK_TEXTURE_CUTOUT = 1, TWK_TEXTURE_ENVIRONMENT = 2


@pytest.mark.parametrize("lens", [1, 2])
def test_lens_shaders(twk, orc, lens):
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (128, 72))

    def edit(st):
        st.lensShader = lens

    dev, ref = _both(twk, orc, app, 2, state_edit=edit)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert np.array_equal(_bits(gpu), _bits(cpu))
    dev.close()


def _intro07(twk, res):
    app = load_app(twk, "system_intro_07.txt", "scene_intro_07.txt", res)
    names = {"floor": 1, "cutout": 4}  # material 0 is the area light's (Application.cpp:640-659), then file order

    def edit(mats):
        assert mats[names["floor"]].indexBSDF == 0 and mats[names["cutout"]].thinwalled == 1
        mats[names["floor"]].useAlbedoTexture = 1
        mats[names["cutout"]].useCutoutTexture = 1

    textures = ((0, albedo_checker()), (1, cutout_slots()), (2, environment_hdr()))
    return app, edit, textures


def test_c3_intro07_with_cutout_opacity(twk, orc):
    """Config C3: intro_07 scene — albedo texture, spherical HDR environment + 4x4 area light (two lights: the light
    pick draws rng), nested dielectrics (glass sphere inside the water box), mirror torus and a sphere with stochastic
    cutout opacity on radiance AND shadow rays. Bit-identical to the oracle (which defines the candidate order)."""
    app, edit, textures = _intro07(twk, (160, 90))
    assert app.info.numLights == 2 and app.info.numInstances == 6
    dev, ref = _both(twk, orc, app, 3, textures=textures, material_edit=edit)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert np.isfinite(cpu).all() and cpu[..., :3].max() > 1.0
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{mism} pixels differ, max |diff| {np.abs(gpu - cpu).max()}"
    # the cutout really cuts: with the texture off the image differs
    dev2, _ = _both(twk, orc, app, 0, textures=textures)
    for it in range(3):
        dev2.render(it)
    assert not np.array_equal(_bits(dev2.getOutputBufferHost()), _bits(gpu))
    dev.close()
    dev2.close()


def test_c3_full_size_crop(twk, orc):
    """C3 at its full 1920x1080: crop parity of a window that straddles the cutout sphere."""
    app, edit, textures = _intro07(twk, None)
    x0, y0, x1, y1 = 930, 500, 994, 540
    dev, ref = _both(twk, orc, app, 2, textures=textures, material_edit=edit, rect=(x0, y0, x1, y1))
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert np.array_equal(_bits(gpu[y0:y1, x0:x1]), _bits(cpu[y0:y1, x0:x1]))
    dev.close()


def test_missing_texture_is_an_error(twk):
    app, edit, textures = _intro07(twk, (32, 18))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    dev.initTexture(2, environment_hdr())
    app.initDevice(dev)
    mats = app.materials
    edit(mats)
    dev.initMaterials(mats)
    with pytest.raises(twk.TwkError) as e:
        dev.render(0)
    assert "texture" in str(e.value)
    dev.close()


def test_c5_tile_set_crop_parity(twk, orc):
    """Config C5: 3840x2160 tiled over 8 devices. Device index 3 of 8 renders its 480x2160 packed share on the GPU;
    the oracle renders a window of the same share."""
    app = load_app(twk, "system_rtigo3_cornell_box_c5.txt", "scene_rtigo3_cornell_box.txt")
    assert list(app.info.resolution) == [3840, 2160] and app.info.strategy == 3
    x0, y0, x1, y1 = 200, 1000, 264, 1048
    dev, ref = _both(twk, orc, app, 2, index=3, count=8, rect=(x0, y0, x1, y1))
    assert dev.launchWidth == 480 == ref.launchWidth
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert gpu.shape == (2160, 480, 4)
    assert np.array_equal(_bits(gpu[y0:y1, x0:x1]), _bits(cpu[y0:y1, x0:x1]))
    dev.close()


def test_compositor_kernel(twk, orc):
    """twk_compositor scatters the gathered [N][H][launchWidth] tile sets into the W x H image like compositor.cu:38-64.
    Device buffers come straight from the HIP runtime the library itself links (ctypes), no torch involved."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    w, h, n = 200, 48, 3
    st = twk.DeviceState()
    st.resolution[0], st.resolution[1] = w, h
    st.tileSize[0], st.tileSize[1] = 8, 8
    st.pathLengths[0], st.pathLengths[1] = 2, 5
    st.distribution, st.samplesSqrt, st.epsilonFactor = 1, 1, 500.0
    dev = twk.Device(ordinal=0, index=0, count=n)
    dev.setState(st)
    lw = dev.launchWidth
    src = np.arange(n * h * lw * 4, dtype=np.float32).reshape(n, h, lw, 4)
    got = np.full((h, w, 4), -1.0, np.float32)
    d_tiles, d_out = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_tiles), C.c_size_t(src.nbytes)) == 0
    assert hip.hipMalloc(C.byref(d_out), C.c_size_t(got.nbytes)) == 0
    assert hip.hipMemcpy(d_tiles, src.ctypes.data_as(C.c_void_p), C.c_size_t(src.nbytes), 1) == 0
    assert hip.hipMemcpy(d_out, got.ctypes.data_as(C.c_void_p), C.c_size_t(got.nbytes), 1) == 0
    dev.compositor(d_tiles.value, d_out.value)
    dev.synchronizeStream()
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d_out, C.c_size_t(got.nbytes), 2) == 0
    hip.hipFree(d_tiles)
    hip.hipFree(d_out)
    expect = np.full((h, w, 4), -1.0, np.float32)
    for d in range(n):
        for y in range(h):
            for x in range(lw):
                px = twk.tile_column(x, y, (8, 8), n, d)
                if px < w:
                    expect[y, px] = src[d, y, x]
    assert np.array_equal(got, expect) and (got >= 0).all()
    assert np.array_equal(got, orc.oracle_compositor(src, w, (8, 8)))  # the oracle's statement-pinned restatement of compositor.cu
    dev.close()


def test_update_camera_light_material_between_launches(twk, orc):
    """≙ Device::updateCamera / updateLight / updateMaterial (Device.cpp:1083-1168), the interactive edit path: after
    an update the caller restarts at iteration 0 (Raytracer.cpp:331-338). Launches recorded BEFORE an update render
    with the old values (the update flushes them); the restarted accumulation equals a fresh device — and the oracle —
    set up with the new values."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (128, 72))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for it in range(3):
        dev.render(it)                       # still pending when the updates arrive
    before = None

    mats = app.materials
    mats[2].albedo[0], mats[2].albedo[1], mats[2].albedo[2] = 0.2, 0.4, 0.9   # repaint a wall
    mats[6].indexBSDF = 3                                                        # mirror sphere → GGX
    mats[6].roughness[0], mats[6].roughness[1] = 0.3, 0.15
    cam = twk.camera_frustum((0.1, 1.0, 0.0), 0.7, 0.55, 50.0, 3.2, 128 / 72)
    (light,) = app.lights
    light.emission[0], light.emission[1], light.emission[2] = 14.0, 11.0, 8.0

    dev.updateMaterial(2, mats[2])
    before = dev.getOutputBufferHost()       # the three old iterations were rendered with the OLD scene
    dev.updateMaterial(6, mats[6])
    dev.updateCamera(0, cam)
    dev.updateLight(0, light)
    for it in range(3):
        dev.render(it)                       # restart: iteration 0 overwrites
    after = dev.getOutputBufferHost()

    old = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(old)
    for it in range(3):
        old.render(it)
    assert np.array_equal(_bits(before), _bits(old.getOutputBufferHost()))
    old.close()

    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    ref.initMaterials(mats)
    ref.initCameras([cam])
    ref.initLights([light])
    for it in range(3):
        ref.render(it)
    cpu = ref.getOutputBufferHost()
    assert not np.array_equal(_bits(before), _bits(after))
    assert np.array_equal(_bits(after), _bits(cpu))
    with pytest.raises(twk.TwkError):
        dev.updateMaterial(99, mats[2])
    with pytest.raises(twk.TwkError):
        dev.updateCamera(5, cam)
    dev.close()


def test_scene_replacement_output_pointer_and_stream_probe(twk, orc):
    """twk_clear_scene + a second initScene on the same handle (≙ Device::initScene called again, Device.cpp:1058-1080)
    renders like a fresh handle; twk_get_output_device_pointer names the accumulation buffer; the stream probe used
    for the measured-bandwidth line of DESIGN.md returns a sane figure."""
    first = load_app(twk, "system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", (96, 54))
    second = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (96, 54))
    dev = twk.Device(ordinal=0, miss=second.info.miss)
    first.initDevice(dev)
    dev.render(0)
    dev.synchronizeStream()
    second.initDevice(dev)                    # clears the first scene, uploads and builds the second
    for it in range(2):
        dev.render(it)
    img = dev.getOutputBufferHost()
    ref = orc.Oracle(miss=second.info.miss)
    ref.loadApplication(second)
    for it in range(2):
        ref.render(it)
    assert np.array_equal(_bits(img), _bits(ref.getOutputBufferHost()))
    ptr, nbytes = dev.outputDevicePointer()
    assert ptr != 0 and nbytes == 96 * 54 * 16
    dev.clearScene()
    with pytest.raises(twk.TwkError):         # nothing to render until the scene is built again
        dev.render(0)
        dev.synchronizeStream()
    gbps = dev.streamPeakGBps(1 << 28, 3)
    assert 500.0 < gbps < 20000.0
    dev.close()


@pytest.mark.parametrize("policy", [(0, 0), (4, 2), (1 << 30, 1 << 30)])
@pytest.mark.parametrize("system,scene,res,iters", [
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (128, 72), 2),
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (96, 54), 2),
])
def test_flatten_policy_two_level_soup_and_mixed(twk, orc, system, scene, res, iters, policy):
    """twk_set_flatten_policy (include/tweeker_hip.h): (0, 0) pure two-level — every instance entered with a transformed
    ray —, the default (walls / light quads and rarely referenced geometries in the world-space soup, spliced into the
    top level next to the entered instances), and everything flattened (one single-level BVH). The oracle takes the
    same policy; images bit-identical in all three, and the instance-entry counter tells which path ran."""
    app = load_app(twk, system, scene, res)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    dev.setFlattenPolicy(*policy)
    app.initDevice(dev)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    ref.setFlattenPolicy(*policy)
    dev.statsEnable(True)
    dev.statsGet(reset=True)
    for it in range(iters):
        dev.render(it)
        ref.render(it)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    st = dev.statsGet(reset=True)
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"policy {policy}: {mism} pixels differ, max |diff| {np.abs(gpu - cpu).max()}"
    entered = st["instancesEntered"]
    if policy == (0, 0):
        assert entered > 0
    if policy[0] > 4 or "cornell" in scene and policy != (0, 0):
        assert entered == 0, "every instance of this scene is flattened under this policy: nothing may be entered"
    if "instances" in scene and policy == (4, 2):
        assert entered > 0, "the grid's shared meshes stay instanced under the default policy"
    dev.close()


def test_c3_optix7gui_light_rule_and_rtigo3_rule(twk, orc):
    """intro_07 is Optix7Gui's scene: scenes/system_intro_07.txt selects its closest-hit rule (a light ends the path on
    either side, black on the back face, apps/Optix7Gui/shaders/closesthit.cu:189-226; rtigo3 lets a back-face hit
    fall through to the light's BSDF, apps/rtigo3/shaders/closesthit.cu:192-222). Both rules against the oracle,
    bit-identical, on intro_07 (where no path reaches the light's back and its material is black anyway: same image)."""
    app, edit, textures = _intro07(twk, (128, 72))
    assert app.info.shaderVariant == 1
    images = {}
    for variant in (1, 0):
        dev, ref = _both(twk, orc, app, 0, textures=textures, material_edit=edit)
        dev.setShaderVariant(variant)
        ref.setShaderVariant(variant)
        for it in range(3):
            dev.render(it)
            ref.render(it)
        gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
        mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
        assert mism == 0, f"variant {variant}: {mism} pixels differ, max |diff| {np.abs(gpu - cpu).max()}"
        images[variant] = gpu
        dev.close()
    assert np.array_equal(_bits(images[0]), _bits(images[1]))


def test_light_back_face_rule_changes_the_image_when_the_light_material_is_not_black(twk, orc):
    """The two apps' rules differ observably once a light's material scatters: a camera above the 1x1 area light of
    `light 1` (facing down, y = 1.95) looks at its BACK. rtigo3 shades it with the light's BSDF (made diffuse here,
    lit by the white environment), Optix7Gui ends the path black. Each variant bit-identical to the oracle."""
    system = "\n".join(["resolution 96 64", "tileSize 8 8", "samplesSqrt 1", "miss 1", "light 1", "pathLengths 2 4", "epsilonFactor 500",
                        "lensShader 0", "center 0 1.95 0", "camera 0.75 0.9 45 4"]) + "\n"  # theta 0.9: the camera is above the light
    scene = "\n".join(["albedo 0.7 0.7 0.7", "material floor brdf_diffuse", "push", "scale 4 1 4", "model plane 1 1 1 floor", "pop"]) + "\n"
    app = twk.Application(system_text=system, scene_text=scene)
    light_material = [m for (_, _, m, l) in app.instances if l >= 0][0]

    def diffuse_light(mats):
        mats[light_material].indexBSDF = 0
        mats[light_material].albedo[0] = mats[light_material].albedo[1] = mats[light_material].albedo[2] = 0.8

    images = {}
    for variant in (0, 1):
        dev, ref = _both(twk, orc, app, 0, material_edit=diffuse_light)
        dev.setShaderVariant(variant)
        ref.setShaderVariant(variant)
        for it in range(2):
            dev.render(it)
            ref.render(it)
        gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
        assert np.array_equal(_bits(gpu), _bits(cpu)), f"variant {variant}"
        images[variant] = gpu
        dev.close()
    black1 = (images[1][..., :3].max(axis=2) == 0.0)
    lit0 = (images[0][..., :3].min(axis=2) > 0.05)
    assert (black1 & lit0).sum() > 50, "Optix7Gui: the light's back is black; rtigo3: it reflects the environment through the light's BSDF"


@pytest.mark.parametrize("scene", ["c3", "c2"])
def test_denoiser_aovs_match_oracle(twk, orc, scene):
    """Albedo and camera-space normal AOVs of Optix7Gui's integrator (raygeneration.cu:125-164,239-262): first diffuse /
    light event's throughput-attenuated albedo, primary-hit shading normal, accumulated like the radiance. Both buffers
    bit-identical to the oracle over four iterations; switching them on does not change the beauty image."""
    if scene == "c3":
        app, edit, textures = _intro07(twk, (128, 72))
    else:
        app, edit, textures = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (128, 72)), None, ()
    plain, _ = _both(twk, orc, app, 0, textures=textures, material_edit=edit)
    dev, ref = _both(twk, orc, app, 0, textures=textures, material_edit=edit)
    dev.enableAov(True)
    ref.enableAov(True)
    dev.setLaunchBatch(3)  # 3 + 1: the running means cross a pass boundary
    for it in range(4):
        plain.render(it)
        dev.render(it)
        ref.render(it)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(ref.getOutputBufferHost()))
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(plain.getOutputBufferHost()))
    for which, name in ((0, "albedo"), (1, "normal")):
        g, c = dev.readAov(which), ref.readAov(which)
        mism = (_bits(g) != _bits(c)).any(axis=2).sum()
        assert mism == 0, f"{name}: {mism} pixels differ, max |diff| {np.abs(g - c).max()}"
    albedo, normal = dev.readAov(0), dev.readAov(1)
    assert (albedo[..., :3] >= 0).all() and (albedo[..., :3] <= 1).all() and (albedo[..., 3] == 1).all()
    n = np.linalg.norm(normal[..., :3], axis=2)
    hit = n > 0
    assert hit.mean() > 0.5 and np.allclose(n[hit], 1.0, atol=1e-5) and (normal[..., 3] == 0).all()
    assert (normal[..., 2][hit] > -0.2).mean() > 0.95  # primary hits face the camera: +z in the right-handed camera space
    with pytest.raises(twk.TwkError):
        plain.readAov(0)  # not enabled on that handle
    dev.close()
    plain.close()


@pytest.mark.parametrize("system,scene,windows,iters", [
    # C4 geometry at its full 1920x1080: box / sphere silhouettes on the floor; rough-glass box next to the glass sphere
    ("system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", [(580, 440, 644, 488), (900, 300, 964, 348), (1300, 420, 1364, 468)], 3),
    # C4 instances at 1920x1080 (101 instances, two-level BVH under the default policy): the busiest window of the grid + a near-field one
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", [(820, 600, 884, 648), (700, 200, 764, 248)], 3),
    # C2: the floor under the glass sphere (caustic: paths through both glass interfaces, volume stack) and the sphere's
    # lower rim, 8 iterations; the mirror sphere's reflection of the GGX wall
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", [(1040, 40, 1136, 120), (700, 250, 764, 298)], 8),
])
def test_full_size_windows(twk, orc, system, scene, windows, iters):
    """Configurations at their FULL 1920x1080 frame on the device; the oracle renders windows of it (pixels are
    independent given pixel and iteration). Bit-identical inside every window."""
    app = load_app(twk, system, scene)
    assert list(app.info.resolution) == [1920, 1080]
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for it in range(iters):
        dev.render(it)
    gpu = dev.getOutputBufferHost()
    assert np.isfinite(gpu).all()
    for (x0, y0, x1, y1) in windows:
        ref = orc.Oracle(miss=app.info.miss)
        ref.loadApplication(app)
        for it in range(iters):
            ref.render(it, rect=(x0, y0, x1, y1), threads=8)
        cpu = ref.getOutputBufferHost()
        g, c = gpu[y0:y1, x0:x1], cpu[y0:y1, x0:x1]
        assert c[..., :3].std() > 1e-3, "window shows something"
        mism = (_bits(g) != _bits(c)).any(axis=2).sum()
        assert mism == 0, f"window {(x0, y0, x1, y1)}: {mism} pixels differ, max |diff| {np.abs(g - c).max()}"
        ref.close()
    dev.close()


@pytest.mark.parametrize("quality,top_cache", [(0, "1"), (1, "0"), (0, "0")])
def test_build_quality_and_top_cache_do_not_change_hit_records(twk, orc, monkeypatch, quality, top_cache):
    """twk_set_build_quality (LBVH / binned SAH) and the LDS top-of-tree cache of the trace kernel (TWK_TOP_CACHE) change
    how rays find their hits, never which: image and first-hit records bit-identical to the oracle in every combination
    (the default SAH + cache is what every other test runs), on a two-level and on a flattened scene."""
    monkeypatch.setenv("TWK_TOP_CACHE", top_cache)
    for system, scene, res in (("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (128, 72)),
                               ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (96, 54))):
        app = load_app(twk, system, scene, res)
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        dev.setBuildQuality(quality)
        app.initDevice(dev)
        info = dev.buildInfo()
        assert info["quality"] == quality and info["trees"] >= 1 and info["sahInnerCost"] > 0
        ref = orc.Oracle(miss=app.info.miss)
        ref.loadApplication(app)
        dev.debugCapture(True)
        ref.captureFirstHits(True)
        dev.setLaunchBatch(1)
        for it in range(2):
            dev.render(it)
            ref.render(it)
        g_tbg, g_ids = dev.debugReadFirstHits()
        o_tbg, o_ids = ref.readFirstHits()
        assert np.array_equal(g_ids, o_ids)
        hit = o_ids[:, 0] >= 0
        assert np.array_equal(_bits(g_tbg[hit]), _bits(o_tbg[hit]))
        assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(ref.getOutputBufferHost()))
        dev.close()


def _decode_children(node):
    """(lo[4,3], hi[4,3], refs[4], unused[4]) of one 64-byte quantised wide node (csrc/device_types.h), float32 arithmetic as the kernel's planes."""
    origin = node[0:3].astype(np.float32)
    cell = node[3:6].astype(np.float32)
    words = node[6:12].view(np.uint32)
    refs = node[12:16].view(np.int32)
    lo = np.zeros((4, 3), np.float32)
    hi = np.zeros((4, 3), np.float32)
    for k in range(4):
        ql = np.array([(int(words[c]) >> (8 * k)) & 0xff for c in range(3)], np.float32)
        qh = np.array([(int(words[3 + c]) >> (8 * k)) & 0xff for c in range(3)], np.float32)
        lo[k] = origin + ql * cell
        hi[k] = origin + qh * cell
    unused = lo[:, 0] > hi[:, 0]
    return lo, hi, refs, unused


@pytest.mark.gpu
@pytest.mark.parametrize("scene_file", ["scene_rtigo3_cornell_box.txt", "scene_rtigo3_instances.txt"])
def test_quantised_boxes_contain_everything_below_them(twk, monkeypatch, scene_file):
    """The boxes of the quantised wide nodes only cull, so all that matters is that they are conservative: walked from the
    root (and from the BVH root of every entered instance), each decoded child box contains every triangle slot — or,
    at the top level, the object-to-world image of the instance's own root box — below that child. Unused entries are
    inverted boxes. Checked on the tree twk_build produced (twk_debug_read_acceleration), float32 as the kernel decodes it."""
    from conftest import scene_path
    system = "\n".join(["resolution 64 40", "tileSize 8 8", "samplesSqrt 1", "miss 0", "light 0", "pathLengths 1 1", "epsilonFactor 500",
                        "lensShader 0", "center 0 1 0", "camera 0.75 0.5 45 3.41"]) + "\n"
    app = twk.Application(system_text=system, scene_text=open(scene_path(scene_file)).read())
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    info, nodes, tris, inst = dev.readAcceleration()
    dev.close()
    assert info["nodeFloats"] == 16
    verts = tris.reshape(-1, 3, 4)[:, :, :3]
    slot_lo, slot_hi = verts.min(axis=1), verts.max(axis=1)
    sys.setrecursionlimit(100000)
    checked = {"nodes": 0, "leaves": 0, "unused": 0, "instances": 0}
    roots_done = set()

    def subtree_bounds(ref, top_level):
        """bounds of the geometry below reference `ref`, in the space the reference lives in"""
        if ref >= 0:
            return node_bounds(ref, top_level)
        payload = ~int(ref)
        if top_level and not (payload & 0x40000000):
            # an entered instance: its own tree is checked in object space (once per geometry root); what the top-level box
            # has to contain is the world-space image of the instance's triangles
            checked["instances"] += 1
            rec = inst[payload]
            root, first, count = (int(v) for v in rec[12:15].view(np.int32))
            if root not in roots_done:
                roots_done.add(root)
                node_bounds(root, False)
            m = rec[16:28].reshape(3, 4).astype(np.float64)
            points = np.concatenate([verts[first:first + count].reshape(-1, 3).astype(np.float64), np.ones((3 * count, 1))], axis=1)
            world = points @ m.T
            return world.min(axis=0), world.max(axis=0)
        first, count = payload & 0x0fffffff, ((payload >> 28) & 3) + 1
        checked["leaves"] += 1
        return slot_lo[first:first + count].min(axis=0).astype(np.float64), slot_hi[first:first + count].max(axis=0).astype(np.float64)

    def node_bounds(index, top_level):
        lo, hi, refs, unused = _decode_children(nodes[index])
        checked["nodes"] += 1
        total_lo, total_hi = np.full(3, np.inf), np.full(3, -np.inf)
        assert (~unused).sum() >= 2, index
        for k in range(4):
            if unused[k]:
                checked["unused"] += 1
                assert np.all(lo[k] > hi[k]), (index, k)
                continue
            # the top level keeps being the top level below inner references (the world-space trees of flattened instances are spliced in)
            b_lo, b_hi = subtree_bounds(int(refs[k]), top_level)
            assert np.all(lo[k].astype(np.float64) <= b_lo) and np.all(b_hi <= hi[k].astype(np.float64)), (index, k, lo[k], hi[k], b_lo, b_hi)
            total_lo, total_hi = np.minimum(total_lo, b_lo), np.maximum(total_hi, b_hi)  # of the GEOMETRY below: grids of different nodes do not nest
        return total_lo, total_hi

    node_bounds(int(info["root"]), True)
    if info["root2"] >= 0:  # the second node of an 8-wide root (bvh_build.hip wideRootKernel)
        node_bounds(int(info["root2"]), True)
    assert checked["nodes"] > 10 and checked["leaves"] > 10
    if info["twoLevel"]:
        assert checked["instances"] > 0


@pytest.mark.gpu
def test_wave_time_profile_of_the_traversal_kernel(twk):
    """TwkLaunchStats.waveCycles: shader-clock time of the waves of the persistent traversal kernel per phase of its outer
    loop, filled by the counting variant only. The phases partition the loop: they add up to (nearly) the whole-kernel
    figure, every phase that must run on a scene with hits has time in it, and nothing is recorded with stats off."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.render(0)
    dev.synchronizeStream()
    assert all(v == 0 for v in dev.statsGet(reset=True)["waveCycles"])
    dev.statsEnable(True)
    dev.statsGet(reset=True)
    for it in range(1, 5):
        dev.render(it)
    dev.synchronizeStream()
    st = dev.statsGet(reset=True)
    refill, node, leaf, tri, write, total = st["waveCycles"]
    assert min(refill, node, tri, write) > 0 and total > 0
    assert 0.6 * total <= refill + node + leaf + tri + write <= total  # the rest: the top-of-tree cache fill, idle turns of the refill, the loop exits
    assert node > tri  # 7.6 node steps against 3 triangle tests per ray
    assert st["nodeWaveSteps"] > 0 and st["nodesVisited"] <= 64 * st["nodeWaveSteps"]
    dev.close()


@pytest.mark.gpu
def test_queue_dealing_modes_at_full_size(twk, orc):
    """The persistent traversal kernel deals its ray queue in three ways depending on the queue's length (one chunk per
    wave for a short queue; interleaved static chunks; the second half in chunk-sized tickets once the queue is longer than
    two chunks per wave — trace_kernels.hip "Wave-uniform pool"). At 1920x1080 a pass of 1 / 2 / 5 / 13 iterations starts with
    2.1 / 4.1 / 10.4 / 27 M rays and shrinks through all three regimes as the bounces go on. Whatever the pass size, the
    window of the frame equals the oracle's bit for bit after the same 13 iterations."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt")
    assert list(app.info.resolution) == [1920, 1080]
    x0, y0, x1, y1 = 1040, 40, 1104, 88
    iters = 13
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    for it in range(iters):
        ref.render(it, rect=(x0, y0, x1, y1), threads=8)
    cpu = ref.getOutputBufferHost()[y0:y1, x0:x1]
    ref.close()
    assert cpu[..., :3].std() > 1e-3
    for batch in (1, 2, 5, 13):
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        app.initDevice(dev)
        dev.setLaunchBatch(batch)
        for it in range(iters):
            dev.render(it)
        gpu = dev.getOutputBufferHost()[y0:y1, x0:x1]
        mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
        assert mism == 0, f"batch {batch}: {mism} pixels differ"
        dev.close()
