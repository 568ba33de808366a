"""-m gpu: the reference's lighting A/B switch and its debug filter on the HIP path (include/tweeker_hip.h
twk_set_next_event_estimation ≙ USE_NEXT_EVENT_ESTIMATION, twk_set_debug_exceptions ≙ USE_DEBUG_EXCEPTIONS; shaders/config.h:50-56).
  * brute-force path tracing (NEE off) bit-identical to the oracle built with -DUSE_NEXT_EVENT_ESTIMATION=0: C1, C2, constant and
    spherical environment scenes;
  * the false-colour filter bit-identical to the oracle built with -DUSE_DEBUG_EXCEPTIONS=1, on lights that produce negative,
    infinite and NaN samples;
  * CONVERGENCE: NEE on and NEE off are two estimators of one image. On C1 at 256x256 their block means agree within the noise
    measured from disjoint iteration ranges of each — the one check of light sampling + MIS weights that needs neither the
    oracle nor the product to be right about them, only to disagree with physics differently.
"""
import numpy as np
import pytest

from conftest import load_app
from procedural import albedo_checker, cutout_slots, environment_hdr
from test_nee_switch import agreement, two_halves

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _pair(twk, orc, app, iterations, nee=True, debug=False, textures=(), light_edit=None):
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    ref = orc.Oracle(miss=app.info.miss, nee=nee, debugExceptions=debug)
    for slot, img in textures:
        dev.initTexture(slot, img)
        ref.initTexture(slot, img)
    app.initDevice(dev)
    ref.loadApplication(app)
    if light_edit:
        lights = app.lights
        light_edit(lights)
        dev.initLights(lights)
        ref.initLights(lights)
    dev.setNextEventEstimation(nee)
    dev.setDebugExceptions(debug)
    for it in range(iterations):
        dev.render(it)
        ref.render(it, threads=8)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    stats = None
    dev.close()
    ref.close()
    return gpu, cpu, stats


@pytest.mark.parametrize("system,scene,res,iters", [
    ("system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (96, 96), 4),
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90), 4),
    ("system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", (160, 90), 3),   # constant environment: miss.cu:62-68
])
def test_brute_force_path_tracing_bit_identical(twk, orc, system, scene, res, iters):
    app = load_app(twk, system, scene, res)
    gpu, cpu, _ = _pair(twk, orc, app, iters, nee=False)
    assert np.isfinite(cpu).all() and cpu[..., :3].max() > 0.0
    assert (_bits(gpu) != _bits(cpu)).any(axis=2).sum() == 0
    # and it is a different estimator: the NEE image of the same iterations differs
    gpu_on, cpu_on, _ = _pair(twk, orc, app, iters, nee=True)
    assert (_bits(gpu_on) != _bits(cpu_on)).any(axis=2).sum() == 0
    assert (_bits(gpu_on) != _bits(gpu)).any(axis=2).mean() > 0.25


def test_brute_force_spherical_environment_bit_identical(twk, orc):
    """intro_07 (C3): spherical HDR environment, miss.cu:92-106 without the MIS weight; textures and cutout opacity as in C3."""
    app = load_app(twk, "system_intro_07.txt", "scene_intro_07.txt", (128, 72))
    textures = ((0, albedo_checker()), (1, cutout_slots()), (2, environment_hdr()))
    gpu, cpu, _ = _pair(twk, orc, app, 3, nee=False, textures=textures)
    assert np.isfinite(cpu).all() and cpu[..., :3].max() > 0.5
    assert (_bits(gpu) != _bits(cpu)).any(axis=2).sum() == 0


def test_no_shadow_rays_without_next_event_estimation(twk):
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.statsEnable(True)
    counts = {}
    for nee in (True, False):
        dev.setNextEventEstimation(nee)
        dev.statsGet(reset=True)
        for it in range(2):
            dev.render(it)
        dev.synchronizeStream()
        counts[nee] = dev.statsGet(reset=True)
    dev.close()
    assert counts[True]["shadowRays"] > 0 and counts[False]["shadowRays"] == 0
    assert counts[False]["radianceRays"] > 0


@pytest.mark.parametrize("emission,colour", [((-10.0, -10.0, -10.0), 2), ((float("inf"),) * 3, 1)], ids=["negative-blue", "infinite-green"])
def test_debug_exceptions_false_colours(twk, orc, emission, colour):
    """raygeneration.cu:205-218. A light with negative emission makes negative samples (super blue); an infinite one makes
    infinite samples (super green) and, where a zero factor meets it, NaN (super red). Bit-identical to the oracle's build of the
    same filter; with the filter off the NaN samples are dropped instead (raygeneration.cu:222)."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (128, 72))

    def edit(lights):
        for l in lights:
            l.emission[0], l.emission[1], l.emission[2] = emission
    gpu, cpu, _ = _pair(twk, orc, app, 1, debug=True, light_edit=edit)   # one iteration: the false colours are not averaged away
    assert (_bits(gpu) != _bits(cpu)).any(axis=2).sum() == 0
    flagged = gpu[..., colour] == 1000000.0
    assert flagged.mean() > 0.05, "the filter coloured the samples it is there for"
    others = [c for c in range(3) if c != colour]
    assert (gpu[flagged][:, others] == 0.0).all()
    gpu_off, cpu_off, _ = _pair(twk, orc, app, 1, debug=False, light_edit=edit)
    assert (_bits(gpu_off) != _bits(cpu_off)).any(axis=2).sum() == 0
    assert not (gpu_off[..., :3] == 1000000.0).any()


def test_nee_on_and_off_converge_to_the_same_image(twk):
    """C1 at 256x256, path length 2..48 (the truncated tail, which the two estimators cut differently, is < 1e-4 of the image).
    NEE on: 2 x 256 iterations; NEE off: 2 x 2048. Block means (16x16 blocks of 16x16 pixels) of the two differ by no more than
    their noise, estimated from the two disjoint halves of each run: R = sum d^2 / sum var(d) ~ 1 (F distributed; a 1 % bias of
    the image would put it above 10 at these counts). The numbers are quoted in DESIGN.md 5."""
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (256, 256))
    st = app.state
    st.pathLengths[0], st.pathLengths[1] = 2, 48
    runs = []
    for nee, n in ((True, 256), (False, 2048)):
        dev = twk.Device(ordinal=0, miss=app.info.miss)
        app.initDevice(dev)
        dev.setState(st)
        dev.setNextEventEstimation(nee)
        runs.append(two_halves(dev.render, dev.getOutputBufferHost, n))
        dev.close()
    R, cross, floor_on, floor_off, bias = agreement(runs[0], runs[1], blocks=16)
    print(f"\nHIP NEE on (2x256 spp) vs off (2x2048 spp), C1 256x256: R {R:.3f}; relative RMSE per pixel: on|off {cross:.4f}, on|on {floor_on:.4f}, off|off {floor_off:.4f}; "
          f"expected on|off from the floors {np.sqrt((floor_on ** 2 + floor_off ** 2) / 4):.4f}; relative bias of the image mean {bias:+.5f}")
    assert np.isfinite(runs[0][2]).all() and np.isfinite(runs[1][2]).all()
    assert R < 1.6, "block means of the NEE and the brute-force image differ by more than their noise"
    assert abs(bias) < 0.003
    assert cross < 1.1 * np.sqrt((floor_on ** 2 + floor_off ** 2) / 4), "per-pixel RMSE between the two images exceeds what their noise floors explain"


def test_switch_from_the_system_description(twk, orc):
    """`nextEventEstimation 0` in the system description (grammar extension) reaches the device through twk_app_init_device: the
    image equals the brute-force oracle's, and the description of a scene without the key renders with NEE as before."""
    from conftest import scene_path
    system = open(scene_path("system_rtigo3_cornell_box.txt")).read() + "nextEventEstimation 0\n"
    app = twk.Application(system_text=system, scene_text=open(scene_path("scene_rtigo3_cornell_box.txt")).read())
    app.setResolution(96, 54)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    ref = orc.Oracle(miss=app.info.miss, nee=False)
    ref.loadApplication(app)
    for it in range(3):
        dev.render(it)
        ref.render(it, threads=8)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    dev.close()
    assert (_bits(gpu) != _bits(cpu)).any(axis=2).sum() == 0
    with pytest.raises(AssertionError):
        orc.Oracle(miss=app.info.miss).loadApplication(app)  # the NEE build of the oracle is not what this description asks for


def test_shade_phase_tallies_of_the_measurement_build(twk):
    """TwkLaunchStats.shadePhase* (ABI 9): per phase of the shading of a segment — wave executions, lanes, shader-clock cycles —
    filled by the measurement builds while statistics are on. Consistency: every wave iteration runs the path phase and the
    append; a phase never has more than 64 lanes per execution; the phases inside shadePath take no more time than the path;
    the miss + hit lanes are the segments shaded; and with next-event estimation off the two NEE phases never run."""
    L = twk._lib if hasattr(twk, "_lib") else None
    PATH, MISS, HIT, NEE_SAMPLE, NEE_EVAL, APPEND, ITERATION = 0, 2, 3, 12, 13, 19, 20
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (320, 180))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.render(0)
    dev.synchronizeStream()
    assert all(v == 0 for v in dev.statsGet(reset=True)["shadePhaseWaveSteps"]), "nothing is tallied with statistics off"
    dev.statsEnable(True)
    out = {}
    for nee in (True, False):
        dev.setNextEventEstimation(nee)
        dev.statsGet(reset=True)
        for it in range(4):
            dev.render(it)
        dev.synchronizeStream()
        out[nee] = dev.statsGet(reset=True)
    dev.close()
    st = out[True]
    ws, ln, cy = st["shadePhaseWaveSteps"], st["shadePhaseLanes"], st["shadePhaseCycles"]
    assert ws[PATH] > 0 and ws[APPEND] == ws[ITERATION] >= ws[PATH]
    for k in range(len(ws)):
        assert ln[k] <= 64 * ws[k] and (ws[k] > 0) == (ln[k] > 0 or k == APPEND), k  # (an append with no appender still runs its barriers)
    assert ln[MISS] + ln[HIT] == st["shadedHits"] + st["missed"] == ln[PATH]
    assert sum(cy[k] for k in (1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 14, 15)) <= 1.02 * cy[PATH] <= 1.02 * cy[ITERATION]
    assert 0.3 < ln[NEE_EVAL] / (64.0 * ws[NEE_EVAL]) <= 1.0 and ws[NEE_SAMPLE] >= ws[NEE_EVAL] > 0
    off = out[False]
    assert off["shadePhaseWaveSteps"][NEE_SAMPLE] == 0 and off["shadePhaseWaveSteps"][NEE_EVAL] == 0 and off["shadowRays"] == 0
    assert off["shadePhaseWaveSteps"][PATH] > 0
