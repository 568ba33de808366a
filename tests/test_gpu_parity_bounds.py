"""-m gpu: where parity was softest ("What's weak" 1 and 8).

1. Object space is the OptiX-faithful definition of an instance intersection (an IAS -> GAS descent transforms the RAY,
   apps/rtigo3/src/Device.cpp:1427-1489); the product's default flattens small / rarely referenced instances into world
   space (include/tweeker_hip.h twk_set_flatten_policy). Here the device under policy (0, 0) must reproduce the
   object-space fixture round 1 committed (tests/golden/oracle_cornell_objectspace.npz) bit for bit, and the device under
   its DEFAULT policy is held against the oracle under policy (0, 0) within SURVEY 8(d)'s stated tolerances.
2. Every configuration at its FULL sample count: C1 512x512 x 1 spp (whole image), C2 64, C3 256, C4 256 (both scenes),
   C5 1024 iterations of one rank's 480x2160 share — a 64x48 oracle window each (pixels are independent given pixel and
   iteration); iterationIndex runs to 1023 in tea<4> and in the running mean's 1 / (i + 1).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_app
from procedural import albedo_checker, cutout_slots, environment_hdr

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_device_under_object_space_policy_matches_the_round1_fixture(twk):
    gold = np.load(os.path.join(GOLDEN, "oracle_cornell_objectspace.npz"))
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (64, 64))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    dev.setFlattenPolicy(0, 0)
    dev.debugCapture(True)
    app.initDevice(dev)
    dev.render(0)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(gold["c1_64_spp1"]))
    tbg, ids = dev.debugReadFirstHits()
    hit = gold["c1_64_firsthit_ids"][:, 0] >= 0
    assert np.array_equal(ids, gold["c1_64_firsthit_ids"]) and np.array_equal(_bits(tbg[hit]), _bits(gold["c1_64_firsthit_tbg"][hit]))
    dev.render(1)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(gold["c1_64_spp2"]))
    dev.close()
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (64, 36))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    dev.setFlattenPolicy(0, 0)
    app.initDevice(dev)
    for it in range(2):
        dev.render(it)
    assert np.array_equal(_bits(dev.getOutputBufferHost()), _bits(gold["c2_64x36_spp2"]))
    dev.close()


# Stated tolerance: per-pixel L2 on linear RGB, mean and 99.9-percentile reported; <= 1e-4 per pixel where the
# RNG stream and the hits are the same, relative RMSE <= 2 % where they are not. Flattening moves a hit distance by ~1e-5
# relative: almost every path keeps its hits (L2 ~ 1e-7), a few paths per thousand pixels change a hit at a silhouette and
# take another route for that sample. Measured on the oracle (default policy vs (0, 0), CPU): C2 160x90 mean 2.5e-7 ..
# 3.3e-7, p99.9 2.3e-5 .. 4.3e-5, relative RMSE 1.1e-6 .. 2.1e-6 over 1..8 spp; C4 instances 128x72 mean ~1e-4, p99.9
# 8e-4 .. 2.3e-2, relative RMSE 4.0e-3 .. 1.6e-3, 97 % of the pixels bit-identical.
@pytest.mark.parametrize("system,scene,res,iters,mean_max,p999_max,rmse_max,identical_min", [
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90), 8, 1.0e-5, 1.0e-4, 1.0e-4, 0.5),
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (128, 72), 8, 1.0e-3, 1.0e-1, 2.0e-2, 0.9),
])
def test_default_policy_is_bounded_by_the_object_space_oracle(twk, orc, system, scene, res, iters, mean_max, p999_max, rmse_max, identical_min):
    app = load_app(twk, system, scene, res)
    dev = twk.Device(ordinal=0, miss=app.info.miss)  # default flatten policy
    app.initDevice(dev)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    ref.setFlattenPolicy(0, 0)                        # every instance in object space
    for it in range(iters):
        dev.render(it)
        ref.render(it, threads=8)
    gpu, cpu = dev.getOutputBufferHost()[..., :3], ref.getOutputBufferHost()[..., :3]
    l2 = np.sqrt(((gpu - cpu) ** 2).sum(axis=2))
    rmse = float(np.sqrt(((gpu - cpu) ** 2).sum()) / np.sqrt((cpu ** 2).sum()))
    report = (f"{scene} {res} {iters} spp, device default policy vs oracle policy (0, 0): per-pixel L2 mean {l2.mean():.3e}, "
              f"p99.9 {np.percentile(l2, 99.9):.3e}, max {l2.max():.3e}; relative RMSE {rmse:.3e}; pixels bit-identical {(l2 == 0).mean():.3f}")
    print(report)
    assert np.isfinite(gpu).all()
    assert l2.mean() <= mean_max and np.percentile(l2, 99.9) <= p999_max and rmse <= rmse_max and (l2 == 0).mean() >= identical_min, report
    dev.close()


def _intro07(twk):
    app = load_app(twk, "system_intro_07.txt", "scene_intro_07.txt")

    def edit(mats):
        mats[1].useAlbedoTexture = 1   # 'floor' (material 0 is the area light's, Application.cpp:640-659)
        mats[4].useCutoutTexture = 1   # 'cutout'

    return app, edit, ((0, albedo_checker()), (1, cutout_slots()), (2, environment_hdr()))


FULL_SPP = [
    # name, system, scene, window (x0, y0, x1, y1), iterations, device count, device index
    ("C2", "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (1000, 300, 1064, 348), 64, 1, 0),
    ("C3", "system_intro_07.txt", "scene_intro_07.txt", (930, 500, 994, 548), 256, 1, 0),
    ("C4-geometry", "system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", (900, 300, 964, 348), 256, 1, 0),
    ("C4-instances", "system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (820, 600, 884, 648), 256, 1, 0),
    ("C5-rank3of8", "system_rtigo3_cornell_box_c5.txt", "scene_rtigo3_cornell_box.txt", (200, 1000, 264, 1048), 1024, 8, 3),
]


@pytest.mark.parametrize("name,system,scene,window,iters,count,index", FULL_SPP, ids=[c[0] for c in FULL_SPP])
def test_full_sample_count_windows(twk, orc, name, system, scene, window, iters, count, index):
    textures, edit = (), None
    if name == "C3":
        app, edit, textures = _intro07(twk)
    else:
        app = load_app(twk, system, scene)
    assert app.info.samplesSqrt ** 2 == iters, "the configuration's own sample count (samplesSqrt^2, Application.cpp:1141,495)"
    dev = twk.Device(ordinal=0, index=index, count=count, miss=app.info.miss)
    ref = orc.Oracle(index=index, count=count, miss=app.info.miss)
    for slot, img in textures:
        dev.initTexture(slot, img)
        ref.initTexture(slot, img)
    app.initDevice(dev, distribution=1 if count > 1 else None)
    st = app.state
    if count > 1:
        st.distribution = 1
    ref.loadApplication(app, state=st)
    if edit:
        mats = app.materials
        edit(mats)
        dev.initMaterials(mats)
        ref.initMaterials(mats)
    dev.setLaunchBatch(32)
    for it in range(iters):
        dev.render(it)
    dev.synchronizeStream()
    x0, y0, x1, y1 = window
    for it in range(iters):
        ref.render(it, rect=window, threads=8)
    gpu, cpu = dev.getOutputBufferHost()[y0:y1, x0:x1], ref.getOutputBufferHost()[y0:y1, x0:x1]
    assert cpu[..., :3].std() > 1e-3 and np.isfinite(gpu).all(), "window shows something"
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{name} after {iters} iterations: {mism} pixels of the window differ, max |diff| {np.abs(gpu - cpu).max()}"
    dev.close()


def test_c1_whole_image_at_its_full_size(twk, orc):
    """C1 = scene_rtigo3_cornell_box.txt Lambert-only, 512x512, 1 spp: the whole image, not a crop."""
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt")
    assert list(app.info.resolution) == [512, 512] and app.info.samplesSqrt == 1
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.render(0)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    ref.render(0, threads=8)
    gpu, cpu = dev.getOutputBufferHost(), ref.getOutputBufferHost()
    assert gpu.shape == (512, 512, 4) and cpu[..., :3].max() > 0.5
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{mism} of 262144 pixels differ"
    dev.close()


def test_c2_whole_frame_at_its_full_sample_count(twk, orc):
    """C2 — the configuration BASELINE.json's metric is quoted on: Cornell box 1920x1080, 64 spp, full BSDF set — the WHOLE
    frame at the full sample count, all 2 073 600 pixels against the oracle (16 host threads: ~133 M samples, about half a minute)."""
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt")
    assert list(app.info.resolution) == [1920, 1080] and app.info.samplesSqrt ** 2 == 64
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    for it in range(64):
        dev.render(it)
    gpu = dev.getOutputBufferHost()
    dev.close()
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    for it in range(64):
        ref.render(it, threads=16)
    cpu = ref.getOutputBufferHost()
    assert gpu.shape == (1080, 1920, 4) and np.isfinite(cpu).all() and cpu[..., :3].mean() > 0.1
    mism = (_bits(gpu) != _bits(cpu)).any(axis=2).sum()
    assert mism == 0, f"{mism} of 2073600 pixels differ after 64 iterations, max |diff| {np.abs(gpu - cpu).max()}"
