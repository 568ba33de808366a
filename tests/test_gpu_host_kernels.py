"""-m gpu: the product's kernels compiled for the HOST (oracle/host_kernels.cpp: shade_device.h / trace_device.h /
device_math.h as they are, g++ -ffp-contract=off) — BASELINE.json north_star's "single-threaded C++ CPU fallback of the same
kernels" — run as the same wavefront on the scene the device built. Three-way, bit for bit: HIP kernels == host build of
the same source == the oracle's restatement of the reference. (Needs a GPU only because twk_build is the one BVH builder.)"""
import numpy as np
import pytest

from conftest import load_app

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("system,scene,res,iters,policy", [
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (96, 54), 3, None),        # C2: every BSDF of rtigo3's Cornell set, glass volume stack, all flattened
    ("system_rtigo3_geometry.txt", "scene_rtigo3_geometry.txt", (96, 54), 2, None),               # constant environment light, rough glass, torus
    ("system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (64, 36), 2, None),             # two-level: 100 entered instances
    ("system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (64, 36), 2, (0, 0)),       # every instance entered in object space
])
def test_host_build_of_the_kernels_equals_the_device_and_the_oracle(twk, orc, system, scene, res, iters, policy):
    app = load_app(twk, system, scene, res)
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    ref = orc.Oracle(miss=app.info.miss)
    if policy is not None:
        dev.setFlattenPolicy(*policy)
    app.initDevice(dev)
    ref.loadApplication(app)
    if policy is not None:
        ref.setFlattenPolicy(*policy)
    host = orc.HostKernels(dev)
    for it in range(iters):
        dev.render(it)
        ref.render(it, threads=8)
        host.render(it)
    gpu, cpu, hst = dev.getOutputBufferHost(), ref.getOutputBufferHost(), host.getOutputBufferHost()
    assert cpu[..., :3].max() > 0.1 and np.isfinite(hst).all()
    assert np.array_equal(_bits(hst), _bits(gpu)), f"host build vs HIP: {(_bits(hst) != _bits(gpu)).any(axis=2).sum()} pixels differ"
    assert np.array_equal(_bits(hst), _bits(cpu)), f"host build vs oracle: {(_bits(hst) != _bits(cpu)).any(axis=2).sum()} pixels differ"
    assert host.counts["radianceRays"] > res[0] * res[1] * iters and host.counts["shadedSegments"] >= host.counts["radianceRays"] * 0.9
    # one pass of several iterations == one pass per iteration (the batching of the device, on the host)
    host2 = orc.HostKernels(dev)
    host2.render(0, batch=iters)
    assert np.array_equal(_bits(host2.getOutputBufferHost()), _bits(gpu))
    dev.close()
