"""-m gpu: the time view (≙ the reference's compile-time USE_TIME_VIEW, shaders/config.h:60, raygeneration.cu:169-171,
231-244): with twk_set_time_view(1) the alpha of the accumulation buffer is the running mean of the sample's clock cycles
x clockFactor x 1e-9 instead of 1, and nothing else changes. Clock cycles are no deterministic quantity: the RGB is
compared bit for bit, the alpha through its properties."""
import numpy as np
import pytest

from conftest import load_app

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _render(dev, iterations):
    for it in range(iterations):
        dev.render(it)
    return dev.getOutputBufferHost().copy()


@pytest.mark.parametrize("batch", [64, 1])
def test_time_view_replaces_alpha_only(twk, batch):
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (160, 90))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    dev.setLaunchBatch(batch)
    plain = _render(dev, 4)
    assert np.all(plain[..., 3] == 1.0)
    dev.setTimeView(True)
    timed = _render(dev, 4)
    assert np.array_equal(_bits(timed[..., :3]), _bits(plain[..., :3]))
    alpha = timed[..., 3]
    assert np.isfinite(alpha).all() and (alpha > 0.0).all()
    assert alpha.max() > 1.5 * alpha.min()          # paths differ in length: the spheres cost more than the walls
    # the state's clockFactor scales it (cycles vary from run to run, a factor of two does not drown in that)
    st = app.state
    st.clockFactor = 2.0 * st.clockFactor
    dev.setState(st)
    doubled = _render(dev, 4)
    assert np.array_equal(_bits(doubled[..., :3]), _bits(plain[..., :3]))
    ratio = float(doubled[..., 3].mean() / alpha.mean())
    assert 1.3 < ratio < 3.0, ratio
    dev.setTimeView(False)
    again = _render(dev, 4)
    assert np.array_equal(_bits(again), _bits(plain))
    dev.close()


def test_time_view_on_a_two_level_scene_with_environment_light(twk):
    app = load_app(twk, "system_rtigo3_instances.txt", "scene_rtigo3_instances.txt", (128, 72))
    dev = twk.Device(ordinal=0, miss=app.info.miss)
    app.initDevice(dev)
    plain = _render(dev, 3)
    dev.setTimeView(True)
    timed = _render(dev, 3)
    assert np.array_equal(_bits(timed[..., :3]), _bits(plain[..., :3]))
    assert np.isfinite(timed[..., 3]).all() and (timed[..., 3] > 0.0).all()
    dev.close()
