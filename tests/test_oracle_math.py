"""CPU: the portable elementary functions the oracle (and the HIP kernels) evaluate stay within a few ulp of libm
on the argument ranges the shaders use — so choosing them over a vendor libm is inside any radiance tolerance."""
import numpy as np


def _ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def test_sin_cos_close_to_libm(orc):
    x = np.linspace(-2 * np.pi, 4 * np.pi, 400001).astype(np.float32)
    for op, f in ((0, np.sin), (1, np.cos)):
        got = orc.oracle_math(op, x)
        ref = f(x.astype(np.float64))
        assert np.max(np.abs(got - ref)) < 2.5e-7  # absolute: near the zeros of sin/cos ulps are meaningless
        big = np.abs(ref) > 0.1
        assert _ulp_diff(got[big], ref[big].astype(np.float32)).max() <= 3


def test_exp_close_to_libm(orc):
    x = -np.logspace(-6, np.log10(80.0), 200001).astype(np.float32)
    got = orc.oracle_math(2, x)
    ref = np.exp(x.astype(np.float64)).astype(np.float32)
    assert _ulp_diff(got, ref).max() <= 2
    assert orc.oracle_math(2, np.array([0.0, -0.0], np.float32)).tolist() == [1.0, 1.0]
    assert orc.oracle_math(2, np.array([-1e27, -100.0], np.float32)).tolist() == [0.0, 0.0]


def test_atan_acos_close_to_libm(orc):
    x = np.linspace(-60, 60, 200001).astype(np.float32)
    assert _ulp_diff(orc.oracle_math(5, x), np.arctan(x.astype(np.float64)).astype(np.float32)).max() <= 3
    c = np.linspace(-1, 1, 200001).astype(np.float32)
    got = orc.oracle_math(4, c)
    ref = np.arccos(c.astype(np.float64))
    assert np.max(np.abs(got - ref)) < 5e-7
    rng = np.random.default_rng(3)
    yy = rng.normal(size=100000).astype(np.float32)
    xx = rng.normal(size=100000).astype(np.float32)
    got = orc.oracle_math(3, yy, xx)
    assert np.max(np.abs(got - np.arctan2(yy.astype(np.float64), xx.astype(np.float64)))) < 6e-7


def test_libm_build_of_the_oracle_agrees_within_tolerance(twk, orc):
    """Same oracle built against glibc's sinf/cosf/expf (-DORC_USE_LIBM): per-image relative L2 <= 2e-2 at 4 spp on a
    64x36 C2 frame (paths diverge at Russian-roulette / Fresnel coin flips when a sample differs by an ulp)."""
    from conftest import load_app
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (64, 36))
    imgs = []
    for libm in (False, True):
        o = orc.Oracle(miss=app.info.miss, libm=libm)
        o.loadApplication(app)
        for it in range(4):
            o.render(it)
        imgs.append(o.getOutputBufferHost()[..., :3].astype(np.float64))
    same = np.isclose(imgs[0], imgs[1], rtol=1e-4, atol=1e-6).all(axis=2).mean()
    assert same > 0.97, f"only {same:.3f} of the pixels agree to 1e-4"
