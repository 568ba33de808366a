"""CPU property tests of the oracle's shading half — the part no reference fixture pins (the reference holds no golden
images and its .cu files do not build here). tests/test_oracle_statement_pin.py shows the oracle SAYS what the reference
says; these show that what it says behaves like a path tracer should: sampling densities integrate to one and agree with
the sample histograms, the Fresnel term has its limits, and a convex object in a white furnace comes out at its albedo
(next-event estimation, BSDF sampling and their MIS weights add up to an unbiased estimate)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

_fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def lib():
    l = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    l.orc_fresnel_dielectric.restype = C.c_float
    l.orc_fresnel_dielectric.argtypes = [C.c_float, C.c_float]
    return l


def _f(a):
    return a.ctypes.data_as(_fp)


def _v(x):
    return (C.c_float * 3)(*[float(k) for k in x])


def _sample(lib, bsdf, params, wo, count, seed=12345):
    out = np.zeros((count, 8), np.float32)
    p = np.asarray(params, np.float32)
    assert lib.orc_prop_bsdf_sample(bsdf, _f(p), _v((0, 0, 1)), _v((1, 0, 0)), _v(wo), C.c_uint(seed), count, _f(out)) == 0
    return out[:, 0:3], out[:, 3:6], out[:, 6], out[:, 7].view(np.uint32)


def _eval(lib, bsdf, params, wo, wi):
    wi = np.ascontiguousarray(wi, np.float32)
    out = np.zeros((wi.shape[0], 4), np.float32)
    p = np.asarray(params, np.float32)
    assert lib.orc_prop_bsdf_eval(bsdf, _f(p), _v((0, 0, 1)), _v((1, 0, 0)), _v(wo), _f(wi), wi.shape[0], _f(out)) == 0
    return out[:, :3], out[:, 3]


def _sphere_grid(n_theta=256, n_phi=512):
    """Midpoint grid over the upper hemisphere: directions and solid angles."""
    ct = (np.arange(n_theta) + 0.5) / n_theta            # cos(theta) in (0, 1)
    ph = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - CT * CT)
    d = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1).reshape(-1, 3)
    return d.astype(np.float32), (1.0 / n_theta) * (2 * np.pi / n_phi), (n_theta, n_phi)


TERMINATE = 0x80000000
WO = np.array([0.5, 0.2, 0.8]) / np.linalg.norm([0.5, 0.2, 0.8])


@pytest.mark.parametrize("bsdf,params,name", [
    (0, (0.8, 0.7, 0.6, 0.1, 0.1, 1.5), "diffuse"),
    (3, (0.9, 0.9, 0.9, 0.35, 0.35, 1.5), "ggx isotropic"),
    (3, (0.9, 0.9, 0.9, 0.5, 0.2, 1.5), "ggx anisotropic"),
])
def test_eval_pdf_integrates_to_the_sampled_mass_and_matches_the_histogram(lib, bsdf, params, name):
    """eval's pdf over the hemisphere = the fraction of sample() draws that are kept (all of them for Lambert; GGX drops
    reflections that dive below the surface), and per solid-angle cell the sample frequency equals the integrated pdf."""
    dirs, dw, (nt, nphi) = _sphere_grid()
    f, pdf = _eval(lib, bsdf, params, WO, dirs)
    mass = float(pdf.astype(np.float64).sum() * dw)
    n = 400_000
    wi, fop, spdf, flags = _sample(lib, bsdf, params, WO, n)
    kept = ((flags & TERMINATE) == 0) & (spdf > 0)
    assert abs(mass - kept.mean()) < 4e-3, f"{name}: integral of pdf {mass} vs kept fraction {kept.mean()}"
    if bsdf == 0:
        assert abs(mass - 1.0) < 2e-3
    # histogram on a coarse 8 x 16 grid of (cos theta, phi)
    w = wi[kept]
    assert np.allclose(np.linalg.norm(w, axis=1), 1.0, atol=1e-4) and (w[:, 2] > -1e-6).all()
    it = np.clip((w[:, 2] * 8).astype(int), 0, 7)
    ip = np.clip(((np.arctan2(w[:, 1], w[:, 0]) % (2 * np.pi)) / (2 * np.pi) * 16).astype(int), 0, 15)
    hist = np.zeros((8, 16))
    np.add.at(hist, (it, ip), 1.0 / n)
    cell = (pdf.astype(np.float64) * dw).reshape(8, nt // 8, 16, nphi // 16).sum(axis=(1, 3))
    sigma = np.sqrt(np.maximum(cell, 1e-9) / n)
    assert (np.abs(hist - cell) < 5 * sigma + 2e-4).all(), f"{name}: worst cell off by {np.abs(hist - cell).max()}"
    # the sample's own pdf and weight agree with eval at the sampled direction: f_over_pdf = f |cos| / pdf
    idx = np.flatnonzero(kept)[:2000]
    f2, pdf2 = _eval(lib, bsdf, params, WO, wi[idx])
    assert np.allclose(pdf2, spdf[idx], rtol=2e-3, atol=1e-6)
    assert np.allclose(f2 * np.abs(wi[idx, 2:3]) / pdf2[:, None], fop[idx], rtol=3e-3, atol=1e-5)


def test_lambert_reflectance_is_the_albedo(lib):
    dirs, dw, _ = _sphere_grid()
    f, pdf = _eval(lib, 0, (0.8, 0.7, 0.6, 0, 0, 1.5), WO, dirs)
    rho = (f.astype(np.float64) * dirs[:, 2:3]).sum(axis=0) * dw
    assert np.allclose(rho, (0.8, 0.7, 0.6), atol=2e-3)


def test_ggx_brdf_conserves_energy_and_is_reciprocal(lib):
    dirs, dw, _ = _sphere_grid()
    params = (1.0, 1.0, 1.0, 0.3, 0.3, 1.5)
    f, _ = _eval(lib, 3, params, WO, dirs)
    rho = float((f[:, 0].astype(np.float64) * dirs[:, 2]).sum() * dw)
    assert 0.6 < rho <= 1.0 + 2e-3, rho  # single-scattering microfacet BRDF: loses a little, never gains
    rng = np.random.default_rng(3)
    a = rng.normal(size=(200, 3)); a[:, 2] = np.abs(a[:, 2]) + 0.05; a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(200, 3)); b[:, 2] = np.abs(b[:, 2]) + 0.05; b /= np.linalg.norm(b, axis=1, keepdims=True)
    for wo_, wi_ in zip(a[:40], b[:40]):
        fab, _ = _eval(lib, 3, params, wo_, wi_[None])
        fba, _ = _eval(lib, 3, params, wi_, wo_[None])
        assert np.allclose(fab, fba, rtol=2e-4, atol=1e-6)


def test_fresnel_dielectric_limits(lib):
    F = lib.orc_fresnel_dielectric
    for et in (1.5, 1.33, 2.4):
        r0 = ((et - 1) / (et + 1)) ** 2
        assert abs(F(et, 1.0) - r0) < 1e-6                      # normal incidence
        assert abs(F(et, 1e-4) - 1.0) < 1e-3                    # grazing
        vals = [F(et, c) for c in np.linspace(1.0, 0.01, 50)]
        assert all(b >= a - 1e-6 for a, b in zip(vals, vals[1:]))  # monotone towards grazing
        assert all(0.0 <= v <= 1.0 for v in vals)
    # from the dense side (eta = 1 / 1.5): total internal reflection beyond the critical angle
    crit = np.sqrt(1 - (1 / 1.5) ** 2)
    assert F(1 / 1.5, float(crit) * 0.98) == 1.0 and F(1 / 1.5, 1.0) == pytest.approx(0.04, abs=1e-6)


SYSTEM = "\n".join(["resolution 48 48", "tileSize 8 8", "samplesSqrt 1", "miss 1", "light 0", "pathLengths 2 6", "epsilonFactor 500",
                    "lensShader 0", "center 0 0 0", "camera 0.75 0.5 30 6"]) + "\n"


@pytest.mark.parametrize("material,expect,tol", [
    ("albedo 0.5 0.5 0.5\nmaterial m brdf_diffuse", 0.5, 0.012),     # Lambert: NEE + BSDF sampling + MIS = albedo
    ("albedo 0.9 0.9 0.9\nmaterial m brdf_specular", 0.9, 1e-6),     # mirror: exactly the albedo
    ("albedo 1 1 1\nroughness 0.3 0.3\nmaterial m brdf_ggx_smith", None, None),
])
def test_white_furnace_convex_object(twk, orc, material, expect, tol):
    """A sphere (convex: no interreflection) in the constant white environment (miss 1, which is also light 0): every
    pixel on it must come out at the directional albedo — for Lambert and the mirror that is the albedo itself."""
    scene = material + "\nmodel sphere 60 30 1.0 m\n"
    app = twk.Application(system_text=SYSTEM, scene_text=scene)
    o = orc.Oracle(miss=1)
    o.loadApplication(app)
    spp = 96
    for it in range(spp):
        o.render(it, threads=8)
    img = o.getOutputBufferHost()[..., :3]
    inner = img[16:32, 16:32]  # well inside the sphere's silhouette
    assert np.isfinite(img).all()
    mean = float(inner.mean())
    if expect is not None:
        assert abs(mean - expect) < tol, mean
        assert np.abs(img[0, 0] - 1.0).max() < 1e-6  # the corner sees the environment itself
    else:
        assert 0.75 < mean <= 1.0 + 5e-3, mean  # single-scattering GGX: a few percent dark, never above the furnace
