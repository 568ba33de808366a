"""CPU: oracle helpers and product host layer against the REFERENCE's own code, live (oracle/_ref/libref_host.so,
compiled from /root/reference by `make -C oracle ref`). Larger random sets than the committed fixtures. Skipped where
the reference-built library is absent."""
import os

import numpy as np
import pytest

from conftest import ROOT

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_host.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (reference absent)")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_units_against_reference(orc):
    ref, u = orc.Reference(), orc.oracle_units()
    rng = np.random.default_rng(99)
    for _ in range(2000):
        a, b = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32))
        assert ref.tea4(a, b) == u.tea4(a, b)
    for seed in rng.integers(0, 2**32, 50):
        r, rf = ref.rng_stream(int(seed), 32)
        o, of = u.rng_stream(int(seed), 32)
        assert np.array_equal(_bits(r), _bits(o)) and rf == of
    a = rng.normal(size=(2000, 3)).astype(np.float32) * rng.choice([1e-3, 1, 1e3], (2000, 1)).astype(np.float32)
    b = rng.normal(size=(2000, 3)).astype(np.float32)
    for op in range(6):
        for x, y in zip(a, b):
            assert np.array_equal(_bits(ref.vec3(op, x, y)), _bits(u.vec3(op, x, y)))
    i = a / np.linalg.norm(a, axis=1, keepdims=True)
    n = b / np.linalg.norm(b, axis=1, keepdims=True)
    for x, y in zip(i.astype(np.float32), n.astype(np.float32)):
        for ior in (1.5, 1 / 1.5, 1.0003):
            r0, r1 = ref.refract(x, y, ior), u.refract(x, y, ior)
            assert r0[0] == r1[0] and np.array_equal(_bits(r0[1]), _bits(r1[1]))
        assert np.array_equal(_bits(ref.tbn(x, y)), _bits(u.tbn(x, y)))


def test_host_layer_against_reference(twk, orc):
    ref = orc.Reference()
    pi = np.float32(np.pi)
    for (u, v, ax) in [(1, 1, 0), (1, 1, 1), (1, 1, 2), (4, 7, 0), (5, 3, 2)]:
        r, p = ref.mesh_plane(u, v, ax), twk.mesh_plane(u, v, ax)
        assert np.array_equal(_bits(r[0]), _bits(p[0])) and np.array_equal(r[1], p[1])
    for (u, v, th) in [(180, 90, pi), (12, 7, np.float32(0.5) * pi), (3, 3, pi)]:
        r, p = ref.mesh_sphere(u, v, 1.0, th), twk.mesh_sphere(u, v, 1.0, th)
        assert np.array_equal(_bits(r[0]), _bits(p[0])) and np.array_equal(r[1], p[1])
    for (u, v) in [(180, 180), (9, 4)]:
        r, p = ref.mesh_torus(u, v, 0.75, 0.25), twk.mesh_torus(u, v, 0.75, 0.25)
        assert np.array_equal(_bits(r[0]), _bits(p[0])) and np.array_equal(r[1], p[1])
    r, p = ref.mesh_box(), twk.mesh_box()
    assert np.array_equal(_bits(r[0]), _bits(p[0])) and np.array_equal(r[1], p[1])
    rng = np.random.default_rng(5)
    for _ in range(50):
        c = rng.uniform(-2, 2, 3).astype(np.float32)
        phi, theta, fov, dist = (np.float32(x) for x in (rng.uniform(0, 1), rng.uniform(0.05, 0.95), rng.uniform(10, 120), rng.uniform(0.5, 20)))
        w, h = int(rng.integers(16, 4000)), int(rng.integers(16, 3000))
        a = ref.camera_frustum(c, phi, theta, fov, dist, w, h)
        k = twk.camera_frustum(c, phi, theta, fov, dist, float(np.float32(w) / np.float32(h)))
        assert np.array_equal(_bits(a), _bits(np.array(list(k.P) + list(k.U) + list(k.V) + list(k.W), np.float32)))
    kinds = {0: "rotate", 1: "scale", 2: "translate"}
    for _ in range(40):
        ops = []
        for _ in range(int(rng.integers(1, 6))):
            k = int(rng.integers(0, 3))
            vals = rng.uniform(-3, 3, 4).astype(np.float32)
            if k == 1:
                vals = np.abs(vals) + np.float32(0.1)
            if k == 0:
                vals[3] = np.float32(rng.uniform(-360, 360))
            ops.append([k] + [float(x) for x in vals])
        lines = ["material m brdf_diffuse"] + [kinds[int(o[0])] + " " + " ".join(repr(x) for x in (o[1:5] if o[0] == 0 else o[1:4])) for o in ops] + ["model box m"]
        app = twk.Application(system_text="light 0\nmiss 0\n", scene_text="\n".join(lines))
        assert np.array_equal(_bits(app.instance(0)[1]), _bits(ref.transform_stack(np.array(ops, np.float32))))


def test_boundary_struct_layouts_equal_the_reference_headers(twk, orc):
    """The structs that cross the replaced boundary by pointer (INTEGRATION.md casts the reference's CameraDefinition /
    LightDefinition / TonemapperGUI / TriangleAttributes to the Twk* types): size and every member offset as the
    reference's own headers lay them out == the ctypes mirrors of include/tweeker_hip.h; FunctionIndex / LightType
    values == the plain ints of the C ABI."""
    import ctypes as C
    lib = orc.Reference().lib
    L = twk._lib if hasattr(twk, "_lib") else __import__("tweeker_raytracer_amd._lib", fromlist=["x"])
    for which, struct in enumerate((L.CameraDefinition, L.LightDefinition, L.TriangleAttributes, L.Tonemapper)):
        out = (C.c_int * 32)()
        n = lib.ref_struct_layout(which, out, 32)
        mine = [C.sizeof(struct)] + [getattr(struct, name).offset for name, _ in struct._fields_]
        assert n == len(mine) and list(out[:n]) == mine, struct.__name__
    out = (C.c_int * 16)()
    assert lib.ref_enum_values(out, 16) == 7 and list(out[:7]) == [0, 1, 2, 3, 4, 0, 1]
