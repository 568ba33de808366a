"""CPU: oracle helper layer and product host scene layer against fixtures generated from the REFERENCE's own
code (tests/golden/reference_helpers.npz, written by tests/golden/make_golden.py from oracle/_ref). Bit-exact."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "reference_helpers.npz"))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_tea_and_rng_streams(orc, gold):
    u = orc.oracle_units()
    got = np.array([u.tea4(int(a), int(b)) for a, b in zip(gold["tea_v0"], gold["tea_v1"])], np.uint32)
    assert np.array_equal(got, gold["tea_out"])
    for seed, stream, final in zip(gold["rng_seeds"], gold["rng_streams"], gold["rng_final"]):
        vals, fin = u.rng_stream(int(seed), 16)
        assert np.array_equal(_bits(vals), _bits(stream)) and fin == final
        assert (vals >= 0).all() and (vals < 1).all()


def test_vector_helpers(orc, gold):
    u = orc.oracle_units()
    for op in range(6):
        got = np.array([u.vec3(op, a, b) for a, b in zip(gold["vec_a"], gold["vec_b"])], np.float32)
        assert np.array_equal(_bits(got), _bits(gold[f"vec_op{op}"])), f"vec3 op {op}"


def test_refract_and_tbn(orc, gold):
    u = orc.oracle_units()
    res = [u.refract(i, n, float(e)) for i, n, e in zip(gold["refract_i"], gold["refract_n"], gold["refract_ior"])]
    assert np.array_equal(np.array([r[0] for r in res], np.int32), gold["refract_ok"])
    assert (gold["refract_ok"] == 0).any(), "fixture must contain total internal reflection"
    assert np.array_equal(_bits(np.array([r[1] for r in res])), _bits(gold["refract_r"]))
    tbn = np.array([u.tbn(i, n) for i, n in zip(gold["refract_i"], gold["refract_n"])], np.float32)
    assert np.array_equal(_bits(tbn), _bits(gold["tbn"]))


def test_mesh_generators(twk, gold):
    pi = np.float32(np.pi)
    small = {
        "plane_1_1_0": twk.mesh_plane(1, 1, 0), "plane_1_1_1": twk.mesh_plane(1, 1, 1), "plane_1_1_2": twk.mesh_plane(1, 1, 2),
        "plane_3_2_1": twk.mesh_plane(3, 2, 1), "box": twk.mesh_box(), "sphere_8_5_1": twk.mesh_sphere(8, 5, 1.0, pi),
        "sphere_6_4_half": twk.mesh_sphere(6, 4, 1.0, np.float32(0.5 * pi)), "torus_6_5": twk.mesh_torus(6, 5, 0.75, 0.25),
        "parallelogram_light1": twk.mesh_parallelogram((-0.5, 1.95, -0.5), (1, 0, 0), (0, 0, 1), (0, -1, 0)),
    }
    for k, (attr, idx) in small.items():
        assert np.array_equal(_bits(attr), _bits(gold[f"mesh_{k}_attr"])), k
        assert np.array_equal(idx, gold[f"mesh_{k}_idx"]), k
    big = {"sphere_180_90_1": twk.mesh_sphere(180, 90, 1.0, pi), "torus_180_180": twk.mesh_torus(180, 180, 0.75, 0.25)}
    for k, (attr, idx) in big.items():
        assert list(gold[f"meshshape_{k}"]) == [attr.shape[0], idx.shape[0]]
        assert [_sha(attr), _sha(idx)] == list(gold[f"meshsha_{k}"]), k
    # Appendix B sizes: sphere (U+1) V verts, 2 U (V-1) tris; torus (U+1)(V+1), 2 U V
    assert big["sphere_180_90_1"][0].shape[0] == 16290 and big["sphere_180_90_1"][1].shape[0] == 3 * 32040
    assert big["torus_180_180"][0].shape[0] == 32761 and big["torus_180_180"][1].shape[0] == 3 * 64800


def test_mesh_generator_rejects_bad_tessellation(twk):
    for call in (lambda: twk.mesh_plane(0, 1, 1), lambda: twk.mesh_plane(1, 1, 3), lambda: twk.mesh_sphere(2, 90, 1, 3.14), lambda: twk.mesh_torus(180, 2, 0.75, 0.25)):
        with pytest.raises(twk.TwkError):
            call()


def test_camera_frustum(twk, gold):
    for cin, cout in zip(gold["camera_in"], gold["camera_out"]):
        c = twk.camera_frustum(cin[:3], cin[3], cin[4], cin[5], cin[6], float(np.float32(int(cin[7])) / np.float32(int(cin[8]))))
        got = np.array(list(c.P) + list(c.U) + list(c.V) + list(c.W), np.float32)
        assert np.array_equal(_bits(got), _bits(cout))


def test_transform_stack_matches_dp_math(twk, gold):
    kinds = {0: "rotate", 1: "scale", 2: "translate"}
    ops = gold["xform_ops"]
    pos = 0
    for count, expect in zip(gold["xform_count"], gold["xform_out"]):
        lines = ["albedo 1 1 1", "material m brdf_diffuse"]
        for op in ops[pos:pos + count]:
            k = int(op[0])
            args = op[1:5] if k == 0 else op[1:4]
            lines.append(kinds[k] + " " + " ".join(repr(float(a)) for a in args))
        lines.append("model box m")
        pos += count
        app = twk.Application(system_text="light 0\nmiss 0\nresolution 8 8\n", scene_text="\n".join(lines))
        g, t, m, l = app.instance(0)
        assert np.array_equal(_bits(t), _bits(expect)), (lines, t, expect)


def test_parser_token_stream(twk, gold):
    """Product tokenizer == the REFERENCE Parser's token stream, on every shipped description file and on a text
    with comments, CR/LF, signed/exponent values and identifier look-alikes."""
    for name in sorted(os.listdir(SCENES)):
        toks = list(gold[f"tokens_{name}"])
        assert len(toks) > 10
        assert all(t[0] in "12" for t in toks), "only identifiers and values in well-formed files"
        with open(os.path.join(SCENES, name), newline="") as f:
            got = [f"{t} {s}" for t, s in twk.parse_tokens(f.read())]
        assert got == toks, name
    tricky = list(gold["tokens_tricky"])
    assert [f"{t} {s}" for t, s in twk.parse_tokens(str(gold["tokens_tricky_text"][0]))] == tricky
    # values: 1 -2.5e3 +.5 ; identifiers: a, model, assimp, my, file, name.obj, 1abc, tab, sep ; 'e5' is an ID (starts with e), '.e' '-e' '5e-' are values
    assert tricky[:4] == ["1 a", "2 1", "2 -2.5e3", "2 +.5"]
    assert "1 1abc" in tricky and "1 e5" in tricky and "2 .e" in tricky and "2 -e" in tricky and "2 5e-" in tricky
    assert not any("comment" in t for t in tricky)
