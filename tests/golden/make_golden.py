#!/usr/bin/env python3
"""Writes the committed fixtures under tests/golden/. Run in the build container (needs oracle/_ref, i.e.
/root/reference, for the reference-derived part):

    python tests/golden/make_golden.py

reference_helpers.npz — outputs of the REFERENCE's own code (oracle/_ref/libref_host.so, compiled from
  /root/reference/apps/rtigo3 by oracle/Makefile): tea<4>/rng streams, refract/TBN/vector-math tables, mesh
  generator dumps (full arrays for the small meshes, SHA-256 + samples for the large ones), Camera::getFrustum,
  the loader's transform stack, Parser token streams. These pin the oracle's helper layer and the product's host
  scene layer on machines where the reference is absent (GPU box).
oracle_cornell.npz — outputs of the ORACLE (brute-force traversal): C1 Cornell box 64x64 first-hit records and
  images at 1 and 2 spp, C2 (full BSDF set) 64x36 image at 2 spp. The reference has no golden images
  (SURVEY.md §4); these pin the oracle and the HIP path against regressions and against each other.
oracle_cornell_objectspace.npz — the same arrays as oracle_cornell.npz made with the flatten policy (0, 0): every instance
  intersected in OBJECT space with the ray taken through the inverse instance transform, which is what an OptiX
  IAS -> GAS descent does (Device.cpp:1427-1489). Byte-identical to the file round 1 committed as oracle_cornell.npz
  (commit a929824) before the flatten policy existed; the default-policy file differs from it by <= 8e-3 per channel.
Fixtures are data only: inputs + expected outputs.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def reference_helpers():
    from oracle import orc
    ref = orc.Reference()
    rng = np.random.default_rng(20261004)
    out = {}

    # tea<4> and the LCG stream
    v0 = rng.integers(0, 2**32, 256, dtype=np.uint64).astype(np.uint32)
    v1 = rng.integers(0, 2**16, 256, dtype=np.uint64).astype(np.uint32)
    v0[:4] = [0, 1, 1920 * 1079 + 1919, 0xFFFFFFFF]
    v1[:4] = [0, 0, 63, 0xFFFFFFFF]
    out["tea_v0"], out["tea_v1"] = v0, v1
    out["tea_out"] = np.array([ref.tea4(int(a), int(b)) for a, b in zip(v0, v1)], np.uint32)
    seeds = np.array([0, 1, 0xDEADBEEF, 0xFFFFFFFF, out["tea_out"][2]], np.uint32)
    streams, finals = [], []
    for s in seeds:
        vals, fin = ref.rng_stream(int(s), 16)
        streams.append(vals)
        finals.append(fin)
    out["rng_seeds"], out["rng_streams"], out["rng_final"] = seeds, np.array(streams, np.float32), np.array(finals, np.uint32)

    # vector helpers: normalize, reflect, cross, dot, length, (powerHeuristic, intensity)
    a = rng.normal(size=(128, 3)).astype(np.float32)
    b = rng.normal(size=(128, 3)).astype(np.float32)
    out["vec_a"], out["vec_b"] = a, b
    for op in range(6):
        out[f"vec_op{op}"] = np.array([ref.vec3(op, x, y) for x, y in zip(a, b)], np.float32)

    # refract (incl. total internal reflection) and TBN
    i = a / np.linalg.norm(a, axis=1, keepdims=True)
    n = b / np.linalg.norm(b, axis=1, keepdims=True)
    ior = rng.choice(np.array([1.5, 1.0 / 1.5, 1.33, 2.4, 1.0], np.float32), 128)
    res = [ref.refract(x, y, float(e)) for x, y, e in zip(i, n, ior)]
    out["refract_i"], out["refract_n"], out["refract_ior"] = i.astype(np.float32), n.astype(np.float32), ior
    out["refract_ok"] = np.array([r[0] for r in res], np.int32)
    out["refract_r"] = np.array([r[1] for r in res], np.float32)
    out["tbn"] = np.array([ref.tbn(x, y) for x, y in zip(i, n)], np.float32)

    # meshes
    small = {
        "plane_1_1_0": ref.mesh_plane(1, 1, 0), "plane_1_1_1": ref.mesh_plane(1, 1, 1), "plane_1_1_2": ref.mesh_plane(1, 1, 2),
        "plane_3_2_1": ref.mesh_plane(3, 2, 1), "box": ref.mesh_box(), "sphere_8_5_1": ref.mesh_sphere(8, 5, 1.0, np.float32(np.pi)),
        "sphere_6_4_half": ref.mesh_sphere(6, 4, 1.0, np.float32(0.5 * np.float32(np.pi))), "torus_6_5": ref.mesh_torus(6, 5, 0.75, 0.25),
        "parallelogram_light1": ref.mesh_parallelogram((-0.5, 1.95, -0.5), (1, 0, 0), (0, 0, 1), (0, -1, 0)),
    }
    for k, (attr, idx) in small.items():
        out[f"mesh_{k}_attr"], out[f"mesh_{k}_idx"] = attr, idx
    big = {"sphere_180_90_1": ref.mesh_sphere(180, 90, 1.0, np.float32(np.pi)), "torus_180_180": ref.mesh_torus(180, 180, 0.75, 0.25)}
    for k, (attr, idx) in big.items():
        out[f"meshsha_{k}"] = np.array([sha(attr), sha(idx)])
        out[f"meshshape_{k}"] = np.array([attr.shape[0], idx.shape[0]], np.int64)
        out[f"meshsample_{k}"] = attr[::997].copy()

    # Camera::getFrustum
    cams = np.array([[0, 1, 0, 0.75, 0.5, 45, 3.41, 1920, 1080], [0, 1, 0, 0.75, 0.5, 45, 3.41, 512, 512],
                     [0, 0, 0, 0.75, 0.6, 60, 10, 1920, 1080], [1, 2, 3, 0.1, 0.9, 30, 5, 640, 480]], np.float32)
    out["camera_in"] = cams
    out["camera_out"] = np.array([ref.camera_frustum(c[:3], c[3], c[4], c[5], c[6], int(c[7]), int(c[8])) for c in cams], np.float32)

    # transform stack (kind, a, b, c, d): 0 rotate, 1 scale, 2 translate
    stacks = [
        [[1, 0.5, 0.5, 0.5, 0], [2, -0.4, 0.5, -0.25, 0]],
        [[0, 1, 0, 0, 180], [2, 0, 2, 0, 0]],
        [[0, 0, 1, 0, 180], [2, 1, 1, 0, 0]],
        [[0, 1, 1, 0, 33.3], [1, 2, 0.5, 1.5, 0], [2, 1, -2, 3, 0], [0, 0.2, -0.7, 0.4, -75]],
        [[2, 1, 2, 3, 0], [0, 0, 0, 1, 90], [1, 3, 3, 3, 0]],
    ]
    out["xform_count"] = np.array([len(s) for s in stacks], np.int32)
    out["xform_ops"] = np.array([op for s in stacks for op in s], np.float32)
    out["xform_out"] = np.array([ref.transform_stack(np.array(s, np.float32)) for s in stacks], np.float32)

    # Parser tokens of the scene files shipped in scenes/
    for name in sorted(os.listdir(os.path.join(ROOT, "scenes"))):
        toks = ref.parse_tokens(os.path.join(ROOT, "scenes", name))
        out[f"tokens_{name}"] = np.array([f"{t} {s}" for t, s in toks])
    # a file exercising comments, CR/LF, values and odd identifiers
    tricky = "a 1 -2.5e3 +.5 # comment with 1 2 3\r\nmodel assimp  my file name.obj  \r\n1abc e5 .e -e 5e- \t tab\tsep\n#only comment"
    path = "/tmp/twk_tricky_tokens.txt"
    with open(path, "w", newline="") as f:
        f.write(tricky)
    out["tokens_tricky_text"] = np.array([tricky])
    out["tokens_tricky"] = np.array([f"{t} {s}" for t, s in ref.parse_tokens(path)])

    np.savez_compressed(os.path.join(HERE, "reference_helpers.npz"), **out)
    print("wrote reference_helpers.npz", len(out), "arrays")


def oracle_cornell(policy=None, name="oracle_cornell.npz"):
    import tweeker_raytracer_amd as twk
    from oracle import orc
    out = {}
    scenes = os.path.join(ROOT, "scenes")

    app = twk.Application(os.path.join(scenes, "system_rtigo3_cornell_box_c1.txt"), os.path.join(scenes, "scene_rtigo3_cornell_box_c1.txt"))
    app.setResolution(64, 64)
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)
    if policy is not None:
        o.setFlattenPolicy(*policy)
    o.setTraceMode(False)  # brute force: the definition
    o.captureFirstHits(True)
    o.render(0)
    out["c1_64_spp1"] = o.getOutputBufferHost()
    tbg, ids = o.readFirstHits()
    out["c1_64_firsthit_tbg"], out["c1_64_firsthit_ids"] = tbg, ids
    o.render(1)
    out["c1_64_spp2"] = o.getOutputBufferHost()

    app = twk.Application(os.path.join(scenes, "system_rtigo3_cornell_box.txt"), os.path.join(scenes, "scene_rtigo3_cornell_box.txt"))
    app.setResolution(64, 36)
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)
    if policy is not None:
        o.setFlattenPolicy(*policy)
    o.setTraceMode(False)
    for it in range(2):
        o.render(it)
    out["c2_64x36_spp2"] = o.getOutputBufferHost()
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name)


if __name__ == "__main__":
    reference_helpers()
    oracle_cornell()
    oracle_cornell(policy=(0, 0), name="oracle_cornell_objectspace.npz")
