"""CPU: the oracle against the committed golden vectors (tests/golden/oracle_cornell.npz) and against itself
(BVH mode == brute force), plus edge cases of the optixTrace contract."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_app


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "oracle_cornell.npz"))


def test_c1_image_and_first_hits_match_golden(twk, orc, gold):
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (64, 64))
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)          # BVH mode; the golden was made by brute force
    o.captureFirstHits(True)
    o.render(0)
    assert np.array_equal(_bits(o.getOutputBufferHost()), _bits(gold["c1_64_spp1"]))
    tbg, ids = o.readFirstHits()
    assert np.array_equal(ids, gold["c1_64_firsthit_ids"])
    hit = ids[:, 0] >= 0
    assert np.array_equal(_bits(tbg[hit]), _bits(gold["c1_64_firsthit_tbg"][hit]))
    o.render(1)
    assert np.array_equal(_bits(o.getOutputBufferHost()), _bits(gold["c1_64_spp2"]))


def test_c2_image_matches_golden(twk, orc, gold):
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (64, 36))
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)
    for it in range(2):
        o.render(it)
    img = o.getOutputBufferHost()
    assert np.array_equal(_bits(img), _bits(gold["c2_64x36_spp2"]))
    assert (img[..., 3] == 1.0).all() and np.isfinite(img).all()


def test_object_space_policy_matches_the_round1_golden(twk, orc):
    """Flatten policy (0, 0) = every instance intersected in object space through the inverse transform (an OptiX
    IAS -> GAS descent, Device.cpp:1427-1489): the oracle under that policy reproduces, bit for bit, the fixture round 1
    committed before the flatten policy existed (tests/golden/oracle_cornell_objectspace.npz). The default-policy images
    stay within 1e-2 per channel of it (the -m gpu tolerance test bounds the device's default against this oracle)."""
    gold = np.load(os.path.join(GOLDEN, "oracle_cornell_objectspace.npz"))
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (64, 64))
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)
    o.setFlattenPolicy(0, 0)
    o.captureFirstHits(True)
    o.render(0)
    assert np.array_equal(_bits(o.getOutputBufferHost()), _bits(gold["c1_64_spp1"]))
    tbg, ids = o.readFirstHits()
    hit = ids[:, 0] >= 0
    assert np.array_equal(ids, gold["c1_64_firsthit_ids"]) and np.array_equal(_bits(tbg[hit]), _bits(gold["c1_64_firsthit_tbg"][hit]))
    o.render(1)
    assert np.array_equal(_bits(o.getOutputBufferHost()), _bits(gold["c1_64_spp2"]))
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (64, 36))
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)
    o.setFlattenPolicy(0, 0)
    for it in range(2):
        o.render(it)
    img = o.getOutputBufferHost()
    assert np.array_equal(_bits(img), _bits(gold["c2_64x36_spp2"]))
    default = np.load(os.path.join(GOLDEN, "oracle_cornell.npz"))
    assert not np.array_equal(_bits(default["c2_64x36_spp2"]), _bits(gold["c2_64x36_spp2"]))
    assert np.abs(default["c2_64x36_spp2"] - gold["c2_64x36_spp2"]).max() < 1e-2


def test_running_mean_is_the_reference_lerp(twk, orc, gold):
    """raygeneration.cu:246-253: dst + (x - dst) / (i + 1) in float, not sum / n."""
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (64, 64))
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)
    o.render(1)  # iteration 1 alone lerps against a zero buffer: out = 0 + 0.5 * (x - 0)
    half = o.getOutputBufferHost()[..., :3]
    a = gold["c1_64_spp1"][..., :3]
    both = gold["c1_64_spp2"][..., :3]
    x1 = half / np.float32(0.5)
    expect = a + np.float32(0.5) * (x1 - a)
    assert np.array_equal(_bits(expect), _bits(both))


def test_trace_contract_edge_cases(twk, orc):
    o = orc.Oracle(miss=0)
    attr, idx = twk.mesh_plane(1, 1, 1)  # y = 0, x,z in [-1,1], normal +y
    g = o.addGeometry(attr, idx)
    ident = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    o.addInstance(g, ident, 0)
    o.addInstance(g, [2, 0, 0, 0, 0, 1, 0, -1, 0, 0, 2, 0], 0)  # scaled copy at y = -1
    rays = np.array([
        [0.25, 1, 0.25, 0, 0, -1, 0, 1e27],     # hits instance 0 at t = 1
        [0.25, 1, 0.25, 0, 0, -1, 0, 1.0],      # tmax exclusive: falls through to ... nothing closer than 1 → miss
        [0.25, 1, 0.25, 1.0, 0, -1, 0, 1e27],   # tmin exclusive: skips t = 1, hits the copy at t = 2
        [1.5, 1, 0.0, 0, 0, -1, 0, 1e27],       # outside the small plane, inside the scaled one
        [0.25, -2, 0.25, 0, 0, 1, 0, 1e27],     # from below: back faces are hit (no culling), closest is the copy
        [5, 1, 5, 0, 0, -1, 0, 1e27],           # miss
        [0.0, 1, 0.0, 0, 0, -1, 0, 1e27],       # exactly on the shared diagonal edge: watertight, lower primitive wins
        [0.25, 1, 0.25, 0, 1, 0, 0, 1e27],      # parallel to the plane
    ], np.float32)
    for mode in (True, False):
        o.setTraceMode(mode)
        tbg, ids = o.traceRays(rays)
        assert ids[0].tolist() == [0, 1] or ids[0].tolist() == [0, 0]
        assert tbg[0, 0] == 1.0
        assert ids[1, 0] == -1
        assert ids[2, 0] == 1 and tbg[2, 0] == 2.0
        assert ids[3, 0] == 1
        assert ids[4, 0] == 1 and tbg[4, 0] == 1.0
        assert ids[5, 0] == -1
        assert ids[6].tolist() == [0, 0] and tbg[6, 0] == 1.0
        assert ids[7, 0] == -1
        occ = o.traceRays(rays, anyHit=True)[1][:, 0]
        assert occ.tolist() == [1, 0, 1, 1, 1, 0, 1, 0]
    # barycentrics: beta weights vertex 1, gamma vertex 2 (closesthit.cu:142-147)
    tbg, ids = o.traceRays(rays[:1])
    tri = idx.reshape(-1, 3)[ids[0, 1]]
    b, c = tbg[0, 1], tbg[0, 2]
    p = attr[tri[0], :3] * (1 - b - c) + attr[tri[1], :3] * b + attr[tri[2], :3] * c
    assert np.allclose(p, [0.25, 0, 0.25], atol=1e-6)


def test_bvh_mode_equals_brute_force_on_random_rays(twk, orc):
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (16, 16))
    o = orc.Oracle(miss=app.info.miss)
    o.loadApplication(app)
    rng = np.random.default_rng(11)
    n = 600
    org = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32)
    org[:, 1] += 1.0
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 5e-5, np.float32), d, np.full((n, 1), 1e27, np.float32)], 1).astype(np.float32)
    o.setTraceMode(True)
    a = o.traceRays(rays)
    o.setTraceMode(False)
    b = o.traceRays(rays)
    assert np.array_equal(a[1], b[1])
    hit = a[1][:, 0] >= 0
    assert np.array_equal(_bits(a[0][hit]), _bits(b[0][hit]))


def test_threaded_render_is_the_same_render(twk, orc):
    """orc_render_rect_threads (bench.py's CPU baseline): rows on several host threads give the image and the ray
    tallies of the single-threaded call."""
    from conftest import load_app
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt", (96, 54))
    out = []
    for threads in (1, 5):
        ref = orc.Oracle(miss=app.info.miss)
        ref.loadApplication(app)
        for it in range(2):
            ref.render(it, threads=threads)
        out.append((ref.getOutputBufferHost(), ref.counters()))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert out[0][1] == out[1][1] and out[0][1]["samples"] == 2 * 96 * 54


def test_device_built_tree_fixture_walked_on_the_cpu(twk, orc):
    """tests/golden/wide4_small_room.npz (made on the GPU box by tests/golden/make_wide4_fixture.py): the quantised 4-ary nodes and
    triangle slots twk_build produced for a small Cornell room, 4 096 rays, and the persistent kernel's hit records and visit counts
    for them. Here, without a GPU: the host walker (oracle/same_bvh_walk.cpp orc_walk_same_bvh) walks that tree — the same hit
    records bit for bit, the same visit counts (up to the reciprocal of the culling test: v_rcp_f32 there, 1 / d here) — and the
    oracle's brute force over the same scene, rebuilt here from its description, gives the same hits: the node format and the
    leaf numbering of csrc/bvh_build.hip are pinned on the CPU side too."""
    import importlib.util
    fx = np.load(os.path.join(GOLDEN, "wide4_small_room.npz"))
    nodes, tris, rays = fx["nodes"], fx["triangles"], fx["rays"]
    assert nodes.shape[1] == 16 and tris.shape[1] == 12
    acc = ({"root": int(fx["root"]), "root2": int(fx["root2"]), "nodeFloats": nodes.shape[1]}, nodes, tris, np.zeros((1, 32), np.float32))
    tbg, ids, counts = orc.walk_same_bvh(acc, rays)
    assert np.array_equal(ids[:, 0], fx["device_instance"]) and np.array_equal(ids[:, 1], fx["device_primitive"])
    hit = ids[:, 0] >= 0
    assert 0.7 < hit.mean() < 1.0
    assert np.array_equal(_bits(tbg[hit]), _bits(fx["device_tbg"][hit]))
    assert abs(int(counts["nodesVisited"]) - int(fx["device_nodes_visited"])) <= 4 and abs(int(counts["trianglesTested"]) - int(fx["device_triangles_tested"])) <= 4, counts
    # any-hit walk: occlusion of the same rays cut short
    shadow = rays.copy()
    shadow[:, 7] = np.random.default_rng(3).uniform(0.1, 3.0, rays.shape[0]).astype(np.float32)
    _, occ, _ = orc.walk_same_bvh(acc, shadow, anyHit=True)
    # the scene the fixture was built from, from its description, through the oracle's brute force
    spec = importlib.util.spec_from_file_location("make_wide4_fixture", os.path.join(GOLDEN, "make_wide4_fixture.py"))
    # (the generator module imports the product library, which loads without a GPU; only its scene text and rays are used here)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert np.array_equal(_bits(gen.fixture_rays()), _bits(rays))
    app = twk.Application(system_text=gen.SYSTEM, scene_text=gen.SCENE)
    app.setResolution(32, 32)
    ref = orc.Oracle(miss=app.info.miss)
    ref.loadApplication(app)
    ref.setTraceMode(False)
    o_tbg, o_ids = ref.traceRays(rays)
    assert np.array_equal(o_ids, ids) and np.array_equal(_bits(o_tbg[hit]), _bits(tbg[hit]))
    _, s_ids = ref.traceRays(shadow, anyHit=True)
    assert np.array_equal(s_ids[:, 0], occ[:, 0])
