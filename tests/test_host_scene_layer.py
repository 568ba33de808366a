"""CPU: the C++ host scene layer (description grammar → cameras, lights, materials, geometries, flattened instances),
mirroring what rtigo3's Application builds (Application.cpp:1046-1299, 572-677, 1397-1878) and Device::traverseNode
flattens (Device.cpp:1283-1331). No GPU calls."""
import numpy as np
import pytest

from conftest import load_app

SYS = "resolution 64 48\nlight 1\nmiss 0\npathLengths 2 5\ncenter 0 1 0\ncamera 0.75 0.5 45 3.41\n"


def test_cornell_box_scene_contents(twk):
    app = load_app(twk, "system_rtigo3_cornell_box.txt", "scene_rtigo3_cornell_box.txt")
    i = app.info
    assert list(i.resolution) == [1920, 1080] and i.samplesSqrt == 8 and list(i.pathLengths) == [2, 10]
    assert i.light == 1 and i.miss == 0 and i.strategy == 0 and list(i.tileSize) == [8, 8]
    # material 0 is the area light's black thin-walled mirror, created before the scene file is read (Application.cpp:640-659)
    mats = app.materials
    assert i.numMaterials == 8 and mats[0].indexBSDF == 1 and mats[0].thinwalled == 1 and list(mats[0].albedo) == [0, 0, 0]
    assert [m.indexBSDF for m in mats[1:]] == [0, 0, 0, 0, 3, 1, 2]
    assert mats[7].absorptionScale == 1.0 and abs(mats[7].ior - 1.5) < 1e-7
    # light 1: 1x1 parallelogram at y = 1.95 facing down, emission 10
    (l,) = app.lights
    assert l.type == 1 and list(l.position) == [-0.5, np.float32(1.95), -0.5] and list(l.normal) == [0, -1, 0] and l.area == 1.0 and list(l.emission) == [10, 10, 10]
    # geometries: light quad, plane y-up, plane z-up, plane x-up, ONE sphere shared by two instances
    assert i.numGeometries == 5 and i.numInstances == 8
    inst = app.instances
    assert [k[0] for k in inst] == [0, 1, 1, 2, 3, 3, 4, 4]
    assert inst[0][3] == 0 and all(k[3] == -1 for k in inst[1:])  # only the light instance carries a light index
    assert sum(app.geometry(g)[1].shape[0] // 3 for g, *_ in inst) == 2 + 5 * 2 + 2 * 32040
    # ceiling: rotate 180 about x then translate (0,2,0): normal (0,1,0) maps to (0,-1,0)
    t = inst[2][1].reshape(3, 4)
    assert np.allclose(t[:, :3] @ np.array([0, 1, 0]), [0, -1, 0], atol=1e-6) and np.allclose(t[:, 3], [0, 2, 0])


def test_state_and_camera(twk):
    app = twk.Application(system_text=SYS, scene_text="material m brdf_diffuse\nmodel box m\n")
    st = app.state
    assert list(st.resolution) == [64, 48] and st.distribution == 0 and st.epsilonFactor == 500.0 and st.lensShader == 0
    c0 = app.cameras[0]
    app.setResolution(128, 48)
    c1 = app.cameras[0]
    assert np.allclose(np.array(list(c1.U)), 2 * np.array(list(c0.U))) and list(c1.V) == list(c0.V) and list(c1.P) == list(c0.P)


def test_system_description_rules(twk):
    app = twk.Application(system_text="tileSize 6 16\nlight 7\nsamplesSqrt 0\nlensShader 9\nresolution 0 -3\nstrategy 9\nunknownKey 3\nenvMap my env file.hdr\nmiss 2\n",
                          scene_text="material m brdf_diffuse\nmodel box m\n")
    i = app.info
    assert list(i.tileSize) == [8, 16]      # non power of two falls back to 8 (Application.cpp:1123-1133)
    assert i.light == 2                     # clamped to [0, 2] (:1172-1180)
    assert i.samplesSqrt == 1 and i.lensShader == 0 and list(i.resolution) == [1, 1] and i.strategy == 0
    assert i.miss == 2 and i.numLights == 2  # environment light first, then the 4x4 area light
    ls = app.lights
    assert ls[0].type == 0 and ls[1].type == 1 and ls[1].area == 16.0 and list(ls[1].position) == [-2, 4, -2]


def test_scene_description_rules(twk):
    scene = """
    albedo 0.5 0.25 1
    roughness 0.3 0.4
    material a brdf_ggx_smith
    ior 1.33 thinwalled 1 absorption 0.1 0.2 0.3 absorptionScale 2
    material b bsdf_specular
    material a brdf_diffuse          # duplicate name: the last one wins
    bogusKeyword 1 2 3
    pop                              # pop on an empty stack resets to identity
    model box a
    push translate 1 2 3 model box nosuchmaterial pop   # unknown reference → 'default' … which does not exist here
    model sphere 8 5 1.0 b
    model sphere 8 5 1 b             # same key "sphere_8_5_1" → shared geometry
    model sphere 8 5 0.5 b           # different theta → new geometry
    model torus 6 5 0.75 0.25 b
    model plane 2 2 1 a
    """
    with pytest.raises(twk.TwkError):
        twk.Application(system_text="light 0\nmiss 0\n", scene_text=scene)  # an instance without material is an error at build
    scene = scene.replace("nosuchmaterial", "b")
    app = twk.Application(system_text="light 0\nmiss 0\n", scene_text=scene)
    i = app.info
    assert i.numLights == 0 and i.numMaterials == 3
    m = app.materials
    assert m[0].indexBSDF == 3 and np.allclose(list(m[0].roughness), [0.3, 0.4]) and np.allclose(list(m[0].albedo), [0.5, 0.25, 1])
    assert m[1].indexBSDF == 2 and m[1].thinwalled == 1 and m[1].absorptionScale == 2.0
    assert m[2].indexBSDF == 0
    inst = app.instances
    assert [k[2] for k in inst] == [2, 1, 1, 1, 1, 1, 2]       # 'a' resolves to the LAST material named a
    assert [k[0] for k in inst] == [0, 0, 1, 1, 2, 3, 4]       # box shared, sphere_8_5_1 shared, half sphere / torus / plane new
    assert np.allclose(inst[1][1].reshape(3, 4)[:, 3], [1, 2, 3])
    assert app.geometry(4)[0].shape[0] == 9 and app.geometry(4)[1].shape[0] == 24


def test_parse_errors_are_reported_not_crashed(twk):
    with pytest.raises(twk.TwkError) as e:
        twk.Application(system_text="resolution 64 abc\n", scene_text="")
    assert "resolution" in str(e.value)
    with pytest.raises(twk.TwkError):
        twk.Application(system_text="", scene_text="material m brdf_diffuse\nscale 1 x 1\n")
    with pytest.raises(twk.TwkError):
        twk.Application("/nonexistent/system.txt", "/nonexistent/scene.txt")
    with pytest.raises(twk.TwkError):
        twk.Application(system_text="", scene_text="")  # no materials


def test_tile_map_is_a_partition(twk):
    """distribute() (raygeneration.cu:152-164): over all devices every pixel column of every row is produced exactly once."""
    for (w, h, n, tile) in [(200, 24, 3, (8, 8)), (1920, 16, 8, (8, 8)), (64, 40, 2, (16, 4)), (37, 9, 4, (8, 8))]:
        lw = twk.launch_width(w, tile[0], n)
        assert lw % tile[0] == 0 and lw * n >= w
        seen = np.zeros((h, w), np.int32)
        for d in range(n):
            for y in range(h):
                for x in range(lw):
                    px = twk.tile_column(x, y, tile, n, d)
                    if px < w:
                        seen[y, px] += 1
        assert (seen == 1).all()
    assert twk.launch_width(3840, 8, 8) == 480  # C5
    with pytest.raises(twk.TwkError):
        twk.tile_column(0, 0, (6, 8), 2, 0)


def test_system_description_round_trip(twk):
    """≙ Application::saveSystemDescription (Application.cpp:1300-1345): keys in the reference's order, and loading the
    text back gives the same settings (numbers go through operator<<: six significant digits, like the reference)."""
    from conftest import scene_path
    scene = open(scene_path("scene_rtigo3_cornell_box_c1.txt")).read()
    src = ("strategy 2\ndevicesMask 15\ninterop 1\npresent 1\nresolution 640 360\ntileSize 16 8\nsamplesSqrt 3\nmiss 2\nenvMap sky.hdr\n"
           "envRotation 0.25\nclockFactor 250\nlight 2\npathLengths 3 7\nepsilonFactor 800\nlensShader 1\ncenter 0.5 1.25 -2\n"
           "camera 0.7 0.45 50 4.5\nprefixScreenshot ./out/shot\ngamma 2.2\ncolorBalance 1 0.9 0.8\nwhitePoint 1.5\nburnHighlights 0.8\n"
           "crushBlacks 0.2\nsaturation 1.2\nbrightness 0.7\n")
    app = twk.Application(system_text=src, scene_text=scene)
    text = app.systemDescription()
    assert text == src
    again = twk.Application(system_text=text, scene_text=scene)
    a, b = app.info, again.info
    for name, _ in a._fields_:
        va, vb = getattr(a, name), getattr(b, name)
        assert (list(va) == list(vb)) if hasattr(va, "__len__") else (va == vb), name
    ta, tb = app.tonemapper, again.tonemapper
    assert bytes(ta) == bytes(tb) and again.environment == "sky.hdr"
    # defaults: no envMap line, neutral tonemapper
    text = twk.Application(system_text="", scene_text=scene).systemDescription()
    assert "envMap" not in text and text.splitlines()[0] == "strategy 0" and "gamma 1\n" in text and text.endswith("brightness 1\n")


def test_config_switches_as_description_keys(twk):
    """Grammar extensions for the reference's compile-time switches (shaders/config.h:50-56): `nextEventEstimation 0|1` (default 1)
    and `debugExceptions 0|1` (default 0); written back only when they differ from the defaults, so a reference file round-trips
    unchanged."""
    from conftest import scene_path
    scene = open(scene_path("scene_rtigo3_cornell_box_c1.txt")).read()
    plain = twk.Application(system_text="resolution 64 64\n", scene_text=scene)
    assert plain.info.nextEventEstimation == 1 and plain.info.debugExceptions == 0
    assert "nextEventEstimation" not in plain.systemDescription() and "debugExceptions" not in plain.systemDescription()
    app = twk.Application(system_text="resolution 64 64\nnextEventEstimation 0\ndebugExceptions 1\n", scene_text=scene)
    assert app.info.nextEventEstimation == 0 and app.info.debugExceptions == 1
    text = app.systemDescription()
    assert "nextEventEstimation 0\n" in text and "debugExceptions 1\n" in text
    again = twk.Application(system_text=text, scene_text=scene)
    assert again.info.nextEventEstimation == 0 and again.info.debugExceptions == 1
