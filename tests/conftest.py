import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    lib = os.path.join(ROOT, "tweeker_raytracer_amd", "libtweeker_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__ as g
        g.build()


_ensure_built()


@pytest.fixture(scope="session")
def twk():
    import tweeker_raytracer_amd
    return tweeker_raytracer_amd


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    return o


def scene_path(name):
    return os.path.join(SCENES, name)


def load_app(twk, system, scene, resolution=None):
    app = twk.Application(scene_path(system), scene_path(scene))
    if resolution is not None:
        app.setResolution(*resolution)
    return app
