"""CPU: libtweeker_hip.so loads without a GPU and exports exactly the functions include/tweeker_hip.h declares."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _header_functions():
    text = open(os.path.join(ROOT, "include", "tweeker_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(twk_[a-z0-9_]+)\s*\(", text)))


def test_header_python_binding_and_library_agree(twk):
    from tweeker_raytracer_amd import _lib
    declared = _header_functions()
    assert len(declared) >= 50
    assert sorted(_lib.SYMBOLS) == declared, set(_lib.SYMBOLS) ^ set(declared)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (twk_[a-z0-9_]+)$", out, flags=re.M)))
    assert exported == declared, set(exported) ^ set(declared)
    for name in declared:
        assert getattr(_lib.lib, name) is not None
    assert _lib.lib.twk_abi_version() == 3


def test_no_gpu_means_loud_failure_not_fallback(twk):
    """Without a HIP device twk_device_create must fail with TWK_ERROR_NO_DEVICE (there is no CPU path)."""
    try:
        n = twk.device_count()
    except twk.TwkError as e:
        assert e.code == 2
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(twk.TwkError) as e:
        twk.Device(ordinal=0)
    assert e.value.code == 2 and "no CPU path" in str(e.value)


def test_product_does_not_link_or_import_the_oracle():
    from tweeker_raytracer_amd import _lib
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "oracle" not in needed
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tweeker_raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "orc_" not in src and "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, os.path.join(dirpath, f)
