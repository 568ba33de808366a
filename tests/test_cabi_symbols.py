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
    assert _lib.lib.twk_abi_version() == 4


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


def test_every_entry_point_is_a_function_try_block():
    """No C++ exception may cross the extern "C" boundary (the reference throws std::runtime_error, inc/CheckMacros.h:38-80;
    a C caller cannot catch it): every definition of an `int twk_*(...)` entry point in the two files that hold them is a
    function-try-block ending in TWK_CATCH("<its own name>"), except the few one-liners that cannot throw."""
    cannot_throw = {"twk_abi_version", "twk_app_destroy"}  # return a constant / delete a pointer
    seen = set()
    for rel in ("csrc/device_api.hip", "csrc/host/host_cabi.cpp"):
        text = open(os.path.join(ROOT, "tweeker_raytracer_amd", rel)).read()
        for m in re.finditer(r"^int (twk_[a-z0-9_]+)\s*\(([^)]*)\)\s*\n(\S+)", text, flags=re.M):
            name, following = m.group(1), m.group(3)
            seen.add(name)
            if name in cannot_throw:
                continue
            assert following == "try", f"{rel}: {name} is not a function-try-block"
            assert f'TWK_CATCH("{name}")' in text, f"{rel}: {name} has no TWK_CATCH of its own"
        for m in re.finditer(r"^int (twk_[a-z0-9_]+)\s*\([^)]*\)\s*\{", text, flags=re.M):  # one-line definitions
            seen.add(m.group(1))
            assert m.group(1) in cannot_throw, f"{rel}: {m.group(1)} is defined without a try block"
    declared = set(_header_functions()) - {"twk_last_error"}
    assert declared <= seen, declared - seen
