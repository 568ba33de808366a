"""CPU: the ctypes mirror of the C ABI's structs (tweeker_raytracer_amd/_lib.py) against the header itself: a C program
generated from include/tweeker_hip.h and compiled with gcc prints sizeof and every field's offsetof; both must equal what
ctypes lays out. (A field appended to a struct in the header but not in the mirror — as TwkBuildInfo grew in ABI 4 and 5 —
would otherwise only show as garbage in whatever follows it.)"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tweeker_raytracer_amd import _lib  # noqa: E402

PAIRS = [("TwkCameraDefinition", _lib.CameraDefinition), ("TwkLightDefinition", _lib.LightDefinition), ("TwkMaterialGUI", _lib.MaterialGUI),
         ("TwkTriangleAttributes", _lib.TriangleAttributes), ("TwkDeviceState", _lib.DeviceState), ("TwkTonemapper", _lib.Tonemapper),
         ("TwkLaunchStats", _lib.LaunchStats), ("TwkAccelerationInfo", _lib.AccelerationInfo), ("TwkBuildInfo", _lib.BuildInfo),
         ("TwkAppInfo", _lib.AppInfo)]


def _header_fields(text, name):
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("unsigned") else decl.split(None, 2)[2]
        for n in names.split(","):
            fields.append(re.sub(r"\[.*\]", "", n).strip().lstrip("*"))
    return fields


def test_ctypes_mirror_has_the_headers_layout(tmp_path):
    header = os.path.join(ROOT, "include", "tweeker_hip.h")
    text = open(header).read()
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{header}"', "int main(void) {"]
    for cname, _ in PAIRS:
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for f in _header_fields(text, cname):
            lines.append(f'  printf("{cname} {f} %zu\\n", offsetof({cname}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    c_layout = {}
    for line in out.splitlines():
        struct, field, value = line.split()
        c_layout.setdefault(struct, {})[field] = int(value)
    for cname, mirror in PAIRS:
        want = c_layout[cname]
        assert C.sizeof(mirror) == want["size"], (cname, C.sizeof(mirror), want["size"])
        got = {name: getattr(mirror, name).offset for name, *_ in mirror._fields_}
        c_fields = {k: v for k, v in want.items() if k != "size"}
        assert got == c_fields, (cname, got, c_fields)
