"""Application — Python mirror of the scene-building half of rtigo3's `Application`
(reference src/Application.cpp: loadSystemDescription :1046-1299, createLights :572-677,
loadSceneDescription :1397-1878). Parsing, mesh generation and flattening run in C++ inside
libtweeker_hip.so; this class only exposes the result.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L


class Application:
    def __init__(self, system_file=None, scene_file=None, system_text=None, scene_text=None):
        self._h = C.c_void_p()
        if system_text is not None or scene_text is not None:
            L.check(L.lib.twk_app_create_from_strings(C.byref(self._h), (system_text or "").encode(), (scene_text or "").encode()))
        else:
            L.check(L.lib.twk_app_create(C.byref(self._h), str(system_file).encode(), str(scene_file).encode()))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib.twk_app_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self):
        i = L.AppInfo()
        L.check(L.lib.twk_app_info(self._h, C.byref(i)))
        return i

    def setResolution(self, width, height):
        L.check(L.lib.twk_app_set_resolution(self._h, int(width), int(height)))

    @property
    def state(self):
        s = L.DeviceState()
        L.check(L.lib.twk_app_get_state(self._h, C.byref(s)))
        return s

    @property
    def cameras(self):
        n = self.info.numCameras
        arr = (L.CameraDefinition * max(1, n))()
        L.check(L.lib.twk_app_get_cameras(self._h, arr, n))
        return list(arr)[:n]

    @property
    def lights(self):
        n = self.info.numLights
        arr = (L.LightDefinition * max(1, n))()
        L.check(L.lib.twk_app_get_lights(self._h, arr, n))
        return list(arr)[:n]

    @property
    def materials(self):
        n = self.info.numMaterials
        arr = (L.MaterialGUI * max(1, n))()
        L.check(L.lib.twk_app_get_materials(self._h, arr, n))
        return list(arr)[:n]

    def geometry(self, idGeometry):
        na, ni = C.c_size_t(0), C.c_size_t(0)
        L.check(L.lib.twk_app_get_geometry_sizes(self._h, int(idGeometry), C.byref(na), C.byref(ni)))
        attr = np.empty((na.value, 12), dtype=np.float32)
        idx = np.empty((ni.value,), dtype=np.uint32)
        L.check(L.lib.twk_app_get_geometry(self._h, int(idGeometry), attr.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p)))
        return attr, idx

    def instance(self, idInstance):
        g, m, l = C.c_int(0), C.c_int(0), C.c_int(0)
        t = (C.c_float * 12)()
        L.check(L.lib.twk_app_get_instance(self._h, int(idInstance), C.byref(g), t, C.byref(m), C.byref(l)))
        return g.value, np.array(list(t), dtype=np.float32), m.value, l.value

    @property
    def instances(self):
        return [self.instance(i) for i in range(self.info.numInstances)]

    def systemDescription(self):
        """≙ Application::saveSystemDescription: the current settings as a system description text."""
        n = C.c_size_t(0)
        L.check(L.lib.twk_app_system_description(self._h, None, C.c_size_t(0), C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        L.check(L.lib.twk_app_system_description(self._h, buf, C.c_size_t(len(buf)), None))
        return buf.value.decode()

    @property
    def tonemapper(self):
        """Tonemapper settings of the system description (Application.cpp:1244-1292)."""
        tm = L.Tonemapper()
        L.check(L.lib.twk_app_get_tonemapper(self._h, C.byref(tm)))
        return tm

    @property
    def environment(self):
        """File name given with "envMap" in the system description (Application.cpp:1151-1156)."""
        buf = C.create_string_buffer(4096)
        L.check(L.lib.twk_app_get_environment(self._h, buf, C.c_size_t(len(buf))))
        return buf.value.decode()

    def screenshotPath(self, tonemap=True):
        """≙ the file name of Application::screenshot: <prefix>_<spp>spp_<date>_<time>_000.png|.hdr."""
        buf = C.create_string_buffer(4096)
        L.check(L.lib.twk_app_screenshot_path(self._h, int(bool(tonemap)), buf, C.c_size_t(len(buf))))
        return buf.value.decode()

    def initDevice(self, device, distribution=None):
        """≙ Application.cpp:303,328-332: initState, initCameras, initLights, initMaterials, initScene."""
        L.check(L.lib.twk_app_init_device(self._h, device.handle))
        st = self.state
        if distribution is not None:
            st.distribution = int(distribution)
            device.setState(st)
        else:
            device.state = st


def load_image(path):
    """≙ Picture::load + Texture::create* format expansion: PNG / Radiance .hdr / PFM → float32 [H, W, 4], row 0 = bottom."""
    w, h = C.c_int(0), C.c_int(0)
    L.check(L.lib.twk_load_image(os.fsencode(path), C.byref(w), C.byref(h), None, C.c_size_t(0)))
    out = np.empty((h.value, w.value, 4), dtype=np.float32)
    L.check(L.lib.twk_load_image(os.fsencode(path), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(out.size)))
    return out


def write_png(path, rgb8, bottomUp=True):
    """8-bit RGB PNG as Application::screenshot(true) stores it; rgb8 uint8 [H, W, 3]."""
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    L.check(L.lib.twk_write_png_rgb8(os.fsencode(path), int(a.shape[1]), int(a.shape[0]), a.ctypes.data_as(C.POINTER(C.c_ubyte)), int(bool(bottomUp))))


def write_hdr(path, rgba, bottomUp=True):
    """Radiance .hdr as Application::screenshot(false) stores it; rgba float32 [H, W, 4]."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    L.check(L.lib.twk_write_hdr_rgba32f(os.fsencode(path), int(a.shape[1]), int(a.shape[0]), a.ctypes.data_as(C.POINTER(C.c_float)), int(bool(bottomUp))))


def _mesh_call(fn, *args):
    na, ni = C.c_size_t(0), C.c_size_t(0)
    L.check(fn(*args, None, C.byref(na), None, C.byref(ni)))
    attr = np.empty((na.value, 12), dtype=np.float32)
    idx = np.empty((ni.value,), dtype=np.uint32)
    L.check(fn(*args, attr.ctypes.data_as(C.c_void_p), C.byref(na), idx.ctypes.data_as(C.c_void_p), C.byref(ni)))
    return attr, idx


def mesh_plane(tessU, tessV, upAxis):
    return _mesh_call(L.lib.twk_mesh_plane, C.c_uint(tessU), C.c_uint(tessV), C.c_uint(upAxis))


def mesh_box():
    return _mesh_call(L.lib.twk_mesh_box)


def mesh_sphere(tessU, tessV, radius, maxTheta):
    return _mesh_call(L.lib.twk_mesh_sphere, C.c_uint(tessU), C.c_uint(tessV), C.c_float(radius), C.c_float(maxTheta))


def mesh_torus(tessU, tessV, innerRadius, outerRadius):
    return _mesh_call(L.lib.twk_mesh_torus, C.c_uint(tessU), C.c_uint(tessV), C.c_float(innerRadius), C.c_float(outerRadius))


def mesh_parallelogram(position, vecU, vecV, normal):
    v = lambda a: (C.c_float * 3)(*[float(x) for x in a])
    return _mesh_call(L.lib.twk_mesh_parallelogram, v(position), v(vecU), v(vecV), v(normal))


def camera_frustum(center, phi, theta, fov, distance, aspect):
    c = L.CameraDefinition()
    L.check(L.lib.twk_camera_frustum((C.c_float * 3)(*center), C.c_float(phi), C.c_float(theta), C.c_float(fov),
                                     C.c_float(distance), C.c_float(aspect), C.byref(c)))
    return c


def tile_column(launchX, launchY, tileSize, deviceCount, deviceIndex):
    px = C.c_int(0)
    L.check(L.lib.twk_tile_column(int(launchX), int(launchY), (C.c_int * 2)(*tileSize), int(deviceCount), int(deviceIndex), C.byref(px)))
    return px.value


def launch_width(width, tileSizeX, deviceCount):
    w = C.c_int(0)
    L.check(L.lib.twk_launch_width(int(width), int(tileSizeX), int(deviceCount), C.byref(w)))
    return w.value


def parse_tokens(text):
    """Token stream of a description text: list of (type, token), type 1 = identifier, 2 = value (Parser.cpp:72-148)."""
    buf = C.create_string_buffer(max(4096, 4 * len(text) + 64))
    n = C.c_int(0)
    L.check(L.lib.twk_parse_tokens(text.encode(), buf, C.c_size_t(len(buf)), C.byref(n)))
    out = []
    for line in buf.value.decode().split("\n"):
        if line:
            t, _, tok = line.partition(" ")
            out.append((int(t), tok))
    assert len(out) == n.value
    return out
