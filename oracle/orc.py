"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes view of oracle/liboracle.so (CPU restatement of the reference's
path tracer, see oracle/orc_shaders.h) and of oracle/_ref/libref_host.so (the reference's own host sources
compiled in the build container). Also of oracle/libhostkernels.so: the PRODUCT's kernel headers compiled for the host (host_kernels.cpp), run on a scene a
Device built. Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.path.join(_HERE, "liboracle.so")
ORACLE_LIBM_PATH = os.path.join(_HERE, "liboracle_libm.so")
ORACLE_NEE0_PATH = os.path.join(_HERE, "liboracle_nee0.so")      # built with -DUSE_NEXT_EVENT_ESTIMATION=0 (config.h:50-52)
ORACLE_DBGEXC_PATH = os.path.join(_HERE, "liboracle_dbgexc.so")  # built with -DUSE_DEBUG_EXCEPTIONS=1 (config.h:54-56)
REF_PATH = os.path.join(_HERE, "_ref", "libref_host.so")


def _load(path):
    if not os.path.exists(path):
        raise ImportError(f"{path} missing: run `make -C oracle` (and `make -C oracle ref` in the build container)")
    return C.CDLL(path)


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _f(a):
    return a.ctypes.data_as(_fp)


class Oracle:
    """Same call sequence as tweeker_raytracer_amd.Device, executed by the CPU restatement."""

    def __init__(self, index=0, count=1, miss=1, libm=False, nee=True, debugExceptions=False):
        assert not (libm and (not nee or debugExceptions)) and not (not nee and debugExceptions), "one compile-time switch per oracle build"
        self.lib = _load(ORACLE_LIBM_PATH if libm else (ORACLE_NEE0_PATH if not nee else (ORACLE_DBGEXC_PATH if debugExceptions else ORACLE_PATH)))
        self.nee, self.debugExceptions = bool(nee), bool(debugExceptions)
        self.lib.orc_last_error.restype = C.c_char_p
        self._h = C.c_void_p()
        self._chk(self.lib.orc_create(C.byref(self._h), int(index), int(count), int(miss)))
        self.state = None

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + (self.lib.orc_last_error() or b"").decode())

    def close(self):
        if self._h.value:
            self.lib.orc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setState(self, state):
        self._chk(self.lib.orc_set_state(self._h, C.byref(state)))
        self.state = state

    def _arr(self, items):
        items = list(items)
        if not items:
            return None, 0
        arr = (type(items[0]) * len(items))(*items)
        return arr, len(items)

    def initCameras(self, cameras):
        a, n = self._arr(cameras)
        self._chk(self.lib.orc_init_cameras(self._h, a, n))

    def initLights(self, lights):
        a, n = self._arr(lights)
        self._chk(self.lib.orc_init_lights(self._h, a, n))

    def initMaterials(self, materials):
        a, n = self._arr(materials)
        self._chk(self.lib.orc_init_materials(self._h, a, n))

    def initTexture(self, slot, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        self._chk(self.lib.orc_init_texture(self._h, int(slot), _f(rgba), int(rgba.shape[1]), int(rgba.shape[0])))

    def envTables(self, width, height):
        u = np.zeros(((width + 1) * height,), np.float32)
        v = np.zeros((height + 1,), np.float32)
        i = C.c_float(0)
        self._chk(self.lib.orc_get_env_tables(self._h, _f(u), _f(v), C.byref(i)))
        return u, v, i.value

    def addGeometry(self, attributes, indices):
        attributes = np.ascontiguousarray(attributes, dtype=np.float32).reshape(-1, 12)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        g = C.c_int(-1)
        self._chk(self.lib.orc_add_geometry(self._h, attributes.ctypes.data_as(C.c_void_p), C.c_size_t(attributes.shape[0]),
                                            indices.ctypes.data_as(C.c_void_p), C.c_size_t(indices.shape[0]), C.byref(g)))
        return g.value

    def addInstance(self, idGeometry, transform, idMaterial, idLight=-1):
        t = (C.c_float * 12)(*[float(x) for x in np.asarray(transform, dtype=np.float32).reshape(12)])
        i = C.c_int(-1)
        self._chk(self.lib.orc_add_instance(self._h, int(idGeometry), t, int(idMaterial), int(idLight), C.byref(i)))
        return i.value

    def clearScene(self):
        self._chk(self.lib.orc_clear_scene(self._h))

    def build(self):
        pass  # geometry BVHs are built on add

    def loadApplication(self, app, distribution=None, state=None):
        """Feed the scene an Application parsed (same inputs twk_app_init_device hands to the HIP device)."""
        st = state if state is not None else app.state
        self.setShaderVariant(getattr(app.info, "shaderVariant", 0))
        # the reference's config.h switches are compile-time here as there: this object must be the build the description asks for
        # (a description that leaves the defaults may still be rendered by another build: the caller flipped the device's switch itself)
        assert (bool(getattr(app.info, "nextEventEstimation", 1)) or not self.nee) and (not getattr(app.info, "debugExceptions", 0) or self.debugExceptions), \
            "the description's nextEventEstimation 0 / debugExceptions 1 need orc.Oracle(nee=False) / orc.Oracle(debugExceptions=True)"
        if distribution is not None:
            st.distribution = int(distribution)
        self.setState(st)
        self.initCameras(app.cameras)
        self.initLights(app.lights)
        self.initMaterials(app.materials)
        self.clearScene()
        for g in range(app.info.numGeometries):
            attr, idx = app.geometry(g)
            assert self.addGeometry(attr, idx) == g
        for (g, t, m, l) in app.instances:
            self.addInstance(g, t, m, l)

    def setFlattenPolicy(self, maxTriangles, maxReferences):
        """≙ Device.setFlattenPolicy: which instances are intersected in world space (include/tweeker_hip.h)."""
        self._chk(self.lib.orc_set_flatten_policy(self._h, int(maxTriangles), int(maxReferences)))

    def setShaderVariant(self, variant):
        self._chk(self.lib.orc_set_shader_variant(self._h, int(variant)))

    def enableAov(self, enable=True):
        self._chk(self.lib.orc_enable_aov(self._h, int(bool(enable))))

    def readAov(self, which):
        h, w = self.state.resolution[1], self.launchWidth
        out = np.empty((h, w, 4), dtype=np.float32)
        self._chk(self.lib.orc_read_aov(self._h, int(which), _f(out), C.c_size_t(out.size)))
        return out

    def setTraceMode(self, use_bvh):
        self._chk(self.lib.orc_set_trace_mode(self._h, int(bool(use_bvh))))

    def captureFirstHits(self, enable=True):
        self._chk(self.lib.orc_capture_first_hits(self._h, int(bool(enable))))

    @property
    def launchWidth(self):
        w = C.c_int(0)
        self._chk(self.lib.orc_get_launch_width(self._h, C.byref(w)))
        return w.value

    def render(self, iterationIndex, rect=None, threads=1):
        """One iteration over the frame or a pixel rectangle; threads > 1 spreads the rows over host threads (same image)."""
        if rect is None and threads <= 1:
            self._chk(self.lib.orc_render(self._h, C.c_uint(int(iterationIndex))))
            return
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, self.launchWidth, self.state.resolution[1])
        if threads <= 1:
            self._chk(self.lib.orc_render_rect(self._h, C.c_uint(int(iterationIndex)), int(x0), int(y0), int(x1), int(y1)))
        else:
            self._chk(self.lib.orc_render_rect_threads(self._h, C.c_uint(int(iterationIndex)), int(x0), int(y0), int(x1), int(y1), int(threads)))

    def debugPath(self, iterationIndex, x, y, capacity=256):
        """Rays the sample (x, y, iteration) traces, in call order: [n, 9] = o.xyz, tmin, d.xyz, tmax, kind (0 radiance, 1 shadow)."""
        rays = np.zeros((capacity, 9), np.float32)
        n = C.c_int(0)
        self._chk(self.lib.orc_debug_path(self._h, C.c_uint(int(iterationIndex)), int(x), int(y), _f(rays), int(capacity), C.byref(n)))
        return rays[:min(n.value, capacity)]

    def getOutputBufferHost(self):
        h, w = self.state.resolution[1], self.launchWidth
        out = np.empty((h, w, 4), dtype=np.float32)
        self._chk(self.lib.orc_read_output(self._h, _f(out), C.c_size_t(out.size)))
        return out

    def readFirstHits(self):
        n = self.state.resolution[1] * self.launchWidth
        tbg = np.empty((n, 3), np.float32)
        ids = np.empty((n, 2), np.int32)
        self._chk(self.lib.orc_read_first_hits(self._h, _f(tbg), ids.ctypes.data_as(_ip), C.c_size_t(n)))
        return tbg, ids

    def counters(self):
        out = (C.c_uint64 * 6)()
        self._chk(self.lib.orc_get_counters(self._h, out))
        return dict(zip(["radianceRays", "shadowRays", "samples", "traceCalls", "boxTests", "triTests"], list(out)))

    def traceRays(self, rays, anyHit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        tbg = np.zeros((n, 3), np.float32)
        ids = np.zeros((n, 2), np.int32)
        self._chk(self.lib.orc_trace_rays(self._h, _f(rays), C.c_size_t(n), int(bool(anyHit)), _f(tbg), ids.ctypes.data_as(_ip)))
        return tbg, ids


HOST_KERNELS_PATH = os.path.join(_HERE, "libhostkernels.so")


class HostKernels:
    """The product's kernels compiled for the host (oracle/host_kernels.cpp) on the scene a Device built: the
    north_star's "single-threaded C++ CPU fallback of the same kernels". render() runs ONE wavefront pass of `batch`
    iterations on the calling thread into the running-mean image it keeps."""

    def __init__(self, device):
        self.lib = _load(HOST_KERNELS_PATH)
        self.lib.hostk_params_bytes.restype = C.c_size_t
        n = self.lib.hostk_params_bytes()
        self._params = C.create_string_buffer(n)
        from tweeker_raytracer_amd import _lib as L
        L.check(L.lib.twk_debug_snapshot_scene(device.handle, self._params, C.c_size_t(n)))
        self._device = device  # owns the host copies of the scene arrays
        h, w = device.state.resolution[1], device.launchWidth
        self.image = np.zeros((h, w, 4), np.float32)
        self.seconds = 0.0
        self.counts = {}

    def render(self, firstIteration, batch=1):
        secs = C.c_double(0.0)
        counts = (C.c_uint64 * 6)()
        rc = self.lib.hostk_render(self._params, C.c_size_t(len(self._params)), C.c_uint(int(firstIteration)), int(batch), _f(self.image), C.byref(secs), counts)
        if rc != 0:
            raise RuntimeError("hostk_render failed (bad arguments, a scene with cutout opacity, or a dropped stack push)")
        self.seconds += secs.value
        names = ("radianceRays", "shadowRays", "nodesVisited", "trianglesTested", "instancesEntered", "shadedSegments")
        self.counts = {k: self.counts.get(k, 0) + int(counts[i]) for i, k in enumerate(names)}
        return secs.value

    def getOutputBufferHost(self):
        return self.image


def walk_same_bvh(acceleration, rays, anyHit=False):
    """Same-BVH host walker (oracle/same_bvh_walk.cpp): `acceleration` = Device.readAcceleration(); rays [n, 8].
    Returns (t/beta/gamma [n, 3], ids [n, 2], counts dict) — single-threaded."""
    lib = _load(ORACLE_PATH)
    info, nodes, tris, inst = acceleration
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    n = rays.shape[0]
    tbg = np.zeros((n, 3), np.float32)
    ids = np.zeros((n, 2), np.int32)
    counts = (C.c_uint64 * 4)()
    assert int(info.get("nodeFloats", 16)) == 16
    rc = lib.orc_walk_same_bvh(_f(nodes), int(info["root"]), int(info.get("root2", -1)), _f(tris), _f(inst), _f(rays), C.c_uint64(n), int(bool(anyHit)),
                               _f(tbg), ids.ctypes.data_as(_ip), counts)
    assert rc == 0
    return tbg, ids, {"nodesVisited": counts[0], "trianglesTested": counts[1], "instancesEntered": counts[2], "deepestStack": counts[3]}


def oracle_math(op, x, y=None, libm=False):
    lib = _load(ORACLE_LIBM_PATH if libm else ORACLE_PATH)
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    yy = x if y is None else np.ascontiguousarray(y, dtype=np.float32).reshape(-1)
    out = np.empty_like(x)
    assert lib.orc_math(int(op), _f(x), _f(yy), _f(out), C.c_size_t(x.size)) == 0
    return out


def oracle_tonemap(rgba, tonemapper=(1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 1.0), libm=False):
    """Application.cpp:2259-2297 on the CPU. tonemapper = gamma, whitePoint, colorBalance r g b, burnHighlights,
    crushBlacks, saturation, brightness (TonemapperGUI field order). rgba float32 [..., 4] → uint8 [..., 3]."""
    lib = _load(ORACLE_LIBM_PATH if libm else ORACLE_PATH)
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    tm = np.asarray(tonemapper, dtype=np.float32)
    assert tm.size == 9 and a.shape[-1] == 4
    out = np.empty(a.shape[:-1] + (3,), dtype=np.uint8)
    assert lib.orc_tonemap(_f(tm), _f(a.reshape(-1)), C.c_size_t(a.size // 4), out.ctypes.data_as(C.POINTER(C.c_ubyte))) == 0
    return out


def oracle_compositor(tiles, width, tileSize=(8, 8)):
    """compositor.cu:38-64 on the CPU, one pass per source device: tiles float32 [deviceCount, height, launchWidth, 4]
    -> image float32 [height, width, 4] (pixels no device owns keep -1)."""
    lib = _load(ORACLE_PATH)
    t = np.ascontiguousarray(tiles, dtype=np.float32)
    n, h, lw, four = t.shape
    assert four == 4
    out = np.full((h, int(width), 4), -1.0, np.float32)
    assert lib.orc_compositor(_f(t.reshape(-1)), _f(out.reshape(-1)), int(width), int(h), int(lw), int(n), int(tileSize[0]), int(tileSize[1])) == 0
    return out


class _Unit:
    """Scalar/vector unit taps shared by liboracle.so (orc_*) and libref_host.so (ref_*)."""

    def __init__(self, lib, prefix):
        self.lib, self.p = lib, prefix
        getattr(lib, prefix + "tea4").restype = C.c_uint
        getattr(lib, prefix + "rng").restype = C.c_float

    def tea4(self, v0, v1):
        return getattr(self.lib, self.p + "tea4")(C.c_uint(v0), C.c_uint(v1))

    def rng_stream(self, seed, n):
        s = C.c_uint(seed)
        f = getattr(self.lib, self.p + "rng")
        vals = [f(C.byref(s)) for _ in range(n)]
        return np.array(vals, np.float32), s.value

    def refract(self, i, n, ior):
        r = (C.c_float * 3)()
        ok = getattr(self.lib, self.p + "refract")((C.c_float * 3)(*i), (C.c_float * 3)(*n), C.c_float(ior), r)
        return ok, np.array(list(r), np.float32)

    def tbn(self, t, n):
        out = (C.c_float * 9)()
        getattr(self.lib, self.p + "tbn")((C.c_float * 3)(*t), (C.c_float * 3)(*n), out)
        return np.array(list(out), np.float32)

    def vec3(self, op, a, b):
        out = (C.c_float * 3)()
        assert getattr(self.lib, self.p + "vec3")(int(op), (C.c_float * 3)(*a), (C.c_float * 3)(*b), out) == 0
        return np.array(list(out), np.float32)


def oracle_units():
    return _Unit(_load(ORACLE_PATH), "orc_")


class Reference(_Unit):
    """The reference's own host code (oracle/_ref/libref_host.so, built from /root/reference by oracle/Makefile)."""

    def __init__(self):
        super().__init__(_load(REF_PATH), "ref_")

    def _mesh(self, fn, *args):
        na, ni = C.c_size_t(0), C.c_size_t(0)
        fn(*args, None, C.byref(na), None, C.byref(ni))
        attr = np.empty((na.value, 12), np.float32)
        idx = np.empty((ni.value,), np.uint32)
        fn(*args, _f(attr), C.byref(na), idx.ctypes.data_as(C.c_void_p), C.byref(ni))
        return attr, idx

    def mesh_plane(self, u, v, axis):
        return self._mesh(self.lib.ref_mesh_plane, C.c_uint(u), C.c_uint(v), C.c_uint(axis))

    def mesh_box(self):
        return self._mesh(self.lib.ref_mesh_box)

    def mesh_sphere(self, u, v, radius, maxTheta):
        return self._mesh(self.lib.ref_mesh_sphere, C.c_uint(u), C.c_uint(v), C.c_float(radius), C.c_float(maxTheta))

    def mesh_torus(self, u, v, ri, ro):
        return self._mesh(self.lib.ref_mesh_torus, C.c_uint(u), C.c_uint(v), C.c_float(ri), C.c_float(ro))

    def mesh_parallelogram(self, p, u, v, n):
        a = lambda x: (C.c_float * 3)(*[float(k) for k in x])
        return self._mesh(self.lib.ref_mesh_parallelogram, a(p), a(u), a(v), a(n))

    def camera_frustum(self, center, phi, theta, fov, distance, width, height):
        out = (C.c_float * 12)()
        self.lib.ref_camera_frustum((C.c_float * 3)(*center), C.c_float(phi), C.c_float(theta), C.c_float(fov),
                                    C.c_float(distance), int(width), int(height), out)
        return np.array(list(out), np.float32)

    def parse_tokens(self, filename):
        buf = C.create_string_buffer(1 << 20)
        n = self.lib.ref_parse_tokens(str(filename).encode(), buf, C.c_size_t(len(buf)))
        assert n >= 0, "reference Parser failed"
        toks = []
        for line in buf.value.decode().split("\n"):
            if line:
                t, _, text = line.partition(" ")
                toks.append((int(t), text))
        return toks

    def transform_stack(self, ops):
        ops = np.ascontiguousarray(ops, np.float32).reshape(-1, 5)
        out = (C.c_float * 12)()
        self.lib.ref_transform_stack(_f(ops), int(ops.shape[0]), out)
        return np.array(list(out), np.float32)
