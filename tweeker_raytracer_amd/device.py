"""Device — Python mirror of rtigo3's per-GPU `Device` interface (reference inc/Device.h:292-404).

Method names follow the reference: initCameras/initLights/initMaterials/initScene-equivalents, setState,
render(iterationIndex), synchronizeStream, getOutputBufferHost. Every method is a direct call into the
C ABI; errors surface as TwkError (≙ the reference's std::runtime_error).
"""
import ctypes as C

import numpy as np

from . import _lib as L

KERNEL_CLASSES = ("generate", "trace", "shade", "accumulate", "tail")


def device_count():
    n = C.c_int(0)
    L.check(L.lib.twk_device_count(C.byref(n)))
    return n.value


def _as_array(ctype, items):
    items = list(items)
    arr = (ctype * max(1, len(items)))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr, len(items)


class Device:
    def __init__(self, ordinal=0, index=0, count=1, miss=1):
        self._h = C.c_void_p()
        L.check(L.lib.twk_device_create(C.byref(self._h), int(ordinal), int(index), int(count), int(miss)))
        self.index, self.count, self.miss = index, count, miss
        self.state = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            L.lib.twk_device_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    # ---- state / scene ------------------------------------------------------------------------
    def setState(self, state):
        L.check(L.lib.twk_set_state(self._h, C.byref(state)))
        self.state = state

    def initCameras(self, cameras):
        arr, n = _as_array(L.CameraDefinition, cameras)
        L.check(L.lib.twk_init_cameras(self._h, arr, n))

    def initLights(self, lights):
        arr, n = _as_array(L.LightDefinition, lights)
        L.check(L.lib.twk_init_lights(self._h, arr, n))

    def initMaterials(self, materials):
        arr, n = _as_array(L.MaterialGUI, materials)
        L.check(L.lib.twk_init_materials(self._h, arr, n))

    def updateCamera(self, idCamera, camera):
        L.check(L.lib.twk_update_camera(self._h, int(idCamera), C.byref(camera)))

    def updateLight(self, idLight, light):
        L.check(L.lib.twk_update_light(self._h, int(idLight), C.byref(light)))

    def updateMaterial(self, idMaterial, material):
        L.check(L.lib.twk_update_material(self._h, int(idMaterial), C.byref(material)))

    def initTexture(self, slot, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        assert rgba.ndim == 3 and rgba.shape[2] == 4, "texture must be [height, width, 4] float32, row 0 = v 0"
        L.check(L.lib.twk_init_texture(self._h, int(slot), rgba.ctypes.data_as(C.POINTER(C.c_float)),
                                       int(rgba.shape[1]), int(rgba.shape[0])))

    def clearScene(self):
        L.check(L.lib.twk_clear_scene(self._h))

    def addGeometry(self, attributes, indices):
        """attributes: float32 [n, 12] (vertex, tangent, normal, texcoord); indices: uint32 [3 m]."""
        attributes = np.ascontiguousarray(attributes, dtype=np.float32).reshape(-1, 12)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        gid = C.c_int(-1)
        L.check(L.lib.twk_add_geometry(self._h, attributes.ctypes.data_as(C.c_void_p), C.c_size_t(attributes.shape[0]),
                                       indices.ctypes.data_as(C.c_void_p), C.c_size_t(indices.shape[0]), C.byref(gid)))
        return gid.value

    def addInstance(self, idGeometry, transform, idMaterial, idLight=-1):
        t = (C.c_float * 12)(*[float(x) for x in np.asarray(transform, dtype=np.float32).reshape(12)])
        iid = C.c_int(-1)
        L.check(L.lib.twk_add_instance(self._h, int(idGeometry), t, int(idMaterial), int(idLight), C.byref(iid)))
        return iid.value

    def setSharedFrame(self, dptr, nbytes):
        """Accumulate into a shared full W x H frame (ZeroCopy / PeerAccess strategies); 0 returns to the packed buffer."""
        L.check(L.lib.twk_set_shared_frame(self._h, C.c_void_p(int(dptr)), C.c_size_t(int(nbytes))))
        self._sharedFrame = bool(dptr)

    def setShaderVariant(self, variant):
        """0 = rtigo3 (a light's back face reflects through its BSDF), 1 = Optix7Gui (any light hit ends the path)."""
        L.check(L.lib.twk_set_shader_variant(self._h, int(variant)))

    def enableAov(self, enable=True):
        L.check(L.lib.twk_enable_aov(self._h, int(bool(enable))))

    def setTimeView(self, enable=True):
        """≙ USE_TIME_VIEW: alpha of the accumulation buffer = running mean of the sample's shader-clock cycles x clockFactor x 1e-9."""
        L.check(L.lib.twk_set_time_view(self._h, int(bool(enable))))

    def setNextEventEstimation(self, enable=True):
        """≙ USE_NEXT_EVENT_ESTIMATION (shaders/config.h:50-52): False = brute-force path tracing without light sampling and MIS weights."""
        L.check(L.lib.twk_set_next_event_estimation(self._h, int(bool(enable))))

    def setDebugExceptions(self, enable=True):
        """≙ USE_DEBUG_EXCEPTIONS (raygeneration.cu:205-218): NaN / Inf / negative samples accumulate as super red / green / blue."""
        L.check(L.lib.twk_set_debug_exceptions(self._h, int(bool(enable))))

    def readAov(self, which):
        """Denoiser AOV running means: which = 0 albedo, 1 camera-space normal; float32 [height, launchWidth, 4]."""
        h, w = self.state.resolution[1], self.launchWidth
        out = np.empty((h, w, 4), dtype=np.float32)
        L.check(L.lib.twk_read_aov(self._h, int(which), out.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(out.size)))
        return out

    def setBuildQuality(self, quality):
        """0 = LBVH (Morton + radix tree), 1 = binned SAH (default)."""
        L.check(L.lib.twk_set_build_quality(self._h, int(quality)))

    def buildInfo(self):
        b = L.BuildInfo()
        L.check(L.lib.twk_get_build_info(self._h, C.byref(b)))
        return {name: getattr(b, name) for name, _ in L.BuildInfo._fields_}

    def setFlattenPolicy(self, maxTriangles, maxReferences):
        """Build option of the next build(): instances of geometries with <= maxTriangles triangles, or referenced by
        <= maxReferences instances, are intersected in world space in one single-level BVH; (0, 0) = pure two-level."""
        L.check(L.lib.twk_set_flatten_policy(self._h, int(maxTriangles), int(maxReferences)))

    def build(self):
        L.check(L.lib.twk_build(self._h))

    # ---- rendering ----------------------------------------------------------------------------
    def render(self, iterationIndex):
        """≙ Device::render(iterationIndex, buffer) → optixLaunch: asynchronous, one sample per pixel."""
        L.check(L.lib.twk_launch(self._h, C.c_uint(int(iterationIndex))))

    def setLaunchBatch(self, iterations):
        """Iterations rendered together per wavefront pass (1..64); results do not depend on it."""
        L.check(L.lib.twk_set_launch_batch(self._h, int(iterations)))

    def reserveLaunchBatch(self, iterations):
        """Allocate the path streams for passes of `iterations` samples per pixel now instead of on demand."""
        L.check(L.lib.twk_reserve_launch_batch(self._h, int(iterations)))

    def synchronizeStream(self):
        L.check(L.lib.twk_sync(self._h))

    @property
    def launchWidth(self):
        w = C.c_int(0)
        L.check(L.lib.twk_get_launch_width(self._h, C.byref(w)))
        return w.value

    def getOutputBufferHost(self):
        """RGBA32F running mean, shape [height, launchWidth, 4] (launchWidth == width unless tiled)."""
        h, w = self.state.resolution[1], (self.state.resolution[0] if getattr(self, "_sharedFrame", False) else self.launchWidth)
        out = np.empty((h, w, 4), dtype=np.float32)
        L.check(L.lib.twk_read_output(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(out.size)))
        return out

    def outputDevicePointer(self):
        p, n = C.c_void_p(), C.c_size_t(0)
        L.check(L.lib.twk_get_output_device_pointer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def setOutputDevicePointer(self, dptr, nbytes):
        L.check(L.lib.twk_set_output_device_pointer(self._h, C.c_void_p(dptr), C.c_size_t(nbytes)))

    def compositor(self, tiles_dptr, output_dptr):
        L.check(L.lib.twk_compositor(self._h, C.c_void_p(tiles_dptr), C.c_void_p(output_dptr)))

    # ---- measurement / parity taps ------------------------------------------------------------
    def profileEnable(self, enable=True):
        L.check(L.lib.twk_profile_enable(self._h, int(bool(enable))))

    def profileReset(self):
        L.check(L.lib.twk_profile_reset(self._h))

    def profileGet(self):
        ms = (C.c_float * len(KERNEL_CLASSES))()
        n = (C.c_int * len(KERNEL_CLASSES))()
        L.check(L.lib.twk_profile_get(self._h, ms, n))
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(KERNEL_CLASSES)}

    def tonemap(self, tonemapper=None, rgbaDevicePointer=None, shape=None):
        """RGBA32F → RGB8 with the reference's tonemapper (Application.cpp:2259-2297) on the device. Without a pointer
        the handle's own accumulation buffer (launchWidth x height) is converted; returns uint8 [H, W, 3], row 0 at the
        bottom like the float buffer."""
        tm = tonemapper if tonemapper is not None else L.Tonemapper()
        if rgbaDevicePointer is None:
            h, w = self.state.resolution[1], self.launchWidth
            ptr = None
        else:
            h, w = shape
            ptr = C.c_void_p(int(rgbaDevicePointer))
        out = np.empty((h, w, 3), dtype=np.uint8)
        L.check(L.lib.twk_tonemap(self._h, C.byref(tm), ptr, C.c_size_t(h * w), out.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return out

    def statsEnable(self, enable=True):
        L.check(L.lib.twk_stats_enable(self._h, int(bool(enable))))

    def statsGet(self, reset=True):
        s = L.LaunchStats()
        L.check(L.lib.twk_stats_get(self._h, C.byref(s), int(bool(reset))))
        return {name: (list(getattr(s, name)) if name in ("waveCycles", "shadePhaseWaveSteps", "shadePhaseLanes", "shadePhaseCycles") else getattr(s, name)) for name, _ in L.LaunchStats._fields_}

    def streamPeakGBps(self, nbytes=1 << 30, repeats=10):
        g = C.c_float(0)
        L.check(L.lib.twk_stream_peak_gbps(self._h, C.c_size_t(nbytes), int(repeats), C.byref(g)))
        return g.value

    def gatherPeak(self, table_bytes=32 << 20):
        """Divergent-gather ceiling in giga lane-loads (16 B each) per second (twk_gather_peak)."""
        g = C.c_float(0)
        L.check(L.lib.twk_gather_peak(self._h, C.c_size_t(int(table_bytes)), C.byref(g)))
        return g.value

    def debugTraceQueue(self, closest=None, shadow=None):
        """One launch of the persistent traversal kernel over explicit rays ([n, 8] each; either may be None).
        Returns (hit records float32 [n, 4] = t, beta, gamma, slot bits; instance int32 [n]; occluded int32 [m])."""
        c = np.zeros((0, 8), np.float32) if closest is None else np.ascontiguousarray(closest, np.float32).reshape(-1, 8)
        s = np.zeros((0, 8), np.float32) if shadow is None else np.ascontiguousarray(shadow, np.float32).reshape(-1, 8)
        rec = np.zeros((max(1, c.shape[0]), 4), np.float32)
        inst = np.full((max(1, c.shape[0]),), -1, np.int32)
        occ = np.zeros((max(1, s.shape[0]),), np.int32)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
        L.check(L.lib.twk_debug_trace_queue(self._h, c.ctypes.data_as(fp), C.c_size_t(c.shape[0]), s.ctypes.data_as(fp), C.c_size_t(s.shape[0]),
                                            rec.ctypes.data_as(fp), inst.ctypes.data_as(ip), occ.ctypes.data_as(ip)))
        return rec[:c.shape[0]], inst[:c.shape[0]], occ[:s.shape[0]]

    def readAcceleration(self):
        """(info dict, quantised wide nodes float32 [n, 16], triangle slots float32 [m, 12], instance records float32 [k, 32]) of the built scene."""
        info = L.AccelerationInfo()
        L.check(L.lib.twk_debug_read_acceleration(self._h, C.byref(info), None, None, None))
        nodes = np.zeros((info.numNodes, info.nodeFloats), np.float32)  # quantised 4-ary nodes (64 B) or compressed 8-ary nodes (80 B): info.nodeFloats
        tris = np.zeros((info.numTriangleSlots, 12), np.float32)
        inst = np.zeros((info.numInstances, 32), np.float32)
        L.check(L.lib.twk_debug_read_acceleration(self._h, C.byref(info), nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p),
                                                  inst.ctypes.data_as(C.c_void_p)))
        return {name: getattr(info, name) for name, _ in L.AccelerationInfo._fields_}, nodes, tris, inst

    def debugCapture(self, enable=True):
        L.check(L.lib.twk_debug_capture(self._h, int(bool(enable))))

    def debugReadFirstHits(self):
        n = self.state.resolution[1] * self.launchWidth
        tbg = np.empty((n, 3), dtype=np.float32)
        ids = np.empty((n, 2), dtype=np.int32)
        L.check(L.lib.twk_debug_read_first_hits(self._h, tbg.ctypes.data_as(C.POINTER(C.c_float)),
                                                ids.ctypes.data_as(C.POINTER(C.c_int)), C.c_size_t(n)))
        return tbg, ids

    def traceRays(self, rays, anyHit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        tbg = np.zeros((n, 3), dtype=np.float32)
        ids = np.zeros((n, 2), dtype=np.int32)
        L.check(L.lib.twk_trace_rays(self._h, rays.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(n), int(bool(anyHit)),
                                     tbg.ctypes.data_as(C.POINTER(C.c_float)), ids.ctypes.data_as(C.POINTER(C.c_int))))
        return tbg, ids

    def debugMath(self, op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        yy = x if y is None else np.ascontiguousarray(y, dtype=np.float32).reshape(-1)
        out = np.empty_like(x)
        L.check(L.lib.twk_debug_math(self._h, int(op), x.ctypes.data_as(C.POINTER(C.c_float)),
                                     yy.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)),
                                     C.c_size_t(x.size)))
        return out
