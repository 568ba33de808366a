"""ctypes loader of libtweeker_hip.so — the C ABI declared in include/tweeker_hip.h.

There is no Python or CPU fallback: if the HIP library is missing the import fails loudly, and every
compute entry point fails with TWK_ERROR_NO_DEVICE on a machine without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TWK_LIB") or os.path.join(_HERE, "libtweeker_hip.so")  # TWK_LIB: an A/B variant built by tools/ab_variant.sh


class TwkError(RuntimeError):
    """≙ the std::runtime_error thrown by CU_CHECK/OPTIX_CHECK (reference inc/CheckMacros.h:38-80)."""

    def __init__(self, code, message):
        super().__init__(f"tweeker_hip error {code}: {message}")
        self.code = code


TWK_SUCCESS, TWK_ERROR_INVALID_VALUE, TWK_ERROR_NO_DEVICE, TWK_ERROR_HIP = 0, 1, 2, 3
TWK_ERROR_INVALID_STATE, TWK_ERROR_OUT_OF_MEMORY, TWK_ERROR_IO, TWK_ERROR_PARSE = 4, 5, 6, 7

f3 = C.c_float * 3
f2 = C.c_float * 2
i2 = C.c_int * 2


class CameraDefinition(C.Structure):
    _fields_ = [("P", f3), ("U", f3), ("V", f3), ("W", f3)]


class LightDefinition(C.Structure):
    _fields_ = [("type", C.c_int), ("position", f3), ("vecU", f3), ("vecV", f3), ("normal", f3),
                ("area", C.c_float), ("emission", f3), ("unused0", C.c_float), ("unused1", C.c_float),
                ("unused2", C.c_float)]


class MaterialGUI(C.Structure):
    _fields_ = [("indexBSDF", C.c_int), ("albedo", f3), ("absorptionColor", f3), ("absorptionScale", C.c_float),
                ("ior", C.c_float), ("thinwalled", C.c_int), ("useAlbedoTexture", C.c_int),
                ("useCutoutTexture", C.c_int), ("roughness", f2)]


class TriangleAttributes(C.Structure):
    _fields_ = [("vertex", f3), ("tangent", f3), ("normal", f3), ("texcoord", f3)]


class DeviceState(C.Structure):
    _fields_ = [("resolution", i2), ("tileSize", i2), ("pathLengths", i2), ("distribution", C.c_int),
                ("samplesSqrt", C.c_int), ("lensShader", C.c_int), ("epsilonFactor", C.c_float),
                ("envRotation", C.c_float), ("clockFactor", C.c_float)]


class Tonemapper(C.Structure):
    """≙ TonemapperGUI (inc/TonemapperGUI.h:34-43); neutral defaults of Application.cpp:111-120."""
    _fields_ = [("gamma", C.c_float), ("whitePoint", C.c_float), ("colorBalance", C.c_float * 3),
                ("burnHighlights", C.c_float), ("crushBlacks", C.c_float), ("saturation", C.c_float),
                ("brightness", C.c_float)]

    def __init__(self, gamma=1.0, whitePoint=1.0, colorBalance=(1.0, 1.0, 1.0), burnHighlights=1.0, crushBlacks=0.0,
                 saturation=1.0, brightness=1.0):
        super().__init__(gamma, whitePoint, (C.c_float * 3)(*colorBalance), burnHighlights, crushBlacks, saturation, brightness)


class LaunchStats(C.Structure):
    _fields_ = [("radianceRays", C.c_uint64), ("shadowRays", C.c_uint64), ("nodesVisited", C.c_uint64),
                ("trianglesTested", C.c_uint64), ("instancesEntered", C.c_uint64), ("shadedHits", C.c_uint64),
                ("missed", C.c_uint64), ("maxNodesPerRay", C.c_uint64), ("tailRays", C.c_uint64),
                ("tailNodesVisited", C.c_uint64), ("tailTrianglesTested", C.c_uint64), ("tailInstancesEntered", C.c_uint64),
                ("overflowRays", C.c_uint64),
                ("nodeWaveSteps", C.c_uint64), ("triangleWaveSteps", C.c_uint64), ("leafWaveSteps", C.c_uint64),
                ("cachedNodesVisited", C.c_uint64), ("droppedStackPushes", C.c_uint64), ("waveCycles", C.c_uint64 * 6),
                ("shadePhaseWaveSteps", C.c_uint64 * 24), ("shadePhaseLanes", C.c_uint64 * 24), ("shadePhaseCycles", C.c_uint64 * 24)]


class AccelerationInfo(C.Structure):
    _fields_ = [("root", C.c_int), ("twoLevel", C.c_int), ("numNodes", C.c_uint64), ("numTriangleSlots", C.c_uint64), ("numInstances", C.c_uint64),
                ("root2", C.c_int), ("nodeFloats", C.c_int)]


class BuildInfo(C.Structure):
    _fields_ = [("quality", C.c_int), ("trees", C.c_int), ("sahInnerCost", C.c_double), ("sahLeafCost", C.c_double),
                ("buildMilliseconds", C.c_double), ("triangleSlots", C.c_uint64), ("nodes", C.c_uint64),
                ("instances", C.c_uint64), ("flattenedInstances", C.c_uint64), ("maxTraversalDepth", C.c_uint64),
                ("directLeafInstances", C.c_uint64), ("traceBlocksPerCU", C.c_uint64),
                ("wide8Nodes", C.c_uint64), ("wide8Levels", C.c_uint64)]


class AppInfo(C.Structure):
    _fields_ = [("strategy", C.c_int), ("devicesMask", C.c_int), ("light", C.c_int), ("miss", C.c_int),
                ("lensShader", C.c_int), ("samplesSqrt", C.c_int), ("resolution", i2), ("tileSize", i2),
                ("pathLengths", i2), ("epsilonFactor", C.c_float), ("envRotation", C.c_float),
                ("clockFactor", C.c_float), ("center", f3), ("phi", C.c_float), ("theta", C.c_float),
                ("fov", C.c_float), ("distance", C.c_float), ("numCameras", C.c_int), ("numLights", C.c_int),
                ("numMaterials", C.c_int), ("numGeometries", C.c_int), ("numInstances", C.c_int),
                ("shaderVariant", C.c_int), ("nextEventEstimation", C.c_int), ("debugExceptions", C.c_int)]


# Every symbol include/tweeker_hip.h declares; tests/test_cabi_symbols.py checks header == this list == the .so.
SYMBOLS = [
    "twk_last_error", "twk_abi_version", "twk_device_count", "twk_device_create", "twk_device_destroy",
    "twk_set_state", "twk_init_cameras", "twk_init_lights", "twk_init_materials", "twk_update_camera",
    "twk_update_light", "twk_update_material", "twk_init_texture", "twk_add_geometry", "twk_add_instance",
    "twk_build", "twk_clear_scene", "twk_set_flatten_policy", "twk_set_build_quality", "twk_get_build_info", "twk_launch", "twk_sync", "twk_set_launch_batch", "twk_reserve_launch_batch", "twk_get_launch_width", "twk_read_output",
    "twk_set_shader_variant", "twk_enable_aov", "twk_read_aov", "twk_set_time_view", "twk_set_next_event_estimation", "twk_set_debug_exceptions", "twk_get_output_device_pointer", "twk_set_output_device_pointer", "twk_set_shared_frame", "twk_compositor", "twk_tonemap", "twk_profile_enable",
    "twk_profile_reset", "twk_profile_get", "twk_stats_enable", "twk_stats_get", "twk_stream_peak_gbps", "twk_gather_peak",
    "twk_debug_capture", "twk_debug_read_first_hits", "twk_trace_rays", "twk_debug_trace_queue", "twk_debug_read_acceleration", "twk_debug_snapshot_scene", "twk_debug_math",
    "twk_app_create", "twk_app_create_from_strings", "twk_app_destroy", "twk_app_info", "twk_app_set_resolution",
    "twk_app_get_state", "twk_app_get_cameras", "twk_app_get_lights", "twk_app_get_materials",
    "twk_app_get_geometry_sizes", "twk_app_get_geometry", "twk_app_get_instance", "twk_app_init_device",
    "twk_app_system_description", "twk_app_get_tonemapper", "twk_app_screenshot_path", "twk_load_image", "twk_app_get_environment",
    "twk_write_png_rgb8", "twk_write_hdr_rgba32f",
    "twk_mesh_plane", "twk_mesh_box", "twk_mesh_sphere", "twk_mesh_torus", "twk_mesh_parallelogram",
    "twk_camera_frustum", "twk_tile_column", "twk_launch_width", "twk_parse_tokens",
]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). tweeker_raytracer_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)
lib.twk_last_error.restype = C.c_char_p
for _name in SYMBOLS:
    if _name != "twk_last_error":
        getattr(lib, _name).restype = C.c_int


def check(code):
    if code != TWK_SUCCESS:
        raise TwkError(code, (lib.twk_last_error() or b"").decode("utf-8", "replace"))
    return code
