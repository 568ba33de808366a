"""tweeker_raytracer_amd — MI355X-native replacement of the rtigo3 / Optix7Gui path-tracing hot path.

Thin Python view of the C ABI in include/tweeker_hip.h:
  Device       ≙ rtigo3's per-GPU `Device` (reference apps/rtigo3/inc/Device.h:292-404)
  Application  ≙ the scene-building part of rtigo3's `Application` (system / scene description files)
All work happens in libtweeker_hip.so (HIP kernels for gfx950 + C++ host scene layer).
"""
from ._lib import (TwkError, CameraDefinition, LightDefinition, MaterialGUI, TriangleAttributes, DeviceState,
                   LaunchStats, AppInfo, Tonemapper, LIB_PATH)
from .device import Device, device_count
from .application import Application, mesh_plane, mesh_box, mesh_sphere, mesh_torus, mesh_parallelogram, \
    camera_frustum, tile_column, launch_width, parse_tokens, write_png, write_hdr, load_image

__all__ = ["Device", "Application", "TwkError", "device_count", "CameraDefinition", "LightDefinition",
           "MaterialGUI", "TriangleAttributes", "DeviceState", "LaunchStats", "AppInfo", "LIB_PATH",
           "mesh_plane", "mesh_box", "mesh_sphere", "mesh_torus", "mesh_parallelogram", "camera_frustum",
           "tile_column", "launch_width", "parse_tokens", "Tonemapper", "write_png", "write_hdr", "load_image"]
