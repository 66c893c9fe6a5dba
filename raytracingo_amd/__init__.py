"""raytracingo_amd -- MI355X-native hot path of RayTracinGO (CHrlS98/raytracingo).

The product is librtgo_hip.so (C ABI in include/rtgo.h: HIP megakernel + on-device LBVH for gfx950) and the C++ host
mirror of the reference's Scene/Shape/Renderer surface (librtgo_host.so, rtgo_engine).  The Python modules here are
thin drivers over those libraries (ctypes); none of them computes pixels.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
