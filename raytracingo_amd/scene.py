"""Scene tables from the PRODUCT host code (librtgo_host.so: engine::host::Scene and friends, C++).

Returns the flattened inputs of the hot path -- what Renderer::CreateShapes / CreateRayGen / CreateMiss / WriteLights
pass to the C ABI -- as numpy arrays, for the Python drivers (bench.py, the band driver).
"""
import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(capi.PKG_DIR, "librtgo_host.so")
SCENES = ["plateau", "slide", "cornell", "mirror_spheres", "soft_mirrors", "window", "balls", "checkered"]  # main.cpp:44-52


class HostScene(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("n_lights", C.c_uint32), ("prims", capi.Prim * capi.RTGO_MAX_PRIMS),
                ("aabbs", capi.Aabb * capi.RTGO_MAX_PRIMS), ("lights", capi.Light * capi.RTGO_MAX_LIGHTS),
                ("eye", C.c_float * 3), ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3),
                ("background", C.c_float * 3)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise capi.RtgoError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        capi.load()  # librtgo_host.so links against librtgo_hip.so
        L = C.CDLL(LIB_PATH)
        L.rtgo_host_scene_build.restype = C.c_int
        L.rtgo_host_scene_build.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(HostScene)]
        L.rtgo_host_render.restype = C.c_int
        L.rtgo_host_render.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.POINTER(capi.Stats)]
        L.rtgo_host_last_error.restype = C.c_char_p
        L.rtgo_host_render_multi.restype = C.c_int
        L.rtgo_host_render_multi.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.POINTER(capi.Stats), C.POINTER(C.c_double)]
        L.rtgo_host_material.restype = C.c_int
        L.rtgo_host_material.argtypes = [C.c_char_p, C.POINTER(C.c_float)]
        _lib = L
    return _lib


def material(name):
    """engine::host::materials::<name> of the product host: kd[3], kr[3], Le[3], specularity"""
    v = np.zeros(10, dtype=np.float32)
    rc = load().rtgo_host_material(name.encode(), v.ctypes.data_as(C.POINTER(C.c_float)))
    if rc:
        raise capi.RtgoError("rtgo_host_material(%s): %s" % (name, load().rtgo_host_last_error().decode()))
    return v


def build(name, width, height):
    """the raw rtgo_host_scene structure"""
    hs = HostScene()
    rc = load().rtgo_host_scene_build(name.encode(), width, height, C.byref(hs))
    if rc != 0:
        raise ValueError("scene %r: %s" % (name, load().rtgo_host_last_error().decode()))
    return hs


def tables(name, width, height):
    """dict(type[n], M[n,16], mat[n,10] = kd kr specularity Le, aabb[n,6], lights[nl,16], cam[12] = eye U V W, bg[3])"""
    hs = build(name, width, height)
    n, nl = hs.n_prims, hs.n_lights
    prims = np.frombuffer(hs.prims, dtype=np.uint32, count=n * 27).reshape(n, 27).copy()
    t = prims[:, 0].astype(np.int32)
    f = prims.view(np.float32)
    M = f[:, 1:17].copy()
    mat = f[:, 17:27].copy()
    aabb = np.frombuffer(hs.aabbs, dtype=np.float32, count=n * 6).reshape(n, 6).copy()
    lights = np.frombuffer(hs.lights, dtype=np.float32, count=nl * 16).reshape(nl, 16).copy()
    cam = np.array(list(hs.eye) + list(hs.U) + list(hs.V) + list(hs.W), dtype=np.float32)
    return {"type": t, "M": M, "mat": mat, "aabb": aabb, "lights": lights, "cam": cam,
            "bg": np.array(list(hs.background), dtype=np.float32)}


def host_render(name, mode, width, height, sample=1, ambient=False, frames=1, device=0):
    """run the headless C++ Renderer; returns (accum[h,w,4] float32, image[h,w,4] uint8, stats dict)"""
    acc = np.empty((height, width, 4), dtype=np.float32)
    img = np.empty((height, width, 4), dtype=np.uint8)
    st = capi.Stats()
    rc = load().rtgo_host_render(name.encode(), mode.encode(), width, height, sample, int(ambient), frames, device,
                                 img.ctypes.data, acc.ctypes.data, C.byref(st))
    if rc != 0:
        raise capi.RtgoError("rtgo_host_render failed (%d): %s" % (rc, load().rtgo_host_last_error().decode()))
    return acc, img, st.as_dict()


def host_render_multi(name, mode, width, height, sample=1, ambient=False, frames=1, devices=(0,), launches_per_device=1,
                      present_every=1, rccl_for_local_shares=False):
    """engine::host::MultiGpuRenderer (C++, RCCL gather) headless: returns (accum, image, stats, ms_per_frame)"""
    L = load()
    img = np.zeros((height, width, 4), dtype=np.uint8)
    acc = np.zeros((height, width, 4), dtype=np.float32)
    st = capi.Stats()
    ms = C.c_double(0.0)
    dev = (C.c_int * len(devices))(*devices)
    rc = L.rtgo_host_render_multi(name.encode(), mode.encode(), width, height, sample, int(ambient), frames, dev, len(devices),
                                  launches_per_device, present_every, int(rccl_for_local_shares), img.ctypes.data, acc.ctypes.data,
                                  C.byref(st), C.byref(ms))
    if rc:
        raise capi.RtgoError("rtgo_host_render_multi: %s" % L.rtgo_host_last_error().decode())
    return acc, img, st.as_dict(), ms.value


class Session:
    """scripted interactive session with the C++ Renderer (rtgo_host_session_*): frames, camera moves, resizes"""

    def __init__(self, name, mode, width, height, sample=1, ambient=False, device=0):
        L = load()
        L.rtgo_host_session_open.restype = C.c_int
        L.rtgo_host_session_open.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.rtgo_host_session_frame.argtypes = [C.c_void_p]
        L.rtgo_host_session_move_camera.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 3
        L.rtgo_host_session_resize.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.rtgo_host_session_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.rtgo_host_session_close.argtypes = [C.c_void_p]
        self._L = L
        self._h = C.c_void_p()
        self.width, self.height = width, height
        self._ok(L.rtgo_host_session_open(name.encode(), mode.encode(), width, height, sample, int(ambient), device, C.byref(self._h)))

    def _ok(self, rc):
        if rc != 0:
            raise capi.RtgoError("host session call failed (%d): %s" % (rc, self._L.rtgo_host_last_error().decode()))

    def frame(self):
        self._ok(self._L.rtgo_host_session_frame(self._h))

    def move_camera(self, eye, lookat, up):
        a = [np.ascontiguousarray(v, dtype=np.float32) for v in (eye, lookat, up)]
        self._ok(self._L.rtgo_host_session_move_camera(self._h, *[x.ctypes.data_as(C.POINTER(C.c_float)) for x in a]))

    def resize(self, width, height):
        self._ok(self._L.rtgo_host_session_resize(self._h, width, height))
        self.width, self.height = width, height

    def read(self):
        acc = np.empty((self.height, self.width, 4), dtype=np.float32)
        img = np.empty((self.height, self.width, 4), dtype=np.uint8)
        fc = C.c_uint32(0)
        self._ok(self._L.rtgo_host_session_read(self._h, img.ctypes.data, acc.ctypes.data, C.byref(fc)))
        return acc, img, int(fc.value)

    def close(self):
        if self._h:
            self._L.rtgo_host_session_close(self._h)
            self._h = None
