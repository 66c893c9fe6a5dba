"""ctypes binding of the C ABI in include/rtgo.h (librtgo_hip.so).

This is plumbing for the Python drivers (bench.py, tests, the multi-GPU band driver): the product is the shared
library.  There is deliberately no fallback: if the library is missing or no gfx950 GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RTGO_HIP_LIB: developer override to A/B an experimental build of the same ABI (never a different backend)
LIB_PATH = os.environ.get("RTGO_HIP_LIB") or os.path.join(PKG_DIR, "librtgo_hip.so")

RTGO_MAX_PRIMS = 512
RTGO_MAX_LIGHTS = 10
CYLINDER, DISK, RECTANGLE, SPHERE = 0, 1, 2, 3

# every symbol include/rtgo.h declares (tests/test_capi_symbols.py checks the header against this list and the .so)
SYMBOLS = [
    "rtgo_create", "rtgo_destroy", "rtgo_last_error", "rtgo_set_stream", "rtgo_set_scene", "rtgo_set_camera",
    "rtgo_set_background", "rtgo_set_lights", "rtgo_resize", "rtgo_bind_output", "rtgo_launch", "rtgo_sync",
    "rtgo_read_image", "rtgo_read_accum", "rtgo_write_accum", "rtgo_get_stats", "rtgo_reset_stats", "rtgo_read_bvh",
    "rtgo_local_rows", "rtgo_abi_version", "rtgo_assemble_bands",
    "rtgo_whitted_set_mesh", "rtgo_whitted_set_lights", "rtgo_whitted_set_miss_color", "rtgo_whitted_launch",
    "rtgo_whitted_set_texcoords", "rtgo_whitted_set_material_textures",
]


class RtgoError(RuntimeError):
    pass


class Prim(C.Structure):
    _fields_ = [("type", C.c_uint32), ("model", C.c_float * 16), ("kd", C.c_float * 3), ("kr", C.c_float * 3),
                ("specularity", C.c_float), ("Le", C.c_float * 3)]


class Aabb(C.Structure):
    _fields_ = [("minX", C.c_float), ("minY", C.c_float), ("minZ", C.c_float), ("maxX", C.c_float),
                ("maxY", C.c_float), ("maxZ", C.c_float)]


class Light(C.Structure):
    _fields_ = [("corner", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("normal", C.c_float * 3),
                ("color", C.c_float * 3), ("falloff", C.c_float)]


class Frame(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("sqrt_spp", C.c_int32),
                ("max_trace_depth", C.c_int32), ("frame_count", C.c_uint32), ("path_tracing", C.c_uint32),
                ("use_ambient", C.c_uint32), ("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32),
                ("h", C.c_uint32), ("band_h", C.c_uint32), ("n_ranks", C.c_uint32), ("rank", C.c_uint32),
                ("collect_stats", C.c_uint32), ("reserve_cus", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_total", C.c_uint64), ("rays_occlusion", C.c_uint64), ("node_visits", C.c_uint64),
                ("prim_tests", C.c_uint64), ("hits", C.c_uint64), ("last_launch_ms", C.c_float),
                ("total_launch_ms", C.c_float), ("launches", C.c_uint32), ("lbvh_depth", C.c_uint32),
                ("dbg_fast_boxes", C.c_uint64), ("dbg_fast_tests", C.c_uint64), ("rays_culled", C.c_uint64),
                ("launches_canonical", C.c_uint32), ("cuboid_groups", C.c_uint32),
                ("guard_reach", C.c_float), ("guard_quadric", C.c_float),
                ("last_variant", C.c_uint32), ("launches_trial", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load():
    """dlopen librtgo_hip.so and declare the prototypes. Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtgoError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                        "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    fp = C.POINTER(C.c_float)
    L.rtgo_abi_version.restype = C.c_uint32
    L.rtgo_last_error.restype = C.c_char_p
    L.rtgo_last_error.argtypes = [vp]
    L.rtgo_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rtgo_destroy.argtypes = [vp]
    L.rtgo_set_stream.argtypes = [vp, vp]
    L.rtgo_set_scene.argtypes = [vp, C.POINTER(Prim), C.POINTER(Aabb), C.c_uint32]
    L.rtgo_set_camera.argtypes = [vp, fp, fp, fp, fp]
    L.rtgo_set_background.argtypes = [vp, fp]
    L.rtgo_set_lights.argtypes = [vp, C.POINTER(Light), C.c_int]
    L.rtgo_resize.argtypes = [vp, C.c_size_t]
    L.rtgo_bind_output.argtypes = [vp, vp, vp, C.c_size_t]
    L.rtgo_launch.argtypes = [vp, C.POINTER(Frame)]
    L.rtgo_sync.argtypes = [vp]
    L.rtgo_read_image.argtypes = [vp, vp, C.c_size_t]
    L.rtgo_read_accum.argtypes = [vp, vp, C.c_size_t]
    L.rtgo_write_accum.argtypes = [vp, vp, C.c_size_t]
    L.rtgo_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rtgo_reset_stats.argtypes = [vp]
    L.rtgo_read_bvh.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t]
    L.rtgo_local_rows.restype = C.c_uint32
    L.rtgo_local_rows.argtypes = [C.c_uint32] * 4
    L.rtgo_assemble_bands.argtypes = [vp, vp, vp, vp] + [C.c_uint32] * 6
    L.rtgo_whitted_set_mesh.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_uint32]
    L.rtgo_whitted_set_lights.argtypes = [vp, vp, C.c_uint32]
    L.rtgo_whitted_set_miss_color.argtypes = [vp, fp]
    L.rtgo_whitted_launch.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.rtgo_whitted_set_texcoords.argtypes = [vp, vp, C.c_uint32]
    L.rtgo_whitted_set_material_textures.argtypes = [vp, C.c_uint32, C.POINTER(Texture), C.POINTER(Texture), C.POINTER(Texture)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("rtgo_last_error", "rtgo_local_rows", "rtgo_abi_version"):
            fn.restype = C.c_int
    _lib = L
    return L


def local_rows(h, band_h, n_ranks, rank):
    return int(load().rtgo_local_rows(h, band_h, n_ranks, rank))


def _f3(v):
    a = np.ascontiguousarray(v, dtype=np.float32).reshape(3)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


class Context:
    """One rtgo_ctx (one GPU). Thin: every method is one C-ABI call plus error translation."""

    def __init__(self, device=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.rtgo_create(int(device), C.byref(h))
        if rc != 0:
            raise RtgoError("rtgo_create(%d) failed (%d): %s" % (device, rc, self._lib.rtgo_last_error(None).decode()))
        self._h = h
        self.pixels = 0

    def _check(self, rc, what):
        if rc != 0:
            raise RtgoError("%s failed (%d): %s" % (what, rc, self._lib.rtgo_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rtgo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.rtgo_set_stream(self._h, C.c_void_p(hip_stream_ptr or 0)), "rtgo_set_stream")

    def set_scene(self, types, models, materials, aabbs=None):
        """types[n] int, models[n,16] float32 row-major, materials[n,10] = kd(3) kr(3) specularity Le(3)."""
        types = np.asarray(types)
        models = np.ascontiguousarray(models, dtype=np.float32).reshape(-1, 16)
        materials = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 10)
        n = len(types)
        arr = (Prim * max(n, 1))()
        for i in range(n):
            p = arr[i]
            p.type = int(types[i])
            p.model[:] = models[i].tolist()
            p.kd[:] = materials[i, 0:3].tolist()
            p.kr[:] = materials[i, 3:6].tolist()
            p.specularity = float(materials[i, 6])
            p.Le[:] = materials[i, 7:10].tolist()
        bb = None
        if aabbs is not None:
            aabbs = np.ascontiguousarray(aabbs, dtype=np.float32).reshape(-1, 6)
            bb = (Aabb * n)()
            for i in range(n):
                (bb[i].minX, bb[i].minY, bb[i].minZ, bb[i].maxX, bb[i].maxY, bb[i].maxZ) = aabbs[i].tolist()
        self._check(self._lib.rtgo_set_scene(self._h, arr, bb, n), "rtgo_set_scene")
        self.n_prims = n

    def set_camera(self, eye, U, V, W):
        a = [_f3(x) for x in (eye, U, V, W)]
        self._check(self._lib.rtgo_set_camera(self._h, a[0][1], a[1][1], a[2][1], a[3][1]), "rtgo_set_camera")

    def set_background(self, rgb):
        a = _f3(rgb)
        self._check(self._lib.rtgo_set_background(self._h, a[1]), "rtgo_set_background")

    def set_lights(self, lights16):
        """lights16[n,16] = corner v1 v2 normal color falloff"""
        lights16 = np.ascontiguousarray(lights16, dtype=np.float32).reshape(-1, 16)
        n = lights16.shape[0]
        arr = (Light * max(n, 1))()
        for i in range(n):
            L = arr[i]
            r = lights16[i].tolist()
            L.corner[:], L.v1[:], L.v2[:], L.normal[:], L.color[:], L.falloff = r[0:3], r[3:6], r[6:9], r[9:12], r[12:15], r[15]
        self._check(self._lib.rtgo_set_lights(self._h, arr, n), "rtgo_set_lights")

    def resize(self, pixels):
        self._check(self._lib.rtgo_resize(self._h, int(pixels)), "rtgo_resize")
        self.pixels = int(pixels)

    def bind_output(self, d_accum_ptr, d_image_ptr, pixels):
        self._check(self._lib.rtgo_bind_output(self._h, C.c_void_p(d_accum_ptr), C.c_void_p(d_image_ptr), int(pixels)),
                    "rtgo_bind_output")
        self.pixels = int(pixels)

    def launch(self, frame):
        self._check(self._lib.rtgo_launch(self._h, C.byref(frame)), "rtgo_launch")

    def sync(self):
        self._check(self._lib.rtgo_sync(self._h), "rtgo_sync")

    def read_accum(self, rows, w):
        out = np.empty((rows, w, 4), dtype=np.float32)
        self._check(self._lib.rtgo_read_accum(self._h, out.ctypes.data, out.nbytes), "rtgo_read_accum")
        return out

    def read_image(self, rows, w):
        out = np.empty((rows, w, 4), dtype=np.uint8)
        self._check(self._lib.rtgo_read_image(self._h, out.ctypes.data, out.nbytes), "rtgo_read_image")
        return out

    def write_accum(self, accum):
        a = np.ascontiguousarray(accum, dtype=np.float32)
        self._check(self._lib.rtgo_write_accum(self._h, a.ctypes.data, a.nbytes), "rtgo_write_accum")

    # ---- the whitted triangle path (cuda/whitted.cu), one call per C-ABI entry ----
    def whitted_set_mesh(self, positions, normals, indices, tri_material, materials):
        pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        tm = None if tri_material is None else np.ascontiguousarray(tri_material, dtype=np.uint32)
        mats = np.ascontiguousarray(materials, dtype=np.float32).reshape(-1, 6)
        self._check(self._lib.rtgo_whitted_set_mesh(self._h, pos.ctypes.data, nrm.ctypes.data if nrm is not None else None, len(pos),
                                                    idx.ctypes.data, tm.ctypes.data if tm is not None else None, len(idx),
                                                    mats.ctypes.data, len(mats)), "rtgo_whitted_set_mesh")

    def whitted_set_texcoords(self, uv):
        if uv is None:
            self._check(self._lib.rtgo_whitted_set_texcoords(self._h, None, 0), "rtgo_whitted_set_texcoords")
            return
        a = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        self._check(self._lib.rtgo_whitted_set_texcoords(self._h, a.ctypes.data, len(a)), "rtgo_whitted_set_texcoords")

    def whitted_set_material_textures(self, material, base_color=None, metallic_roughness=None, normal=None):
        """each texture: uint8 array [h, w, 4] (row 0 first) or None"""
        keep, ptrs = [], []
        for t in (base_color, metallic_roughness, normal):
            if t is None:
                ptrs.append(None)
                continue
            a = np.ascontiguousarray(t, dtype=np.uint8)
            assert a.ndim == 3 and a.shape[2] == 4
            keep.append(a)
            ptrs.append(C.pointer(Texture(a.ctypes.data, a.shape[1], a.shape[0])))
        self._check(self._lib.rtgo_whitted_set_material_textures(self._h, int(material), ptrs[0], ptrs[1], ptrs[2]), "rtgo_whitted_set_material_textures")

    def whitted_set_lights(self, lights8):
        l = np.ascontiguousarray(lights8, dtype=np.float32).reshape(-1, 8)
        self._check(self._lib.rtgo_whitted_set_lights(self._h, l.ctypes.data if len(l) else None, len(l)), "rtgo_whitted_set_lights")

    def whitted_set_miss_color(self, rgb):
        a, p = _f3(rgb)
        self._check(self._lib.rtgo_whitted_set_miss_color(self._h, p), "rtgo_whitted_set_miss_color")

    def whitted_launch(self, width, height, subframe):
        self._check(self._lib.rtgo_whitted_launch(self._h, width, height, subframe), "rtgo_whitted_launch")

    def stats(self):
        s = Stats()
        self._check(self._lib.rtgo_get_stats(self._h, C.byref(s)), "rtgo_get_stats")
        return s.as_dict()

    def reset_stats(self):
        self._check(self._lib.rtgo_reset_stats(self._h), "rtgo_reset_stats")

    def read_bvh(self):
        n = self.n_prims
        nodes = np.empty((2 * n - 1, 8), dtype=np.float32)
        inv = np.empty((n, 12), dtype=np.float32)
        aabb = np.empty((n, 6), dtype=np.float32)
        self._check(self._lib.rtgo_read_bvh(self._h, nodes.ctypes.data, nodes.nbytes, inv.ctypes.data, inv.nbytes,
                                            aabb.ctypes.data, aabb.nbytes), "rtgo_read_bvh")
        links = nodes.view(np.int32)[:, [3, 7]].copy()
        boxes = nodes[:, [0, 1, 2, 4, 5, 6]].copy()
        return boxes, links, inv, aabb


def make_frame(width, height, sqrt_spp=1, frame_count=0, path=True, ambient=False, window=None, bands=(4, 1, 0),
               max_depth=5, stats=False, reserve_cus=0):
    x0, y0, w, h = window if window is not None else (0, 0, width, height)
    band_h, n_ranks, rank = bands
    return Frame(width, height, sqrt_spp, max_depth, frame_count, int(path), int(ambient), x0, y0, w, h, band_h,
                 n_ranks, rank, int(stats), int(reserve_cus))
