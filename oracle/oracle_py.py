"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
raytracingo_amd/ may (tests/test_layout_rules.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PROBE_PATH = os.path.join(HERE, "_ref", "libref_probe.so")

MAX_PRIMS = 512
MAX_LIGHTS = 10
SCENES = ["cornell", "slide", "mirror_spheres", "plateau", "window", "checkered", "balls", "soft_mirrors"]
CYLINDER, DISK, RECTANGLE, SPHERE = 0, 1, 2, 3


class Prim(C.Structure):
    _fields_ = [("type", C.c_int32), ("M", C.c_float * 16), ("kd", C.c_float * 3), ("kr", C.c_float * 3),
                ("specularity", C.c_float), ("Le", C.c_float * 3)]


class Light(C.Structure):
    _fields_ = [("corner", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("normal", C.c_float * 3),
                ("color", C.c_float * 3), ("falloff", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("n_prims", C.c_int32), ("n_lights", C.c_int32), ("prims", Prim * MAX_PRIMS),
                ("aabb", (C.c_float * 6) * MAX_PRIMS), ("lights", Light * MAX_LIGHTS), ("eye", C.c_float * 3),
                ("U", C.c_float * 3), ("V", C.c_float * 3), ("W", C.c_float * 3), ("bg", C.c_float * 3)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sqrt_spp", C.c_int32), ("max_depth", C.c_int32),
                ("frame_count", C.c_uint32), ("path_tracing", C.c_int32), ("use_ambient", C.c_int32),
                ("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32), ("band_h", C.c_uint32),
                ("n_ranks", C.c_uint32), ("rank", C.c_uint32), ("mode", C.c_int32), ("threads", C.c_int32),
                ("host_double_trig", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("rays_radiance", C.c_uint64 * 8), ("rays_occlusion", C.c_uint64), ("rays_total", C.c_uint64),
                ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64), ("hits", C.c_uint64)]

    def as_dict(self):
        return {"rays_radiance": list(self.rays_radiance), "rays_occlusion": int(self.rays_occlusion),
                "rays_total": int(self.rays_total), "node_visits": int(self.node_visits),
                "prim_tests": int(self.prim_tests), "hits": int(self.hits)}


class Node(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("left", C.c_int32), ("right", C.c_int32)]


class WhittedScene(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("indices", C.c_void_p), ("tri_material", C.c_void_p),
                ("materials", C.c_void_p), ("lights", C.c_void_p), ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("n_materials", C.c_uint32), ("n_lights", C.c_uint32), ("eye", C.c_float * 3), ("U", C.c_float * 3),
                ("V", C.c_float * 3), ("W", C.c_float * 3), ("miss", C.c_float * 3), ("texcoords", C.c_void_p), ("mat_tex", C.c_void_p)]


class Tex(C.Structure):
    _fields_ = [("px", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32)]


class MatTex(C.Structure):
    _fields_ = [("base_color", Tex), ("metallic_roughness", Tex), ("normal", Tex)]


def build(force=False):
    """Compile the oracle (and the reference probe when /root/reference is mounted). Building the checker is not using it."""
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in ("rtgo_oracle.c", "rtgo_oracle_scenes.c", "rtgo_oracle_whitted.c", "rtgo_oracle.h")):
        subprocess.check_call(["make", "-C", HERE, "-s", "all"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        u32p = C.POINTER(C.c_uint32)
        L.oracle_tea16.restype = C.c_uint32
        L.oracle_tea16.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_lcg.restype = C.c_uint32
        L.oracle_lcg.argtypes = [u32p]
        L.oracle_rnd.restype = C.c_float
        L.oracle_rnd.argtypes = [u32p]
        L.oracle_mat_mul.argtypes = [fp, fp, fp]
        L.oracle_mat_vec4.argtypes = [fp, fp, fp]
        L.oracle_mat_transpose.argtypes = [fp, fp]
        L.oracle_mat_det.restype = C.c_float
        L.oracle_mat_det.argtypes = [fp]
        L.oracle_mat_inverse.argtypes = [fp, fp]
        L.oracle_mat_rotate.argtypes = [C.c_float] * 4 + [fp]
        L.oracle_mat_translate.argtypes = [C.c_float] * 3 + [fp]
        L.oracle_mat_scale.argtypes = [C.c_float] * 3 + [fp]
        L.oracle_normalize3.argtypes = [fp, fp]
        L.oracle_camera_uvw.argtypes = [fp, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.oracle_prim_aabb.argtypes = [fp, fp]
        L.oracle_light_from_matrix.argtypes = [fp, fp, C.c_float, C.POINTER(Light)]
        L.oracle_intersect.restype = C.c_int
        L.oracle_intersect.argtypes = [C.POINTER(Prim), fp, fp, fp, fp]
        L.oracle_hemisphere.argtypes = [fp, fp, C.c_float, u32p, fp]
        L.oracle_scene_create.restype = C.c_int
        L.oracle_scene_create.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(Scene)]
        L.oracle_tea4.restype = C.c_uint32
        L.oracle_tea4.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_tri_intersect.restype = C.c_int
        L.oracle_tri_intersect.argtypes = [fp, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.oracle_whitted_render.restype = C.c_int
        L.oracle_whitted_render.argtypes = [C.POINTER(WhittedScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_uint64), C.c_int]
        L.oracle_tex2d.restype = None
        L.oracle_tex2d.argtypes = [C.POINTER(Tex), C.c_float, C.c_float, fp]
        L.oracle_material.restype = C.c_int
        L.oracle_material.argtypes = [C.c_char_p, fp]
        L.oracle_lbvh_build.restype = C.c_int
        L.oracle_lbvh_build.argtypes = [C.c_void_p, C.c_int, C.POINTER(Node), u32p, C.POINTER(C.c_int32)]
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.POINTER(Scene), C.POINTER(Frame), C.c_void_p, C.c_void_p, C.POINTER(Counters)]
        L.oracle_local_rows.restype = C.c_uint32
        L.oracle_local_rows.argtypes = [C.c_uint32] * 4
        _lib = L
    return _lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def scene(name, width, height):
    sc = Scene()
    if lib().oracle_scene_create(name.encode(), width, height, C.byref(sc)) != 0:
        raise ValueError("unknown scene %r" % name)
    return sc


def material(name):
    """engine/materials.h constant `name` as the oracle states it: kd[3], kr[3], Le[3], specularity (None: unknown name)"""
    v = np.zeros(10, dtype=np.float32)
    return v if lib().oracle_material(name.encode(), fptr(v)) == 0 else None


def scene_tables(sc):
    """numpy views of a Scene: dict(type[n], M[n,16], mat[n,10] (kd,kr,spec,Le), aabb[n,6], lights[nl,16], cam[12], bg[3])"""
    n = sc.n_prims
    t = np.array([sc.prims[i].type for i in range(n)], dtype=np.int32)
    M = np.array([list(sc.prims[i].M) for i in range(n)], dtype=np.float32)
    mat = np.array([list(sc.prims[i].kd) + list(sc.prims[i].kr) + [sc.prims[i].specularity] + list(sc.prims[i].Le)
                    for i in range(n)], dtype=np.float32)
    aabb = np.array([list(sc.aabb[i]) for i in range(n)], dtype=np.float32)
    lights = np.array([list(L.corner) + list(L.v1) + list(L.v2) + list(L.normal) + list(L.color) + [L.falloff]
                       for L in list(sc.lights)[:sc.n_lights]], dtype=np.float32)
    cam = np.array(list(sc.eye) + list(sc.U) + list(sc.V) + list(sc.W), dtype=np.float32)
    return {"type": t, "M": M, "mat": mat, "aabb": aabb, "lights": lights, "cam": cam,
            "bg": np.array(list(sc.bg), dtype=np.float32)}


def frame(width, height, sqrt_spp=1, frame_count=0, path=True, ambient=False, window=None, bands=(1, 1, 0), mode=1,
          threads=0, max_depth=5, host_double_trig=False):
    x0, y0, w, h = window if window is not None else (0, 0, width, height)
    band_h, n_ranks, rank = bands
    return Frame(width, height, sqrt_spp, max_depth, frame_count, int(path), int(ambient), x0, y0, w, h, band_h,
                 n_ranks, rank, mode, threads, int(host_double_trig))


def local_rows(h, band_h, n_ranks, rank):
    return int(lib().oracle_local_rows(h, band_h, n_ranks, rank))


def render(sc, fr, accum_prev=None):
    """returns (accum float32 [rows,w,4], image uint8 [rows,w,4], counters dict)"""
    rows = local_rows(fr.h, fr.band_h or 1, fr.n_ranks or 1, fr.rank)
    accum = np.zeros((rows, fr.w, 4), dtype=np.float32) if accum_prev is None else np.array(accum_prev, dtype=np.float32)
    assert accum.shape == (rows, fr.w, 4)
    image = np.zeros((rows, fr.w, 4), dtype=np.uint8)
    ctr = Counters()
    lib().oracle_render(C.byref(sc), C.byref(fr), accum.ctypes.data, image.ctypes.data, C.byref(ctr))
    return accum, image, ctr.as_dict()


def lbvh(aabb):
    aabb = f32(aabb)
    n = aabb.shape[0]
    nodes = (Node * (2 * n))()
    code = np.zeros(n, dtype=np.uint32)
    order = np.zeros(n, dtype=np.int32)
    lib().oracle_lbvh_build(aabb.ctypes.data, n, nodes, code.ctypes.data_as(C.POINTER(C.c_uint32)),
                            order.ctypes.data_as(C.POINTER(C.c_int32)))
    arr = np.zeros((2 * n - 1, 8), dtype=np.float32)
    links = np.zeros((2 * n - 1, 2), dtype=np.int32)
    for i in range(2 * n - 1):
        arr[i, :3] = list(nodes[i].bmin)
        arr[i, 3:6] = list(nodes[i].bmax)
        links[i] = (nodes[i].left, nodes[i].right)
    return arr[:, :6], links, code, order


def scene_from_tables(types, M, mat, lights16, cam12, bg=(0.0, 0.0, 0.0), aabb=None):
    """build an oracle Scene from flattened tables (the same arrays the C ABI takes); AABBs default to the CubeBox rule"""
    sc = Scene()
    n = len(types)
    assert n <= MAX_PRIMS
    sc.n_prims = n
    M = f32(M).reshape(n, 16)
    mat = f32(mat).reshape(n, 10)
    for i in range(n):
        p = sc.prims[i]
        p.type = int(types[i])
        p.M[:] = M[i].tolist()
        p.kd[:] = mat[i, 0:3].tolist()
        p.kr[:] = mat[i, 3:6].tolist()
        p.specularity = float(mat[i, 6])
        p.Le[:] = mat[i, 7:10].tolist()
        if aabb is None:
            bb = np.zeros(6, dtype=np.float32)
            lib().oracle_prim_aabb(fptr(M[i].copy()), fptr(bb))
        else:
            bb = f32(aabb[i])
        sc.aabb[i][:] = bb.tolist()
    lights16 = f32(lights16).reshape(-1, 16)
    sc.n_lights = lights16.shape[0]
    for i in range(sc.n_lights):
        r = lights16[i].tolist()
        L = sc.lights[i]
        L.corner[:], L.v1[:], L.v2[:], L.normal[:], L.color[:], L.falloff = r[0:3], r[3:6], r[6:9], r[9:12], r[12:15], r[15]
    cam12 = f32(cam12)
    sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = cam12[0:3].tolist(), cam12[3:6].tolist(), cam12[6:9].tolist(), cam12[9:12].tolist()
    sc.bg[:] = list(bg)
    return sc


def light_from_matrix(M, color=(1.0, 1.0, 1.0), falloff=0.0):
    """SurfaceLight record (16 floats) of a unit rectangle under M (light.cpp:9-28)"""
    L = Light()
    lib().oracle_light_from_matrix(fptr(f32(M).reshape(16).copy()), fptr(f32(color)), float(falloff), C.byref(L))
    return np.array(list(L.corner) + list(L.v1) + list(L.v2) + list(L.normal) + list(L.color) + [L.falloff], dtype=np.float32)


def tex2d(texture, u, v):
    """oracle_tex2d on a uint8 [h, w, 4] array"""
    a = np.ascontiguousarray(texture, dtype=np.uint8)
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_tex2d(C.byref(Tex(a.ctypes.data, a.shape[1], a.shape[0])), float(u), float(v), fptr(out))
    return out


def whitted_render(mesh, cam12, width, height, subframes=1, threads=0):
    """the oracle's whitted path (rtgo_oracle_whitted.c) over `subframes` accumulated subframes.
    mesh: dict(positions [nv,3] f32, normals [nv,3] f32 or None, indices [nt,3] u32, tri_material [nt] u32 or None,
    materials [nm,6] f32 (base_color rgba, metallic, roughness), lights [nl,8] (color rgb, intensity, position xyz, falloff as int bits),
    miss (3,)).  Returns (accum [h,w,4] f32, image [h,w,4] u8, {rays_total, rays_occlusion})."""
    pos = np.ascontiguousarray(mesh["positions"], dtype=np.float32)
    nrm = None if mesh.get("normals") is None else np.ascontiguousarray(mesh["normals"], dtype=np.float32)
    idx = np.ascontiguousarray(mesh["indices"], dtype=np.uint32)
    tm = None if mesh.get("tri_material") is None else np.ascontiguousarray(mesh["tri_material"], dtype=np.uint32)
    mats = np.ascontiguousarray(mesh["materials"], dtype=np.float32)
    lights = np.ascontiguousarray(mesh["lights"], dtype=np.float32).reshape(-1, 8)
    s = WhittedScene()
    s.positions, s.normals = pos.ctypes.data, (nrm.ctypes.data if nrm is not None else None)
    s.indices, s.tri_material = idx.ctypes.data, (tm.ctypes.data if tm is not None else None)
    s.materials, s.lights = mats.ctypes.data, (lights.ctypes.data if len(lights) else None)
    s.n_vertices, s.n_triangles, s.n_materials, s.n_lights = len(pos), len(idx), len(mats), len(lights)
    cam = f32(cam12)
    s.eye[:], s.U[:], s.V[:], s.W[:] = cam[0:3].tolist(), cam[3:6].tolist(), cam[6:9].tolist(), cam[9:12].tolist()
    s.miss[:] = f32(mesh["miss"]).tolist()
    # optional: texcoords [nv, 2] and textures = {material index: (base_color, metallic_roughness, normal)}, each uint8 [h, w, 4] or None
    uv = None if mesh.get("texcoords") is None else np.ascontiguousarray(mesh["texcoords"], dtype=np.float32)
    s.texcoords = uv.ctypes.data if uv is not None else None
    keep = []
    mt = None
    if mesh.get("textures"):
        mt = (MatTex * len(mats))()
        for mi, triple in mesh["textures"].items():
            for name, t in zip(("base_color", "metallic_roughness", "normal"), triple):
                if t is None:
                    continue
                a = np.ascontiguousarray(t, dtype=np.uint8)
                keep.append(a)
                setattr(mt[mi], name, Tex(a.ctypes.data, a.shape[1], a.shape[0]))
        s.mat_tex = C.addressof(mt)
    acc = np.zeros((height, width, 4), dtype=np.float32)
    img = np.zeros((height, width, 4), dtype=np.uint8)
    rays = (C.c_uint64 * 2)(0, 0)
    if threads <= 0:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1
    for sf in range(subframes):
        rc = lib().oracle_whitted_render(C.byref(s), width, height, sf, acc.ctypes.data, img.ctypes.data, rays, threads)
        assert rc == 0
    return acc, img, {"rays_total": int(rays[0]), "rays_occlusion": int(rays[1])}
