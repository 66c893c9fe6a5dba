"""The oracle's building blocks against the REFERENCE'S OWN CODE (tests/golden/ref_blocks.json, produced from
cuda/random.h, sutil/Matrix.h, sutil/vec_math.h, sutil/Camera.cpp and glm by oracle/gen_golden.py). Bit-exact."""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(GOLD, "ref_blocks.json")) as f:
        return json.load(f)


def f(bits):
    return np.array(bits, dtype=np.uint32).view(np.float32)


def b(arr):
    return np.ascontiguousarray(arr, dtype=np.float32).view(np.uint32).reshape(-1).tolist()


def test_tea16(oracle, ref):
    for c in ref["tea16"]:
        assert oracle.lib().oracle_tea16(c["in"][0], c["in"][1]) == c["out"]


def test_rnd_sequences(oracle, ref):
    for c in ref["rnd"]:
        s = C.c_uint32(c["seed"])
        got = [b(np.float32(oracle.lib().oracle_rnd(C.byref(s))))[0] for _ in c["rnd_bits"]]
        assert got == c["rnd_bits"]
        assert s.value == c["final_state"]
        assert all(0.0 <= v < 1.0 for v in f(got))


def test_rotate_translate_scale(oracle, ref):
    L = oracle.lib()
    for c in ref["rotate"]:
        a = f(c["in"])
        o = np.zeros(16, dtype=np.float32)
        L.oracle_mat_rotate(a[0], a[1], a[2], a[3], oracle.fptr(o))
        assert b(o) == c["out"]
    for c in ref["translate_scale"]:
        v = f(c["in"])
        o = np.zeros(16, dtype=np.float32)
        L.oracle_mat_translate(v[0], v[1], v[2], oracle.fptr(o))
        assert b(o) == c["translate"]
        L.oracle_mat_scale(v[0], v[1], v[2], oracle.fptr(o))
        assert b(o) == c["scale"]


def test_matrix_mul_inverse_det_transpose_vec(oracle, ref):
    L = oracle.lib()
    for c in ref["matrix"]:
        m, other = f(c["m"]).copy(), f(c["other"]).copy()
        o = np.zeros(16, dtype=np.float32)
        L.oracle_mat_mul(oracle.fptr(m), oracle.fptr(other), oracle.fptr(o))
        assert b(o) == c["mul"]
        L.oracle_mat_inverse(oracle.fptr(m), oracle.fptr(o))
        assert b(o) == c["inverse"]
        L.oracle_mat_transpose(oracle.fptr(m), oracle.fptr(o))
        assert b(o) == c["transpose"]
        assert b(np.float32(L.oracle_mat_det(oracle.fptr(m))))[0] == c["det"]
        v4, o4 = f(c["v4"]).copy(), np.zeros(4, dtype=np.float32)
        L.oracle_mat_vec4(oracle.fptr(m), oracle.fptr(v4), oracle.fptr(o4))
        assert b(o4) == c["m_v4"]


def test_normalize_and_light_normal(oracle, ref):
    L = oracle.lib()
    for c in ref["vec"]:
        a, bb = f(c["a"]).copy(), f(c["b"]).copy()
        o = np.zeros(3, dtype=np.float32)
        L.oracle_normalize3(oracle.fptr(a), oracle.fptr(o))
        assert b(o) == c["normalize_a"]
        # SurfaceLight: v1 = M*(1,0,0,0) and v2 = M*(0,0,-1,0) are column 0 and minus column 2 of M; build such an M
        M = np.zeros(16, dtype=np.float32)
        M[[0, 4, 8]] = a
        M[[2, 6, 10]] = -bb
        M[15] = 1
        col = np.zeros(3, dtype=np.float32)
        light = oracle.Light()
        L.oracle_light_from_matrix(oracle.fptr(M), oracle.fptr(col), 0.0, C.byref(light))
        assert b(list(light.v1)) == c["a"] and b(list(light.v2)) == c["b"]
        assert b(list(light.normal)) == c["glm_normal"]


def test_camera_uvw(oracle, ref):
    L = oracle.lib()
    eye, look, up = oracle.f32([0, 0, 14]), oracle.f32([0, 0, 0]), oracle.f32([0, 1, 0])
    for c in ref["camera"]:
        U, V, W = [np.zeros(3, dtype=np.float32) for _ in range(3)]
        asp = np.float32(np.float32(c["w"]) / np.float32(c["h"]))
        L.oracle_camera_uvw(oracle.fptr(eye), oracle.fptr(look), oracle.fptr(up), 60.0, asp, oracle.fptr(U), oracle.fptr(V), oracle.fptr(W))
        assert (b(U), b(V), b(W)) == (c["U"], c["V"], c["W"])
        sc = oracle.scene("cornell", c["w"], c["h"])
        assert b(list(sc.U)) == c["U"] and b(list(sc.V)) == c["V"] and b(list(sc.W)) == c["W"]


def test_scene_inverses_use_the_pinned_inverse(oracle, ref):
    """cos/sin of M_PIf/3 agree between the float and double overloads (scene.cpp:97,124: overload is toolchain-dependent)"""
    import math
    x = np.float32(3.14159265358979323846) / np.float32(3.0)
    assert np.float32(math.cos(float(x))) == np.cos(x, dtype=np.float32)
    assert np.float32(math.sin(float(x))) == np.sin(x, dtype=np.float32)


def test_materials_are_the_references(oracle, ref):
    """the 33 constants of engine/materials.h:13-283, read through the reference's own BasicMaterial getters
    (oracle/ref_probe.cpp over materials.h + basicmaterial.cpp), against the oracle's table: bit for bit, name for name"""
    assert len(ref["materials"]) == 33
    for name, want in ref["materials"].items():
        got = oracle.material(name)
        assert got is not None, name
        assert b(got) == want, (name, got.tolist(), f(want).tolist())
    assert oracle.material("teapot") is None


def test_every_scene_material_is_a_pinned_constant(oracle, ref):
    """each primitive of each scene carries one of the pinned constants (scene.cpp picks them by name; balls picks them
    through the scene RNG, scene.cpp:592-604)"""
    pinned = {tuple(v) for v in ref["materials"].values()}
    for name in oracle.SCENES:
        t = oracle.scene_tables(oracle.scene(name, 64, 64))
        m = t["mat"]    # kd, kr, specularity, Le (rtgo_prim order)
        for row in m:
            key = tuple(b(np.concatenate([row[0:6], row[7:10], row[6:7]])))
            assert key in pinned, (name, row.tolist())
