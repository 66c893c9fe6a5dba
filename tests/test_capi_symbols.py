"""The C ABI library loads on a CPU-only box and exports exactly what include/rtgo.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from raytracingo_amd import _build, capi as m
    _build.build_all()
    m.load()
    return m


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtgo_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(capi):
    names = declared("rtgo.h")
    assert names == sorted(capi.SYMBOLS)
    L = capi.load()
    for n in names:
        assert getattr(L, n) is not None
    assert L.rtgo_abi_version() == 6


def test_host_header_symbols_are_exported(capi):
    from raytracingo_amd import scene
    L = scene.load()
    for n in declared("rtgo_host.h"):
        assert getattr(L, n) is not None


def test_struct_layouts(capi):
    # rtgo_prim = PRIMITIVE_TYPE + HitGroupData (104 B, params.h:103-110); SurfaceLight 64 B; OptixAabb 24 B
    assert C.sizeof(capi.Prim) == 108 and C.sizeof(capi.Light) == 64 and C.sizeof(capi.Aabb) == 24
    assert C.sizeof(capi.Frame) == 16 * 4 and C.sizeof(capi.Stats) == 104
    assert capi.Prim.kd.offset == 68 and capi.Prim.Le.offset == 96


def test_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.RtgoError) as e:
        capi.Context(0)
    assert "no CPU path" in str(e.value) or "no HIP device" in str(e.value)


def test_local_rows_matches_oracle(capi, oracle):
    for h, b, g in [(1080, 4, 8), (1080, 4, 3), (2160, 4, 8), (90, 4, 3), (7, 4, 2), (5, 1, 3), (4, 4, 8)]:
        tot = 0
        for r in range(g):
            assert capi.local_rows(h, b, g, r) == oracle.local_rows(h, b, g, r)
            tot += capi.local_rows(h, b, g, r)
        assert tot == h
