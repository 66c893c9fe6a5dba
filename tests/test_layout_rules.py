"""Structural rules of the tier: the product never touches the oracle, and fails loudly without its HIP library."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(top, exts):
    for d, _, fs in os.walk(os.path.join(ROOT, top)):
        for f in fs:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_never_references_the_oracle():
    for path in _files("raytracingo_amd", (".py", ".h", ".hip", ".cpp")):
        src = open(path, errors="ignore").read()
        code = "\n".join(l for l in src.splitlines() if not l.strip().startswith(("#", "//", "*", '"""')))
        assert not re.search(r"\boracle_py\b|liboracle|rtgo_oracle", code), path


def test_oracle_says_it_is_test_infrastructure():
    for f in ("rtgo_oracle.h", "rtgo_oracle.c", "rtgo_oracle_scenes.c", "oracle_py.py", "ref_probe.cpp"):
        head = open(os.path.join(ROOT, "oracle", f)).read(1500).upper()
        assert "TEST INFRASTRUCTURE" in head or "ORACLE-SIDE ONLY" in head, f


def test_no_reference_sources_in_repo():
    # nothing under the repo may be a copy of the reference's kernel: spot-check distinctive identifiers
    for path in list(_files("raytracingo_amd", (".h", ".hip", ".cpp"))) + list(_files("oracle", (".c", ".h", ".cpp"))):
        src = open(path, errors="ignore").read()
        assert "optixTrace(" not in src and "__raygen__rg()" not in src and "optixReportIntersection(\n" not in src, path


def test_missing_library_raises(monkeypatch):
    from raytracingo_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", os.path.join(ROOT, "raytracingo_amd", "does_not_exist.so"))
    try:
        capi.load()
    except capi.RtgoError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("capi.load() must fail when librtgo_hip.so is absent")
