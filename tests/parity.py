"""Parity metric shared by the GPU tests.

Tolerance, stated: per channel |gpu - ref| <= 1e-4 * max(1, |ref|) on the float accumulation buffer.  Pixels outside
it are path flips (an ulp at an edge changes which primitive a bounce hits); their share is bounded and the image mean
must stay within 0.5 % per channel.  The 8-bit image may differ by at most 1 LSB on the in-tolerance pixels.
"""
import numpy as np

ABS_REL_TOL = 1e-4
MEAN_TOL = 5e-3


def compare(accum, ref, image=None, ref_image=None):
    a = np.asarray(accum, dtype=np.float64)[..., :3]
    r = np.asarray(ref, dtype=np.float64)[..., :3]
    tol = ABS_REL_TOL * np.maximum(1.0, np.abs(r))
    ok_px = (np.abs(a - r) <= tol).all(axis=-1)
    out = {
        "pixels": int(ok_px.size),
        "frac_within": float(ok_px.mean()) if ok_px.size else 1.0,
        "frac_bit_exact": float((np.asarray(accum)[..., :3] == np.asarray(ref)[..., :3]).all(axis=-1).mean()) if ok_px.size else 1.0,
        "max_abs": float(np.abs(a - r).max()) if ok_px.size else 0.0,
    }
    ma, mr = a.reshape(-1, 3).mean(0), r.reshape(-1, 3).mean(0)
    out["mean_rel"] = float(np.max(np.abs(ma - mr) / np.maximum(np.abs(mr), 1e-6)))
    if image is not None:
        d = np.abs(np.asarray(image, dtype=np.int32)[..., :3] - np.asarray(ref_image, dtype=np.int32)[..., :3]).max(axis=-1)
        out["image_max_lsb_within"] = int(d[ok_px].max()) if ok_px.any() else 0
    return out


def assert_parity(accum, ref, image=None, ref_image=None, min_frac=0.99, what=""):
    m = compare(accum, ref, image, ref_image)
    assert np.isfinite(np.asarray(accum)[..., :3]).all() == np.isfinite(np.asarray(ref)[..., :3]).all(), what
    assert m["frac_within"] >= min_frac, "%s: only %.4f of pixels within tolerance (%r)" % (what, m["frac_within"], m)
    assert m["mean_rel"] <= MEAN_TOL, "%s: image mean off by %.4g (%r)" % (what, m["mean_rel"], m)
    if image is not None:
        assert m["image_max_lsb_within"] <= 1, "%s: 8-bit image differs by %d LSB" % (what, m["image_max_lsb_within"])
    return m
