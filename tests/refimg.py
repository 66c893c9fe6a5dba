"""Helpers of tests/test_reference_renders.py: the decoded reference renders (tests/golden/ref_images/, made by make_fixtures.py from
/root/reference/images/*.jpg), region / block statistics, and the two renderers the pictures are held against -- the CPU oracle and the
HIP path through the C ABI -- under the camera and the white of each picture.  Statistics only: a screenshot of unknown frame count, JPEG
coded, rendered with --use_fast_math, cannot be compared pixel by pixel."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "golden", "ref_images")

_cache = {}


def fixtures():
    if "fx" not in _cache:
        meta = json.load(open(os.path.join(DIR, "ref_images.json")))
        arr = np.load(os.path.join(DIR, "ref_images.npz"))           # plain uint8 arrays (allow_pickle stays False)
        _cache["fx"] = (meta, {k: arr[k] for k in arr.files})
    return _cache["fx"]


def display(image_or_accum):
    """our buffers (bottom row first, kernel.cu:187-204: pixel (0,0) = bottom left) -> display orientation, RGB"""
    return np.asarray(image_or_accum)[::-1, :, :3]


def to8(accum):
    """make_color (kernel.cu:90-98) of a float accumulation buffer: clamp, x 255, truncate; NO gamma"""
    return (np.clip(np.asarray(accum, dtype=np.float32)[..., :3], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)


def region_mean(img, box, scale=1):
    """mean RGB over a region given in 600 x 600 display coordinates [x0, y0, x1, y1); scale = 600 / our image size"""
    x0, y0, x1, y1 = [int(round(v / scale)) for v in box]
    return np.asarray(img, dtype=np.float64)[y0:y1, x0:x1, :3].reshape(-1, 3).mean(axis=0)


def block_means(img, n):
    a = np.asarray(img, dtype=np.float64)[..., :3]
    h, w, _ = a.shape
    return a.reshape(h // n, n, w // n, n, 3).mean(axis=(1, 3))


def block_mad(a, b, n=20):
    """mean absolute difference of n x n block means, in 8-bit units, over the whole frame"""
    return float(np.abs(block_means(a, n) - block_means(b, n)).mean())


def room_edges(img):
    """first / last lit column on rows 280-320 and the columns where the red / blue walls end (600 x 600 display image): the
    silhouette quantities make_fixtures.py fitted eye.z to"""
    a = np.asarray(img)[..., :3].astype(int)
    s = a.sum(axis=2)
    row = s[280:320].mean(axis=0)
    xs = np.where(row > 12)[0]
    r, g, b = a[280:320, :, 0].mean(axis=0), a[280:320, :, 1].mean(axis=0), a[280:320, :, 2].mean(axis=0)
    red = np.where((r > 40) & (g < 0.3 * r) & (b < 0.3 * r))[0]
    blue = np.where((b > 40) & (g < 0.3 * b) & (r < 0.3 * b))[0]
    return [int(xs.min()), int(xs.max()), int(red.max()), int(blue.min())]


def luminance_correlation(a, b, box):
    """normalised cross-correlation of the two images' luminance over a display-space box"""
    x0, y0, x1, y1 = box
    la = np.asarray(a, dtype=np.float64)[y0:y1, x0:x1, :3].mean(axis=2)
    lb = np.asarray(b, dtype=np.float64)[y0:y1, x0:x1, :3].mean(axis=2)
    la -= la.mean()
    lb -= lb.mean()
    return float((la * lb).sum() / np.sqrt((la * la).sum() * (lb * lb).sum()))


def camera_uvw(oracle, eye, lookat, up, fov, aspect):
    """sutil::Camera::UVWFrame (sutil/Camera.cpp:34-45) through the oracle's pinned restatement"""
    e, l, u = oracle.f32(eye), oracle.f32(lookat), oracle.f32(up)
    U, V, W = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    oracle.lib().oracle_camera_uvw(oracle.fptr(e), oracle.fptr(l), oracle.fptr(u), float(fov), float(aspect), oracle.fptr(U), oracle.fptr(V), oracle.fptr(W))
    return e, U, V, W


def oracle_render(oracle, name, size, spp_sqrt, frames, white=None, first_frame=0):
    """the CPU oracle under picture `name`'s scene / mode / camera; `white`: cornellWhite's kd replaced (reference.jpg's revision);
    frames first_frame .. first_frame + frames - 1 averaged (independent seeds per frame: tea<16>(pixel, frameCount), kernel.cu:203-204).
    Returns the float accumulation buffer [size, size, 3], bottom row first."""
    meta = fixtures()[0]["images"][name]
    sc = oracle.scene(meta["scene"], size, size)
    e, U, V, W = camera_uvw(oracle, meta["eye"], meta["lookat"], meta["up"], meta["fov"], 1.0)
    sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = e.tolist(), U.tolist(), V.tolist(), W.tolist()
    if white is not None:
        for i in range(sc.n_prims):
            p = sc.prims[i]
            if abs(p.kd[0] - 0.8) < 1e-6 and abs(p.kd[1] - 0.8) < 1e-6 and abs(p.kd[2] - 0.8) < 1e-6:
                p.kd[:] = [white] * 3
    total = np.zeros((size, size, 3), dtype=np.float64)
    for f in range(first_frame, first_frame + frames):
        acc, _, _ = oracle.render(sc, oracle.frame(size, size, sqrt_spp=spp_sqrt, frame_count=f, path=meta["mode"] == "path"), None)
        total += acc[..., :3].astype(np.float64) * (f + 1)    # (a lone frame f is lerped against an empty buffer: cur / (f + 1), kernel.cu:239-245)
    return (total / frames).astype(np.float32)


def gpu_render(capi, hscene, oracle, name, size, spp_sqrt, frames, white=None):
    """the HIP path through the C ABI under the same conditions; scene tables from the PRODUCT host (librtgo_host.so).
    Progressive frames 0 .. frames-1 accumulate on the device like the reference's (kernel.cu:239-245).  Returns (accum, stats)."""
    meta = fixtures()[0]["images"][name]
    t = hscene.tables(meta["scene"], size, size)
    mat = np.array(t["mat"], dtype=np.float32)
    if white is not None:
        sel = (np.abs(mat[:, 0:3] - 0.8) < 1e-6).all(axis=1)
        mat[sel, 0:3] = white
    e, U, V, W = camera_uvw(oracle, meta["eye"], meta["lookat"], meta["up"], meta["fov"], 1.0)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], mat, t["aabb"])
    ctx.set_camera(e, U, V, W)
    ctx.set_background(t["bg"])
    ctx.set_lights(t["lights"])
    ctx.resize(size * size)
    ctx.reset_stats()
    for f in range(frames):
        ctx.launch(capi.make_frame(size, size, spp_sqrt, f, meta["mode"] == "path"))
    ctx.sync()
    acc = ctx.read_accum(size, size)[..., :3].copy()
    st = ctx.stats()
    ctx.close()
    return acc, st
