"""The oracle's whitted path (oracle/rtgo_oracle_whitted.c): tea<4> against the reference's own cuda/random.h, known answers of
the triangle test, and properties of whitted.cu's pipeline the restatement must show (CPU only)."""
import ctypes as C
import json
import os

import numpy as np

import whitted_scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tea4_is_the_references(oracle):
    with open(os.path.join(GOLD, "ref_blocks.json")) as f:
        ref = json.load(f)
    assert len(ref["tea4"]) >= 30
    for c in ref["tea4"]:
        assert oracle.lib().oracle_tea4(c["in"][0], c["in"][1]) == c["out"]


def _isect(oracle, p0, p1, p2, o, d, tmin=0.0, tmax=1e16):
    t, u, v = [np.zeros(1, np.float32) for _ in range(3)]
    a = [oracle.f32(x) for x in (p0, p1, p2, o, d)]
    hit = oracle.lib().oracle_tri_intersect(*[oracle.fptr(x) for x in a], tmin, tmax, oracle.fptr(t), oracle.fptr(u), oracle.fptr(v))
    return bool(hit), float(t[0]), float(u[0]), float(v[0])


def test_triangle_known_answers(oracle):
    p0, p1, p2 = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    assert _isect(oracle, p0, p1, p2, (0.25, 0.25, 1), (0, 0, -1)) == (True, 1.0, 0.25, 0.25)
    assert _isect(oracle, p0, p1, p2, (0.25, 0.25, -1), (0, 0, 1))[0]            # two-sided
    assert not _isect(oracle, p0, p1, p2, (0.75, 0.75, 1), (0, 0, -1))[0]        # u + v > 1
    assert not _isect(oracle, p0, p1, p2, (-0.1, 0.2, 1), (0, 0, -1))[0]         # u < 0
    assert not _isect(oracle, p0, p1, p2, (0.25, 0.25, 1), (1, 0, 0))[0]         # parallel: det == 0
    assert not _isect(oracle, p0, p1, p2, (0.25, 0.25, 1), (0, 0, -1), tmin=1.0)[0]   # t must exceed tmin strictly
    assert not _isect(oracle, p0, p1, p2, (0.25, 0.25, 1), (0, 0, -1), tmax=1.0)[0]   # ... and stay below tmax
    assert _isect(oracle, p0, p1, p2, (0, 0, 1), (0, 0, -1)) == (True, 1.0, 0.0, 0.0)   # a vertex belongs to the triangle


def test_pipeline_properties(oracle):
    W, H = 64, 40
    mesh = whitted_scene.build()
    cam = whitted_scene.camera(oracle, W, H)
    acc0, img0, c0 = oracle.whitted_render(mesh, cam, W, H, 1)
    assert np.isfinite(acc0).all() and (acc0[..., 3] == 1.0).all() and (img0[..., 3] == 255).all()
    # every pixel traces one primary ray; a hit adds at most one occlusion ray per light
    assert c0["rays_total"] == W * H + c0["rays_occlusion"] and 0 < c0["rays_occlusion"] <= 2 * W * H
    # the top rows see only the miss colour (whitted.cu:243-246); the 8-bit image is make_color with gamma 2.2 (:164-173)
    assert np.array_equal(acc0[-1, :, :3], np.tile(mesh["miss"], (W, 1)))
    exp = (np.power(np.clip(acc0[..., :3], 0, 1), np.float32(1.0 / np.float32(2.2))) * np.float32(255)).astype(np.uint8)
    assert np.abs(exp.astype(int) - img0[..., :3].astype(int)).max() <= 1
    # subframe 0 has no jitter (:198-200): a second render is identical; accumulating 4 subframes is a running mean of jittered frames
    acc0b, _, _ = oracle.whitted_render(mesh, cam, W, H, 1)
    assert np.array_equal(acc0, acc0b)
    acc4, _, c4 = oracle.whitted_render(mesh, cam, W, H, 4)
    assert not np.array_equal(acc4, acc0) and np.abs(acc4[..., :3].mean() - acc0[..., :3].mean()) < 0.02
    # shading: the lit ground is brighter than its shadowed part; without lights every hit is black
    dark = dict(mesh, lights=np.zeros((0, 8), np.float32))
    accd, _, cd = oracle.whitted_render(dark, cam, W, H, 1)
    assert cd["rays_occlusion"] == 0 and accd[:H // 3, :, :3].max() == 0.0
    # without vertex normals the sphere is faceted: N = Ng (LocalGeometry.h:113-116)
    flat = dict(mesh, normals=None)
    accf, _, _ = oracle.whitted_render(flat, cam, W, H, 1)
    assert not np.array_equal(accf, acc0)


def test_committed_whitted_render_reproduces_bitwise(oracle):
    """tests/golden/oracle_whitted.npz (oracle/gen_golden.py): the restatement is deterministic, threads or not"""
    z = np.load(os.path.join(GOLD, "oracle_whitted.npz"))
    W, H = 64, 40
    mesh = whitted_scene.build()
    cam = whitted_scene.camera(oracle, W, H)
    assert np.array_equal(cam.view(np.uint32), z["cam"].view(np.uint32))
    acc, img, c = oracle.whitted_render(mesh, cam, W, H, 2)
    assert np.array_equal(acc.view(np.uint32), z["accum"].view(np.uint32)) and np.array_equal(img, z["image"])
    assert [c["rays_total"], c["rays_occlusion"]] == z["rays"].tolist()
