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


def test_tex2d_known_answers(oracle):
    """tex2D<float4> as the CUDA programming guide states linear filtering, on the texture objects of sutil::Scene::addSampler
    (normalised coordinates and reads, wrap addressing -- whatever the glTF sampler says: Scene.cpp:517-524 compares CUDA enums with GL
    constants): texel centres return the texel, halfway between two the mean, (0, 0) the mean of the four corners of a 2 x 2 image
    (wrap), the result is periodic, and the weights have 8 fractional bits"""
    t = np.zeros((2, 2, 4), np.uint8)
    t[0, 0], t[0, 1], t[1, 0], t[1, 1] = [255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [255, 255, 255, 255]
    assert oracle.tex2d(t, 0.25, 0.25).tolist() == [1.0, 0.0, 0.0, 1.0]
    assert oracle.tex2d(t, 0.75, 0.25).tolist() == [0.0, 1.0, 0.0, 1.0]
    assert oracle.tex2d(t, 0.25, 0.75).tolist() == [0.0, 0.0, 1.0, 1.0]
    assert oracle.tex2d(t, 0.5, 0.25).tolist() == [0.5, 0.5, 0.0, 1.0]
    assert oracle.tex2d(t, 0.0, 0.0).tolist() == [0.5, 0.5, 0.5, 1.0]
    for du, dv in ((1.0, 0.0), (-2.0, 3.0)):
        assert np.array_equal(oracle.tex2d(t, 0.3 + du, 0.6 + dv), oracle.tex2d(t, 0.3, 0.6))
    # 1.8 fixed point: moving u by less than half a weight step changes nothing; the weight moves in steps of 1/256
    g = np.zeros((1, 2, 4), np.uint8)
    g[0, 1] = 255
    base = float(oracle.tex2d(g, 0.25 + 0.5 * 100 / 256, 0.5)[0])
    assert base == 100 / 256
    assert float(oracle.tex2d(g, 0.25 + 0.5 * 100.4 / 256, 0.5)[0]) == base
    assert float(oracle.tex2d(g, 0.25 + 0.5 * 101 / 256, 0.5)[0]) == 101 / 256


def test_waterbottle_fixture_and_its_render(oracle):
    """the reference's triangle asset through the oracle: the fixture is what the glTF says (2549 vertices, 4510 triangles, unit normals,
    texture coordinates in [0, 1]), textures change the picture, and the normal map changes it again"""
    mesh = whitted_scene.waterbottle()
    assert mesh["positions"].shape == (2549, 3) and mesh["indices"].shape == (4510, 3) and mesh["indices"].max() == 2548
    assert np.allclose(np.linalg.norm(mesh["normals"], axis=1), 1.0, atol=1e-3)
    assert mesh["texcoords"].min() >= 0.0 and mesh["texcoords"].max() <= 1.0
    assert abs(mesh["positions"][:, 1]).max() < 0.14 and mesh["textures"][0][0].shape == (256, 256, 4)
    W, H = 64, 48
    cam = whitted_scene.camera(oracle, W, H, eye=(0.12, 0.08, 0.42), lookat=(0.0, 0.0, 0.0), fov=40.0)
    full, _, rc = oracle.whitted_render(mesh, cam, W, H, 1)
    bc, mr, nm = mesh["textures"][0]
    plain, _, _ = oracle.whitted_render(dict(mesh, textures=None), cam, W, H, 1)
    no_nm, _, _ = oracle.whitted_render(dict(mesh, textures={0: (bc, mr, None)}), cam, W, H, 1)
    on = (full[..., :3] != np.float32(mesh["miss"])).any(axis=-1)
    assert 0.1 < on.mean() < 0.9 and rc["rays_total"] > W * H
    assert np.abs(full[on] - plain[on]).mean() > 1e-2 and np.abs(full[on] - no_nm[on]).mean() > 1e-4
    # without texture coordinates UV = the barycentrics (LocalGeometry.h:97-102): another picture, still finite
    bary, _, _ = oracle.whitted_render(dict(mesh, texcoords=None), cam, W, H, 1)
    assert np.isfinite(bary).all() and np.abs(bary[on] - full[on]).mean() > 1e-3
