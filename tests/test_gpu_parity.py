"""GPU parity: the HIP megakernel, called through the C ABI (include/rtgo.h), against the CPU oracle on the same
seeded inputs.  Needs a real MI355X: run with `pytest -m gpu` on the GPU box.  Scene tables fed to the ABI here come
from the oracle so that this file isolates the kernel; the C++ host's tables are checked in test_host_scene.py."""
import json
import os

import numpy as np
import pytest

from parity import assert_parity, compare

pytestmark = pytest.mark.gpu

ALL_SCENES = ["cornell", "slide", "mirror_spheres", "plateau", "window", "checkered", "balls", "soft_mirrors"]


@pytest.fixture(scope="module")
def capi():
    from raytracingo_amd import capi as m
    m.load()
    return m


def upload(capi, oracle, name, W, H, with_aabbs=True):
    sc = oracle.scene(name, W, H)
    t = oracle.scene_tables(sc)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"] if with_aabbs else None)
    ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"])
    ctx.set_lights(t["lights"])
    return sc, t, ctx


def gpu_render(capi, ctx, W, H, n, frame, path, ambient=False, window=None, bands=(4, 1, 0), stats=False, prev=None):
    x0, y0, w, h = window if window else (0, 0, W, H)
    rows = capi.local_rows(h, bands[0], bands[1], bands[2])
    if ctx.pixels < rows * w:
        ctx.resize(max(rows * w, 1))
    if prev is not None:
        ctx.write_accum(prev)
    ctx.launch(capi.make_frame(W, H, n, frame, path, ambient, window, bands, stats=stats))
    ctx.sync()
    return ctx.read_accum(rows, w), ctx.read_image(rows, w)


def test_device_lbvh_and_inverses_bitwise(capi, oracle):
    """what rtgo_set_scene builds on the device == the oracle's canonical LBVH and hoisted inverses, bit for bit"""
    import ctypes as C
    for name in ALL_SCENES:
        for with_aabbs in (True, False):
            sc, t, ctx = upload(capi, oracle, name, 64, 64, with_aabbs)
            boxes, links, inv, aabb = ctx.read_bvh()
            oboxes, olinks, _, _ = oracle.lbvh(t["aabb"])
            assert np.array_equal(aabb.view(np.uint32), t["aabb"].view(np.uint32)), name
            assert np.array_equal(links, olinks), name
            assert np.array_equal(boxes.view(np.uint32), oboxes.view(np.uint32)), name
            oinv = np.zeros((sc.n_prims, 16), dtype=np.float32)
            for i in range(sc.n_prims):
                m = np.ascontiguousarray(t["M"][i])
                oracle.lib().oracle_mat_inverse(oracle.fptr(m), oracle.fptr(oinv[i]))
            assert np.array_equal(inv.view(np.uint32), oinv[:, :12].view(np.uint32)), name
            ctx.close()


def test_c1_cornell_256_distributed(capi, oracle):
    """BASELINE config 1: cornell 256x256 --mode=distributed --sample=1, frame 0"""
    sc, t, ctx = upload(capi, oracle, "cornell", 256, 256)
    acc, img = gpu_render(capi, ctx, 256, 256, 1, 0, path=False, stats=True)
    racc, rimg, rc = oracle.render(sc, oracle.frame(256, 256, 1, 0, path=False, mode=1))
    m = assert_parity(acc, racc, img, rimg, what="C1")
    st = ctx.stats()
    assert abs(st["rays_total"] - rc["rays_total"]) <= 0.002 * rc["rays_total"]
    assert abs(st["rays_occlusion"] - rc["rays_occlusion"]) <= 0.002 * rc["rays_occlusion"]
    print("C1", m, st, rc)


@pytest.mark.parametrize("name", ["cornell", "balls", "mirror_spheres", "plateau"])
def test_path_n4_progressive(capi, oracle, name):
    """BASELINE scenes, path mode, N=4 (16 spp), frames 0..3 with running-average accumulation (kernel.cu:239-245)"""
    W, H = 128, 72
    sc, t, ctx = upload(capi, oracle, name, W, H)
    racc = None
    acc = None
    for f in range(4):
        racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, 4, f, path=True, mode=1), accum_prev=racc)
        acc, img = gpu_render(capi, ctx, W, H, 4, f, path=True)
        m = assert_parity(acc, racc, img, rimg, what="%s frame %d" % (name, f))
    print(name, m)


@pytest.mark.parametrize("name", ALL_SCENES)
@pytest.mark.parametrize("mode", ["path", "distributed", "distributed_ambient"])
def test_all_scenes_small(capi, oracle, name, mode):
    W, H, n = 96, 64, 2
    sc, t, ctx = upload(capi, oracle, name, W, H)
    path = mode == "path"
    amb = mode == "distributed_ambient"
    ctx.reset_stats()
    acc, img = gpu_render(capi, ctx, W, H, n, 0, path, amb, stats=True)
    racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, ambient=amb, mode=1))
    m = assert_parity(acc, racc, img, rimg, what="%s %s" % (name, mode))
    st = ctx.stats()
    for k, tol in (("rays_total", 0.01), ("node_visits", 0.02), ("prim_tests", 0.02), ("hits", 0.01)):
        assert abs(st[k] - rc[k]) <= tol * max(rc[k], 1), (k, st[k], rc[k])
    # the timed kernel (fast walk over the collapsed LBVH) must give the canonical walk's pixels bit for bit
    ctx.reset_stats()
    acc_f, img_f = gpu_render(capi, ctx, W, H, n, 0, path, amb, stats=False)
    assert np.array_equal(acc_f.view(np.uint32), acc.view(np.uint32)), "fast walk != canonical walk"
    assert np.array_equal(img_f, img)
    assert ctx.stats()["rays_total"] == st["rays_total"]
    print(name, mode, m, {k: (st[k], rc[k]) for k in ("rays_total", "node_visits", "prim_tests", "hits")})


def test_window_and_bands_are_bitwise_the_full_frame(capi, oracle):
    """pixels are independent (seed = tea<16>(W*y+x, frame)): any window / band split reproduces the full launch bitwise"""
    W, H, n = 160, 90, 2
    sc, t, ctx = upload(capi, oracle, "cornell", W, H)
    full, fimg = gpu_render(capi, ctx, W, H, n, 0, True)
    win = (48, 20, 70, 37)
    part, _ = gpu_render(capi, ctx, W, H, n, 0, True, window=win)
    assert np.array_equal(part.view(np.uint32), full[20:57, 48:118].view(np.uint32))
    for G in (2, 3, 8):
        out = np.zeros_like(full)
        for g in range(G):
            a, _ = gpu_render(capi, ctx, W, H, n, 0, True, bands=(4, G, g))
            rows = [r for r in range(H) if (r // 4) % G == g]
            assert a.shape[0] == len(rows)
            out[rows] = a
        assert np.array_equal(out.view(np.uint32), full.view(np.uint32)), G
    # and the oracle agrees on the same split
    ra, _, _ = oracle.render(sc, oracle.frame(W, H, n, 0, path=True, bands=(4, 3, 1), mode=1))
    a, _ = gpu_render(capi, ctx, W, H, n, 0, True, bands=(4, 3, 1))
    assert_parity(a, ra, what="band 1/3")


def test_full_size_properties(capi, oracle):
    """BASELINE config 2 size (1920x1080, N=4): size-independent properties + a cropped oracle comparison"""
    W, H, n = 1920, 1080, 4
    sc, t, ctx = upload(capi, oracle, "cornell", W, H)
    ctx.reset_stats()
    acc, img = gpu_render(capi, ctx, W, H, n, 0, True)
    st = ctx.stats()
    assert np.isfinite(acc).all() and (acc[..., 3] == 1.0).all() and (img[..., 3] == 255).all()
    # the 8-bit image is make_color(accum) exactly (kernel.cu:90-98, 246)
    exp = (np.clip(acc[..., :3], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(exp, img[..., :3])
    # every pixel traces N*N primary rays; at most 6 radiance rays per sample
    assert 16 * W * H <= st["rays_total"] <= 6 * 16 * W * H
    # determinism: a second launch of frame 0 is bitwise identical
    acc2, _ = gpu_render(capi, ctx, W, H, n, 0, True)
    assert np.array_equal(acc.view(np.uint32), acc2.view(np.uint32))
    # crop vs oracle
    win = (800, 400, 320, 180)
    racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 0, path=True, window=win, mode=1))
    m = assert_parity(acc[400:580, 800:1120], racc, img[400:580, 800:1120], rimg, what="1080p crop")
    print("1080p", m, st)


def test_error_behaviour(capi, oracle):
    ctx = capi.Context(0)
    with pytest.raises(capi.RtgoError):
        ctx.launch(capi.make_frame(64, 64))  # no scene yet
    sc, t, ctx2 = upload(capi, oracle, "cornell", 64, 64)
    with pytest.raises(capi.RtgoError):
        ctx2.launch(capi.make_frame(64, 64))  # no output yet
    ctx2.resize(64 * 64)
    with pytest.raises(capi.RtgoError):
        ctx2.launch(capi.make_frame(64, 64, max_depth=9))
    with pytest.raises(capi.RtgoError):
        ctx2.launch(capi.make_frame(64, 64, window=(10, 10, 64, 64)))
    with pytest.raises(capi.RtgoError):
        ctx.set_scene(np.zeros(0, np.int32), np.zeros((0, 16), np.float32), np.zeros((0, 10), np.float32))
    # a singular or non-finite model matrix has no object space to intersect in: refused, not rendered
    ident = np.eye(4, dtype=np.float32).reshape(1, 16)
    mat = np.array([[0.5, 0.5, 0.5, 0, 0, 0, 1, 0, 0, 0]], np.float32)
    flat = ident.copy(); flat[0, 5] = 0.0
    nan = ident.copy(); nan[0, 3] = np.nan
    for M in (flat, nan):
        with pytest.raises(capi.RtgoError):
            ctx.set_scene(np.array([2], np.int32), M, mat)
    with pytest.raises(capi.RtgoError):
        ctx.set_scene(np.full(513, 3, np.int32), np.repeat(ident, 513, 0), np.repeat(mat, 513, 0))


def test_config3_balls_1080p_properties(capi, oracle, monkeypatch):
    """BASELINE config 3 size (balls 1920x1080 path N=4): determinism + crop vs oracle at full scale, on each of the structures the
    launch-time trial chooses from (the 36 % tree and the uniform grid that keeps the job: 10.2 ms against 13.5) -- one frame, bit for bit"""
    W, H, n = 1920, 1080, 4
    sc, t, ctx = upload(capi, oracle, "balls", W, H)
    frames = {}
    for pin in ("0", "2"):
        monkeypatch.setenv("RTGO_TREE", pin)
        ctx.reset_stats()
        acc, img = gpu_render(capi, ctx, W, H, n, 0, True)
        st = ctx.stats()
        assert bool(st["last_variant"] & 16) == (pin == "2") and st["launches_canonical"] == 0, st
        assert np.isfinite(acc).all() and 16 * W * H <= st["rays_total"] <= 6 * 16 * W * H
        assert np.array_equal((np.clip(acc[..., :3], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8), img[..., :3])
        frames[pin] = (acc.copy(), img.copy(), st["rays_total"])
    monkeypatch.delenv("RTGO_TREE", raising=False)
    assert np.array_equal(frames["0"][0].view(np.uint32), frames["2"][0].view(np.uint32)) and frames["0"][2] == frames["2"][2]
    acc, img = frames["2"][0], frames["2"][1]
    win = (900, 500, 160, 90)
    racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 0, path=True, window=win, mode=1))
    assert_parity(acc[500:590, 900:1060], racc, img[500:590, 900:1060], rimg, what="balls 1080p crop")


def _full_frame_properties(acc, img, st, W, H, n):
    """size-independent properties of a whole frame (kernel.cu:236-246): finite, alpha 1 / 255, image == make_color(accum),
    ray-count bounds (N*N primary rays per pixel, at most 6 radiance rays per sample in path mode)"""
    assert np.isfinite(acc).all() and (acc[..., 3] == 1.0).all() and (img[..., 3] == 255).all()
    exp = (np.clip(acc[..., :3], 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    assert np.array_equal(exp, img[..., :3])
    assert n * n * W * H <= st["rays_total"] <= 6 * n * n * W * H


def test_config4_mirror_spheres_4k_spp64_bands_of_8(capi, oracle):
    """BASELINE config 4 at its real size and sample count: mirror_spheres 3840x2160 --sample=8 (64 spp = 4 passes of 16 samples
    per pixel, kernel.cu:206-246), framebuffer tiled over 8 ranks.  Whole frame: properties + determinism; a 128x72 crop
    against the oracle; the 8 band launches reassemble to the whole-image launch bit for bit and trace exactly its rays."""
    W, H, n = 3840, 2160, 8
    sc, t, ctx = upload(capi, oracle, "mirror_spheres", W, H)
    ctx.reset_stats()
    full, fimg = gpu_render(capi, ctx, W, H, n, 0, True)
    st = ctx.stats()
    _full_frame_properties(full, fimg, st, W, H, n)
    again, _ = gpu_render(capi, ctx, W, H, n, 0, True)
    assert np.array_equal(again.view(np.uint32), full.view(np.uint32))           # determinism
    win = (1850, 1040, 128, 72)                                                  # the two spheres and the floor between them
    racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 0, path=True, window=win, mode=1))
    x0, y0, w, h = win
    m = assert_parity(full[y0:y0 + h, x0:x0 + w], racc, fimg[y0:y0 + h, x0:x0 + w], rimg, what="C4 mirror_spheres 4K spp 64 crop")
    out, oimg = np.zeros_like(full), np.zeros_like(fimg)
    ctx.reset_stats()
    for g in range(8):
        a, i8 = gpu_render(capi, ctx, W, H, n, 0, True, bands=(4, 8, g))
        rows = [r for r in range(H) if (r // 4) % 8 == g]
        out[rows], oimg[rows] = a, i8
    assert np.array_equal(out.view(np.uint32), full.view(np.uint32))
    assert np.array_equal(oimg, fimg)
    assert ctx.stats()["rays_total"] == st["rays_total"]          # the 8 bands trace exactly the rays of the whole frame
    print("C4", m, st["rays_total"], st["last_launch_ms"])
    ctx.close()


def test_config5_plateau_4k_spp256_progressive_bands_of_8(capi, oracle):
    """BASELINE config 5 at its real size and sample count: plateau 3840x2160 --sample=16 (256 spp = 16 passes), progressive
    frames 0 and 1 (running average, kernel.cu:239-245), tiled over 8 ranks whose bands keep their own accumulation rows."""
    W, H, n = 3840, 2160, 16
    sc, t, ctx = upload(capi, oracle, "plateau", W, H)
    ctx.reset_stats()
    f0, i0 = gpu_render(capi, ctx, W, H, n, 0, True)
    st0 = ctx.stats()
    _full_frame_properties(f0, i0, st0, W, H, n)
    f0 = f0.copy()
    ctx.reset_stats()
    f1, i1 = gpu_render(capi, ctx, W, H, n, 1, True)          # accumulates onto frame 0 in place
    st1 = ctx.stats()
    _full_frame_properties(f1, i1, st1, W, H, n)
    win = (1800, 820, 128, 72)                                # cylinders, a sphere, the plate's rim
    x0, y0, w, h = win
    r0, ri0, _ = oracle.render(sc, oracle.frame(W, H, n, 0, path=True, window=win, mode=1))
    assert_parity(f0[y0:y0 + h, x0:x0 + w], r0, i0[y0:y0 + h, x0:x0 + w], ri0, what="C5 plateau 4K spp 256 frame 0 crop")
    r1, ri1, _ = oracle.render(sc, oracle.frame(W, H, n, 1, path=True, window=win, mode=1), accum_prev=r0)
    m = assert_parity(f1[y0:y0 + h, x0:x0 + w], r1, i1[y0:y0 + h, x0:x0 + w], ri1, what="C5 plateau 4K spp 256 frames 0-1 crop")
    # 8 ranks, two progressive frames each on its own compact band buffer
    out0, out1, oimg1 = np.zeros_like(f0), np.zeros_like(f1), np.zeros_like(i1)
    rays = 0
    for g in range(8):
        rows = [r for r in range(H) if (r // 4) % 8 == g]
        ctx.reset_stats()
        a0, _ = gpu_render(capi, ctx, W, H, n, 0, True, bands=(4, 8, g))
        out0[rows] = a0
        a1, b1 = gpu_render(capi, ctx, W, H, n, 1, True, bands=(4, 8, g))     # onto the band's own frame 0, resident in its buffer
        out1[rows], oimg1[rows] = a1, b1
        rays += ctx.stats()["rays_total"]
    assert np.array_equal(out0.view(np.uint32), f0.view(np.uint32))
    assert np.array_equal(out1.view(np.uint32), f1.view(np.uint32))
    assert np.array_equal(oimg1, i1)
    assert rays == st0["rays_total"] + st1["rays_total"]
    print("C5", m, st0["rays_total"], st1["rays_total"], st1["last_launch_ms"])
    ctx.close()


def test_config5_plateau_progressive_deep(capi, oracle):
    """the shape of config 5 (plateau, N=16 = 256 spp, progressive frames) as a WHOLE small image: 3 accumulated frames"""
    W, H, n = 96, 54, 16
    sc, t, ctx = upload(capi, oracle, "plateau", W, H)
    racc = None
    for f in range(3):
        racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, n, f, path=True, mode=1), accum_prev=racc)
        acc, img = gpu_render(capi, ctx, W, H, n, f, True)
    assert_parity(acc, racc, img, rimg, what="plateau N=16 frames 0-2")


def test_host_renderer_end_to_end(capi, oracle):
    """the C++ engine::host::Renderer (headless Display loop) over the ABI == the oracle, and == driving the ABI from Python"""
    from raytracingo_amd import scene as hscene
    W, H = 96, 64
    acc, img, st = hscene.host_render("cornell", "path", W, H, sample=2, frames=3)
    sc = oracle.scene("cornell", W, H)
    racc = None
    for f in range(3):
        racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, 2, f, path=True, mode=1), accum_prev=racc)
    assert_parity(acc, racc, img, rimg, what="host Renderer")
    assert st["launches"] == 3
    acc2, img2, _ = hscene.host_render("window", "distributed", W, H, sample=1, ambient=True, frames=1)
    r2, i2, _ = oracle.render(oracle.scene("window", W, H), oracle.frame(W, H, 1, 0, path=False, ambient=True, mode=1))
    assert_parity(acc2, r2, img2, i2, what="host Renderer distributed+ambient")


def test_golden_oracle_renders(capi, oracle):
    """committed fixtures (tests/golden/oracle_renders.npz): the GPU against stored oracle output, no oracle run needed"""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_renders.npz"))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_renders.json")) as f:
        meta = json.load(f)["cases"]
    for i, m in enumerate(meta):
        sc, t, ctx = upload(capi, oracle, m["name"], m["W"], m["H"])
        for fr in range(m["frames"]):
            acc, img = gpu_render(capi, ctx, m["W"], m["H"], m["N"], fr, m["path"], m["ambient"])
        assert_parity(acc, z["accum_%d" % i], img, z["image_%d" % i], what=str(m))
        ctx.close()


def _random_scene(oracle, seed, W, H):
    """a random scene through the plugin surface: all four primitive types under random translate*rotate*scale (non-uniform),
    diffuse / glossy / mirror / emissive materials, one to three rectangular lights, a non-black background"""
    rng = np.random.RandomState(seed)
    L = oracle.lib()

    def trs():
        T, R, S, TR, out = [np.zeros(16, dtype=np.float32) for _ in range(5)]
        L.oracle_mat_translate(*[np.float32(v) for v in rng.uniform(-3.5, 3.5, 3)], oracle.fptr(T))
        ax = rng.uniform(-1, 1, 3)
        ax = ax / np.linalg.norm(ax)
        L.oracle_mat_rotate(np.float32(rng.uniform(-3.1, 3.1)), *[np.float32(v) for v in ax], oracle.fptr(R))
        L.oracle_mat_scale(*[np.float32(v) for v in rng.uniform(0.3, 2.2, 3)], oracle.fptr(S))
        L.oracle_mat_mul(oracle.fptr(T), oracle.fptr(R), oracle.fptr(TR))
        L.oracle_mat_mul(oracle.fptr(TR), oracle.fptr(S), oracle.fptr(out))
        return out

    n = int(rng.randint(12, 70))
    types = rng.randint(0, 4, n)
    M = np.stack([trs() for _ in range(n)])
    mat = np.zeros((n, 10), dtype=np.float32)
    mat[:, 0:3] = rng.uniform(0.1, 1.0, (n, 3))
    mat[:, 3:6] = rng.uniform(0.0, 0.8, (n, 1))
    mat[:, 6] = rng.choice([0.0, 1.0, 100.0, 10000.0], n)
    # a big floor and a big emissive ceiling light so that most paths end somewhere
    floor, ceil = np.zeros(16, dtype=np.float32), np.zeros(16, dtype=np.float32)
    S = np.zeros(16, dtype=np.float32)
    T = np.zeros(16, dtype=np.float32)
    R = np.zeros(16, dtype=np.float32)
    L.oracle_mat_translate(0.0, -4.0, 0.0, oracle.fptr(T))
    L.oracle_mat_scale(14.0, 1.0, 14.0, oracle.fptr(S))
    L.oracle_mat_mul(oracle.fptr(T), oracle.fptr(S), oracle.fptr(floor))
    L.oracle_mat_translate(0.0, 4.5, 0.0, oracle.fptr(T))
    L.oracle_mat_rotate(np.float32(np.pi), 1.0, 0.0, 0.0, oracle.fptr(R))
    TR = np.zeros(16, dtype=np.float32)
    L.oracle_mat_mul(oracle.fptr(T), oracle.fptr(R), oracle.fptr(TR))
    L.oracle_mat_scale(6.0, 1.0, 6.0, oracle.fptr(S))
    L.oracle_mat_mul(oracle.fptr(TR), oracle.fptr(S), oracle.fptr(ceil))
    types[0], M[0], mat[0] = 2, floor, [0.7, 0.7, 0.7, 0.3, 0.3, 0.3, 1.0, 0, 0, 0]
    types[1], M[1], mat[1] = 2, ceil, [0, 0, 0, 0, 0, 0, 1.0, 12.0, 12.0, 12.0]
    lights = [oracle.light_from_matrix(ceil, falloff=0.02)]
    for k in range(int(rng.randint(0, 3))):
        idx = 2 + k
        types[idx] = 2
        mat[idx] = [0, 0, 0, 0, 0, 0, 1.0, 9.0, 9.0, 9.0]
        lights.append(oracle.light_from_matrix(M[idx], falloff=float(rng.uniform(0, 0.1))))
    cam = oracle.scene_tables(oracle.scene("cornell", W, H))["cam"]
    bg = (0.05, 0.07, 0.1)
    sc = oracle.scene_from_tables(types, M, mat, np.stack(lights), cam, bg)
    return sc, oracle.scene_tables(sc)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_scenes_through_the_plugin_surface(capi, oracle, seed):
    W, H, n = 80, 60, 2
    sc, t = _random_scene(oracle, seed, W, H)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], None)   # AABBs derived on the device (CubeBox rule)
    ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"])
    ctx.set_lights(t["lights"])
    boxes, links, inv, aabb = ctx.read_bvh()
    assert np.array_equal(aabb.view(np.uint32), t["aabb"].view(np.uint32))
    for path, amb in ((True, False), (False, False), (False, True)):
        ctx.reset_stats()
        acc, img = gpu_render(capi, ctx, W, H, n, 0, path, amb, stats=True)
        racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, ambient=amb, mode=1))
        r0, _, _ = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, ambient=amb, mode=0))
        assert np.array_equal(r0.view(np.uint32), racc.view(np.uint32)), "oracle literal != oracle LBVH"
        assert_parity(acc, racc, img, rimg, what="random scene %d %s" % (seed, (path, amb)))
        fast, fimg = gpu_render(capi, ctx, W, H, n, 0, path, amb, stats=False)
        assert np.array_equal(fast.view(np.uint32), acc.view(np.uint32)), "fast walk != canonical walk"
    ctx.close()


def test_cli_headless_ppm_and_pfm(tmp_path, capi, oracle):
    """the `engine` command line (engine/main.cpp flags + --frames/--out/--out-accum) end to end: P6 file with rows flipped and
    alpha dropped (sutil.cpp:81-101, 377-388), PFM of the accumulation buffer"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracingo_amd", "rtgo_engine")
    W, H = 80, 48
    ppm, pfm = str(tmp_path / "o.ppm"), str(tmp_path / "o.pfm")
    r = subprocess.run([exe, "--scene=mirror_spheres", "--mode=path", "--dim=%dx%d" % (W, H), "--sample=2", "--frames=2",
                        "--out=" + ppm, "--out-accum=" + pfm], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sc = oracle.scene("mirror_spheres", W, H)
    racc = None
    for f in range(2):
        racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, 2, f, path=True, mode=1), accum_prev=racc)
    raw = open(ppm, "rb").read()
    head = b"P6\n%d %d\n255\n" % (W, H)
    assert raw.startswith(head) and len(raw) == len(head) + W * H * 3
    img = np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(H, W, 3)[::-1]     # file is top row first
    rawf = open(pfm, "rb").read()
    headf = b"PF\n%d %d\n-1.0\n" % (W, H)
    assert rawf.startswith(headf)
    acc = np.frombuffer(rawf[len(headf):], dtype="<f4").reshape(H, W, 3)
    acc4 = np.concatenate([acc, np.ones((H, W, 1), np.float32)], axis=2)
    img4 = np.concatenate([img, np.full((H, W, 1), 255, np.uint8)], axis=2)
    assert_parity(acc4, racc, img4, rimg, what="CLI")


def test_interactive_session_camera_and_resize(capi, oracle):
    """the caller side of the boundary without a window: a camera move or a resize restarts the running
    average at frameCount 0 (renderer.cpp:682) with the new raygen record / buffers (renderer.cpp:703-747)"""
    from raytracingo_amd import scene as hscene
    import ctypes as C
    W, H = 96, 64
    s = hscene.Session("cornell", "path", W, H, sample=2)
    for _ in range(3):
        s.frame()
    acc, img, fc = s.read()
    assert fc == 2
    sc = oracle.scene("cornell", W, H)
    racc = None
    for f in range(3):
        racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, 2, f, path=True, mode=1), accum_prev=racc)
    assert_parity(acc, racc, img, rimg, what="session frames 0-2")
    # move the camera: next frame is frame 0 of the new view
    eye, look, up = [3.0, 1.0, 12.0], [0.0, -0.5, 0.0], [0.0, 1.0, 0.0]
    s.move_camera(eye, look, up)
    s.frame()
    acc, img, fc = s.read()
    assert fc == 0
    U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    oracle.lib().oracle_camera_uvw(oracle.fptr(oracle.f32(eye)), oracle.fptr(oracle.f32(look)), oracle.fptr(oracle.f32(up)), 60.0,
                                   np.float32(np.float32(W) / np.float32(H)), oracle.fptr(U), oracle.fptr(V), oracle.fptr(Wv))
    sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = eye, U.tolist(), V.tolist(), Wv.tolist()
    racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, 2, 0, path=True, mode=1))
    assert_parity(acc, racc, img, rimg, what="after camera move")
    s.frame()
    assert s.read()[2] == 1
    # resize: new buffers, new aspect ratio, frame 0 again
    W2, H2 = 128, 48
    s.resize(W2, H2)
    s.frame()
    acc, img, fc = s.read()
    assert fc == 0 and acc.shape == (H2, W2, 4)
    oracle.lib().oracle_camera_uvw(oracle.fptr(oracle.f32(eye)), oracle.fptr(oracle.f32(look)), oracle.fptr(oracle.f32(up)), 60.0,
                                   np.float32(np.float32(W2) / np.float32(H2)), oracle.fptr(U), oracle.fptr(V), oracle.fptr(Wv))
    sc.U[:], sc.V[:], sc.W[:] = U.tolist(), V.tolist(), Wv.tolist()
    racc, rimg, _ = oracle.render(sc, oracle.frame(W2, H2, 2, 0, path=True, mode=1))
    assert_parity(acc, racc, img, rimg, what="after resize")
    s.close()


@pytest.mark.parametrize("max_depth", [0, 1, 3])
def test_shallower_trace_depths(capi, oracle, max_depth):
    """Params::maxTraceDepth below the reference's 5: the depth cut-off of kernel.cu:465 / :506 at every level"""
    W, H = 96, 64
    sc, t, ctx = upload(capi, oracle, "cornell", W, H)
    for path in (True, False):
        rows = H
        if ctx.pixels < W * H:
            ctx.resize(W * H)
        ctx.reset_stats()
        ctx.launch(capi.make_frame(W, H, 2, 0, path, False, None, (4, 1, 0), max_depth=max_depth))
        ctx.sync()
        acc, img = ctx.read_accum(H, W), ctx.read_image(H, W)
        racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, 2, 0, path=path, mode=1, max_depth=max_depth))
        assert_parity(acc, racc, img, rimg, what="max_depth %d path=%s" % (max_depth, path))
        if path and max_depth == 0:
            # no bounce at all: exactly the N*N primary rays of every pixel, and only emitters are visible
            assert rc["rays_total"] == W * H * 4 and ctx.stats()["rays_total"] == W * H * 4
    ctx.close()


def test_camera_inside_the_scene(capi, oracle):
    """eye inside the room (and inside several primitives' AABBs): the scheduling rectangle degenerates to the whole window,
    rays start inside boxes; also a non-square aspect and a tilted up vector"""
    W, H, n = 112, 64, 2
    sc, t, ctx = upload(capi, oracle, "window", W, H)
    eye, look, up = oracle.f32([1.5, -1.0, 3.0]), oracle.f32([-4.0, 0.5, -7.0]), oracle.f32([0.1, 1.0, 0.0])
    U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    oracle.lib().oracle_camera_uvw(oracle.fptr(eye), oracle.fptr(look), oracle.fptr(up), 60.0, np.float32(np.float32(W) / np.float32(H)),
                                   oracle.fptr(U), oracle.fptr(V), oracle.fptr(Wv))
    sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = eye.tolist(), U.tolist(), V.tolist(), Wv.tolist()
    ctx.set_camera(eye, U, V, Wv)
    for path in (True, False):
        acc, img = gpu_render(capi, ctx, W, H, n, 0, path)
        racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, mode=1))
        assert_parity(acc, racc, img, rimg, what="camera inside, path=%s" % path)
        canon, _ = gpu_render(capi, ctx, W, H, n, 0, path, stats=True)
        assert np.array_equal(canon.view(np.uint32), acc.view(np.uint32))
    ctx.close()


def test_odd_sample_counts_and_tiny_images(capi, oracle):
    """N = 3 (9 spp: 7 pixels x 9 samples per wave), N = 5 (25 spp: two passes, the second half empty), 1x1 and 3x2 images"""
    for (W, H, n) in [(40, 30, 3), (33, 17, 5), (1, 1, 4), (3, 2, 1)]:
        sc, t, ctx = upload(capi, oracle, "mirror_spheres", W, H)
        acc, img = gpu_render(capi, ctx, W, H, n, 2, True, prev=np.full((H, W, 4), 0.5, np.float32))
        racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 2, path=True, mode=1), accum_prev=np.full((H, W, 4), 0.5, np.float32))
        assert_parity(acc, racc, img, rimg, what="%dx%d N=%d" % (W, H, n))
        assert ctx.stats()["rays_total"] == rc["rays_total"] or abs(ctx.stats()["rays_total"] - rc["rays_total"]) <= 0.01 * rc["rays_total"]
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("plateau", 16, True), ("cornell", 5, True), ("mirror_spheres", 8, True), ("cornell", 6, False), ("slide", 7, False)])
def test_streaming_and_lock_step_loops_are_bitwise_the_same(capi, oracle, case, monkeypatch):
    """frames of more than 16 spp have two kernels (lanes streaming through their samples / the wave in lock-step, pass by pass):
    both must give the oracle's frame, bit for bit the same as each other -- and so must whatever the launch-time trial between
    them picks (launches 0-3 of a configuration alternate, later ones use the faster)"""
    name, n, path = case
    W, H = 112, 63
    sc, t, ctx = upload(capi, oracle, name, W, H)
    racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 3, path=path, mode=1))
    out = {}
    for force in ("1", "0"):
        monkeypatch.setenv("RTGO_STREAM", force)
        ctx.reset_stats()
        out[force] = gpu_render(capi, ctx, W, H, n, 3, path, prev=np.zeros((H, W, 4), np.float32))
        assert_parity(out[force][0], racc, out[force][1], rimg, what="%s N=%d RTGO_STREAM=%s" % (name, n, force))
    assert np.array_equal(out["1"][0].view(np.uint32), out["0"][0].view(np.uint32)) and np.array_equal(out["1"][1], out["0"][1])
    monkeypatch.delenv("RTGO_STREAM")
    for launch in range(7):
        acc, img = gpu_render(capi, ctx, W, H, n, 3, path, prev=np.zeros((H, W, 4), np.float32))
        assert np.array_equal(acc.view(np.uint32), out["0"][0].view(np.uint32)) and np.array_equal(img, out["0"][1]), launch
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "plateau", "mirror_spheres", "slide"])
def test_background_culling_random_cameras(capi, oracle, name):
    """The timed kernel traces nothing for pixels outside the screen rectangle of the scene's bounds (their primary rays are
    misses by construction) while the instrumented kernel traces every pixel: both must give the same frame bit for bit, and
    the same ray count, wherever the camera stands -- scene small in view, partly off screen, behind the eye, sheared axes --
    and for windows and band shares that cut the rectangle."""
    W, H, n = 144, 80, 2
    sc, t, ctx = upload(capi, oracle, name, W, H)
    bb = np.asarray(t["aabb"], dtype=np.float64).reshape(-1, 6)
    lo, hi = bb[:, :3].min(axis=0), bb[:, 3:].max(axis=0)
    centre, size = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))   # (str hashes change from run to run)
    prev = rng.random((H, W, 4), dtype=np.float32)
    culled = 0
    for trial in range(14):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        eye = oracle.f32(centre + d * size * rng.uniform(0.8, 6.0))
        look = oracle.f32(centre + rng.normal(size=3) * size * rng.choice([0.05, 0.4, 1.5]))
        up = oracle.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
        U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
        oracle.lib().oracle_camera_uvw(oracle.fptr(eye), oracle.fptr(look), oracle.fptr(up), float(rng.uniform(15.0, 100.0)),
                                       np.float32(np.float32(W) / np.float32(H)), oracle.fptr(U), oracle.fptr(V), oracle.fptr(Wv))
        if trial == 12:
            U = (U + np.float32(0.2) * V).astype(np.float32)     # sheared axes: the rectangle cannot be trusted, nothing is culled
        if trial == 13:
            eye = oracle.f32(centre)                              # eye inside the bounds
        ctx.set_camera(eye, U, V, Wv)
        for (window, bands) in [(None, (4, 1, 0)), ((30, 10, 75, 50), (4, 1, 0)), (None, (4, 3, 2))]:
            if trial % 3 and window is not None:
                continue
            x0, y0, w, h = window if window else (0, 0, W, H)
            rows = [r for r in range(h) if (r // bands[0]) % bands[1] == bands[2]]
            pv = np.ascontiguousarray(prev[y0:y0 + h, x0:x0 + w][rows])
            ctx.reset_stats()
            fast, fimg = gpu_render(capi, ctx, W, H, n, 3, True, window=window, bands=bands, prev=pv)
            rays_fast = ctx.stats()["rays_total"]
            ctx.reset_stats()
            canon, cimg = gpu_render(capi, ctx, W, H, n, 3, True, window=window, bands=bands, stats=True, prev=pv)
            st = ctx.stats()
            assert np.array_equal(fast.view(np.uint32), canon.view(np.uint32)), (name, trial, window, bands)
            assert np.array_equal(fimg, cimg)
            assert rays_fast == st["rays_total"], (name, trial, rays_fast, st["rays_total"])
            culled += int(st["hits"] == 0 or st["rays_total"] < 2 * n * n * len(rows) * w)
    ctx.close()
    print(name, "launches dominated by background:", culled)


def _box_scene(oracle, seed, W, H, n_boxes, with_spheres, tilt):
    """a room (five big rectangles + ceiling light) holding random boxes of six rectangles each, optionally spheres between
    them (mixed leaves) and a pair of nearly -- not exactly -- opposite rectangles (both can face a grazing ray)"""
    rng = np.random.RandomState(1000 + seed)
    L = oracle.lib()
    f = oracle.fptr

    def mat(kind, *a):
        out = np.zeros(16, dtype=np.float32)
        getattr(L, "oracle_mat_" + kind)(*[np.float32(v) for v in a], f(out))
        return out

    def mul(*ms):
        acc = ms[0]
        for m in ms[1:]:
            out = np.zeros(16, dtype=np.float32)
            L.oracle_mat_mul(f(acc), f(m), f(out))
            acc = out
        return acc

    hp = np.pi / 2
    faces = [mul(mat("translate", 0, .5, 0)), mul(mat("translate", 0, -.5, 0), mat("rotate", np.pi, 1, 0, 0)),
             mul(mat("translate", .5, 0, 0), mat("rotate", -hp, 0, 0, 1)), mul(mat("translate", -.5, 0, 0), mat("rotate", hp, 0, 0, 1)),
             mul(mat("translate", 0, 0, .5), mat("rotate", hp, 1, 0, 0)), mul(mat("translate", 0, 0, -.5), mat("rotate", -hp, 1, 0, 0))]
    types, M, mats = [], [], []

    def add(t, m, kd=(0.7, 0.7, 0.7), kr=0.2, spec=1.0, Le=0.0):
        types.append(t)
        M.append(m)
        mats.append([kd[0], kd[1], kd[2], kr, kr, kr, spec, Le, Le, Le])

    # room: inward-facing walls (the unit rectangle faces +y), floor, and the light as the ceiling
    add(2, mul(mat("translate", 0, -4, 0), mat("scale", 12, 1, 12)))
    ceil = mul(mat("translate", 0, 4, 0), mat("rotate", np.pi, 1, 0, 0), mat("scale", 12, 1, 12))
    add(2, ceil, kd=(0, 0, 0), kr=0, Le=6.0)
    add(2, mul(mat("translate", -6, 0, 0), mat("rotate", -hp, 0, 0, 1), mat("scale", 8, 1, 12)), kd=(0.8, 0.2, 0.2))
    add(2, mul(mat("translate", 6, 0, 0), mat("rotate", hp, 0, 0, 1), mat("scale", 8, 1, 12)), kd=(0.2, 0.8, 0.2))
    add(2, mul(mat("translate", 0, 0, -6), mat("rotate", hp, 1, 0, 0), mat("scale", 12, 1, 8)))
    for b in range(n_boxes):
        ax = rng.uniform(-1, 1, 3)
        ax /= np.linalg.norm(ax)
        box = mul(mat("translate", *rng.uniform(-4, 4, 3)), mat("rotate", rng.uniform(-3, 3), *ax), mat("scale", *rng.uniform(0.5, 2.5, 3)))
        kd = rng.uniform(0.2, 1.0, 3)
        for face in faces:
            add(2, mul(box, face), kd=kd, kr=float(rng.choice([0.0, 0.6])), spec=float(rng.choice([1.0, 200.0])))
    if with_spheres:
        for s in range(8):
            add(3, mul(mat("translate", *rng.uniform(-4, 4, 3)), mat("scale", *([rng.uniform(0.3, 0.9)] * 3))), kd=rng.uniform(0.2, 1, 3))
    if tilt:
        base = mul(mat("translate", 2.5, -1.0, 2.0), mat("scale", 3, 1, 3))
        add(2, base, kd=(0.9, 0.9, 0.2))
        add(2, mul(mat("translate", 2.5, -1.6, 2.0), mat("rotate", np.pi + 0.006, 1, 0, 0), mat("scale", 3, 1, 3)), kd=(0.2, 0.9, 0.9))
    cam = oracle.scene_tables(oracle.scene("cornell", W, H))["cam"]
    sc = oracle.scene_from_tables(np.array(types), np.stack(M), np.array(mats, dtype=np.float32),
                                  np.stack([oracle.light_from_matrix(ceil, falloff=0.02)]), cam, (0.02, 0.02, 0.05))
    return sc, oracle.scene_tables(sc)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 2, False, False), (2, 5, False, True), (3, 9, True, False), (4, 14, True, True), (5, 30, False, False)])
def test_paired_rectangles_in_boxes(capi, oracle, case):
    """the fast walk tests rectangles with opposite normals as one pair (up-front list always; leaves when every leaf is a whole
    box): same pixels as the canonical walk bit for bit, with whole-box leaves, mixed leaves, boxes cut by Morton leaves
    and a pair that is not quite parallel; and both agree with the oracle"""
    seed, n_boxes, with_spheres, tilt = case
    W, H, n = 96, 72, 2
    sc, t = _box_scene(oracle, seed, W, H, n_boxes, with_spheres, tilt)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], None)
    ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"])
    ctx.set_lights(t["lights"])
    prev = np.full((H, W, 4), 0.25, np.float32)
    for path, amb in ((True, False), (False, False)):
        canon, cimg = gpu_render(capi, ctx, W, H, n, 1, path, amb, stats=True, prev=prev)
        fast, fimg = gpu_render(capi, ctx, W, H, n, 1, path, amb, stats=False, prev=prev)
        assert np.array_equal(fast.view(np.uint32), canon.view(np.uint32)), "fast walk != canonical walk"
        assert np.array_equal(fimg, cimg)
        racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 1, path=path, ambient=amb, mode=1), accum_prev=prev)
        assert_parity(canon, racc, cimg, rimg, what="boxes %s path=%s" % (case, path))
    ctx.close()


@pytest.mark.gpu
def test_bench_contract_and_rccl_path_on_one_gpu():
    """bench.py prints exactly one JSON line on stdout with the contract's keys (+ roofline), and its N>1 machinery -- RCCL
    process group, comm stream, triple-buffered bands, gather + de-interleave of every frame -- runs with a world of one rank
    and hands back the frame that was rendered"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--width", "320", "--height", "180", "--no-cpu-baseline"]
    for extra in ([], ["--single-rank-collectives"], ["--single-rank-collectives", "--launches-per-frame", "2"]):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        j = json.loads(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline"):
            assert k in j, k
        assert j["n_gpus"] == 1 and j["steps"] == 4 and j["value"] > 0 and j["roofline"]["frac"] > 0
        if extra:
            assert j["gathered_frame_matches_single_launch"] is True
            assert j["config"]["launches_per_frame_per_gpu"] == (2 if len(extra) > 1 else 1)


@pytest.mark.gpu
def test_far_camera_where_the_sphere_quadratic_loses_its_digits(capi, oracle):
    """1200 units from the slide scene the sphere's b*b - 4ac is good to a tenth of a radius: the reference's arithmetic then
    reports a hit 0.08 units OFF a rotated sphere, inside its CubeBox box but outside a tight one.  The timed kernel must give
    what that arithmetic gives (found by tools/fuzz_cameras.py: slide, seeds 51/8 and 9/12)."""
    W, H, n = 144, 80, 2
    sc, t, ctx = upload(capi, oracle, "slide", W, H)
    cams = [([1157.8185, 86.236015, 436.8588], [-7.5457845, 36.456894, -7.860523], [0.04177115, 1.0, 0.16844608], 21.766),
            ([-1089.4718, 357.51648, 1021.0891], [35.332947, 62.253723, 28.36545], [-0.01468868, 1.0, -0.27898717], 31.809)]
    for eye, look, up, fov in cams:
        eye, look, up = oracle.f32(eye), oracle.f32(look), oracle.f32(up)
        U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
        oracle.lib().oracle_camera_uvw(oracle.fptr(eye), oracle.fptr(look), oracle.fptr(up), fov, np.float32(np.float32(W) / np.float32(H)),
                                       oracle.fptr(U), oracle.fptr(V), oracle.fptr(Wv))
        sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = eye.tolist(), U.tolist(), V.tolist(), Wv.tolist()
        ctx.set_camera(eye, U, V, Wv)
        for path in (False, True):
            fast, fimg = gpu_render(capi, ctx, W, H, n, 0, path)
            canon, cimg = gpu_render(capi, ctx, W, H, n, 0, path, stats=True)
            assert np.array_equal(fast.view(np.uint32), canon.view(np.uint32)), (eye, path)
            racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, mode=1))
            assert_parity(canon, racc, cimg, rimg, what="far camera path=%s" % path)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "checkered"])
def test_precomputed_shading_frames_are_bitwise_the_per_hit_ones(capi, oracle, name, monkeypatch):
    """scenes of flat primitives only: the path-mode kernel takes N = normalize(TransformNormal(0,1,0)) (kernel.cu:428) and
    GetRayOnHemisphere's tangent (kernel.cu:105) from the frames build_kernel computed once per primitive (render_kernel<.., FRAMES>);
    RTGO_NO_FRAMES launches the instantiation that computes them per hit: the same bits, at 16 and at 36 spp (lock-step and streaming)"""
    W, H = 320, 180
    sc, t, ctx = upload(capi, oracle, name, W, H)
    for n in (4, 6):
        prev = None
        outs = []
        for no_frames in (False, True):
            if no_frames:
                monkeypatch.setenv("RTGO_NO_FRAMES", "1")
            else:
                monkeypatch.delenv("RTGO_NO_FRAMES", raising=False)
            acc = None
            for f in range(2):
                acc, img = gpu_render(capi, ctx, W, H, n, f, True, prev=acc)
            outs.append((acc.copy(), img.copy()))
        assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32)) and np.array_equal(outs[0][1], outs[1][1]), (name, n)
        canon, cimg = None, None
        for f in range(2):
            canon, cimg = gpu_render(capi, ctx, W, H, n, f, True, stats=True, prev=canon)
        assert np.array_equal(outs[0][0].view(np.uint32), canon.view(np.uint32)), (name, n)
    monkeypatch.delenv("RTGO_NO_FRAMES", raising=False)
    ctx.close()


def _shifted_upload(capi, oracle, name, W, H, shift, eye_offset, fov=60.0):
    """scene `name` moved by `shift`, seen from shift + eye_offset looking at its centre: (oracle scene, context)"""
    sc = oracle.scene(name, W, H)
    t = oracle.scene_tables(sc)
    M = np.array(t["M"], dtype=np.float32).reshape(-1, 16).copy()
    for k, col in enumerate((3, 7, 11)):
        M[:, col] = (M[:, col] + np.float32(shift[k])).astype(np.float32)
    L = np.array(t["lights"], dtype=np.float32).reshape(-1, 16).copy()
    L[:, 0:3] = (L[:, 0:3] + oracle.f32(shift)).astype(np.float32)
    eye, look, up = oracle.f32(np.asarray(shift) + np.asarray(eye_offset)), oracle.f32(shift), oracle.f32([0, 1, 0])
    U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    oracle.lib().oracle_camera_uvw(oracle.fptr(eye), oracle.fptr(look), oracle.fptr(up), fov, np.float32(np.float32(W) / np.float32(H)),
                                   oracle.fptr(U), oracle.fptr(V), oracle.fptr(Wv))
    cam = np.concatenate([eye, U, V, Wv]).astype(np.float32)
    sc2 = oracle.scene_from_tables(t["type"], M, t["mat"], L, cam, t["bg"])     # (boxes by the CubeBox rule, like the device's)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], M, t["mat"], None)
    ctx.set_camera(eye, U, V, Wv)
    ctx.set_background(t["bg"])
    ctx.set_lights(L)
    return sc2, ctx


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("cornell", (290.0, 0.0, 0.0), (0.0, 0.0, 14.0), 60.0, False),      # reach ~ 304: flat primitives only
                                  ("checkered", (-470.0, 10.0, 30.0), (0.0, 2.0, 14.0), 60.0, False),  # reach ~ 490, just inside
                                  ("cornell", (0.0, 505.0, 0.0), (0.0, 0.0, 14.0), 60.0, True),        # reach ~ 509: beyond the reach guard
                                  ("mirror_spheres", (0.0, 0.0, 0.0), (0.0, 0.0, 75.0), 12.0, False),  # quadric ~ 6.5e3: inside
                                  ("mirror_spheres", (0.0, 0.0, 0.0), (0.0, 0.0, 95.0), 10.0, True),   # quadric ~ 1e4: beyond the quadric guard
                                  ("plateau", (20.0, -15.0, 10.0), (30.0, 20.0, 40.0), 20.0, False),
                                  ("balls", (0.0, 0.0, 0.0), (0.0, 0.0, 14.0), 60.0, False),           # the reference's own view: quadric ~ 2.1e3
                                  ("slide", (0.0, 0.0, 0.0), (0.0, 0.0, 14.0), 60.0, True)])           # ... and the one reference view beyond: 4.3e4
def test_far_field_guard(capi, oracle, case):
    """The guard of rtgo_launch (rtgo_capi.hip: kGuardReach 500, kGuardQuadric 8000; evidence: profiles/r03a/farfield_*.log).  Inside it the
    timed kernel walks the fast structure and its frame is the canonical walk's bit for bit; beyond it the product launch takes the
    canonical walk (without counters) -- rtgo_stats says which ran and carries the two quantities."""
    name, shift, eye_off, fov, beyond = case
    W, H, n = 160, 90, 2
    sc, ctx = _shifted_upload(capi, oracle, name, W, H, shift, eye_off, fov)
    for path in (True, False):
        ctx.reset_stats()
        fast, fimg = gpu_render(capi, ctx, W, H, n, 0, path)
        st = ctx.stats()
        assert (st["guard_reach"] > 500.0 or st["guard_quadric"] > 8000.0) == beyond, st
        assert st["launches_canonical"] == (1 if beyond else 0), st
        canon, cimg = gpu_render(capi, ctx, W, H, n, 0, path, stats=True)
        assert np.array_equal(fast.view(np.uint32), canon.view(np.uint32)) and np.array_equal(fimg, cimg), (case, path, st)
        racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, mode=1))
        assert_parity(canon, racc, cimg, rimg, what="far field %s path=%s" % (name, path))
    print("far-field guard", name, "reach %.1f quadric %.0f" % (st["guard_reach"], st["guard_quadric"]), "canonical" if beyond else "fast")
    ctx.close()


@pytest.mark.gpu
def test_smallest_and_largest_scenes(capi, oracle):
    """one primitive (no tree at all: the fast walk is its up-front list or a root leaf) and the maximum of 512 (every LDS array
    of the build at its limit; workgroup size and stack chosen for it), both kernels against each other and the oracle"""
    W, H, n = 80, 60, 2
    L = oracle.lib()
    cam = oracle.scene_tables(oracle.scene("cornell", W, H))["cam"]
    rng = np.random.RandomState(7)

    def trs(t, s):
        T, S, out = [np.zeros(16, dtype=np.float32) for _ in range(3)]
        L.oracle_mat_translate(*[np.float32(v) for v in t], oracle.fptr(T))
        L.oracle_mat_scale(*[np.float32(v) for v in s], oracle.fptr(S))
        L.oracle_mat_mul(oracle.fptr(T), oracle.fptr(S), oracle.fptr(out))
        return out

    light = trs((0, 6, 0), (8, 1, 8))
    R = np.zeros(16, dtype=np.float32); out = np.zeros(16, dtype=np.float32)
    L.oracle_mat_rotate(np.float32(np.pi), 1.0, 0.0, 0.0, oracle.fptr(R))
    L.oracle_mat_mul(oracle.fptr(light), oracle.fptr(R), oracle.fptr(out))
    light = out.copy()
    cases = []
    for ptype, m in ((3, trs((0, 0, 0), (3, 3, 3))), (2, trs((0, -1, 0), (9, 1, 9)))):
        cases.append((np.array([ptype, 2]), np.stack([m, light]), np.array([[0.8, 0.3, 0.3, 0.2, 0.2, 0.2, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 1, 8, 8, 8]], np.float32)))
    types = rng.randint(0, 4, 512); types[0] = 2
    M = np.stack([light] + [trs(rng.uniform(-6, 6, 3), rng.uniform(0.15, 0.6, 3)) for _ in range(511)])
    mats = np.zeros((512, 10), np.float32); mats[:, 0:3] = rng.uniform(0.2, 1, (512, 3)); mats[:, 6] = 1.0
    mats[0] = [0, 0, 0, 0, 0, 0, 1, 8, 8, 8]
    cases.append((types, M, mats))
    for types, M, mats in cases:
        sc = oracle.scene_from_tables(types, M, mats, np.stack([oracle.light_from_matrix(light, falloff=0.02)]), cam, (0.1, 0.1, 0.2))
        t = oracle.scene_tables(sc)
        ctx = capi.Context(0)
        ctx.set_scene(t["type"], t["M"], t["mat"], None)
        ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12]); ctx.set_background(t["bg"]); ctx.set_lights(t["lights"])
        for path in (True, False):
            ctx.reset_stats()
            canon, cimg = gpu_render(capi, ctx, W, H, n, 0, path, stats=True)
            st = ctx.stats()
            fast, fimg = gpu_render(capi, ctx, W, H, n, 0, path)
            assert np.array_equal(fast.view(np.uint32), canon.view(np.uint32)), (len(types), path)
            racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, mode=1))
            assert_parity(canon, racc, cimg, rimg, what="%d primitives path=%s" % (len(types), path))
            assert abs(st["rays_total"] - rc["rays_total"]) <= 0.01 * rc["rays_total"]
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "checkered", "window"])
def test_rays_with_an_exactly_zero_direction_component(capi, oracle, name):
    """A hemisphere sample has sin(phi) = 0 exactly whenever its random number is 0 (one bounce in 2^24: a few rays per 1080p
    frame), and then one direction component is exactly zero; 1/0 in the fast walk's slab test once dropped every box that
    straddles the origin on that axis (found by tools/fuzz_scenes.py, one pixel in 3e7).  Here every primary ray has it: a
    camera whose three axes lie in a coordinate plane (the ABI takes any U, V, W)."""
    W, H, n = 96, 64, 2
    sc, t, ctx = upload(capi, oracle, name, W, H)
    for axis in range(3):
        a, b = [k for k in range(3) if k != axis]
        eye = np.zeros(3, np.float32); eye[axis] = 0.3; eye[a] = 0.2; eye[b] = -0.4
        U, V, Wv = [np.zeros(3, np.float32) for _ in range(3)]
        U[a], V[b] = 1.0, 1.0
        Wv[a], Wv[b] = 0.35, 0.15
        sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = eye.tolist(), U.tolist(), V.tolist(), Wv.tolist()
        ctx.set_camera(eye, U, V, Wv)
        for path in (True, False):
            fast, fimg = gpu_render(capi, ctx, W, H, n, 0, path)
            canon, cimg = gpu_render(capi, ctx, W, H, n, 0, path, stats=True)
            assert np.array_equal(fast.view(np.uint32), canon.view(np.uint32)), (name, axis, path)
            racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, mode=1))
            assert_parity(canon, racc, cimg, rimg, what="%s zero component %d path=%s" % (name, axis, path))
    ctx.close()


def _cuboid_scene(oracle, W, H, variant):
    """a closed room of six inward-facing walls (three pairs: the up-front list) holding two boxes of six rectangles each.
    variant "good": both are cuboids; "wide_face": one face of the second box is 1.5 x too wide (it sticks out past its neighbours:
    the point where a ray meets its plane can be beyond another face's plane and still inside the face); "sheared": the second box is
    a parallelepiped (still the image of the unit cube); "inside_out": its six faces look into it"""
    L = oracle.lib()
    f = oracle.fptr

    def mat(kind, *a):
        out = np.zeros(16, dtype=np.float32)
        getattr(L, "oracle_mat_" + kind)(*[np.float32(v) for v in a], f(out))
        return out

    def mul(*ms):
        acc = ms[0]
        for m in ms[1:]:
            out = np.zeros(16, dtype=np.float32)
            L.oracle_mat_mul(f(acc), f(m), f(out))
            acc = out
        return acc

    hp = np.pi / 2
    faces = [mul(mat("translate", 0, .5, 0)), mul(mat("translate", 0, -.5, 0), mat("rotate", np.pi, 1, 0, 0)),
             mul(mat("translate", .5, 0, 0), mat("rotate", -hp, 0, 0, 1)), mul(mat("translate", -.5, 0, 0), mat("rotate", hp, 0, 0, 1)),
             mul(mat("translate", 0, 0, .5), mat("rotate", hp, 1, 0, 0)), mul(mat("translate", 0, 0, -.5), mat("rotate", -hp, 1, 0, 0))]
    types, M, mats = [], [], []

    def add(m, kd=(0.7, 0.7, 0.7), kr=0.2, spec=1.0, Le=0.0):
        types.append(2)
        M.append(m)
        mats.append([kd[0], kd[1], kd[2], kr, kr, kr, spec, Le, Le, Le])

    add(mul(mat("translate", 0, -4, 0), mat("scale", 12, 1, 12)))
    add(mul(mat("translate", 0, 4, 0), mat("rotate", np.pi, 1, 0, 0), mat("scale", 12, 1, 12)))
    add(mul(mat("translate", -6, 0, 0), mat("rotate", -hp, 0, 0, 1), mat("scale", 8, 1, 12)), kd=(0.8, 0.2, 0.2))
    add(mul(mat("translate", 6, 0, 0), mat("rotate", hp, 0, 0, 1), mat("scale", 8, 1, 12)), kd=(0.2, 0.8, 0.2))
    add(mul(mat("translate", 0, 0, -6), mat("rotate", hp, 1, 0, 0), mat("scale", 12, 1, 8)))
    add(mul(mat("translate", 0, 0, 6), mat("rotate", -hp, 1, 0, 0), mat("scale", 12, 1, 8)))
    light = mul(mat("translate", 0, 3.9, 0), mat("rotate", np.pi, 1, 0, 0), mat("scale", 3, 1, 3))
    add(light, kd=(0, 0, 0), kr=0, Le=6.0)
    box_a = mul(mat("translate", -2.2, -2.5, 0.5), mat("rotate", 0.4, 0, 1, 0), mat("scale", 2.5, 3.0, 2.5))
    box_b = mul(mat("translate", 2.0, -3.0, -1.0), mat("rotate", -0.3, 0, 1, 0), mat("scale", 2.0, 2.0, 2.0))
    if variant == "sheared":
        shear = np.eye(4, dtype=np.float32)
        shear[0, 1] = 0.35
        box_b = mul(box_b, shear.reshape(16).copy())
    if variant == "tiny_far":
        box_b = mul(mat("translate", 2.0, -3.9, -1.0), mat("rotate", -0.3, 0, 1, 0), mat("scale", 0.12, 0.12, 0.12))
    for face in faces:
        add(mul(box_a, face), kd=(0.9, 0.8, 0.3))
    for k, face in enumerate(faces):
        m = mul(box_b, face)
        if variant == "wide_face" and k == 0:
            m = mul(m, mat("scale", 1.5, 1, 1))
        if variant == "inside_out":
            m = mul(m, mat("translate", 0, 0, 0), mat("rotate", np.pi, 1, 0, 0))   # the rectangle flipped in place
        add(m, kd=(0.3, 0.5, 0.9), kr=0.5)
    cam = np.array(oracle.scene_tables(oracle.scene("cornell", W, H))["cam"], dtype=np.float32).copy()
    if variant in ("far", "tiny_far"):
        # the same view from 12 x the distance (eye 168 units out, still the fast walk's domain): the margin of the cuboid test grows with
        # the rays' reach -- a fat margin for "far", beyond the 0.02 at which the launch stops using the test for "tiny_far"
        cam[0:12] *= 12.0
    sc = oracle.scene_from_tables(np.array(types), np.stack(M), np.array(mats, dtype=np.float32),
                                  np.stack([oracle.light_from_matrix(light, falloff=0.02)]), cam, (0.02, 0.02, 0.05))
    return sc, oracle.scene_tables(sc)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,groups", [("good", 3), ("sheared", 3), ("wide_face", 2), ("inside_out", 2), ("far", 3), ("tiny_far", 3)])
def test_cuboid_certificate_and_test(capi, oracle, variant, groups, monkeypatch):
    """three rectangle pairs the build certifies as the faces of one box (leaves) or one room (the up-front list) go through the
    fast walk's cuboid test (at most one of the three front-facing faces can be hit: the unit-square half of the test runs once).
    The certificate is geometric -- every corner of every face on the inner side of the other pairs' planes -- so a box with a face
    that sticks out, or one turned inside out, must not get it, a sheared one may; the pixels are the canonical walk's bit for bit
    either way, and equal those of the pair-by-pair tests (RTGO_NO_CUBOID)"""
    W, H, n = 128, 96, 3
    sc, t = _cuboid_scene(oracle, W, H, variant)
    prev = np.full((H, W, 4), 0.25, np.float32)
    frames = {}
    for knob in ("", "1"):
        if knob:
            monkeypatch.setenv("RTGO_NO_CUBOID", knob)
        else:
            monkeypatch.delenv("RTGO_NO_CUBOID", raising=False)
        ctx = capi.Context(0)
        ctx.set_scene(t["type"], t["M"], t["mat"], None)
        ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
        ctx.set_background(t["bg"])
        ctx.set_lights(t["lights"])
        assert ctx.stats()["cuboid_groups"] == (0 if knob else groups), (variant, knob, ctx.stats()["cuboid_groups"])
        for path in (True, False):
            canon, cimg = gpu_render(capi, ctx, W, H, n, 2, path, stats=True, prev=prev)
            fast, fimg = gpu_render(capi, ctx, W, H, n, 2, path, stats=False, prev=prev)
            assert np.array_equal(fast.view(np.uint32), canon.view(np.uint32)), "fast walk != canonical walk (%s, path=%s)" % (variant, path)
            assert np.array_equal(fimg, cimg)
            frames[(knob, path)] = (fast, fimg)
        assert ctx.stats()["launches_canonical"] == 2, "the timed launches must have taken the fast walk"
        ctx.close()
    for path in (True, False):
        assert np.array_equal(frames[("", path)][0].view(np.uint32), frames[("1", path)][0].view(np.uint32))
    racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, n, 2, path=True, mode=1), accum_prev=prev)
    assert_parity(frames[("", True)][0], racc, frames[("", True)][1], rimg, what="cuboid scene " + variant)


def test_cuboid_certificates_of_the_reference_scenes(capi, oracle):
    """cornell: the room and the two boxes; checkered: its 64 cubes; balls: the room"""
    expect = {"cornell": 3, "balls": 1, "checkered": 64}
    for name, want in expect.items():
        sc, t, ctx = upload(capi, oracle, name, 64, 64)
        got = ctx.stats()["cuboid_groups"]
        assert got >= want, (name, got)
        ctx.close()


@pytest.mark.gpu
def test_lean_ops_are_ieee(tmp_path):
    """div_cr / sqrt_cr (rtgo_device.h: the compiler's own correctly rounded division and square root without the range plumbing around
    them) against the plain operators on the device, bit for bit, over 2^30 operand pairs per range drawn log-uniformly from the ranges
    the kernels feed them, and sincos_cr against the library's sincosf on every float of [0, 8]: tools/lean_ops_probe.hip, exit code 0 iff none differs"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from raytracingo_amd import _build
    exe = str(tmp_path / "lean_ops_probe")
    subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(root, "tools", "lean_ops_probe.hip")])
    r = subprocess.run([exe, str(1 << 30)], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "division differs on 0, sqrt on 0" in r.stdout.splitlines()[0] and "division differs on 0, sqrt on 0" in r.stdout.splitlines()[1], r.stdout
    assert any(l.startswith("sincos_cr against sincosf on every float") and l.rstrip().endswith("differs on 0") for l in r.stdout.splitlines()), r.stdout


@pytest.mark.gpu
def test_both_walks_agree_on_every_ray(tmp_path):
    """the -DRTGO_CMPWALK build of the library runs the fast walk next to the canonical one on EVERY ray of an instrumented launch
    and records the rays on which hit, t, primitive or normal differ (tools/cmp_walks.py): all 8 scenes x 3 modes, none"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from raytracingo_amd import _build
    lib = str(tmp_path / "librtgo_hip_cmpwalk.so")
    subprocess.check_call([_build.HIPCC] + _build.HIP_FLAGS + ["-DRTGO_CMPWALK", "-o", lib, os.path.join(root, "raytracingo_amd", "csrc", "rtgo_capi.hip")])
    env = dict(os.environ, RTGO_HIP_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "cmp_walks.py"), "480", "270", "3", "2"], capture_output=True, text=True,
                       timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.startswith("total:")][-1]
    rays, bad = int(last.split()[1]), int(last.split()[3])
    assert rays > 100_000_000 and bad == 0, r.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("rccl_local", [False, True])
def test_cpp_multigpu_driver_two_shares_on_one_gpu(capi, oracle, rccl_local):
    """engine::host::MultiGpuRenderer (C++ host, raytracingo_amd/host/multigpu.cpp) on a one-GPU box: two shares of the band
    interleave on GPU 0, each with its own context, stream and resident accumulation bands; the 8-bit bands are gathered to the
    root every frame -- by device-to-device copies, or (rccl_local) through RCCL's grouped ncclSend/ncclRecv on a one-rank
    communicator -- and assembled by rtgo_assemble_bands.  The presented frame and the gathered accumulation buffer must equal
    the single-launch Renderer's bit for bit (pixels are independent: kernel.cu:187-217), and the oracle's within tolerance."""
    from raytracingo_amd import scene as hscene
    W, H, n, frames = 200, 90, 2, 3        # 90 rows: 23 bands of 4 rows over 2 shares, the last band 2 rows high
    acc1, img1, st1 = hscene.host_render("mirror_spheres", "path", W, H, sample=n, frames=frames)
    acc2, img2, st2, ms = hscene.host_render_multi("mirror_spheres", "path", W, H, sample=n, frames=frames, devices=(0,),
                                                   launches_per_device=2, present_every=1, rccl_for_local_shares=rccl_local)
    assert np.array_equal(img2, img1)
    assert np.array_equal(acc2.view(np.uint32), acc1.view(np.uint32))
    assert st2["rays_total"] == st1["rays_total"] and st2["launches"] == frames
    # a frame that is not presented is still accumulated: present only the last of the three
    acc3, img3, _, _ = hscene.host_render_multi("mirror_spheres", "path", W, H, sample=n, frames=frames, devices=(0,),
                                                launches_per_device=2, present_every=5, rccl_for_local_shares=rccl_local)
    assert np.array_equal(img3, img1) and np.array_equal(acc3.view(np.uint32), acc1.view(np.uint32))
    sc = oracle.scene("mirror_spheres", W, H)
    racc = None
    for f in range(frames):
        racc, rimg, _ = oracle.render(sc, oracle.frame(W, H, n, f, path=True, mode=1), accum_prev=racc)
    assert_parity(acc2, racc, img2, rimg, what="C++ multi-GPU driver")
    print("multi-GPU driver, 2 shares on one GPU, rccl_local=%s: %.3f ms/frame host wall" % (rccl_local, ms))


@pytest.mark.gpu
def test_assemble_bands_odd_shapes(capi):
    """rtgo_assemble_bands against numpy for widths that force the 4-byte path, band heights 1..5 and 1..7 ranks"""
    import torch
    ctx = capi.Context(0)
    L = capi.load()
    rng = np.random.default_rng(5)
    for (w, h, band_h, G, elem) in [(37, 19, 4, 3, 4), (64, 64, 4, 8, 4), (5, 7, 1, 7, 16), (130, 33, 5, 2, 4), (3, 2, 4, 1, 16), (21, 40, 3, 4, 16)]:
        rows = [[r for r in range(h) if (r // band_h) % G == g] for g in range(G)]
        rows_pad = max(len(r) for r in rows) + 1
        full = rng.integers(0, 255, size=(h, w * elem), dtype=np.uint8)
        gathered = np.zeros((G * rows_pad, w * elem), dtype=np.uint8)
        for g in range(G):
            gathered[g * rows_pad:g * rows_pad + len(rows[g])] = full[rows[g]]
        d_g = torch.from_numpy(gathered).cuda()
        d_f = torch.zeros((h, w * elem), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        rc = L.rtgo_assemble_bands(ctx._h, None, d_g.data_ptr(), d_f.data_ptr(), w, h, band_h, G, rows_pad, elem)
        assert rc == 0
        ctx.sync()
        assert np.array_equal(d_f.cpu().numpy(), full), (w, h, band_h, G, elem)
    assert L.rtgo_assemble_bands(ctx._h, None, d_g.data_ptr(), d_f.data_ptr(), 21, 40, 3, 4, 2, 16) != 0   # rows_pad too small
    ctx.close()


# ---- the whitted triangle path (cuda/whitted.cu): SURVEY 8f row f4 --------------------------------------------------------------
def _whitted_ctx(capi, mesh, cam, W, H):
    ctx = capi.Context(0)
    ctx.whitted_set_mesh(mesh["positions"], mesh.get("normals"), mesh["indices"], mesh.get("tri_material"), mesh["materials"])
    if mesh.get("texcoords") is not None:
        ctx.whitted_set_texcoords(mesh["texcoords"])
    for mi, (bc, mr, nm) in (mesh.get("textures") or {}).items():
        ctx.whitted_set_material_textures(mi, bc, mr, nm)
    ctx.whitted_set_lights(mesh["lights"])
    ctx.whitted_set_miss_color(mesh["miss"])
    ctx.set_camera(cam[0:3], cam[3:6], cam[6:9], cam[9:12])
    ctx.resize(W * H)
    return ctx


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["smooth", "faceted", "one_light_fine"])
def test_whitted_triangles_against_the_oracle(capi, oracle, variant):
    """__raygen__pinhole / __closesthit__radiance / __closesthit__occlusion / __miss__constant_radiance of cuda/whitted.cu over a
    procedural triangle scene (tests/whitted_scene.py), subframes 0..3 accumulated (whitted.cu:226-239), against the oracle's
    brute-force restatement.  The device walks an LBVH it built itself; the triangle test and the closest-hit rule are shared
    operation for operation, so which triangle a ray hits, and where, agree exactly -- what differs is the device libm
    (powf in schlick and make_color, sqrtf): tolerance 1e-4 relative on the float buffer, 1 LSB on the 8-bit image."""
    import whitted_scene
    W, H = 160, 100
    mesh = whitted_scene.build(n_lat=40, n_lon=48) if variant == "one_light_fine" else whitted_scene.build()
    if variant == "faceted":
        mesh = dict(mesh, normals=None)
    if variant == "one_light_fine":
        mesh = dict(mesh, lights=mesh["lights"][:1])          # ~3.8 k triangles: close to RTGO_MAX_TRIANGLES
        assert 3000 < len(mesh["indices"]) <= 4096
    cam = whitted_scene.camera(oracle, W, H)
    ctx = _whitted_ctx(capi, mesh, cam, W, H)
    ctx.reset_stats()
    for sf in range(4):
        ctx.whitted_launch(W, H, sf)
    ctx.sync()
    acc, img = ctx.read_accum(H, W), ctx.read_image(H, W)
    st = ctx.stats()
    racc, rimg, rc = oracle.whitted_render(mesh, cam, W, H, 4)
    m = assert_parity(acc, racc, img, rimg, what="whitted " + variant)
    assert abs(st["rays_total"] - rc["rays_total"]) <= 0.002 * rc["rays_total"]
    assert abs(st["rays_occlusion"] - rc["rays_occlusion"]) <= 0.002 * rc["rays_occlusion"]
    # subframe 0 alone: no jitter, so primary hits are identical and the ray counts must agree exactly
    ctx.reset_stats()
    ctx.whitted_launch(W, H, 0)
    ctx.sync()
    r0, _, c0 = oracle.whitted_render(mesh, cam, W, H, 1)
    st0 = ctx.stats()
    assert st0["rays_total"] == c0["rays_total"] and st0["rays_occlusion"] == c0["rays_occlusion"]
    a0 = ctx.read_accum(H, W)
    assert np.array_equal(a0[..., 3], r0[..., 3])
    # pixels that see only the miss colour are exact
    miss_px = (r0[..., :3] == mesh["miss"]).all(axis=-1)
    assert miss_px.any() and np.array_equal(a0[miss_px], r0[miss_px])
    print("whitted", variant, m, st0["last_launch_ms"], "ms,", len(mesh["indices"]), "triangles")
    ctx.close()


@pytest.mark.gpu
def test_whitted_the_three_residency_modes_agree(capi, oracle, monkeypatch):
    """the render kernel walks one of three forms of the same structure -- everything in LDS on a 16-bit grid (what the other tests
    run, the meshes being small enough), fp32 records in LDS with the triangles in L2, everything in L2: closest hits are
    closest hits, so the three frames must be bitwise one frame"""
    import whitted_scene
    W, H = 160, 100
    mesh = whitted_scene.build(n_lat=24, n_lon=32)
    cam = whitted_scene.camera(oracle, W, H)
    frames = []
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("RTGO_WHITTED_MODE", mode)
        ctx = _whitted_ctx(capi, mesh, cam, W, H)
        ctx.reset_stats()
        for sf in range(3):
            ctx.whitted_launch(W, H, sf)
        ctx.sync()
        frames.append((ctx.read_accum(H, W), ctx.read_image(H, W), ctx.stats()["rays_total"]))
        ctx.close()
    for acc, img, rays in frames[1:]:
        assert np.array_equal(acc.view(np.uint32), frames[0][0].view(np.uint32)) and np.array_equal(img, frames[0][1]) and rays == frames[0][2]
    racc, rimg, rc = oracle.whitted_render(mesh, cam, W, H, 3)
    assert_parity(frames[0][0], racc, frames[0][1], rimg, what="whitted, three modes")


@pytest.mark.gpu
def test_whitted_surface_area_tree_and_morton_tree_agree(capi, oracle, monkeypatch):
    """rtgo_whitted_set_mesh rebuilds the walk's records over the Morton hierarchy's leaves with the surface-area heuristic
    (sah_kernel); RTGO_WHITTED_NO_SAH keeps the Morton topology.  Any tree over the same leaves returns the same hits: the frames are
    bitwise one frame in every residency mode, for a mesh of a few leaves and for the largest mesh the build takes"""
    import whitted_scene
    W, H = 128, 80
    cam = whitted_scene.camera(oracle, W, H)
    for n_lat, n_lon in ((2, 3), (6, 8), (43, 48)):   # 20, 94 and 4046 triangles (the build takes 4096)
        mesh = whitted_scene.build(n_lat=n_lat, n_lon=n_lon)
        frames = []
        for no_sah in (False, True):
            for mode in ("2", "0"):
                monkeypatch.setenv("RTGO_WHITTED_MODE", mode)
                if no_sah:
                    monkeypatch.setenv("RTGO_WHITTED_NO_SAH", "1")
                else:
                    monkeypatch.delenv("RTGO_WHITTED_NO_SAH", raising=False)
                ctx = _whitted_ctx(capi, mesh, cam, W, H)
                ctx.reset_stats()
                for sf in range(2):
                    ctx.whitted_launch(W, H, sf)
                ctx.sync()
                frames.append((ctx.read_accum(H, W), ctx.read_image(H, W), ctx.stats()["rays_total"]))
                ctx.close()
        for acc, img, rays in frames[1:]:
            assert np.array_equal(acc.view(np.uint32), frames[0][0].view(np.uint32)) and np.array_equal(img, frames[0][1]) and rays == frames[0][2], (n_lat, n_lon)
    monkeypatch.delenv("RTGO_WHITTED_NO_SAH", raising=False)


@pytest.mark.gpu
def test_whitted_against_the_committed_fixture(capi):
    """tests/golden/oracle_whitted.npz: the GPU against stored oracle output (no oracle run needed)"""
    import whitted_scene
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_whitted.npz"))
    W, H = 64, 40
    ctx = _whitted_ctx(capi, whitted_scene.build(), z["cam"], W, H)
    ctx.reset_stats()
    ctx.whitted_launch(W, H, 0)
    ctx.whitted_launch(W, H, 1)
    ctx.sync()
    assert_parity(ctx.read_accum(H, W), z["accum"], ctx.read_image(H, W), z["image"], what="whitted fixture")
    st = ctx.stats()
    assert abs(st["rays_total"] - int(z["rays"][0])) <= 0.002 * int(z["rays"][0])
    ctx.close()


@pytest.mark.gpu
def test_whitted_edge_cases(capi, oracle):
    """one triangle (the tree is a single leaf), a camera inside the mesh's box, no lights, and the argument checks"""
    import whitted_scene
    W, H = 64, 48
    one = {"positions": np.array([[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], np.float32), "normals": None, "indices": np.array([[0, 1, 2]], np.uint32),
           "tri_material": None, "materials": np.array([[0.7, 0.7, 0.7, 1, 0.0, 0.5]], np.float32),
           "lights": np.array([[1, 1, 1, 3.0, 0.5, 1.0, 3.0, 0]], np.float32), "miss": np.array([0.0, 0.1, 0.0], np.float32)}
    cam = whitted_scene.camera(oracle, W, H, eye=(0.2, 0.6, 4.0), lookat=(0, 0.6, 0))
    ctx = _whitted_ctx(capi, one, cam, W, H)
    ctx.whitted_launch(W, H, 0)
    ctx.sync()
    racc, rimg, _ = oracle.whitted_render(one, cam, W, H, 1)
    assert_parity(ctx.read_accum(H, W), racc, ctx.read_image(H, W), rimg, what="one triangle")
    ctx.close()
    mesh = whitted_scene.build()
    cam = whitted_scene.camera(oracle, W, H, eye=(-1.2, 1.2, 0.3), lookat=(2.0, 1.0, 0.3))     # inside the sphere
    for lights in (mesh["lights"], np.zeros((0, 8), np.float32)):
        m2 = dict(mesh, lights=lights)
        ctx = _whitted_ctx(capi, m2, cam, W, H)
        ctx.whitted_launch(W, H, 0)
        ctx.whitted_launch(W, H, 1)
        ctx.sync()
        racc, rimg, _ = oracle.whitted_render(m2, cam, W, H, 2)
        assert_parity(ctx.read_accum(H, W), racc, ctx.read_image(H, W), rimg, what="camera inside, %d lights" % len(lights))
        ctx.close()
    ctx = capi.Context(0)
    with pytest.raises(capi.RtgoError):
        ctx.whitted_launch(W, H, 0)                                          # no mesh
    with pytest.raises(capi.RtgoError):
        ctx.whitted_set_mesh(one["positions"], None, np.array([[0, 1, 3]], np.uint32), None, one["materials"])    # index beyond the vertices
    big = np.zeros((8193, 3), np.uint32)
    with pytest.raises(capi.RtgoError):
        ctx.whitted_set_mesh(one["positions"], None, big, None, one["materials"])                                # too many triangles
    with pytest.raises(capi.RtgoError):
        ctx.whitted_set_mesh(one["positions"], None, one["indices"], np.array([1], np.uint32), one["materials"])  # material beyond the table
    ctx.close()


@pytest.mark.gpu
def test_whitted_coincident_triangles(capi, oracle):
    """200 copies of one triangle (every leaf box of the surface-area rebuild equal: its sweep ties at every split position) beside a
    few distinct ones: the build must not turn the ties into a chain deeper than the walk's stack (ties go to the median; a tree that
    still comes out too deep falls back to the Morton records), and the closest hit keeps the lowest triangle index"""
    W, H = 64, 48
    base = np.array([[-1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.5, 0.0]], np.float32)
    extra = np.array([[-2.0, 0.0, -1.0], [2.0, 0.0, -1.0], [0.0, 2.5, -1.0], [-0.5, 0.2, 0.5], [0.5, 0.2, 0.5], [0.0, 0.9, 0.5]], np.float32)
    pos = np.concatenate([base, extra])
    idx = np.array([[0, 1, 2]] * 200 + [[3, 4, 5], [6, 7, 8]], np.uint32)
    mats = np.array([[0.7, 0.3, 0.2, 1, 0.0, 0.5], [0.2, 0.6, 0.8, 1, 0.3, 0.4]], np.float32)
    mesh = {"positions": pos, "normals": None, "indices": idx, "tri_material": (np.arange(len(idx)) % 2).astype(np.uint32), "materials": mats,
            "lights": np.array([[1, 1, 1, 4.0, 0.5, 2.0, 3.0, 0]], np.float32), "miss": np.array([0.02, 0.02, 0.05], np.float32)}
    import whitted_scene
    cam = whitted_scene.camera(oracle, W, H, eye=(0.3, 0.8, 4.0), lookat=(0, 0.7, 0))
    ctx = _whitted_ctx(capi, mesh, cam, W, H)
    ctx.whitted_launch(W, H, 0)
    ctx.whitted_launch(W, H, 1)
    ctx.sync()
    racc, rimg, _ = oracle.whitted_render(mesh, cam, W, H, 2)
    assert_parity(ctx.read_accum(H, W), racc, ctx.read_image(H, W), rimg, what="coincident triangles")
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["waterbottle", "quads", "quads_barycentric_uv"])
def test_whitted_textures_against_the_oracle(capi, oracle, variant, monkeypatch):
    """the three tex2D branches of __closesthit__radiance (whitted.cu:264-292: base colour with linearize, metallic-roughness, normal map
    over dp/du, dp/dv of LocalGeometry.h:118-134) and getLocalGeometry's UV (:88-102): the reference's own asset data/WaterBottle (4510
    triangles: past round 2's cap of 4096; fixture by tests/golden/waterbottle/make_fixture.py), and procedural quads whose texture
    coordinates wrap on both sides under textures of odd size, with and without per-vertex coordinates.  Against the oracle at 1e-4;
    the three residency modes give one frame bit for bit."""
    import whitted_scene
    W, H = (160, 120) if variant == "waterbottle" else (96, 72)
    if variant == "waterbottle":
        mesh = whitted_scene.waterbottle()
        cam = whitted_scene.camera(oracle, W, H, eye=(0.12, 0.08, 0.42), lookat=(0.0, 0.0, 0.0), fov=40.0)
    else:
        mesh = whitted_scene.textured_quad(with_texcoords=variant == "quads")
        cam = whitted_scene.camera(oracle, W, H, eye=(0.3, 1.4, 5.0), lookat=(0.0, 0.9, -1.0), fov=45.0)
    frames = {}
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("RTGO_WHITTED_MODE", mode)
        ctx = _whitted_ctx(capi, mesh, cam, W, H)
        for sf in range(2):
            ctx.whitted_launch(W, H, sf)
        ctx.sync()
        frames[mode] = (ctx.read_accum(H, W).copy(), ctx.read_image(H, W).copy(), ctx.stats())
        ctx.close()
    monkeypatch.delenv("RTGO_WHITTED_MODE", raising=False)
    for mode in ("1", "0"):
        assert np.array_equal(frames["2"][0].view(np.uint32), frames[mode][0].view(np.uint32)), (variant, mode)
    racc, rimg, rc = oracle.whitted_render(mesh, cam, W, H, 2)
    m = assert_parity(frames["2"][0], racc, frames["2"][1], rimg, what="whitted textures " + variant)
    assert frames["2"][2]["rays_total"] == rc["rays_total"]
    on = (racc[..., :3] != np.float32(mesh["miss"])).any(axis=-1)
    assert on.mean() > 0.1 and racc[on][:, :3].mean() > 0.01     # (the mesh is in view and shaded: the comparison is not background against background)
    print("whitted textures", variant, m)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL_SCENES)
def test_the_two_fast_structures_and_the_trial_are_bitwise_the_same(capi, oracle, name, monkeypatch):
    """rtgo_set_scene builds the fast walk's structure twice (a primitive is tested up front by every ray from 36 % / from 15 % of the scene's
    extent on two axes) and the first launches of a (frame geometry, spp, mode) try the candidates -- structure x loop -- and keep the fastest
    (rtgo_ctx::Trial).  Whatever runs, the frame is the canonical walk's bit for bit: RTGO_TREE pins either structure, no pin lets the trial
    run through its candidates and settle (12 launches)."""
    W, H = 192, 108
    sc, t, ctx = upload(capi, oracle, name, W, H)
    for n, path in ((2, True), (5, True), (3, False)):
        canon, cimg = gpu_render(capi, ctx, W, H, n, 1, path, stats=True, prev=np.full((H, W, 4), 0.5, np.float32))
        for pin in ("0", "1", "2", None):          # ("2": the uniform grid, in the scenes that have one -- structure 0 otherwise)
            if pin is None:
                monkeypatch.delenv("RTGO_TREE", raising=False)
            else:
                monkeypatch.setenv("RTGO_TREE", pin)
            for rep in range(16 if pin is None else 1):
                acc, img = gpu_render(capi, ctx, W, H, n, 1, path, prev=np.full((H, W, 4), 0.5, np.float32))
                assert np.array_equal(acc.view(np.uint32), canon.view(np.uint32)) and np.array_equal(img, cimg), (name, n, path, pin, rep)
                if pin == "2" and name == "balls":
                    assert ctx.stats()["last_variant"] & 16, "balls has a grid"

    monkeypatch.delenv("RTGO_TREE", raising=False)
    ctx.close()


@pytest.mark.gpu
def test_the_launch_whose_rejection_loop_never_ended(capi, oracle):
    """tests/golden/rejection_loop: the tests' random scene 45 moved ~330 units off the origin, distributed mode, 16 spp -- the launch on which
    GetRayOnHemisphere's unbounded rejection loop (kernel.cu:109-120) hung the GPU until rtgo::hemisphere stopped after 1024 draws
    (found by tools/fuzz_farfield.py).  It returns, the product launch (beyond the far-field guard: canonical walk) and the
    instrumented one give one frame, and the oracle -- same bound -- agrees within the contract's tolerance and traces the same number of rays to 0.1 %."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rejection_loop", "random45_far.npz"))
    W, H, n, path = int(fx["W"]), int(fx["H"]), int(fx["n"]), bool(fx["path"])
    ctx = capi.Context(0)
    ctx.set_scene(fx["type"], fx["M"], fx["mat"], None)
    cam = fx["cam"]
    ctx.set_camera(cam[0:3], cam[3:6], cam[6:9], cam[9:12])
    ctx.set_background(fx["bg"])
    ctx.set_lights(fx["lights"])
    ctx.reset_stats()
    acc, img = gpu_render(capi, ctx, W, H, n, 0, path)
    st = ctx.stats()
    assert st["launches_canonical"] == 1 and st["guard_quadric"] > 8000.0, st
    acc2, img2 = gpu_render(capi, ctx, W, H, n, 0, path, stats=True)
    assert np.array_equal(acc.view(np.uint32), acc2.view(np.uint32)) and np.array_equal(img, img2)
    assert np.isfinite(acc).all()
    sc = oracle.scene_from_tables(fx["type"], fx["M"], fx["mat"], fx["lights"], cam, fx["bg"])
    racc, rimg, rc = oracle.render(sc, oracle.frame(W, H, n, 0, path=path, mode=1))
    m = assert_parity(acc, racc, img, rimg, what="rejection-loop launch")                 # (measured: 99.98 % within 1e-4, 98.3 % bit-exact)
    print("rejection-loop launch:", m, st["rays_total"], rc["rays_total"])
    assert abs(st["rays_total"] - rc["rays_total"]) <= 0.001 * rc["rays_total"]
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["random", "boxes", "boxes_spheres"])
def test_uniform_grid_walk_on_random_scenes(capi, oracle, kind, monkeypatch):
    """rtgo::fast_grid (the fast walk's third structure: a uniform grid over the small primitives, built by rtgo_set_scene) on the
    tests' random scenes -- all four primitive types under random non-uniform transforms / rooms full of boxes of six rectangles, with spheres
    between them -- with the grid forced onto scenes it would not normally be built for (RTGO_GRID_MIN, RTGO_GRID_MAX_DUP) and pinned
    (RTGO_TREE=2): the frame is the canonical walk's bit for bit, from inside and from 30 scene sizes away, and beyond the reach its pad was
    sized for the launch walks the tree instead.  tools/fuzz_scenes.py / fuzz_farfield.py / cmp_walks.py with the same pins compare ray by ray
    (profiles/r03s)."""
    monkeypatch.setenv("RTGO_GRID_MIN", "4")
    monkeypatch.setenv("RTGO_GRID_MAX_DUP", "8")
    W, H = 96, 64
    on_grid = 0
    for seed in range(8):
        if kind == "random":
            sc, t = _random_scene(oracle, 40 + seed, W, H)
        else:
            sc, t = _box_scene(oracle, 40 + seed, W, H, 3 + 4 * seed, kind == "boxes_spheres", bool(seed & 1))
        ctx = capi.Context(0)
        ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"] if seed & 2 else None)
        ctx.set_background(t["bg"])
        ctx.set_lights(t["lights"])
        bb = np.asarray(t["aabb"], dtype=np.float64).reshape(-1, 6)
        centre, size = 0.5 * (bb[:, :3].min(axis=0) + bb[:, 3:].max(axis=0)), float(np.abs(bb).max())
        rng = np.random.default_rng(seed)
        for dist, fov in ((0.2, 100.0), (1.5, 50.0), (30.0, 4.0)):
            d = rng.normal(size=3)
            eye = oracle.f32(centre + d / np.linalg.norm(d) * size * dist)
            U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
            oracle.lib().oracle_camera_uvw(oracle.fptr(eye), oracle.fptr(oracle.f32(centre)), oracle.fptr(oracle.f32([0.1, 1, 0.05])), fov,
                                           np.float32(np.float32(W) / np.float32(H)), oracle.fptr(U), oracle.fptr(V), oracle.fptr(Wv))
            ctx.set_camera(eye, U, V, Wv)
            for n, path in ((2, True), (3, False)):
                monkeypatch.delenv("RTGO_TREE", raising=False)
                canon, cimg = gpu_render(capi, ctx, W, H, n, 2, path, stats=True, prev=np.full((H, W, 4), 0.25, np.float32))
                monkeypatch.setenv("RTGO_TREE", "2")
                acc, img = gpu_render(capi, ctx, W, H, n, 2, path, prev=np.full((H, W, 4), 0.25, np.float32))
                st = ctx.stats()
                assert np.array_equal(acc.view(np.uint32), canon.view(np.uint32)) and np.array_equal(img, cimg), (kind, seed, dist, n, path, st)
                if st["last_variant"] & 16:
                    on_grid += 1
                    assert dist < 4.0 and not (st["last_variant"] & 4), st      # (30 scene sizes away: beyond the grid's reach, the tree walks)
        ctx.close()
    monkeypatch.delenv("RTGO_TREE", raising=False)
    assert on_grid >= 8, on_grid


@pytest.mark.gpu
def test_which_scenes_get_a_grid_and_at_what_resolution(capi, oracle):
    """build_grid (rtgo_capi.hip) on the reference's scenes: balls -- 256 spheres, one per 1 x 1 column of the room -- gets the resolution the
    cost search is there to find (16 columns each way: one sphere per column, at most a quarter of them listed twice: profiles/r03r, r03v);
    checkered's 384 tile faces would be listed eight times each and get none; scenes of a few dozen primitives get none."""
    import ctypes as C
    got = {}
    for name in ALL_SCENES:
        sc, t, ctx = upload(capi, oracle, name, 64, 64)
        out = (C.c_int32 * 6)()
        ctx._lib.rtgo_debug_grid.restype = C.c_int
        ctx._lib.rtgo_debug_grid.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        assert ctx._lib.rtgo_debug_grid(ctx._h, out) == 0
        got[name] = list(out)
        ctx.close()
    print(got)
    assert got["balls"][0] == 1 and got["balls"][1] == 16 and got["balls"][3] == 16 and 1 <= got["balls"][2] <= 4 and 256 <= got["balls"][4] <= 330, got["balls"]
    for name in ALL_SCENES:
        if name != "balls":
            assert got[name][0] == 0, (name, got[name])


@pytest.mark.gpu
def test_uniform_grid_walk_on_balls(capi, oracle, monkeypatch):
    """balls (256 spheres in a room: the one reference scene whose grid the build keeps by default, 16 x 4 x 16 cells) where the reference puts it and moved
    off the origin (inside +-50: beyond, the CubeBox rule's boxes reach back to +-50, primitive.cpp:35-60, and the launch is the canonical walk's):
    grid == both trees == canonical, bit for bit, in every mode."""
    W, H = 240, 136
    for shift in ((0.0, 0.0, 0.0), (30.0, -20.0, 25.0)):
        sc, ctx = _shifted_upload(capi, oracle, "balls", W, H, shift, (0.0, 0.0, 14.0))
        for n, path, amb in ((4, True, False), (2, False, False), (5, False, True)):
            monkeypatch.delenv("RTGO_TREE", raising=False)
            canon, cimg = gpu_render(capi, ctx, W, H, n, 1, path, ambient=amb, stats=True, prev=np.full((H, W, 4), 0.5, np.float32))
            for pin in ("2", "0", "1"):
                monkeypatch.setenv("RTGO_TREE", pin)
                acc, img = gpu_render(capi, ctx, W, H, n, 1, path, ambient=amb, prev=np.full((H, W, 4), 0.5, np.float32))
                st = ctx.stats()
                assert np.array_equal(acc.view(np.uint32), canon.view(np.uint32)) and np.array_equal(img, cimg), (shift, n, path, pin)
                assert bool(st["last_variant"] & 16) == (pin == "2"), st
        monkeypatch.delenv("RTGO_TREE", raising=False)
        ctx.close()


@pytest.mark.gpu
def test_the_trial_settles_for_a_caller_that_never_synchronises(capi, oracle, monkeypatch):
    """a job enqueued without a single rtgo_sync (bench.py's spin-up, a batch render) must not run to its end on the stand-in of an undecided
    trial (profiles/r03p: 90 of 100 launches of mirror_spheres 4K on the slowest candidate): the launch after the 2 x candidates trial launches
    waits for their times, so launch 2 x candidates + 1 already runs what every later launch runs.  rtgo_stats.last_variant / launches_trial."""
    monkeypatch.delenv("RTGO_TREE", raising=False)
    monkeypatch.delenv("RTGO_STREAM", raising=False)
    W, H = 480, 270
    sc, t, ctx = upload(capi, oracle, "plateau", W, H)
    ctx.resize(W * H)
    fr = capi.make_frame(W, H, 5, 1, True, False, None, (4, 1, 0))          # 25 spp: two passes, so both loops are candidates
    ctx.reset_stats()
    for k in range(9):
        ctx.launch(fr)
    st = ctx.stats()
    assert st["launches"] == 9 and st["launches_canonical"] == 0
    assert st["launches_trial"] in (4, 8), st                             # (4: the two structures came out identical, loops only)
    n_trial = st["launches_trial"]
    if n_trial == 4:
        for k in range(4):
            ctx.launch(fr)
    decided = ctx.stats()["last_variant"]
    assert decided & 8 == 0, "launch 2 x candidates + 1 must be a decided one"
    for k in range(6):
        ctx.launch(fr)
        assert ctx.stats()["last_variant"] == decided
    assert ctx.stats()["launches_trial"] == n_trial
    ctx.close()
