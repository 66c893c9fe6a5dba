"""The oracle AND the HIP path against the only outputs the reference itself holds: its renders (/root/reference/images/*.jpg, decoded
into tests/golden/ref_images/ by make_fixtures.py; nothing here reads /root/reference).

What is compared, and what it can resolve (measured by tests/golden/ref_images/derive_floors.py, recorded in ref_images.json "floors"):
  reference.jpg          cornell, path mode, eye (0, 0, 16.2).  With cornellWhite's kd at 0.9 -- the picture's revision; HEAD's materials.h
                         says 0.8 -- a converged render agrees with the picture to 0.46 / 255 in mean absolute difference of 20 x 20
                         block means over the whole frame (Monte-Carlo floor of two independent 256-spp renders 0.36, JPEG round trip
                         0.16), image means within 0.5 %, seven named regions within 2.7 %, silhouette edges to the pixel.
                         With HEAD's 0.8 the direct-lit walls stay within 4 % and white, indirectly lit surfaces come out 13-22 % darker,
                         as a darker white must.  This pins, against the reference's own output: raygen and the pinhole frame, the
                         rectangle intersector and its one-sidedness (black surround: the front wall is invisible from outside), both
                         boxes' model matrices, the emissive test and Le = 15, the estimator of kernel.cu:456-473 (uniform hemisphere,
                         weight dot(N, Ra) kd, no pdf, no 1/pi), make_color WITHOUT gamma, accumulation.
  reference_mirror.jpg   same shell, older contents (mirror panel, other boxes), Scene::SetupCamera's own eye: shell regions within 4 %
                         of HEAD's constants.
  distributed_rt.jpg     distributed mode, older shader (no kr term on the walls, brighter direct term): factor-of-two statements only --
                         lit, unshadowed surfaces are BRIGHT, which is SURVEY Q2 (an occlusion miss that wrote the background colour
                         would leave everything but the light black).
A screenshot of unknown frame count, JPEG coded, from a --use_fast_math build cannot be compared pixel by pixel; block means, region
means and silhouettes can, and the thresholds below are the measured values with the stated margins.
CPU tests render the oracle at 300 x 300 (seconds); the -m gpu tests render 600 x 600 at 1024-4096 spp through the C ABI.
"""
import numpy as np
import pytest

import refimg

WHITE_OF_THE_PICTURE = 0.9      # cornellWhite kd of reference.jpg's revision (fitted on the back wall; make_fixtures.py)


def _downsample(img, k):
    a = np.asarray(img, dtype=np.float64)[..., :3]
    h, w, _ = a.shape
    return a.reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))


def _check_surround_and_light(ours8_disp, regions, scale, what):
    for k, box in regions.items():
        if k.startswith("surround"):
            # one-sided rectangles (kernel.cu:372-416, d.y < 0 only): from outside the room its walls do not exist; miss = black background
            assert refimg.region_mean(ours8_disp, box, scale).max() == 0.0, (what, k)
        if k == "light":
            # Le = 15 > 0.01 ends the path with Le (kernel.cu:458-461) / (1,1,1) in distributed mode (:479-482); make_color clamps
            assert refimg.region_mean(ours8_disp, box, scale).min() == 255.0, (what, k)


def _ratios(ours_disp_f255, ref, regions, scale):
    out = {}
    for k, box in regions.items():
        if k.startswith("surround") or k == "light":
            continue
        r, m = refimg.region_mean(ref, box), refimg.region_mean(ours_disp_f255, box, scale)
        out[k] = [float(m[c] / r[c]) for c in range(3) if r[c] > 10.0]
    return out


def check_reference_jpg(acc, white, scale, block_mad_max, tol=0.04):
    """acc: float accumulation buffer [S, S, 3] (bottom row first) of cornell / path / eye z = 16.2 with cornellWhite = `white`"""
    meta, arr = refimg.fixtures()
    ref, regions = arr["reference"], meta["regions"]["cornell_z162"]
    ours8 = refimg.display(refimg.to8(acc))
    oursf = refimg.display(acc) * 255.0
    _check_surround_and_light(ours8, regions, scale, "reference.jpg")
    ratios = _ratios(oursf, ref, regions, scale)
    refd = _downsample(ref, scale) if scale > 1 else ref.astype(np.float64)
    mad = refimg.block_mad(ours8, refd, 20 // scale)
    mean_rel = np.abs(ours8.reshape(-1, 3).mean(axis=0) / ref.reshape(-1, 3).mean(axis=0) - 1.0).max()
    print("reference.jpg, white %.1f: block MAD %.3f, image mean off by %.4f, ratios %s" % (white, mad, mean_rel, {k: [round(v, 3) for v in r] for k, r in ratios.items()}))
    if white == WHITE_OF_THE_PICTURE:
        for k, r in ratios.items():
            assert all(abs(v - 1.0) <= tol for v in r), (k, r)          # measured <= 0.027 at convergence (derive_floors.py)
        assert mean_rel <= 0.02, mean_rel                                 # measured 0.005
        assert mad <= block_mad_max, mad
    else:
        # HEAD's white (0.8): surfaces whose light is mostly direct keep the picture's level ...
        assert all(abs(v - 1.0) <= 0.06 for v in ratios["left_wall"] + ratios["right_wall"]), ratios      # measured 0.960 / 0.962
        # ... white ones lit through other white ones are darker by about one (0.8 / 0.9) per white bounce: between 0.75 and 0.92
        for k in ("back_wall", "floor_front", "ceiling", "tall_box_front", "short_box_top"):
            assert all(0.75 <= v <= 0.92 for v in ratios[k]), (k, ratios[k])
        # and no gamma (kernel.cu:90-98): a 1/2.2 curve would put the back wall at 1.27 x the picture's level, not below it
    return ratios, mad


def check_shell(acc, name, scale):
    meta, arr = refimg.fixtures()
    ref, regions = arr[name], meta["regions"]["shell_z14"]
    ours8 = refimg.display(refimg.to8(acc))
    _check_surround_and_light(ours8, regions, scale, name)
    ratios = _ratios(refimg.display(acc) * 255.0, ref, regions, scale)
    print(name, {k: [round(v, 3) for v in r] for k, r in ratios.items()})
    return ratios


def check_edges(img8_disp_600, name):
    meta, _ = refimg.fixtures()
    want = meta["images"][name]["room_edges_rows_280_320"]
    got = refimg.room_edges(img8_disp_600)
    assert all(abs(a - b) <= 1 for a, b in zip(got, want)), (name, got, want)


# ------------------------------------------------------------------ CPU: the oracle ----------------------------------------------------------

def test_fixture_provenance():
    meta, arr = refimg.fixtures()
    assert set(arr) == {"reference", "reference_mirror", "distributed_rt"}
    for name, a in arr.items():
        m = meta["images"][name]
        assert a.shape == (600, 600, 3) and a.dtype == np.uint8 and len(m["sha256"]) == 64
        assert m["jpeg"]["quantization_all_ones"]                       # quality 100: what is lost is chroma resolution (4:2:0) and rounding
        # the pictures themselves: left wall red, right wall blue, black surround, a saturated light
        reg = meta["regions"]["cornell_z162" if name == "reference" else "shell_z14"]
        lw, rw = refimg.region_mean(a, reg["left_wall"]), refimg.region_mean(a, reg["right_wall"])
        assert lw[0] > 80 and lw[1] < 6 and lw[2] < 6 and rw[2] > 80 and rw[0] < 6 and rw[1] < 6
        assert refimg.region_mean(a, reg["light"]).min() > 254.0
        assert max(refimg.region_mean(a, reg[k]).max() for k in reg if k.startswith("surround")) < 1.0


@pytest.mark.parametrize("white", [WHITE_OF_THE_PICTURE, None])
def test_oracle_against_reference_jpg(oracle, white):
    acc = refimg.oracle_render(oracle, "reference", 300, 4, 8, white=white)          # 128 spp
    # (128 spp at 300 x 300: the Monte-Carlo floor of the block statistic is ~1, the smallest region -- 245 pixels of ceiling -- is good to ~5 %)
    check_reference_jpg(acc, white if white is not None else 0.8, 2, block_mad_max=2.0, tol=0.09)


def test_oracle_silhouettes(oracle):
    for name in ("reference", "reference_mirror", "distributed_rt"):
        acc = refimg.oracle_render(oracle, name, 600, 2, 1)
        check_edges(refimg.display(refimg.to8(acc)), name)


def test_oracle_against_reference_mirror_shell(oracle):
    ratios = check_shell(refimg.oracle_render(oracle, "reference_mirror", 300, 4, 4), "reference_mirror", 2)
    for k, r in ratios.items():
        assert all(abs(v - 1.0) <= 0.06 for v in r), (k, r)            # measured 0.960 .. 1.037 at convergence


def test_oracle_distributed_is_lit(oracle):
    """the occlusion miss keeps the payload's (1,1,1) (kernel.cu:74,498 + renderer.cpp:383-397)"""
    ratios = check_shell(refimg.oracle_render(oracle, "distributed_rt", 300, 2, 2), "distributed_rt", 2)
    for k, r in ratios.items():
        assert all(0.45 <= v <= 1.4 for v in r), (k, r)                 # measured 0.49 .. 1.30 (older shader in the picture)


# ------------------------------------------------------------------ GPU: the HIP path -------------------------------------------------------

@pytest.fixture(scope="module")
def gpu():
    from raytracingo_amd import capi, scene as hscene
    capi.load()
    return capi, hscene


@pytest.mark.gpu
@pytest.mark.parametrize("white", [WHITE_OF_THE_PICTURE, None])
def test_gpu_against_reference_jpg(gpu, oracle, white):
    capi, hscene = gpu
    acc, st = refimg.gpu_render(capi, hscene, oracle, "reference", 600, 8, 64, white=white)      # 4096 spp
    assert st["launches_canonical"] == 0
    # whole frame, 20 x 20 blocks: measured 0.46 with the oracle at 512 spp (floors: noise 0.36, JPEG 0.16); twice that is the bar.
    # Regions: 0.05 -- the smallest one (the short box's top: 560 pixels under the light, a heavy-tailed estimator: 512-spp renders
    # with 1, 4, 16, 64 samples per frame put it at 1.040, 1.024, 1.023, 1.041 of the picture) sets it; the others are within 0.03
    check_reference_jpg(acc, white if white is not None else 0.8, 1, block_mad_max=1.0, tol=0.05)
    check_edges(refimg.display(refimg.to8(acc)), "reference")
    if white == WHITE_OF_THE_PICTURE:
        _, arr = refimg.fixtures()
        assert refimg.luminance_correlation(refimg.display(refimg.to8(acc)), arr["reference"], [100, 100, 500, 500]) >= 0.96


@pytest.mark.gpu
def test_gpu_against_reference_mirror_shell(gpu, oracle):
    capi, hscene = gpu
    acc, _ = refimg.gpu_render(capi, hscene, oracle, "reference_mirror", 600, 8, 16)
    ratios = check_shell(acc, "reference_mirror", 1)
    for k, r in ratios.items():
        assert all(abs(v - 1.0) <= 0.05 for v in r), (k, r)
    check_edges(refimg.display(refimg.to8(acc)), "reference_mirror")


@pytest.mark.gpu
def test_gpu_distributed_is_lit(gpu, oracle):
    capi, hscene = gpu
    acc, _ = refimg.gpu_render(capi, hscene, oracle, "distributed_rt", 600, 4, 8)
    ratios = check_shell(acc, "distributed_rt", 1)
    for k, r in ratios.items():
        assert all(0.45 <= v <= 1.4 for v in r), (k, r)
    check_edges(refimg.display(refimg.to8(acc)), "distributed_rt")


@pytest.mark.gpu
def test_gpu_converged_equals_oracle_converged(gpu, oracle):
    """the two renderers the pictures are held against agree with each other far below what the pictures resolve"""
    capi, hscene = gpu
    g, _ = refimg.gpu_render(capi, hscene, oracle, "reference", 300, 4, 4, white=WHITE_OF_THE_PICTURE)
    o = refimg.oracle_render(oracle, "reference", 300, 4, 4, white=WHITE_OF_THE_PICTURE)
    # same seeds (frames 0..3 of 16 spp): the accumulated frames differ only by the path flips of tests/parity.py
    # (oracle_render averages lone frames; the device accumulates by running mean: equal up to float rounding of the mean)
    assert np.abs(g - o).mean() < 2e-4 and float((np.abs(g - o) <= 1e-3).mean()) > 0.99
