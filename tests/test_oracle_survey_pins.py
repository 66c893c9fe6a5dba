"""Known answers SURVEY.md recorded from a run of the reference's own device code (see tests/golden/survey_pins.json for
provenance): primitive counts, ray counts by depth, BASELINE config 1 (cornell 256x256 distributed) image statistics."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pins():
    with open(os.path.join(GOLD, "survey_pins.json")) as f:
        return json.load(f)


def test_primitive_counts(oracle, pins):
    for name, n in pins["primitive_counts"].items():
        assert oracle.scene(name, 256, 256).n_prims == n


def test_first_ball(oracle, pins):
    t = oracle.scene_tables(oracle.scene("balls", 960, 540))
    pos = t["M"][1][[3, 7, 11]]  # primitive 0 is the floor
    assert np.allclose(pos, pins["first_ball_left_to_right"]["pos"], atol=1e-4)
    assert np.allclose(t["mat"][1][:3], [0.16, 0.83, 0.18])  # platePrettyGreen


def test_c1_cornell_256_distributed(oracle, pins):
    """BASELINE.json configs[0]: the reference's own CPU-runnable case. Exact ray counts, image statistics to 6 digits."""
    p = pins["c1_cornell_256_distributed"]
    sc = oracle.scene("cornell", 256, 256)
    for mode in (0, 1):
        acc, img, c = oracle.render(sc, oracle.frame(256, 256, 1, 0, path=False, mode=mode))
        assert c["rays_total"] == p["total"]
        assert c["rays_radiance"][:6] == p["by_depth"]
        assert c["rays_occlusion"] == p["occlusion"]
        assert np.allclose(acc[..., :3].reshape(-1, 3).astype(np.float64).mean(0), p["mean_accum"], atol=2e-6)
        assert np.allclose(acc[128, 128, :3], p["centre_accum"], atol=1e-6)
        assert img[128, 128, :3].tolist() == p["centre_image"]


def test_cornell_256_path(oracle, pins):
    p = pins["cornell_256_path"]
    acc, img, c = oracle.render(oracle.scene("cornell", 256, 256), oracle.frame(256, 256, 1, 0, path=True, mode=1))
    assert c["rays_total"] == p["total"]
    assert np.allclose(acc[..., :3].reshape(-1, 3).astype(np.float64).mean(0), p["mean_accum"], atol=2e-6)


def test_full_size_ray_counts_exact(oracle, pins):
    """1080p / 960x540 path-mode ray counts by depth, reproduced EXACTLY when sin/cos take the host-build's double route;
    the canonical float route (CUDA device semantics) differs by a handful of paths out of millions."""
    for p in pins["ray_counts_path_1spp"]:
        sc = oracle.scene(p["scene"], p["W"], p["H"])
        _, _, c = oracle.render(sc, oracle.frame(p["W"], p["H"], 1, 0, path=True, mode=1, host_double_trig=True))
        assert c["rays_total"] == p["total"], p["scene"]
        assert c["rays_radiance"][:6] == p["by_depth"], p["scene"]
        _, _, c = oracle.render(sc, oracle.frame(p["W"], p["H"], 1, 0, path=True, mode=1))
        assert abs(c["rays_total"] - p["total"]) <= 1e-5 * p["total"], p["scene"]
