"""developer tool: time the whitted triangle path (rtgo_whitted_launch) on the procedural scene of tests/whitted_scene.py:
   python tools/whitted_perf.py [W] [H] [n_lat] [n_lon]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import whitted_scene
from raytracingo_amd import capi

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
n_lat = int(sys.argv[3]) if len(sys.argv) > 3 else 40
n_lon = int(sys.argv[4]) if len(sys.argv) > 4 else 48
mesh = whitted_scene.build(n_lat=n_lat, n_lon=n_lon)
# pinhole frame like sutil::Camera::UVWFrame (float64 here: a timing tool, not a parity test)
eye, look, up, fov = np.array([0.5, 3.0, 7.0]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 0.0]), 45.0
Wv = look - eye
wlen = np.linalg.norm(Wv)
U = np.cross(Wv, up); U /= np.linalg.norm(U)
V = np.cross(U, Wv); V /= np.linalg.norm(V)
vlen = wlen * np.tan(0.5 * np.radians(fov))
V *= vlen
U *= vlen * W / H
ctx = capi.Context(0)
ctx.whitted_set_mesh(mesh["positions"], mesh["normals"], mesh["indices"], mesh["tri_material"], mesh["materials"])
ctx.whitted_set_lights(mesh["lights"])
ctx.whitted_set_miss_color(mesh["miss"])
ctx.set_camera(eye.astype(np.float32), U.astype(np.float32), V.astype(np.float32), Wv.astype(np.float32))
ctx.resize(W * H)
for sf in range(3):
    ctx.whitted_launch(W, H, sf)
ctx.sync()
ctx.reset_stats()
K = 20
for sf in range(K):
    ctx.whitted_launch(W, H, 3 + sf)
ctx.sync()
st = ctx.stats()
ms = st["total_launch_ms"] / K
print("whitted %dx%d, %d triangles, %d lights: %.3f ms/subframe, %.1f Mray/s (%.2f rays per pixel, %.0f %% occlusion rays)" %
      (W, H, len(mesh["indices"]), len(mesh["lights"]), ms, st["rays_total"] / K / ms / 1e3, st["rays_total"] / K / (W * H),
       100.0 * st["rays_occlusion"] / max(st["rays_total"], 1)))
if st["node_visits"]:
    waves = min((((W + 7) // 8) * ((H + 7) // 8) + 15) // 16, 256) * 16
    print("  diagnostic build: per wave and launch: alive %.1f us, inside tiles %.1f us, %.1f tiles" %
          (st["node_visits"] / K / waves / 100.0, st["hits"] / K / waves / 100.0, st["dbg_fast_boxes"] / K / waves))
    print("  longest tile of the %d launches: %.1f us" % (K, st["dbg_fast_tests"] / 100.0))
    ticks = ctx.read_accum(H, W)[::8, ::8, 3] / 100.0       # one value per tile, us
    print("  tile duration [us]: p50 %.1f p90 %.1f p99 %.1f max %.1f; tiles over 100 us: %d of %d" %
          (np.percentile(ticks, 50), np.percentile(ticks, 90), np.percentile(ticks, 99), ticks.max(), (ticks > 100).sum(), ticks.size))
    th, tw = ticks.shape
    print("  map of tile durations (rows top to bottom; . < 25 us, - < 50, + < 100, * < 200, # more):")
    for yy in range(th - 1, -1, -max(1, th // 34)):
        print("    " + "".join(".-+*#"[int(np.searchsorted([25, 50, 100, 200], ticks[yy, xx:xx + max(1, tw // 80)].max()))] for xx in range(0, tw, max(1, tw // 80))))
