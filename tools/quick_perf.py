"""quick perf probe (developer tool): python tools/quick_perf.py [scene] [W] [H] [N] [mode]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracingo_amd import capi, scene as hscene

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4
path = (sys.argv[5] if len(sys.argv) > 5 else "path") == "path"
G, g = [int(x) for x in os.environ.get("BANDS", "1,0").split(",")]   # render only rank g's share of a G-way band split
t = hscene.tables(name, W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"])
ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"])
ctx.resize(W * H)
_mk = capi.make_frame
capi.make_frame = lambda *a, **k: _mk(*a, bands=(4, G, g), **k)
for f in range(3):
    ctx.launch(capi.make_frame(W, H, N, f, path)); ctx.sync()
ctx.reset_stats()
ctx.launch(capi.make_frame(W, H, N, 3, path, stats=True)); ctx.sync()
st = ctx.stats(); print("stats launch", st)
V, T, h = st["node_visits"] / st["rays_total"], st["prim_tests"] / st["rays_total"], st["hits"] / st["rays_total"]
A = 64 + 32 * V + 64 * T + 40 * h
ctx.reset_stats()
K = 10
for f in range(K):
    ctx.launch(capi.make_frame(W, H, N, 4 + f, path)); ctx.sync()
st = ctx.stats()
ms = st["total_launch_ms"] / K
rays = st["rays_total"] / K
print("%s %dx%d N=%d %s: %.3f ms/frame, %.1f Mray/s, V=%.2f T=%.2f h=%.3f A_ray=%.0f B -> %.1f GB/s algorithmic" %
      (name, W, H, N, "path" if path else "dist", ms, rays / ms / 1e3, V, T, h, A, (rays * A + W * H * 36) / ms / 1e6))
