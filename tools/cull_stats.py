"""developer tool: how many primary rays the background culling answers (scene rectangle, per-strip mask) and the frame time"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracingo_amd import capi, scene as hscene
name, W, H, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
t = hscene.tables(name, W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
for f in range(3):
    ctx.launch(capi.make_frame(W, H, N, f, True)); ctx.sync()
ctx.reset_stats()
for f in range(3, 8):
    ctx.launch(capi.make_frame(W, H, N, f, True))
ctx.sync()
st = ctx.stats()
print("%s %dx%d N=%d: %.3f ms/frame, rays/frame %.1f M, culled %.1f M (%.1f %% of the primary rays), traversed %.1f M" %
      (name, W, H, N, st["total_launch_ms"] / 5, st["rays_total"] / 5e6, st["rays_culled"] / 5e6, 100.0 * st["rays_culled"] / 5 / (W * H * N * N),
       (st["rays_total"] - st["rays_culled"]) / 5e6))
