"""developer tool: kernel time of very small launches (fixed per-launch cost of the megakernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracingo_amd import capi, scene as hscene
for (W, H, N) in [(16, 16, 1), (64, 64, 1), (256, 256, 1), (640, 360, 1), (640, 360, 4), (1920, 1080, 1)]:
    t = hscene.tables("cornell", W, H)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    for f in range(3):
        ctx.launch(capi.make_frame(W, H, N, f, True)); ctx.sync()
    ctx.reset_stats()
    for f in range(10):
        ctx.launch(capi.make_frame(W, H, N, 3 + f, True)); ctx.sync()
    st = ctx.stats()
    print("%4dx%-4d N=%d: %.4f ms/launch, %d rays/launch" % (W, H, N, st["total_launch_ms"] / 10, st["rays_total"] // 10))
    ctx.close()
