"""developer tool: launch floor. A 16x16 corner window of the 1080p cornell frame (all rays miss) vs a 16x16 window in the box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracingo_amd import capi, scene as hscene
W, H = 1920, 1080
t = hscene.tables("cornell", W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(64 * 64)
for name, win, N in [("corner 16x16 N=1", (0, 0, 16, 16), 1), ("corner 64x64 N=4", (0, 0, 64, 64), 4), ("centre 16x16 N=1", (952, 532, 16, 16), 1),
                     ("centre 4x1 N=4 (one unit)", (958, 540, 4, 1), 4), ("centre 64x64 N=4", (928, 508, 64, 64), 4)]:
    for f in range(3):
        ctx.launch(capi.make_frame(W, H, N, f, True, window=win)); ctx.sync()
    ctx.reset_stats()
    for f in range(20):
        ctx.launch(capi.make_frame(W, H, N, 3 + f, True, window=win)); ctx.sync()
    st = ctx.stats()
    print("%-28s %.4f ms/launch, %d rays/launch" % (name, st["total_launch_ms"] / 20, st["rays_total"] // 20))
