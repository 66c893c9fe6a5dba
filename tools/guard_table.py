"""developer tool: the far-field guard's two quantities (rtgo_stats.guard_reach / guard_quadric) for the reference's eight scenes under
Scene::SetupCamera's camera, and which walk a product launch takes.   python tools/guard_table.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracingo_amd import capi, scene as hscene
W, H = 320, 180
for name in hscene.SCENES:
    t = hscene.tables(name, W, H)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    ctx.reset_stats()
    ctx.launch(capi.make_frame(W, H, 2, 0, True)); ctx.sync()
    st = ctx.stats()
    print("%-15s %3d primitives  reach %7.2f  quadric %9.1f  -> %s walk" % (name, len(t["type"]), st["guard_reach"], st["guard_quadric"], "canonical" if st["launches_canonical"] else "fast"))
    ctx.close()
