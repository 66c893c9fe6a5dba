"""developer tool: per-ray work of the FAST walk (needs a -DRTGO_FAST_COUNTERS build pointed to by RTGO_HIP_LIB)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from raytracingo_amd import capi, scene as hscene
for name in sys.argv[1:] or ["cornell", "balls", "checkered"]:
    W, H, N = 960, 540, 2
    t = hscene.tables(name, W, H)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    ctx.reset_stats(); ctx.launch(capi.make_frame(W, H, N, 0, True)); ctx.sync()
    # raw counters: read the 8 u64 through a private peek (diagnostic): re-use get_stats fields + hipMemcpy not exposed -> use stats of a stats launch for canonical
    st = ctx.stats()
    r = st["rays_total"]
    print("%-14s fast walk: %.2f boxes/ray, %.2f leaf tests/ray (LBVH depth %d)" % (name, st["dbg_fast_boxes"] / r, st["dbg_fast_tests"] / r, st["lbvh_depth"]))
