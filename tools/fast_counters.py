"""developer tool: per-ray work of the FAST walk (needs a -DRTGO_FAST_COUNTERS build pointed to by RTGO_HIP_LIB; RTGO_TREE pins the structure).
-DRTGO_FAST_COUNTERS=1: per lane -- boxes tested (tree) / cells stepped (grid) and primitive tests per ray, the up-front list included.
-DRTGO_FAST_COUNTERS=2: per WAVE -- node steps / cell steps and leaf phases a wave executes, whatever the number of its lanes that are still in them
   (printed per 64 rays; the ratio to the per-lane figures is how much of a wave's walk runs for its longest ray alone).
   python tools/fast_counters.py [scene ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracingo_amd import capi, scene as hscene
for name in sys.argv[1:] or ["cornell", "balls", "checkered"]:
    W, H, N = 960, 540, 4
    t = hscene.tables(name, W, H)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    ctx.reset_stats(); ctx.launch(capi.make_frame(W, H, N, 0, True)); ctx.sync()
    st = ctx.stats()
    r = st["rays_total"] - st["rays_culled"]
    print("%-14s %.2f boxes or cells / ray, %.2f tests or leaf phases / ray; x 64: %.1f, %.1f  (variant %d)" %
          (name, st["dbg_fast_boxes"] / r, st["dbg_fast_tests"] / r, 64.0 * st["dbg_fast_boxes"] / r, 64.0 * st["dbg_fast_tests"] / r, st["last_variant"]))
    ctx.close()
