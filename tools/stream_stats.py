"""developer tool: census of the streaming loop (needs a -DRTGO_STREAM_STATS build pointed to by RTGO_HIP_LIB): python tools/stream_stats.py scene W H N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["RTGO_STREAM"] = "1"
from raytracingo_amd import capi, scene as hscene
name, W, H, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
t = hscene.tables(name, W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
for f in range(2):
    ctx.launch(capi.make_frame(W, H, N, f, True)); ctx.sync()
ctx.reset_stats()
ctx.launch(capi.make_frame(W, H, N, 2, True)); ctx.sync()
st = ctx.stats()
it = max(st["node_visits"], 1)
print("%s %dx%d N=%d: %.3f ms; traversed rays %.1f M; loop iterations %.2f M; lanes with a ray per iteration %.1f; iterations with a hit %.2f M, %.1f hits each; iterations stalled on the ring %.1f %%" %
      (name, W, H, N, st["last_launch_ms"], (st["rays_total"] - st["rays_culled"]) / 1e6, it / 1e6, st["prim_tests"] / it, st["hits"] / 1e6,
       st["dbg_fast_boxes"] / max(st["hits"], 1), 100.0 * st["dbg_fast_tests"] / it))
