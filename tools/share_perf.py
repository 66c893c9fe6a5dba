"""developer tool: one GPU's share of a G-way band split as ONE launch per frame, steady state (the launch-time trial has settled):
   BANDS=G,g python tools/share_perf.py scene W H N mode   ->   ms per launch (HIP events), rays per launch
The multi-GPU projections of DESIGN.md section 6 divide the whole frame's time by these."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracingo_amd import capi, scene as hscene
name = sys.argv[1]; W = int(sys.argv[2]); H = int(sys.argv[3]); N = int(sys.argv[4]); path = (sys.argv[5] if len(sys.argv) > 5 else "path") == "path"
G, g = [int(x) for x in os.environ.get("BANDS", "1,0").split(",")]
t = hscene.tables(name, W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"])
rows = capi.local_rows(H, 4, G, g)
ctx.resize(rows * W)
f = 0
for k in range(16):                      # (2 x up to 6 candidates of the trial, and the clocks)
    ctx.launch(capi.make_frame(W, H, N, f, path, bands=(4, G, g))); f += 1
ctx.sync(); ctx.reset_stats()
K = 30
for k in range(K):
    ctx.launch(capi.make_frame(W, H, N, f, path, bands=(4, G, g))); f += 1
ctx.sync()
st = ctx.stats()
print("%s %dx%d N=%d %s share 1/%d: %.4f ms per launch, %.1f Mrays per launch, variant %d, trial launches %d" %
      (name, W, H, N, "path" if path else "dist", G, st["total_launch_ms"] / K, st["rays_total"] / K / 1e6, st["last_variant"], st["launches_trial"]))
