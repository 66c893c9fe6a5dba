"""developer tool: host time of one rtgo_launch (enqueue only: the loop never waits for the GPU until the end), on a frame small enough that
the GPU keeps up.   python tools/host_cost.py [launches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from raytracingo_amd import capi, scene as hscene
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60      # <= 64: the context harvests its event ring (and waits for the GPU) beyond that
W, H, N = 64, 16, 1
for name in ("cornell", "balls", "plateau"):
    t = hscene.tables(name, W, H)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    fr = [capi.make_frame(W, H, N, f, True) for f in range(K)]
    for f in range(50):
        ctx.launch(fr[f])
    ctx.sync()
    best = 1e9
    for rep in range(20):
        t0 = time.perf_counter()
        for f in range(K):
            ctx.launch(fr[f])
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        best = min(best, (t1 - t0) / K)
    print("%-8s %d launches of a %dx%d frame: %.2f us of host time per rtgo_launch through ctypes (%.2f us per launch until the GPU has drained)" %
          (name, K, W, H, best * 1e6, (t2 - t0) / K * 1e6))
    ctx.close()
