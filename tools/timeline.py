"""developer tool: per-wave timeline of one launch.  Needs the -DRTGO_TIMELINE build:
   hipcc ... -DRTGO_TIMELINE -o tools/_diag/librtgo_hip_timeline.so raytracingo_amd/csrc/rtgo_capi.hip
   RTGO_HIP_LIB=tools/_diag/librtgo_hip_timeline.so BANDS=8,0 python tools/timeline.py [scene] [W] [H] [N] [mode]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("RTGO_HIP_LIB", os.path.join(ROOT, "tools/_diag/librtgo_hip_timeline.so"))
import numpy as np
from raytracingo_amd import capi, scene as hscene

name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4
path = (sys.argv[5] if len(sys.argv) > 5 else "path") == "path"
G, g = [int(x) for x in os.environ.get("BANDS", "1,0").split(",")]
t = hscene.tables(name, W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
for f in range(4):
    ctx.reset_stats()
    ctx.launch(capi.make_frame(W, H, N, f, path, bands=(4, G, g))); ctx.sync()
st = ctx.stats()
lib = ctx._lib
buf = np.zeros((16384, 16), dtype=np.uint64)
lib.rtgo_debug_timeline.restype = C.c_int
lib.rtgo_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
n = lib.rtgo_debug_timeline(ctx._h, buf.ctypes.data, buf.nbytes)
r = buf[:n].astype(np.int64)
us = lambda x: x / 100.0   # wall_clock64: 100 MHz
t00 = r[:, 0].min()
start, staged, first, end = [us(r[:, i] - t00) for i in range(4)]
units, hot, iters, lanes = r[:, 4] & 0xFFFFFFFF, r[:, 4] >> 32, r[:, 5], r[:, 6]
qwait, cold = us(r[:, 7] & 0xFFFFFFFF), us(r[:, 7] >> 32) + start
pc = lambda a: " ".join("%7.1f" % np.percentile(a, q) for q in (0, 10, 50, 90, 99, 100))
print("%s %dx%d N=%d %s share 1/%d: HIP-event %.1f us; %d waves; span first start -> last end %.1f us" %
      (name, W, H, N, "path" if path else "dist", G, st["last_launch_ms"] * 1e3, n, end.max()))
print("percentiles                 min     p10     p50     p90     p99     max")
print("wave start        [us]  " + pc(start))
print("scene staged      [us]  " + pc(staged))
print("first unit taken  [us]  " + pc(np.where(units > 0, first, np.nan)[units > 0] if (units > 0).any() else start))
print("first pull known  [us]  " + pc(us(r[:, 8] - t00)))
print("first seeds hashed [us] " + pc(us(r[:, 9] - t00)))
print("first ray iter done [us]" + pc(us(r[:, 10] - t00)[r[:, 10] > 0]))
print("first unit done   [us]  " + pc(us(r[:, 11] - t00)[r[:, 11] > 0]))
print("wave end          [us]  " + pc(end))
print("units per wave          " + pc(units))
print("hot strips per wave     " + pc(hot))
print("first cold strip  [us]  " + pc(cold[cold > start]) if (cold > start).any() else "no cold strips")
print("queue wait / wave [us]  " + pc(qwait))
print("ray iterations per wave " + pc(iters))
print("waves with no unit: %d; total units %d; iterations/unit %.2f; live lanes/iteration %.1f of 64" %
      ((units == 0).sum(), units.sum(), iters.sum() / max(units.sum(), 1), lanes.sum() / max(iters.sum(), 1)))
big, tree, loop = r[:, 12].sum(), r[:, 13].sum(), r[:, 14].sum()
if loop > 0:
    print("ray-loop time: up-front list %.1f %%, tree walk %.1f %%, shading + bookkeeping %.1f %%; per iteration %.2f us" % (100.0 * big / loop, 100.0 * tree / loop, 100.0 * (loop - big - tree) / loop, us(loop) / max(iters.sum(), 1)))
busy = us(r[:, 3] - r[:, 2])[units > 0]
print("busy time first unit -> end per wave [us] " + pc(busy) + "   sum/(waves*span) = %.2f" % (busy.sum() / (n * end.max())))
# how much of the span is tail: time after which fewer than half of the waves are still running
e = np.sort(end)
print("50 %% of waves done at %.1f us, 90 %% at %.1f us, last at %.1f us" % (e[n // 2], e[int(n * 0.9)], e[-1]))
