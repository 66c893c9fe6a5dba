"""developer tool: where the STREAMING loop (frames of more than 16 spp) spends its time.  Needs the -DRTGO_TIMELINE build (see timeline.py);
RTGO_STREAM=1 pins the streaming loop.   python tools/timeline_stream.py [scene] [W] [H] [N]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("RTGO_HIP_LIB", os.path.join(ROOT, "tools/_diag/librtgo_hip_timeline.so"))
os.environ.setdefault("RTGO_STREAM", "1")
import numpy as np
from raytracingo_amd import capi, scene as hscene
name = sys.argv[1] if len(sys.argv) > 1 else "plateau"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
N = int(sys.argv[4]) if len(sys.argv) > 4 else 16
t = hscene.tables(name, W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
for f in range(3):
    ctx.reset_stats()
    ctx.launch(capi.make_frame(W, H, N, f, True)); ctx.sync()
st = ctx.stats()
lib = ctx._lib
buf = np.zeros((16384, 16), dtype=np.uint64)
lib.rtgo_debug_timeline.restype = C.c_int; lib.rtgo_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
n = lib.rtgo_debug_timeline(ctx._h, buf.ctypes.data, buf.nbytes)
r = buf[:n].astype(np.float64)
busy = (r[:, 3] - r[:, 1]).sum()     # scene staged -> wave end
regen, trace, shade, lanes_trace, lanes_regen, regens, iters = r[:, 8].sum(), r[:, 9].sum(), r[:, 10].sum(), r[:, 11].sum(), r[:, 15].sum(), r[:, 14].sum(), r[:, 5].sum()
big, tree = r[:, 12].sum(), r[:, 13].sum()
print("%s %dx%d N=%d streaming loop: HIP-event %.2f ms; %d waves; %.0f iterations, %.1f lanes hold a ray per iteration; %.0f regenerations of %.1f lanes" %
      (name, W, H, N, st["last_launch_ms"], n, iters, lanes_trace / max(iters, 1), regens, lanes_regen / max(regens, 1)))
print("share of the waves' busy time: task regeneration %.1f %%, trace + shading + ring write %.1f %% (lane 0's view of the trace: up-front list %.1f %%, tree %.1f %%), fold %.1f %%, the rest (seeds, pixel writes, queue) %.1f %%" %
      (100 * regen / busy, 100 * trace / busy, 100 * big / busy, 100 * tree / busy, 100 * shade / busy, 100 * (busy - regen - trace - shade) / busy))
print("per iteration: %.2f us (regeneration %.2f, trace + shading %.2f, fold %.2f)" % (busy / iters / 100.0, regen / iters / 100.0, trace / iters / 100.0, shade / iters / 100.0))
