"""developer tool: tools/cmp_walks.py for ONE scene and mode (needs the -DRTGO_CMPWALK build): python tools/cmp_one.py scene mode W H N"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("RTGO_HIP_LIB", os.path.join(ROOT, "tools/_diag/librtgo_hip_cmpwalk.so"))
import numpy as np
from raytracingo_amd import capi, scene as hscene
name, mode = sys.argv[1], sys.argv[2]
W, H, N = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
t = hscene.tables(name, W, H)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
lib = ctx._lib
lib.rtgo_debug_cmpwalk.restype = C.c_int; lib.rtgo_debug_cmpwalk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
buf = np.zeros((256, 16), np.float32)
print(name, mode, "launching", flush=True)
ctx.launch(capi.make_frame(W, H, N, 0, mode == "path", mode == "ambient", stats=True)); ctx.sync()
lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)
print(name, mode, ctx.stats()["rays_total"], "rays,", int(buf[0].view(np.uint32)[0]), "disagreements", flush=True)
for r in buf[1:1 + min(int(buf[0].view(np.uint32)[0]), 6)]:
    print("    o", r[0:3], "d", r[3:6], "tmin", r[6], "tmax", r[7], "canonical (t, prim)", r[8], int(r[9]), "fast", r[10], int(r[11]), "depth", int(r[12]), "phase", int(r[13]), "guard", r[14], r[15], flush=True)
