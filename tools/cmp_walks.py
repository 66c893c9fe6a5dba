"""developer tool (needs the -DRTGO_CMPWALK build: tools/_diag/librtgo_hip_cmpwalk.so): full-size frames with the instrumented
kernel, which in that build also runs the fast walk on every ray and records the rays on which the two walks disagree.
   python tools/cmp_walks.py [W] [H] [N] [frames]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("RTGO_HIP_LIB", os.path.join(ROOT, "tools/_diag/librtgo_hip_cmpwalk.so"))
import numpy as np
from raytracingo_amd import capi, scene as hscene
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
F = int(sys.argv[4]) if len(sys.argv) > 4 else 3
total_rays = 0; total_bad = 0
for name in ["cornell", "slide", "mirror_spheres", "plateau", "window", "checkered", "balls", "soft_mirrors"]:
    t = hscene.tables(name, W, H)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    lib = ctx._lib
    lib.rtgo_debug_cmpwalk.restype = C.c_int; lib.rtgo_debug_cmpwalk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    buf = np.zeros((256, 16), np.float32)
    for mode in ("path", "distributed", "ambient"):
        ctx.reset_stats(); lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)
        for f in range(F):
            ctx.launch(capi.make_frame(W, H, N, f, mode == "path", mode == "ambient", stats=True))
        ctx.sync()
        st = ctx.stats()
        rays = st["rays_total"]
        lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)
        bad = int(buf[0].view(np.uint32)[0])
        total_rays += rays; total_bad += bad
        print("%-14s %-11s %12d rays, %d on which the walks disagree%s" % (name, mode, rays, bad, " (fast walk: the uniform grid)" if st["last_variant"] & 16 else ""), flush=True)
        for r in buf[1:1 + min(bad, 4)]:
            print("    o", r[0:3], "d", r[3:6], "tmin", r[6], "tmax", r[7], "canonical (t, prim)", r[8], int(r[9]), "fast", r[10], int(r[11]), "depth", int(r[12]), "phase", int(r[13]))
    ctx.close()
print("total: %d rays, %d disagreements" % (total_rays, total_bad))
