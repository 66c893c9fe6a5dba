"""developer tool: a hash of the accumulation buffer after a few progressive frames of every scene x mode, one line each -- two builds of
the library (RTGO_HIP_LIB) give the same lines iff they give the same pixels.   python tools/frame_hashes.py [W] [H] [N] [frames]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from raytracingo_amd import capi, scene as hscene
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
F = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for name in ["cornell", "slide", "mirror_spheres", "plateau", "window", "checkered", "balls", "soft_mirrors"]:
    t = hscene.tables(name, W, H)
    for mode in ("path", "distributed", "ambient"):
        for stats in (False, True):
            ctx = capi.Context(0)
            ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
            ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
            for f in range(F):
                ctx.launch(capi.make_frame(W, H, N, f, mode == "path", mode == "ambient", stats=stats))
            ctx.sync()
            acc = ctx.read_accum(H, W)
            print("%-14s %-11s %-9s %dx%d N=%d x%d  %s  rays %d" % (name, mode, "canonical" if stats else "fast", W, H, N, F,
                  hashlib.sha1(np.ascontiguousarray(acc).tobytes()).hexdigest()[:16], ctx.stats()["rays_total"]), flush=True)
            ctx.close()

# the whitted triangle path on the procedural mesh of the tests (three residency modes)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import whitted_scene
mesh = whitted_scene.build(n_lat=40, n_lon=48)
eye, look, up, fov = np.array([0.5, 3.0, 7.0]), np.array([0.0, 1.0, 0.0]), np.array([0.0, 1.0, 0.0]), 45.0
Wv = look - eye
U = np.cross(Wv, up); U /= np.linalg.norm(U)
V = np.cross(U, Wv); V /= np.linalg.norm(V)
vlen = np.linalg.norm(Wv) * np.tan(0.5 * np.radians(fov))
V *= vlen; U *= vlen * W / H
for mode in ("2", "1", "0"):
    os.environ["RTGO_WHITTED_MODE"] = mode
    ctx = capi.Context(0)
    ctx.whitted_set_mesh(mesh["positions"], mesh["normals"], mesh["indices"], mesh["tri_material"], mesh["materials"])
    ctx.whitted_set_lights(mesh["lights"]); ctx.whitted_set_miss_color(mesh["miss"])
    ctx.set_camera(eye.astype(np.float32), U.astype(np.float32), V.astype(np.float32), Wv.astype(np.float32)); ctx.resize(W * H)
    for sf in range(F):
        ctx.whitted_launch(W, H, sf)
    ctx.sync()
    print("whitted mode %s %dx%d x%d  %s  rays %d" % (mode, W, H, F, hashlib.sha1(np.ascontiguousarray(ctx.read_accum(H, W)).tobytes()).hexdigest()[:16], ctx.stats()["rays_total"]), flush=True)
    ctx.close()
