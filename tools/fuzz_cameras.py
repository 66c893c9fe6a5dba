"""developer tool: random cameras, timed kernel (culling, fast walk) vs instrumented kernel (traces everything), bitwise.
   python tools/fuzz_cameras.py [scene] [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
from raytracingo_amd import capi
O.build(); O.lib()
name = sys.argv[1] if len(sys.argv) > 1 else "slide"
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, n = 144, 80, 2
sc = O.scene(name, W, H); t = O.scene_tables(sc)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
bb = np.asarray(t["aabb"], dtype=np.float64).reshape(-1, 6)
lo, hi = bb[:, :3].min(axis=0), bb[:, 3:].max(axis=0)
centre, size = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
bad = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(seed)
    for trial in range(14):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        eye = O.f32(centre + d * size * rng.uniform(0.8, 6.0))
        look = O.f32(centre + rng.normal(size=3) * size * rng.choice([0.05, 0.4, 1.5]))
        up = O.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
        U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
        fov = float(rng.uniform(15.0, 100.0))
        O.lib().oracle_camera_uvw(O.fptr(eye), O.fptr(look), O.fptr(up), fov, np.float32(np.float32(W) / np.float32(H)), O.fptr(U), O.fptr(V), O.fptr(Wv))
        ctx.set_camera(eye, U, V, Wv)
        for path in (True, False):
            ctx.launch(capi.make_frame(W, H, n, 0, path)); ctx.sync(); fast = ctx.read_accum(H, W).copy()
            ctx.launch(capi.make_frame(W, H, n, 0, path, stats=True)); ctx.sync(); canon = ctx.read_accum(H, W)
            if not np.array_equal(fast.view(np.uint32), canon.view(np.uint32)):
                diff = np.argwhere((fast.view(np.uint32) != canon.view(np.uint32)).any(axis=-1))
                bad += 1
                print("MISMATCH scene %s seed %d trial %d path %s: %d pixels, first %s; eye %s look %s up %s fov %.3f" %
                      (name, seed, trial, path, len(diff), diff[0], eye, look, up, fov))
print("%s: %d seeds x 14 cameras x 2 modes, %d mismatching launches" % (name, n_seeds, bad))
