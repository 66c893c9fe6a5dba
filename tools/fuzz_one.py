"""developer tool: replay one camera of tools/fuzz_cameras.py and hold both GPU kernels against the oracle at the differing pixels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
from raytracingo_amd import capi
O.build(); O.lib()
name, seed, trial_want, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "path"
W, H, n = 144, 80, 2
sc = O.scene(name, W, H); t = O.scene_tables(sc)
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
bb = np.asarray(t["aabb"], dtype=np.float64).reshape(-1, 6)
lo, hi = bb[:, :3].min(axis=0), bb[:, 3:].max(axis=0)
centre, size = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
rng = np.random.default_rng(seed)
for trial in range(trial_want + 1):
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    eye = O.f32(centre + d * size * rng.uniform(0.8, 6.0))
    look = O.f32(centre + rng.normal(size=3) * size * rng.choice([0.05, 0.4, 1.5]))
    up = O.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
    fov = float(rng.uniform(15.0, 100.0))
U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
O.lib().oracle_camera_uvw(O.fptr(eye), O.fptr(look), O.fptr(up), fov, np.float32(np.float32(W) / np.float32(H)), O.fptr(U), O.fptr(V), O.fptr(Wv))
ctx.set_camera(eye, U, V, Wv)
sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = eye.tolist(), U.tolist(), V.tolist(), Wv.tolist()
ctx.launch(capi.make_frame(W, H, n, 0, path)); ctx.sync(); fast = ctx.read_accum(H, W).copy()
ctx.reset_stats(); ctx.launch(capi.make_frame(W, H, n, 0, path, stats=True)); ctx.sync(); canon = ctx.read_accum(H, W).copy(); st = ctx.stats()
o1, _, c1 = O.render(sc, O.frame(W, H, n, 0, path=path, mode=1))
o0, _, c0 = O.render(sc, O.frame(W, H, n, 0, path=path, mode=0))
print("oracle literal == oracle LBVH:", np.array_equal(o0.view(np.uint32), o1.view(np.uint32)), " rays gpu/oracle", st["rays_total"], c1["rays_total"])
diff = np.argwhere((fast.view(np.uint32) != canon.view(np.uint32)).any(axis=-1))
for (y, x) in diff:
    print("pixel", (y, x), "fast", fast[y, x, :3], "canon", canon[y, x, :3], "oracle", o1[y, x, :3], "literal", o0[y, x, :3])
print("canon vs oracle: max abs diff", np.abs(canon[..., :3] - o1[..., :3]).max(), " fast vs oracle", np.abs(fast[..., :3] - o1[..., :3]).max())
print("eye", eye, "size", size)
for (y, x) in diff[:1]:
    for md in range(6):
        fr = lambda stats: capi.make_frame(W, H, n, 0, path, False, (int(x), int(y), 1, 1), max_depth=md, stats=stats)
        ctx.reset_stats(); ctx.launch(fr(False)); ctx.sync(); a = ctx.read_accum(1, 1).copy(); ra = ctx.stats()["rays_total"]
        ctx.reset_stats(); ctx.launch(fr(True)); ctx.sync(); b = ctx.read_accum(1, 1).copy(); sb = ctx.stats()
        print("max_depth", md, "fast", a[0, 0, :3], "rays", ra, "| canon", b[0, 0, :3], "rays", sb["rays_total"], "hits", sb["hits"])
