"""developer tool: replay one launch pair of tools/fuzz_scenes.py (seed, trial) and localise the difference"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import oracle_py as O
from raytracingo_amd import capi
O.build(); O.lib()
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
seed, trial_want = int(sys.argv[1]), int(sys.argv[2])
W, H = 96, 64
rng = np.random.default_rng(seed)
kind = seed % 3
if kind == 0:
    sc, t = tg._random_scene(O, seed, W, H)
else:
    sc, t = tg._box_scene(O, seed, W, H, int(rng.integers(1, 40)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
use_aabb = None if seed % 2 else t["aabb"]
ctx = capi.Context(0)
ctx.set_scene(t["type"], t["M"], t["mat"], use_aabb); ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
bb = np.asarray(t["aabb"], dtype=np.float64).reshape(-1, 6)
centre = 0.5 * (bb[:, :3].min(axis=0) + bb[:, 3:].max(axis=0)); size = 12.0
for trial in range(trial_want + 1):
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    eye = O.f32(centre + d * size * rng.choice([0.05, 0.3, 1.0, 3.0, 30.0, 300.0]))
    look = O.f32(centre + rng.normal(size=3) * size * rng.choice([0.05, 0.4]))
    up = O.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
    U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    fov = float(rng.uniform(10.0, 120.0))
    O.lib().oracle_camera_uvw(O.fptr(eye), O.fptr(look), O.fptr(up), fov, np.float32(np.float32(W) / np.float32(H)), O.fptr(U), O.fptr(V), O.fptr(Wv))
    n = int(rng.choice([1, 2, 3, 5])); frame = int(rng.choice([0, 3])); path = bool(rng.integers(0, 2)); amb = bool(rng.integers(0, 2)) and not path
    md = int(rng.choice([5, 5, 2]))
    win = None if rng.integers(0, 2) else (int(rng.integers(0, 40)), int(rng.integers(0, 30)), int(rng.integers(1, 56)), int(rng.integers(1, 34)))
    G = int(rng.choice([1, 1, 2, 3])); g = int(rng.integers(0, G))
    h = win[3] if win else H; w = win[2] if win else W
    rows = capi.local_rows(h, 4, G, g)
    if rows == 0:
        continue
    prev = rng.random((rows, w, 4), dtype=np.float32)
ctx.set_camera(eye, U, V, Wv)
sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = eye.tolist(), U.tolist(), V.tolist(), Wv.tolist()
print("seed", seed, "trial", trial_want, "kind", kind, "prims", len(t["type"]), "n", n, "frame", frame, "path", path, "amb", amb, "md", md, "win", win, "bands", (G, g), "eye", eye, "fov", fov)
outs = []
for stats in (False, True):
    ctx.write_accum(prev)
    ctx.launch(capi.make_frame(W, H, n, frame, path, amb, win, (4, G, g), max_depth=md, stats=stats)); ctx.sync()
    outs.append(ctx.read_accum(rows, w).copy())
diff = np.argwhere((outs[0].view(np.uint32) != outs[1].view(np.uint32)).any(axis=-1))
print("differing pixels (local row, col):", diff.tolist())
racc, _, _ = O.render(sc, O.frame(W, H, n, frame, path=path, ambient=amb, window=win, bands=(4, G, g), mode=1, max_depth=md) if "max_depth" in O.frame.__code__.co_varnames else O.frame(W, H, n, frame, path=path, ambient=amb, window=win, bands=(4, G, g), mode=1), accum_prev=prev)
for (y, x) in diff[:2]:
    print("pixel", (y, x), "fast", outs[0][y, x, :3], "canon", outs[1][y, x, :3], "oracle", racc[y, x, :3])
    # global pixel coordinates of that local pixel
    wr = [r for r in range(h) if (r // 4) % G == g][y]
    gx, gy = (win[0] if win else 0) + x, (win[1] if win else 0) + wr
    for mdd in range(md + 1):
        res = []
        for stats in (False, True):
            ctx.reset_stats(); ctx.write_accum(np.zeros((1, 1, 4), np.float32))
            ctx.launch(capi.make_frame(W, H, n, 0, path, amb, (int(gx), int(gy), 1, 1), max_depth=mdd, stats=stats)); ctx.sync()
            res.append((ctx.read_accum(1, 1)[0, 0, :3].copy(), ctx.stats()["rays_total"]))
        print("   global pixel", (gx, gy), "max_depth", mdd, "fast", res[0], "canon", res[1])
# with the -DRTGO_CMPWALK build (RTGO_HIP_LIB=tools/_diag/librtgo_hip_cmpwalk.so) the instrumented launch above also ran the
# fast walk on every ray: print the rays on which the two walks disagreed
import ctypes as C
lib = ctx._lib
if hasattr(lib, "rtgo_debug_cmpwalk"):
    lib.rtgo_debug_cmpwalk.restype = C.c_int
    lib.rtgo_debug_cmpwalk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.rtgo_debug_cmpwalk(ctx._h, None, 0) if False else None
    buf = np.zeros((256, 16), np.float32)
    # discard what the localisation launches above recorded, then repeat the original instrumented launch
    lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)
    ctx.write_accum(prev)
    ctx.launch(capi.make_frame(W, H, n, frame, path, amb, win, (4, G, g), max_depth=md, stats=True)); ctx.sync()
    lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)
    cnt = int(buf[0].view(np.uint32)[0])
    print("rays on which the walks disagree:", cnt)
    types = list(t["type"])
    np.set_printoptions(precision=9, suppress=False)
    for r in buf[1:1 + min(cnt, 8)]:
        pc, pf = int(r[9]), int(r[11])
        print("  o", r[0:3], "d", r[3:6], "tmin", r[6], "tmax", r[7], "| canonical t", r[8], "prim", pc, "(type %s)" % (types[pc] if pc >= 0 else "-"),
              "| fast t", r[10], "prim", pf, "(type %s)" % (types[pf] if pf >= 0 else "-"), "| depth", int(r[12]), "phase", int(r[13]))
        for k in (pc, pf):
            if k >= 0:
                print("     prim", k, "M", np.asarray(t["M"]).reshape(-1, 16)[k].tolist(), "aabb", np.asarray(t["aabb"]).reshape(-1, 6)[k].tolist())
