"""developer tool: random scenes (tests' generators) x random cameras / windows / sample counts: timed kernel vs instrumented kernel, bitwise.
   python tools/fuzz_scenes.py [first_seed] [count]"""
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
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
W, H = 96, 64
bad = 0; launches = 0; ray_bad = 0; grid_launches = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    kind = seed % 3
    if kind == 0:
        sc, t = tg._random_scene(O, seed, W, H)
    else:
        sc, t = tg._box_scene(O, seed, W, H, int(rng.integers(1, 40)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], None if seed % 2 else t["aabb"]); ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    bb = np.asarray(t["aabb"], dtype=np.float64).reshape(-1, 6)
    centre = 0.5 * (bb[:, :3].min(axis=0) + bb[:, 3:].max(axis=0)); size = 12.0
    for trial in range(6):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        eye = O.f32(centre + d * size * rng.choice([0.05, 0.3, 1.0, 3.0, 30.0, 300.0]))
        look = O.f32(centre + rng.normal(size=3) * size * rng.choice([0.05, 0.4]))
        up = O.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
        U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
        O.lib().oracle_camera_uvw(O.fptr(eye), O.fptr(look), O.fptr(up), float(rng.uniform(10.0, 120.0)), np.float32(np.float32(W) / np.float32(H)), O.fptr(U), O.fptr(V), O.fptr(Wv))
        ctx.set_camera(eye, U, V, Wv)
        n = int(rng.choice([1, 2, 3, 5, 6, 9])); frame = int(rng.choice([0, 3])); path = bool(rng.integers(0, 2)); amb = bool(rng.integers(0, 2)) and not path
        md = int(rng.choice([5, 5, 2]))
        win = None if rng.integers(0, 2) else (int(rng.integers(0, 40)), int(rng.integers(0, 30)), int(rng.integers(1, 56)), int(rng.integers(1, 34)))
        G = int(rng.choice([1, 1, 2, 3])); g = int(rng.integers(0, G))
        h = win[3] if win else H; w = win[2] if win else W
        rows = capi.local_rows(h, 4, G, g)
        if rows == 0:
            continue
        prev = rng.random((rows, w, 4), dtype=np.float32)
        outs = []
        for stats in (False, True):
            ctx.write_accum(prev)
            ctx.launch(capi.make_frame(W, H, n, frame, path, amb, win, (4, G, g), max_depth=md, stats=stats)); ctx.sync()
            outs.append(ctx.read_accum(rows, w).copy())
            grid_launches += 1 if ctx.stats()["last_variant"] & 16 else 0
        launches += 1
        if not np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)):
            bad += 1
            print("MISMATCH seed %d trial %d: kind %d n %d frame %d path %s amb %s md %d win %s bands %s eye %s: %d pixels" %
                  (seed, trial, kind, n, frame, path, amb, md, win, (G, g), eye, int((outs[0].view(np.uint32) != outs[1].view(np.uint32)).any(axis=-1).sum())))
    if hasattr(ctx._lib, "rtgo_debug_cmpwalk"):   # -DRTGO_CMPWALK build: the instrumented launches compared the walks ray by ray
        import ctypes as C
        ctx._lib.rtgo_debug_cmpwalk.restype = C.c_int; ctx._lib.rtgo_debug_cmpwalk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        buf = np.zeros((256, 16), np.float32)
        ctx._lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)
        k = int(buf[0].view(np.uint32)[0])
        ray_bad += k
        if k:
            print("seed %d: %d rays on which the walks disagree; first: o %s d %s tmin %g tmax %g canonical (%g, %d) fast (%g, %d)" %
                  (seed, k, buf[1, 0:3], buf[1, 3:6], buf[1, 6], buf[1, 7], buf[1, 8], int(buf[1, 9]), buf[1, 10], int(buf[1, 11])))
    ctx.close()
print("%d scenes, %d launch pairs, %d mismatches; rays on which the walks disagree (RTGO_CMPWALK build only): %d; launches on the uniform grid (RTGO_TREE=2): %d of %d" % (count, launches, bad, ray_bad, grid_launches, 2 * launches))
