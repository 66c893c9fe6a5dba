"""developer tool: random scenes x random cameras, whole frames: GPU (timed kernel) vs the CPU oracle under tests/parity.py's rule, ray counts.
   python tools/fuzz_oracle.py [first_seed] [count]"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import oracle_py as O
from parity import compare
from raytracingo_amd import capi
O.build(); O.lib()
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H = 96, 64
worst = 1.0; bad = 0; launches = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    if seed % 3 == 0:
        sc, t = tg._random_scene(O, seed, W, H)
    else:
        sc, t = tg._box_scene(O, seed, W, H, int(rng.integers(1, 30)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); ctx.set_background(t["bg"]); ctx.set_lights(t["lights"]); ctx.resize(W * H)
    bb = np.asarray(t["aabb"], dtype=np.float64).reshape(-1, 6)
    centre = 0.5 * (bb[:, :3].min(axis=0) + bb[:, 3:].max(axis=0)); size = 12.0
    for trial in range(3):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        eye = O.f32(centre + d * size * rng.choice([0.3, 1.0, 3.0, 20.0]))
        look = O.f32(centre + rng.normal(size=3) * size * 0.2)
        up = O.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
        U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
        O.lib().oracle_camera_uvw(O.fptr(eye), O.fptr(look), O.fptr(up), float(rng.uniform(20.0, 100.0)), np.float32(np.float32(W) / np.float32(H)), O.fptr(U), O.fptr(V), O.fptr(Wv))
        ctx.set_camera(eye, U, V, Wv)
        sc.eye[:], sc.U[:], sc.V[:], sc.W[:] = eye.tolist(), U.tolist(), V.tolist(), Wv.tolist()
        n = int(rng.choice([1, 2, 3])); path = bool(rng.integers(0, 2)); amb = bool(rng.integers(0, 2)) and not path
        ctx.reset_stats()
        ctx.launch(capi.make_frame(W, H, n, 0, path, amb)); ctx.sync()
        acc, img, st = ctx.read_accum(H, W), ctx.read_image(H, W), ctx.stats()
        racc, rimg, rc = O.render(sc, O.frame(W, H, n, 0, path=path, ambient=amb, mode=1))
        m = compare(acc, racc, img, rimg)
        launches += 1
        worst = min(worst, m["frac_within"])
        ray_off = abs(st["rays_total"] - rc["rays_total"]) / max(rc["rays_total"], 1)
        if m["frac_within"] < 0.98 or m["mean_rel"] > 5e-3 or m.get("image_max_lsb_within", 0) > 1 or ray_off > 0.01:
            bad += 1
            print("OUT OF TOLERANCE seed %d trial %d n %d path %s amb %s: %r rays %d vs %d" % (seed, trial, n, path, amb, m, st["rays_total"], rc["rays_total"]))
    ctx.close()
print("%d scenes, %d frames against the oracle: %d out of tolerance; lowest share of pixels within 1e-4: %.4f" % (count, launches, bad, worst))
