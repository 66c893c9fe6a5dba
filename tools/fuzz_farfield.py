"""developer tool (needs the -DRTGO_CMPWALK build: tools/_diag/librtgo_hip_cmpwalk.so): the fast walk against the canonical walk FAR
from the origin -- the band the 500-unit guard of rtgo_launch (rtgo_capi.hip) rests on.  The reference's scenes and the tests' random
scenes are translated off the origin (up to `max_shift` units per axis) and seen from eye distances drawn log-uniformly from
[d_lo, d_hi] with a field of view that keeps the scene in frame, so that primary rays start far away and bounce rays carry large
coordinates.  The instrumented launch of that build runs both walks on every ray (also beyond the guard, where the product takes the
canonical one) and records the rays on which they differ; the table is rays and disagreements by `reach` = max(|bounds|, |eye|), the
quantity the guard tests.
   python tools/fuzz_farfield.py [first_seed] [count] [d_lo] [d_hi] [max_shift]"""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("RTGO_HIP_LIB", os.path.join(ROOT, "tools/_diag/librtgo_hip_cmpwalk.so"))
import numpy as np
import oracle_py as O
from raytracingo_amd import capi
O.build(); O.lib()
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d_lo = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
d_hi = float(sys.argv[4]) if len(sys.argv) > 4 else 600.0
max_shift = float(sys.argv[5]) if len(sys.argv) > 5 else 400.0
W, H = 512, 288
VERBOSE = bool(os.environ.get("FUZZ_VERBOSE"))
EDGES = [0, 20, 50, 100, 170, 300, 400, 500, 700, 1000, 2000, 5000, 1e9]
QEDGES = [0, 1e3, 2e3, 4e3, 8e3, 1.6e4, 3.2e4, 6.4e4, 1.28e5, 2.56e5, 1e6, 1e7, 1e30]
rays_by = np.zeros(len(EDGES) - 1, dtype=np.int64); bad_by = np.zeros(len(EDGES) - 1, dtype=np.int64); launches_by = np.zeros(len(EDGES) - 1, dtype=np.int64)
flat_by = np.zeros(len(EDGES) - 1, dtype=np.int64)    # disagreements (first 255 per launch) between two flat primitives / a hit and a miss of one
qrays_by = np.zeros(len(QEDGES) - 1, dtype=np.int64); qbad_by = np.zeros(len(QEDGES) - 1, dtype=np.int64); qlaunches_by = np.zeros(len(QEDGES) - 1, dtype=np.int64)
px_bad = 0
grid_cmp = 0; grid_product = 0
worst = []
rows = []


def shifted(t, s):
    """the scene tables moved by s: model matrices, boxes, light corners (row-major 4x4: translation in 3, 7, 11)"""
    M = np.array(t["M"], dtype=np.float32).reshape(-1, 16).copy()
    M[:, 3] = (M[:, 3] + np.float32(s[0])).astype(np.float32); M[:, 7] = (M[:, 7] + np.float32(s[1])).astype(np.float32); M[:, 11] = (M[:, 11] + np.float32(s[2])).astype(np.float32)
    L = np.array(t["lights"], dtype=np.float32).reshape(-1, 16).copy()
    L[:, 0:3] = (L[:, 0:3] + np.float32(s)).astype(np.float32)
    return M, L


for seed in range(first, first + count):
    if (seed - first) % 10 == 0:
        print("... seed %d, %d launches so far, %d rays, %d disagreements" % (seed, int(launches_by.sum()), int(rays_by.sum()), int(bad_by.sum())), flush=True)
    rng = np.random.default_rng(77000 + seed)
    kind = seed % 4
    if kind == 0:
        name = O.SCENES[(seed // 4) % len(O.SCENES)]
        sc = O.scene(name, W, H); t = O.scene_tables(sc)
    elif kind == 1:
        sc, t = tg._random_scene(O, seed, W, H); name = "random%d" % seed
    else:
        sc, t = tg._box_scene(O, seed, W, H, int(rng.integers(1, 40)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))); name = "boxes%d" % seed
    shift = O.f32(rng.uniform(-1.0, 1.0, 3) * max_shift * rng.choice([0.0, 0.25, 1.0]))
    M, Ls = shifted(t, shift)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], M, t["mat"], None)        # boxes by the CubeBox rule on the device (primitive.cpp:35-118)
    ctx.set_background(t["bg"]); ctx.set_lights(Ls); ctx.resize(W * H)
    _, _, _, bb = ctx.read_bvh()
    lo, hi = bb[:, :3].min(axis=0).astype(np.float64), bb[:, 3:].max(axis=0).astype(np.float64)
    # (CubeBox seeds min/max with +-50: a shifted scene's boxes reach back to +-50; the part that holds geometry is what the camera frames)
    centre = np.asarray(shift, dtype=np.float64) + 0.0
    size = 14.0
    lib = ctx._lib
    if not hasattr(lib, "rtgo_debug_cmpwalk"):          # (the product build: frames only, no ray-by-ray comparison)
        class _NoCmp:
            def rtgo_debug_cmpwalk(self, *a):
                return 0
        lib = _NoCmp()
    else:
        lib.rtgo_debug_cmpwalk.restype = C.c_int; lib.rtgo_debug_cmpwalk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    buf = np.zeros((256, 16), np.float32)
    for trial in range(6):
        dist = float(np.exp(rng.uniform(np.log(d_lo), np.log(d_hi))))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        eye = O.f32(centre + d * dist)
        look = O.f32(centre + rng.normal(size=3) * 1.5)
        up = O.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
        fov = float(np.degrees(2.0 * np.arctan(size * rng.uniform(0.4, 1.2) / dist)))
        U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
        O.lib().oracle_camera_uvw(O.fptr(eye), O.fptr(look), O.fptr(up), fov, np.float32(np.float32(W) / np.float32(H)), O.fptr(U), O.fptr(V), O.fptr(Wv))
        ctx.set_camera(eye, U, V, Wv)
        n = int(rng.choice([2, 3, 4])); path = bool(rng.integers(0, 4) != 0); amb = bool(rng.integers(0, 2)) and not path
        if os.environ.get("FUZZ_ONLY_TRIAL") and trial != int(os.environ["FUZZ_ONLY_TRIAL"]):
            continue          # (reproduce one launch: every random number above has been drawn)
        lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)    # clear
        ctx.reset_stats()
        if VERBOSE:
            print("   %s seed %d trial %d: n %d path %s amb %s eye %s dist %.1f shift %s" % (name, seed, trial, n, path, amb, eye, dist, shift), flush=True)
        ctx.launch(capi.make_frame(W, H, n, 0, path, amb, stats=True)); ctx.sync()
        canon = ctx.read_accum(H, W).copy()
        st = ctx.stats()
        grid_cmp += 1 if st["last_variant"] & 16 else 0          # (RTGO_TREE=2: the walk compared with the canonical one was the grid's)
        rays, reach, quad = st["rays_total"], float(st["guard_reach"]), float(st["guard_quadric"])   # the guard's two quantities, as rtgo_launch computed them
        b = int(np.searchsorted(EDGES, reach, side="right") - 1); qb = int(np.searchsorted(QEDGES, quad, side="right") - 1)
        lib.rtgo_debug_cmpwalk(ctx._h, buf.ctypes.data, buf.nbytes)
        k = int(buf[0].view(np.uint32)[0])
        rays_by[b] += rays; bad_by[b] += k; launches_by[b] += 1
        qrays_by[qb] += rays; qbad_by[qb] += k; qlaunches_by[qb] += 1
        types = list(t["type"])
        kflat = 0
        for r in buf[1:1 + min(k, 255)]:
            pc, pf = int(r[9]), int(r[11])
            if all(types[q] in (1, 2) for q in (pc, pf) if q >= 0):
                kflat += 1
        flat_by[b] += kflat
        rows.append((name, seed, trial, reach, quad, rays, k, kflat))
        if k:
            worst.append((reach, quad, name, seed, trial, k, kflat, buf[1].copy()))
        # the product launch: the timed kernel inside the guard, the canonical walk without counters beyond it -- bit for bit the
        # instrumented frame either way
        ctx.reset_stats()
        if VERBOSE:
            print("      instrumented launch done: %d rays, %d disagreements; product launch" % (rays, k), flush=True)
        ctx.launch(capi.make_frame(W, H, n, 0, path, amb, stats=False)); ctx.sync()
        fast = ctx.read_accum(H, W)
        pst = ctx.stats()
        took_canonical = pst["launches_canonical"]
        grid_product += 1 if pst["last_variant"] & 16 else 0
        if not np.array_equal(fast.view(np.uint32), canon.view(np.uint32)):
            px_bad += 1
            print("PIXEL MISMATCH %s seed %d trial %d reach %.1f quadric %.0f (product launch walked %s)" % (name, seed, trial, reach, quad, "canonical" if took_canonical else "fast"), flush=True)
    ctx.close()
print("reach = max(|scene bounds|, |eye|); rays of the instrumented launches (both walks on every ray)")
for i in range(len(EDGES) - 1):
    if launches_by[i]:
        print("  reach [%6g, %6g): %5d launches %14d rays %6d disagreements (%d of the recorded ones between flat primitives only)" % (EDGES[i], EDGES[i + 1], launches_by[i], rays_by[i], bad_by[i], flat_by[i]))
print("quadric = max over spheres / cylinders of D^2 smax / smin^2 (rtgo_stats.guard_quadric)")
for i in range(len(QEDGES) - 1):
    if qlaunches_by[i]:
        print("  quadric [%8g, %8g): %5d launches %14d rays %6d disagreements" % (QEDGES[i], QEDGES[i + 1], qlaunches_by[i], qrays_by[i], qbad_by[i]))
print("product launches whose frame differs from the instrumented (canonical) frame: %d" % px_bad)
print("launches that walked the uniform grid (RTGO_TREE=2): %d of the instrumented comparisons, %d of the product launches" % (grid_cmp, grid_product))
print("launches with disagreements, by reach:")
for (reach, quad, name, seed, trial, k, kflat, r) in sorted(worst, key=lambda w: w[0])[:16]:
    print("  reach %.1f quadric %.0f %s seed %d trial %d: %d rays (%d flat); first o %s d %s canonical (%g, %d) fast (%g, %d)" %
          (reach, quad, name, seed, trial, k, kflat, r[0:3], r[3:6], r[8], int(r[9]), r[10], int(r[11])))
print("launches with disagreements, by quadric:")
for (reach, quad, name, seed, trial, k, kflat, r) in sorted(worst, key=lambda w: w[1])[:16]:
    print("  quadric %.0f reach %.1f %s seed %d trial %d: %d rays (%d flat)" % (quad, reach, name, seed, trial, k, kflat))
if os.environ.get("FUZZ_CSV"):
    with open(os.environ["FUZZ_CSV"], "a") as fh:
        for r in rows:
            fh.write("%s,%d,%d,%.3f,%.1f,%d,%d,%d\n" % r)
