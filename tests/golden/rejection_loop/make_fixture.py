"""The launch on which GetRayOnHemisphere's rejection loop (kernel.cu:109-120) never ends -- found by tools/fuzz_farfield.py (seed 45, trial 5
of `fuzz_farfield.py 0 N 2 400 300`): the tests' random scene 45 moved ~330 units off the origin, seen from 11.5 units, distributed mode,
16 spp.  This script redraws exactly that tool's random numbers and stores the tables of the launch (scene, lights, camera) as data:
   python tests/golden/rejection_loop/make_fixture.py"""
import importlib.util, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import oracle_py as O
O.build(); O.lib()
spec = importlib.util.spec_from_file_location("tg", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
W, H, seed, the_trial, d_lo, d_hi, max_shift = 512, 288, 45, 5, 2.0, 400.0, 300.0
rng = np.random.default_rng(77000 + seed)
sc, t = tg._random_scene(O, seed, W, H)
shift = O.f32(rng.uniform(-1.0, 1.0, 3) * max_shift * rng.choice([0.0, 0.25, 1.0]))
M = np.array(t["M"], dtype=np.float32).reshape(-1, 16).copy()
M[:, 3] = (M[:, 3] + np.float32(shift[0])).astype(np.float32); M[:, 7] = (M[:, 7] + np.float32(shift[1])).astype(np.float32); M[:, 11] = (M[:, 11] + np.float32(shift[2])).astype(np.float32)
L = np.array(t["lights"], dtype=np.float32).reshape(-1, 16).copy()
L[:, 0:3] = (L[:, 0:3] + shift).astype(np.float32)
centre, size = np.asarray(shift, dtype=np.float64) + 0.0, 14.0
for trial in range(6):
    dist = float(np.exp(rng.uniform(np.log(d_lo), np.log(d_hi))))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    eye = O.f32(centre + d * dist)
    look = O.f32(centre + rng.normal(size=3) * 1.5)
    up = O.f32([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)])
    fov = float(np.degrees(2.0 * np.arctan(size * rng.uniform(0.4, 1.2) / dist)))
    U, V, Wv = [np.zeros(3, dtype=np.float32) for _ in range(3)]
    O.lib().oracle_camera_uvw(O.fptr(eye), O.fptr(look), O.fptr(up), fov, np.float32(np.float32(W) / np.float32(H)), O.fptr(U), O.fptr(V), O.fptr(Wv))
    n = int(rng.choice([2, 3, 4])); path = bool(rng.integers(0, 4) != 0); amb = bool(rng.integers(0, 2)) and not path
    if trial == the_trial:
        break
print("shift", shift, "eye", eye, "n", n, "path", path, "ambient", amb, "primitives", len(t["type"]))
np.savez(os.path.join(HERE, "random45_far.npz"), type=np.asarray(t["type"], np.uint32), M=M, mat=np.asarray(t["mat"], np.float32), lights=L,
         bg=np.asarray(t["bg"], np.float32), cam=np.concatenate([eye, U, V, Wv]).astype(np.float32), W=W, H=H, n=n, path=path, ambient=amb)
