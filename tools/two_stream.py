"""developer experiment: one GPU's share of a G-way tiled frame rendered as ONE launch per frame, or as TWO half-share launches
per frame on two contexts / HIP streams (the tail and launch latency of one overlap the body of the other).
   python tools/two_stream.py [G] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from raytracingo_amd import capi, scene as hscene

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, N = 1920, 1080, 4
t = hscene.tables("cornell", W, H)


def make():
    c = capi.Context(0)
    c.set_scene(t["type"], t["M"], t["mat"], t["aabb"]); c.set_camera(t["cam"][0:3], t["cam"][3:6], t["cam"][6:9], t["cam"][9:12])
    c.set_background(t["bg"]); c.set_lights(t["lights"]); c.resize(W * H)
    return c


def run(ctxs, parts):
    # parts: list of (n_ranks, rank) per context
    for f in range(5):
        for c, (n, r) in zip(ctxs, parts):
            c.launch(capi.make_frame(W, H, N, f, True, bands=(4, n, r)))
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for f in range(5, 5 + K):
        for c, (n, r) in zip(ctxs, parts):
            c.launch(capi.make_frame(W, H, N, f, True, bands=(4, n, r)))
    for c in ctxs:
        c.sync()
    return (time.perf_counter() - t0) / K * 1e3


cs = [make() for _ in range(4)]
a, b = cs[0], cs[1]
one = run([a], [(G, 0)])
two = run([a, b], [(2 * G, 0), (2 * G, 1)])
three = run(cs[:3], [(3 * G, i) for i in range(3)])
four = run(cs, [(4 * G, i) for i in range(4)])
print("share 1/%d of cornell 1080p spp16: one launch per frame %.4f ms/frame; 2 / 3 / 4 part launches on as many streams %.4f / %.4f / %.4f ms/frame" % (G, one, two, three, four))
two = run([a, b], [(2 * G, 0), (2 * G, 1)])
# the two halves together are the same rows as the single share
ra = a.read_accum(capi.local_rows(H, 4, 2 * G, 0), W); rb = b.read_accum(capi.local_rows(H, 4, 2 * G, 1), W)
print("rows:", ra.shape[0] + rb.shape[0], "of", capi.local_rows(H, 4, G, 0))
