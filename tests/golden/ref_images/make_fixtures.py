"""Build-container script: decode the renders the reference itself holds (/root/reference/images/*.jpg) into data fixtures.

    python tests/golden/ref_images/make_fixtures.py            # decode + provenance  -> ref_images.npz / ref_images.json
    python tests/golden/ref_images/derive_floors.py            # the comparison's noise floors and measured values (oracle renders; minutes)

The reference has no tests and no golden vectors; the only outputs it holds are screenshots of its own window.  Three of
them are framebuffer-sized (600 x 600, the default --dim of engine/main.cpp:60) JPEGs without window chrome, so nothing is cropped:
every pixel of the file is a pixel of the reference's `image` buffer (make_color, kernel.cu:90-98), top row first (the files are the
display orientation; our buffers are bottom row first like the reference's, so the loader flips).  What is stored is decoded PIXELS
(uint8 [600, 600, 3]) plus what a reader needs to judge the comparison: file hash, JPEG parameters (quality-100 tables, 4:2:0 chroma),
which scene revision the picture shows and which camera reproduces its geometry.  No source text of the reference is stored.

Cameras.  The pictures were taken in an interactive window (trackball, renderer.cpp:36-145), so the eye is not always Scene::SetupCamera's
(0, 0, 14) (scene.cpp:660-671).  For each file the eye is FITTED to silhouettes only (edges of the room / of the plate), never to colours:
  reference.jpg          room edges at x = 129/470 (front), 196/403 (back wall) <=> eye (0, 0, 16.2), fov 60 (measured below by `room_edges`)
  reference_mirror.jpg   room edges 91/508, 183/416                             <=> eye (0, 0, 14): SetupCamera's own
  distributed_rt.jpg     same edges as reference_mirror.jpg                     <=> eye (0, 0, 14)
(16.2 = 14 + 11 steps of the window's W/S keys, MOVEMENT_SPEED 0.2: renderer.cpp keyCallback -> Trackball::moveBackward.)
Scene revisions.  reference.jpg shows HEAD's CreateCornellBox geometry (scene.cpp:279-336: tall box back-left, short box front-right)
under a WHITE of 0.9 where HEAD's materials.h says 0.8 (cornellWhite kd): with 0.8 the direct-lit walls agree (0.96) but every
white, indirectly lit surface is 13-22 % darker than the picture; with 0.9 -- one constant, fitted on the back wall -- all seven named
regions agree within 2 % (tests/golden/ref_images/ref_images.json, "floors").  Recursion depth was ruled out (5, 7, 10, 15, 31 give the
same picture within 0.5 %).  reference_mirror.jpg and distributed_rt.jpg show an OLDER cornell (a mirror panel and two other boxes)
inside the SAME shell (walls, ceiling, floor, light: same edges), so only the shell regions listed in ref_images.json are compared for
them; reference_mirror.jpg's shell agrees with HEAD's constants (white 0.8) within 4 %.  distributed_rt.jpg comes from an older
distributed-mode shader as well (its walls carry no kr term: pure red where HEAD adds kr * reflected light; its back wall is twice as
bright): it supports only factor-of-two statements -- which is all SURVEY Q2 needs (an occlusion miss that wrote the background would
leave every surface black).

Excluded, and why: drt_plateau.jpg / drt_plateau2.jpg (the plate carries its objects in another arrangement than HEAD's
CreateFunPlate, scene.cpp:200-277: a five-parameter camera fit to the silhouette stops at an intersection-over-union of 0.81 from either
side of the plate, the disk matching and the objects not); balls_path.png, global_illumination.PNG, path_tracing_filip*.jpg,
balls_drt*.png (scene revisions that no longer exist: emissive balls, a single box); mirroir_spheres.jpg and the other
1920 x 1017 files (hand-moved cameras inside the scene, depth-of-field-like blur from a revision with lens sampling); every *.PNG
screenshot with window chrome (bug reports on revisions before the epsilons of kernel.cu:155,273,292,336,374).
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
SRC = "/root/reference/images"

# name -> (file, scene, mode, eye, what of it is compared)
SOURCES = {
    "reference": dict(file="reference.jpg", scene="cornell", mode="path", eye=[0.0, 0.0, 16.2], lookat=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fov=60.0,
                      revision="HEAD (scene.cpp:279-336)", compare="whole frame",
                      camera_fit="eye.z fitted to the room's silhouette edges (front 129..470, back wall 196..403 on rows 280-320); fov, lookat, up as SetupCamera"),
    "reference_mirror": dict(file="reference_mirror.jpg", scene="cornell", mode="path", eye=[0.0, 0.0, 14.0], lookat=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fov=60.0,
                             revision="older cornell (mirror panel + two boxes) in HEAD's shell", compare="shell regions only",
                             camera_fit="none needed: SetupCamera's own camera reproduces the edges (91..508, 183..416)"),
    "distributed_rt": dict(file="distributed_rt.jpg", scene="cornell", mode="distributed", eye=[0.0, 0.0, 14.0], lookat=[0.0, 0.0, 0.0], up=[0.0, 1.0, 0.0], fov=60.0,
                           revision="older cornell (mirror panel + two boxes) in HEAD's shell", compare="shell regions only",
                           camera_fit="none needed: SetupCamera's own camera reproduces the edges (91..508, 183..416)"),
}

# Regions, in DISPLAY coordinates (x right, y down, half-open), used by tests/test_reference_renders.py.
# cornell shell at eye z = 14 (reference_mirror.jpg, distributed_rt.jpg): parts of the picture no inner object covers in either revision.
SHELL_Z14 = {
    "left_wall": [100, 150, 175, 450],     # x0, y0, x1, y1
    "right_wall": [425, 150, 500, 450],
    "ceiling_left": [120, 100, 205, 125],
    "ceiling_right": [395, 100, 480, 125],
    "back_wall_top": [200, 195, 400, 280],
    "floor_front": [150, 480, 450, 500],
    "light": [240, 135, 360, 165],
    "surround_top": [0, 0, 600, 85],
    "surround_bottom": [0, 515, 600, 600],
    "surround_left": [0, 0, 85, 600],
    "surround_right": [515, 0, 600, 600],
}
# cornell at eye z = 16.2 (reference.jpg): named regions for the sharp checks (the whole frame is compared block by block as well)
REGIONS_Z162 = {
    "left_wall": [135, 180, 190, 420],
    "right_wall": [410, 180, 465, 420],
    "back_wall": [205, 200, 395, 290],
    "floor_front": [180, 440, 290, 462],
    "ceiling": [150, 135, 220, 150],
    "light": [245, 160, 355, 183],
    "tall_box_front": [245, 310, 295, 400],
    "short_box_top": [310, 368, 380, 376],
    "surround_top": [0, 0, 600, 120],
    "surround_bottom": [0, 480, 600, 600],
    "surround_left": [0, 0, 120, 600],
    "surround_right": [480, 0, 600, 600],
}


def room_edges(img):
    """first/last lit column on rows 280-320 and the columns where the red / blue walls end (the back wall's edges): what eye.z was fitted to"""
    a = img.astype(int)
    s = a.sum(axis=2)
    row = s[280:320].mean(axis=0)
    xs = np.where(row > 12)[0]
    r, g, b = a[280:320, :, 0].mean(axis=0), a[280:320, :, 1].mean(axis=0), a[280:320, :, 2].mean(axis=0)
    red = np.where((r > 40) & (g < 0.3 * r) & (b < 0.3 * r))[0]
    blue = np.where((b > 40) & (g < 0.3 * b) & (r < 0.3 * b))[0]
    return [int(xs.min()), int(xs.max()), int(red.max()), int(blue.min())]


def main():
    from PIL import Image
    arrays, meta = {}, {}
    for name, s in SOURCES.items():
        path = os.path.join(SRC, s["file"])
        raw = open(path, "rb").read()
        im = Image.open(path)
        assert im.size == (600, 600) and im.mode == "RGB", (name, im.size, im.mode)
        px = np.asarray(im, dtype=np.uint8)
        arrays[name] = px
        m = dict(s)
        m["sha256"] = hashlib.sha256(raw).hexdigest()
        m["shape"] = list(px.shape)
        m["jpeg"] = {"quantization_all_ones": all(all(v == 1 for v in t) for t in im.quantization.values()),
                     "sampling": [list(l) for l in getattr(im, "layer", [])]}
        m["mean_rgb"] = [float(v) for v in px.reshape(-1, 3).mean(axis=0)]
        if s["scene"] == "cornell":
            m["room_edges_rows_280_320"] = room_edges(px)
        meta[name] = m
    out = {"images": meta, "regions": {"shell_z14": SHELL_Z14, "cornell_z162": REGIONS_Z162},
           "orientation": "arrays are in display orientation (row 0 = top); the renderer's buffers are bottom row first"}
    prev_path = os.path.join(HERE, "ref_images.json")
    if os.path.exists(prev_path):   # keep what the slow steps wrote earlier (camera fit, floors)
        prev = json.load(open(prev_path))
        for k in ("floors",):
            if k in prev:
                out[k] = prev[k]
    np.savez_compressed(os.path.join(HERE, "ref_images.npz"), **arrays)
    json.dump(out, open(prev_path, "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "ref_images.npz"), {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
