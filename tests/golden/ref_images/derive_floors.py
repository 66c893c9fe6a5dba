"""Build-container script: what the comparison with the reference's renders can resolve, measured -- written into ref_images.json
under "floors" and quoted by tests/test_reference_renders.py and DESIGN.md section 4.

For each picture the CPU oracle renders the same view twice with independent seeds (frames 0..F-1 and F..2F-1 of 16 spp each), and
  noise_block_mad     block-mean (20 x 20) mean absolute difference between the two halves: what is left of Monte-Carlo noise
  jpeg_block_mad      ... between our 8-bit picture and its own JPEG round trip with the reference files' parameters
                      (quality-100 tables, 4:2:0 chroma): what the container costs
  ours_vs_ref_*       the same statistic, and the per-region ratios ours / reference, against the decoded reference picture
The test thresholds are these measured values with the margins stated in the test.   python tests/golden/ref_images/derive_floors.py [F]
"""
import io
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle_py as O  # noqa: E402
import refimg  # noqa: E402


def jpeg_round_trip(img8):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img8).save(b, "JPEG", quality=100, subsampling=2)
    b.seek(0)
    return np.asarray(Image.open(b))


def ratios(ours_disp_f, ref, regions):
    out = {}
    for k, box in regions.items():
        if k.startswith("surround") or k == "light":
            continue
        r, m = refimg.region_mean(ref, box), refimg.region_mean(ours_disp_f, box)
        out[k] = [round(float(m[c] / r[c]), 4) if r[c] > 10.0 else None for c in range(3)]
    return out


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    O.build()
    meta, arr = refimg.fixtures()
    floors = {"frames_per_half": F, "spp_per_frame": 16, "size": 600}
    t0 = time.time()
    for name, white, regions in (("reference", 0.9, "cornell_z162"), ("reference", None, "cornell_z162"), ("reference_mirror", None, "shell_z14"),
                                 ("distributed_rt", None, "shell_z14")):
        A = refimg.oracle_render(O, name, 600, 4, F, white=white, first_frame=0)
        B = refimg.oracle_render(O, name, 600, 4, F, white=white, first_frame=F)
        M = (A.astype(np.float64) + B) / 2
        a8, b8, m8 = refimg.display(refimg.to8(A)), refimg.display(refimg.to8(B)), refimg.display(refimg.to8(M))
        ref = arr[name]
        key = name + ("_white_%.1f" % white if white is not None else "")
        floors[key] = {
            "noise_block_mad": round(refimg.block_mad(a8, b8), 4),
            "jpeg_block_mad": round(refimg.block_mad(m8, jpeg_round_trip(np.ascontiguousarray(m8))), 4),
            "ours_vs_ref_block_mad": round(refimg.block_mad(m8, ref), 4),
            "ours_vs_ref_block_max": round(float(np.abs(refimg.block_means(m8, 20) - refimg.block_means(ref, 20)).max()), 2),
            "ours_vs_ref_mean_rgb": [round(float(v), 3) for v in m8.reshape(-1, 3).mean(axis=0)],
            "ref_mean_rgb": [round(float(v), 3) for v in ref.reshape(-1, 3).mean(axis=0)],
            "region_ratio_ours_over_ref": ratios(refimg.display(M) * 255.0, ref, meta["regions"][regions]),
            "room_edges_ours": refimg.room_edges(m8),
            "luminance_correlation_room": round(refimg.luminance_correlation(m8, ref, [100, 100, 500, 500]), 5),
        }
        print(key, json.dumps(floors[key]), "%.0f s" % (time.time() - t0), flush=True)
    path = os.path.join(HERE, "ref_images.json")
    j = json.load(open(path))
    j["floors"] = floors
    json.dump(j, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
