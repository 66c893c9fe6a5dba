#!/bin/bash
# developer tool, ON THE GPU BOX: the fast-walk build knobs (RTGO_LEAF_BUDGET, RTGO_BIG_PERCENT) on a few scenes, and the scene build time
mkdir -p gpurun_out/r2
O=gpurun_out/r2/knobs.log; : > $O
for b in 1 6 12 33 66; do for sc in balls checkered plateau; do echo "$sc leaf budget $b: $(RTGO_LEAF_BUDGET=$b python tools/quick_perf.py $sc 1920 1080 4 path 2>&1 | grep ms/frame | sed 's/,.*//')" >> $O; done; done
for pc in 25 40 60; do for sc in cornell balls checkered; do echo "$sc big percent $pc: $(RTGO_BIG_PERCENT=$pc python tools/quick_perf.py $sc 1920 1080 4 path 2>&1 | grep ms/frame | sed 's/,.*//')" >> $O; done; done
python - >> $O <<'PY'
import time, sys
sys.path.insert(0, ".")
from raytracingo_amd import capi, scene as hscene
for name in ("cornell", "balls", "checkered", "plateau"):
    t = hscene.tables(name, 64, 64)
    ctx = capi.Context(0)
    ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"])
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.set_scene(t["type"], t["M"], t["mat"], t["aabb"])
    print("rtgo_set_scene %s: %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
PY
cat $O
