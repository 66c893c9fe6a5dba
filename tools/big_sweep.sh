#!/bin/bash
# developer tool, ON THE GPU BOX: the up-front list's size threshold (RTGO_BIG_PERCENT: a primitive whose box spans at least that share of the
# scene on two axes is tested up front by every ray instead of sitting in the tree) over the reference's scenes
OUT=${1:-gpurun_out/big_sweep.log}; : > $OUT
for pct in 5 10 15 20 25 30 36 45; do
  for args in "plateau 3840 2160 16 path" "cornell 1920 1080 4 path" "balls 1920 1080 4 path" "mirror_spheres 3840 2160 8 path" "checkered 1920 1080 4 path" "window 1920 1080 4 path" "slide 1920 1080 4 path" "soft_mirrors 1920 1080 4 path" "plateau 1920 1080 4 path" "plateau 1920 1080 4 dist"; do
    r=$(RTGO_GUARD_QUADRIC=1e9 RTGO_BIG_PERCENT=$pct timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    echo "big $pct | $r" >> $OUT
  done
done
sort -t'|' -k2,2 -s $OUT
