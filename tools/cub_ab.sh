#!/bin/bash
# developer tool, ON THE GPU BOX: cuboid_range against the pair-by-pair tests (RTGO_NO_CUBOID=1), two rounds
mkdir -p gpurun_out/s2
OUT=gpurun_out/s2/cub_ab_${1:-x}.log; : > $OUT
for round in 1 2; do
for V in cuboid pairs; do
  for args in "cornell 1920 1080 4 path" "checkered 1920 1080 4 path" "window 1920 1080 4 path" "mirror_spheres 3840 2160 8 path" "plateau 3840 2160 16 path" "soft_mirrors 1920 1080 4 path" "cornell 1920 1080 6 dist" "balls 1920 1080 4 path"; do
    if [ $V = pairs ]; then export RTGO_NO_CUBOID=1; else unset RTGO_NO_CUBOID; fi
    r=$(timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    echo "$V | $r" >> $OUT
  done
done
done
sort $OUT
