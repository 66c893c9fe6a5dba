#!/bin/bash
# developer tool, ON THE GPU BOX: time every tools/_diag/ab_*.so on a few workloads (quick_perf), two rounds to see the noise
mkdir -p gpurun_out/r2
OUT=gpurun_out/r2/ab_${1:-x}.log; : > $OUT
for round in 1 2; do
for lib in tools/_diag/ab_*.so; do
  for args in "cornell 1920 1080 4 path" "balls 1920 1080 4 path" "checkered 1920 1080 4 path" "mirror_spheres 3840 2160 8 path" "plateau 3840 2160 16 path" "window 1920 1080 4 path" "slide 1920 1080 4 path"; do
    r=$(RTGO_HIP_LIB=$lib timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    echo "$(basename $lib .so) | $r" >> $OUT
  done
done
done
sort $OUT | awk -F'|' '{print}' 
