#!/bin/bash
# developer tool, ON THE GPU BOX: the streaming loop (RTGO_STREAM=1) of every tools/_diag/ab_*.so on the multi-pass workloads
mkdir -p gpurun_out/r2
OUT=gpurun_out/r2/abstream_${1:-x}.log; : > $OUT
for lib in tools/_diag/ab_*.so; do
  for args in "plateau 3840 2160 16 path" "mirror_spheres 3840 2160 8 path" "slide 1920 1080 8 path" "cornell 1920 1080 8 path"; do
    r=$(RTGO_STREAM=1 RTGO_HIP_LIB=$lib timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    echo "$(basename $lib .so) | $r" >> $OUT
  done
done
sort $OUT
