#!/bin/bash
# developer tool, ON THE GPU BOX: GPU suite, then the streaming loop of multi-pass frames against the lock-step pass loop
mkdir -p gpurun_out/r2
OUT=gpurun_out/r2/stream_${1:-x}.log; : > $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2/stream_tests_${1:-x}.log 2>&1 || { tail -30 gpurun_out/r2/stream_tests_${1:-x}.log; exit 1; }
tail -3 gpurun_out/r2/stream_tests_${1:-x}.log
for round in 1 2; do
  for args in "plateau 3840 2160 16 path" "mirror_spheres 3840 2160 8 path" "slide 1920 1080 8 path" "cornell 1920 1080 8 path" "cornell 1920 1080 6 distributed" "balls 1920 1080 8 path" "cornell 1920 1080 4 path"; do
    a=$(RTGO_STREAM=1 timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    b=$(RTGO_STREAM=0 timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    c=$(timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed "s/,.*//")
    echo "stream: $a | lock-step: $b | trial: $c" >> $OUT
  done
done
cat $OUT
