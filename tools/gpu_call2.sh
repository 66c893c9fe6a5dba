# round-2 GPU call 2: 4-wide tree + rectangles-only kernels: parity first, then timings
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r2/t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/t2.log
grep -E "passed|failed|error" gpurun_out/r2/t2.log | tail -3
if grep -q "pytest rc=0" gpurun_out/r2/t2.log; then
  P=gpurun_out/r2/perf2.log; : > $P
  for args in "cornell 1920 1080 4 path" "balls 1920 1080 4 path" "checkered 1920 1080 4 path" "mirror_spheres 3840 2160 8 path" "plateau 3840 2160 16 path" "cornell 1920 1080 4 distributed" "window 1920 1080 4 path" "soft_mirrors 1920 1080 4 distributed"; do
    timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" >> $P
  done
  for w in 4 5 6; do echo "cornell MAX_WPE=$w" >> $P; RTGO_MAX_WPE=$w timeout -k 10 120 python tools/quick_perf.py cornell 1920 1080 4 path 2>&1 | grep "ms/frame" >> $P; done


  for sc in cornell balls checkered; do
    RTGO_HIP_LIB=tools/_diag/librtgo_hip_timeline.so timeout -k 10 120 python tools/timeline.py $sc 1920 1080 4 path 2>&1 | grep "ray-loop\|HIP-event" >> $P
  done
  cat $P
else
  grep -n "Error\|assert\|FAILED" gpurun_out/r2/t2.log | head -20
fi
