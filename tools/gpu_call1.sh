# round-2 GPU call 1 (run ON THE GPU BOX through gpurun): whole GPU suite, timelines, per-scene timings, the bench line
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2/t1.log
for sc in cornell balls checkered; do
  RTGO_HIP_LIB=tools/_diag/librtgo_hip_timeline.so timeout -k 10 120 python tools/timeline.py $sc 1920 1080 4 path > gpurun_out/r2/timeline_$sc.log 2>&1
done
for args in "cornell 1920 1080 4 path" "balls 1920 1080 4 path" "checkered 1920 1080 4 path" "mirror_spheres 3840 2160 8 path" "plateau 3840 2160 16 path" "cornell 1920 1080 4 distributed"; do
  timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" >> gpurun_out/r2/perf_base.log
done
timeout -k 10 300 python bench.py > gpurun_out/r2/bench1.json 2> gpurun_out/r2/bench1.err
grep -E "passed|failed|error" gpurun_out/r2/t1.log | tail -3; cat gpurun_out/r2/perf_base.log; cat gpurun_out/r2/bench1.json
