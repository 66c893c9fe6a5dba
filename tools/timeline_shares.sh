#!/bin/bash
# developer tool, ON THE GPU BOX: per-wave timelines (diagnostic build) of a full cornell frame and of 1/2 .. 1/16 band shares
mkdir -p gpurun_out/r2
for G in 1 2 4 8 16; do
  BANDS=$G,0 RTGO_HIP_LIB=$PWD/tools/_diag/librtgo_hip_timeline.so timeout -k 10 120 python tools/timeline.py cornell 1920 1080 4 path > gpurun_out/r2/timeline_share_$G.log 2>&1
  tail -22 gpurun_out/r2/timeline_share_$G.log
done
