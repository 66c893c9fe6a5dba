#!/bin/bash
# developer tool, ON THE GPU BOX: HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the bench workload for the register-budget variants
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
for w in 4 5 6; do
  export RTGO_MAX_WPE=$w
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/r2/traffic_w${w}_$C -- python3 tools/quick_perf.py cornell 1920 1080 4 path > gpurun_out/r2/traffic_w${w}_$C.log 2>&1 || echo "pass failed"
    echo "== wpe $w $C"; python tools/pmc_summary.py gpurun_out/r2/traffic_w${w}_$C "false" 2>&1 | tail -2
  done
done
