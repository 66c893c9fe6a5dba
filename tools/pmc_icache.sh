#!/bin/bash
# developer tool, ON THE GPU BOX: instruction-cache and scalar-cache counters of the megakernel on cornell / balls (quick_perf workload)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
mkdir -p gpurun_out/r2
for sc in cornell balls; do
  i=0
  for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_WAVE_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/r2/pmc_ic_${sc}_$i -- python3 tools/quick_perf.py $sc 1920 1080 4 path > gpurun_out/r2/pmc_ic_${sc}_$i.log 2>&1 || echo "pass failed"
    echo "== $sc"; python tools/pmc_summary.py gpurun_out/r2/pmc_ic_${sc}_$i "false" 2>&1 | tail -9
  done
done
