#!/bin/bash
# developer tool, ON THE GPU BOX: timings of every tools/_diag/ab_*.so + one SQ counter pass per variant on cornell and balls
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
mkdir -p gpurun_out/r2
OUT=gpurun_out/r2/ab_${1:-x}.log; : > $OUT
for round in 1 2; do
for lib in tools/_diag/ab_*.so; do
  for args in "cornell 1920 1080 4 path" "cornell 1920 1080 4 distributed" "balls 1920 1080 4 path" "checkered 1920 1080 4 path" "mirror_spheres 3840 2160 8 path"; do
    r=$(RTGO_HIP_LIB=$lib timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    echo "$(basename $lib .so) | $r" >> $OUT
  done
done
done
sort $OUT
for lib in tools/_diag/ab_*.so; do
  n=$(basename $lib .so)
  export RTGO_HIP_LIB=$PWD/$lib
  for sc in cornell balls; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r2/pmc_${n}_$sc -- python3 tools/quick_perf.py $sc 1920 1080 4 path > gpurun_out/r2/pmc_${n}_$sc.log 2>&1 || echo "pmc $n $sc failed"
    echo "== $n $sc"; python tools/pmc_summary.py gpurun_out/r2/pmc_${n}_$sc "false" 2>&1 | tail -12
  done
done
