#!/bin/bash
# developer tool, ON THE GPU BOX: the megakernel's SQ counters on one workload:  tools/pmc_scene.sh <scene> <W> <H> <N> <mode>
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
mkdir -p gpurun_out/r2
tag=$1_$2x$3_N$4_$5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r2/pmc_$tag -- python3 tools/quick_perf.py $1 $2 $3 $4 $5 > gpurun_out/r2/pmc_$tag.log 2>&1 || echo "pmc failed"
grep "ms/frame\|stats launch" gpurun_out/r2/pmc_$tag.log
python tools/pmc_summary.py gpurun_out/r2/pmc_$tag "false" 2>&1 | tail -10
