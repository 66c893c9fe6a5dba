#!/bin/bash
# developer tool, ON THE GPU BOX: SQ counters of the whitted render kernel (1920x1080, the fine procedural mesh)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
mkdir -p gpurun_out/r2
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/r2/whitted_pmc$i -- python3 tools/whitted_perf.py 1920 1080 > gpurun_out/r2/whitted_pmc$i.log 2>&1 || echo "pass $i failed"
  python tools/pmc_summary.py gpurun_out/r2/whitted_pmc$i "whitted::render" | tail -12
done
