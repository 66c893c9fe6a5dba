#!/bin/bash
# tools/profile_round.sh <tag> [bench.py arguments, e.g. --scene balls]  -- run ON THE GPU BOX (via gpurun).  Collects, for the bench workload:
#   1. rocprofv3 --kernel-trace --stats   (average kernel duration, to compare with bench.py's HIP-event figure)
#   2. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM traffic; MI355X_MICROARCH.md "HBM"), SQ issue/wait/lane counters
# and leaves raw CSVs in gpurun_out/prof_<tag>/ plus a summary JSON/markdown that tools/profile_summary.py writes.
set -u
TAG=${1:-round}
shift
ARGS="$@"
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS > $OUT/bench_traced.json 2> $OUT/trace.err
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SMEM" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_ACTIVE_INST_SCA" \
         "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err || echo "pmc pass $i failed"
done
python tools/profile_summary.py $OUT $TAG
