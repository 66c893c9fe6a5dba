#!/bin/bash
# developer tool, ON THE GPU BOX: the grid's resolution on one workload (RTGO_GRID_DIMS=nx,ny,nz; RTGO_TREE=2 pins the grid)
OUT=${1:-gpurun_out/grid_dims.log}; shift; ARGS=${1:-"balls 1920 1080 4 path"}; shift
: > $OUT
for dims in "$@"; do
  r=$(RTGO_GRID_DIMS=$dims RTGO_TREE=2 RTGO_DEBUG=1 timeout -k 10 120 python tools/quick_perf.py $ARGS 2>&1 | grep -o -e "[a-z_]* [0-9]*x[0-9]* N=[0-9]* [a-z]*: [0-9.]* ms/frame" -e "grid [0-9]* x [0-9]* x [0-9]* over [0-9]* primitives, [0-9]* list entries" | tr '\n' ' ')
  echo "dims $dims | $r" >> $OUT
done
cat $OUT
