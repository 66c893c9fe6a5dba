#!/bin/bash
# developer tool, ON THE GPU BOX: time library variants / environment knobs side by side (quick_perf), two rounds to see the noise.
#   tools/ab_r3.sh out.log "label|ENV=val ...|lib.so|scene W H N mode" ...
OUT=$1; shift; : > $OUT
for round in 1 2; do
  for cfg in "$@"; do
    IFS='|' read -r label envs lib args <<< "$cfg"
    r=$(env $envs RTGO_HIP_LIB=$lib timeout -k 10 120 python tools/quick_perf.py $args 2>&1 | grep "ms/frame" | sed 's/,.*//')
    echo "$label | $r" >> $OUT
  done
done
sort $OUT
