#!/bin/bash
# developer tool, ON THE GPU BOX: the fast-walk structures (RTGO_TREE=0 / 1 / 2 = the 36 % tree, the 15 % tree, the uniform grid) and the
# launch-time trial side by side on the scenes that have a grid.  (RTGO_NO_FRAMES: the flat-primitives instantiation has no grid variant.)
OUT=${1:-gpurun_out/grid_sweep.log}; : > $OUT
run() { env "$@" 2>&1 | grep -o -e "[a-z_]* [0-9]*x[0-9]* N=[0-9]* [a-z]*: [0-9.]* ms/frame" -e "grid [0-9]* x [0-9]* x [0-9]* over [0-9]* primitives, [0-9]* list entries" | tr '\n' ' '; }
for args in "balls 1920 1080 4 path" "balls 1920 1080 4 dist" "balls 1920 1080 8 path" "checkered 1920 1080 4 path" "checkered 1920 1080 4 dist" "slide 1920 1080 4 path" "plateau 3840 2160 16 path" "mirror_spheres 3840 2160 8 path" "window 1920 1080 4 path"; do
  for tree in 0 1 2 trial; do
    if [ $tree = trial ]; then e="RTGO_X=0"; else e="RTGO_TREE=$tree"; fi
    echo "tree $tree | $(run RTGO_DEBUG=1 RTGO_NO_FRAMES=1 RTGO_GUARD_QUADRIC=1e9 $e timeout -k 10 120 python tools/quick_perf.py $args)" >> $OUT
  done
done
cat $OUT
