O=gpurun_out/r2/knobs2.log; mkdir -p gpurun_out/r2; : > $O
for pc in 30 33 36 40; do for sc in "cornell 1920 1080 4" "balls 1920 1080 4" "checkered 1920 1080 4" "window 1920 1080 4" "slide 1920 1080 4" "soft_mirrors 1920 1080 4" "mirror_spheres 3840 2160 8" "plateau 3840 2160 16"; do echo "big $pc: $(RTGO_BIG_PERCENT=$pc python tools/quick_perf.py $sc path 2>&1 | grep ms/frame | sed 's/,.*//')" >> $O; done; done
sort -k3,3 -k2,2n $O
