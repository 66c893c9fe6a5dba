mkdir -p gpurun_out/r2
O=gpurun_out/r2/misc10.log; : > $O
for b in 6 40 70 140; do echo "balls leaf budget $b: $(RTGO_LEAF_BUDGET=$b python tools/quick_perf.py balls 1920 1080 4 path 2>&1 | grep ms/frame | sed 's/,.*//')" >> $O; done
for b in 3 6 12 24; do echo "checkered leaf budget $b: $(RTGO_LEAF_BUDGET=$b python tools/quick_perf.py checkered 1920 1080 4 path 2>&1 | grep ms/frame | sed 's/,.*//')" >> $O; done
for pc in 25 40 60 90; do echo "cornell big percent $pc: $(RTGO_BIG_PERCENT=$pc python tools/quick_perf.py cornell 1920 1080 4 path 2>&1 | grep ms/frame | sed 's/,.*//')" >> $O; done
E=raytracingo_amd/rtgo_engine
for args in "" "--gpus=1 --launches-per-gpu=1" "--gpus=1 --launches-per-gpu=2" "--gpus=1 --launches-per-gpu=2 --present-every=8"; do
  echo "engine cornell 1080p spp16 200 frames [$args]: $($E --scene=cornell --mode=path --dim=1920x1080 --sample=4 --frames=200 $args 2>&1 | tail -1)" >> $O
done
for args in "" "--gpus=1 --launches-per-gpu=2"; do
  echo "engine mirror_spheres 4K spp64 40 frames [$args]: $($E --scene=mirror_spheres --mode=path --dim=3840x2160 --sample=8 --frames=40 $args 2>&1 | tail -1)" >> $O
done
cat $O
