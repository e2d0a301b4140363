# quick_bench (render call only: set-up + lists + render kernel) for the tree's library with several tuning values, and
# for tools/probes/ab_libs/libnfi_good.so, on ONE box
cp nerf_from_image_amd/libnfi_hip.so /tmp/libnfi_new.so
echo "== new"; NFI_TUNING=${NFI_TUNING:-0,32,64,96} NFI_ITERS=50 timeout 300 python tools/quick_bench.py 2>&1 | grep "B=8"
cp tools/probes/ab_libs/libnfi_good.so nerf_from_image_amd/libnfi_hip.so
echo "== good"; NFI_TUNING=0 NFI_ITERS=50 timeout 300 python tools/quick_bench.py 2>&1 | grep "B=8"
cp /tmp/libnfi_new.so nerf_from_image_amd/libnfi_hip.so
