# PMC summary of the render kernel for two prebuilt libraries on ONE box (see ab_two_libs.sh)
R=$PWD; mkdir -p gpurun_out/ab
cp nerf_from_image_amd/libnfi_hip.so /tmp/libnfi_new.so
for which in new good; do
  if [ $which = good ]; then cp tools/probes/ab_libs/libnfi_good.so nerf_from_image_amd/libnfi_hip.so; else cp /tmp/libnfi_new.so nerf_from_image_amd/libnfi_hip.so; fi
  timeout 900 python tools/pmc_collect.py --kernel render_fwd_kernel --out $R/gpurun_out/ab/pmc_$which.json --marched-from-bench -- \
     python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-variants > gpurun_out/ab/pmc_$which.log 2>&1
  python - $which <<'PY'
import json, sys
j = json.load(open('gpurun_out/ab/pmc_%s.json' % sys.argv[1]))
keep = {k: v for k, v in j.items() if not isinstance(v, (dict, list)) and k not in ('command',)}
print(sys.argv[1], json.dumps(keep))
PY
done
cp /tmp/libnfi_new.so nerf_from_image_amd/libnfi_hip.so
rm -rf gpurun_out/ab/*_passes
