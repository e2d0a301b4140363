# A/B of two prebuilt libraries on ONE box: tools/probes/ab_libs/libnfi_good.so (built from another revision) against the
# tree's own.  bash tools/probes/ab_two_libs.sh   (through gpurun)
mkdir -p gpurun_out/ab
cp nerf_from_image_amd/libnfi_hip.so /tmp/libnfi_new.so
for round in 1 2; do
for which in new good; do
  if [ $which = good ]; then cp tools/probes/ab_libs/libnfi_good.so nerf_from_image_amd/libnfi_hip.so; else cp /tmp/libnfi_new.so nerf_from_image_amd/libnfi_hip.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/ab/bench_$which.json 2> gpurun_out/ab/bench_$which.err
  python - $which <<'PY'
import json, sys
j=json.loads(open("gpurun_out/ab/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print('%-5s value %.1f pipelined %.1f exact %.1f allhit %.1f kernel %.4f ms' % (sys.argv[1], j["value"]/1e6, j["value_pipelined"]/1e6, j["value_mlp_exact_fp32"]/1e6, j["value_all_rays_hit"]/1e6, j["kernel_ms_stats"]["median"]))
PY
done
done
cp /tmp/libnfi_new.so nerf_from_image_amd/libnfi_hip.so
