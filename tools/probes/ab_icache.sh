# instruction-cache counters of the render kernel for two prebuilt libraries on ONE box (see ab_two_libs.sh)
R=$PWD; O=$R/gpurun_out/ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cp $R/nerf_from_image_amd/libnfi_hip.so /tmp/libnfi_new.so
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | head -40 > $O/sqc_counters.txt
for which in new good; do
  if [ $which = good ]; then cp $R/tools/probes/ab_libs/libnfi_good.so $R/nerf_from_image_amd/libnfi_hip.so; else cp /tmp/libnfi_new.so $R/nerf_from_image_amd/libnfi_hip.so; fi
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
    tag=$(echo $set | cut -d' ' -f1)
    NFI_ITERS=5 NFI_TUNING=0 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/ic_${which}_$tag -o x -- python $R/tools/quick_bench.py > /dev/null 2>&1
  done
done
cp /tmp/libnfi_new.so $R/nerf_from_image_amd/libnfi_hip.so
cd $R
python - <<'PY'
import csv, glob, collections
for which in ('new', 'good'):
    acc = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/ab/ic_%s_*/**/*counter_collection.csv' % which, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'render_fwd_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) >= 512 * 256:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(which, {k: '%.3g (n=%d)' % (sum(v) / len(v), len(v)) for k, v in sorted(acc.items())})
PY
cat gpurun_out/ab/sqc_counters.txt | tr '\n' ' '
rm -rf gpurun_out/ab/ic_*
