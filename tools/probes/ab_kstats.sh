# per-kernel durations (rocprofv3 --kernel-trace --stats) of quick_bench for two prebuilt libraries on ONE box
R=$PWD; O=$R/gpurun_out/ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cp $R/nerf_from_image_amd/libnfi_hip.so /tmp/libnfi_new.so
for which in new good; do
  if [ $which = good ]; then cp $R/tools/probes/ab_libs/libnfi_good.so $R/nerf_from_image_amd/libnfi_hip.so; else cp /tmp/libnfi_new.so $R/nerf_from_image_amd/libnfi_hip.so; fi
  NFI_ITERS=20 NFI_TUNING=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ks_$which -o x -- python $R/tools/quick_bench.py > /dev/null 2>&1
done
cp /tmp/libnfi_new.so $R/nerf_from_image_amd/libnfi_hip.so
cd $R
python - <<'PY'
import csv, glob, collections
for which in ('new', 'good'):
    for f in glob.glob('gpurun_out/ab/ks_%s/**/*kernel_trace.csv' % which, recursive=True):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(list)
        for r in rows:
            acc[(r['Kernel_Name'][:60], r.get('Grid_Size', r.get('Grid_Size_X', '?')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        print('==', which)
        for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:9]:
            v.sort()
            print('  %-62s grid %8s n=%4d median %8.1f us  min %8.1f' % (name, grid, len(v), v[len(v) // 2], v[0]))
PY
rm -rf gpurun_out/ab/ks_*
