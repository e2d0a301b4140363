# A/B of the variant builds under build/variants (tools/probes/scatter_variants.py build ...) on ONE box: whole field backward +
# scatter by HIP events (two alternating rounds), then per-kernel medians from a rocprofv3 kernel trace of each.
# usage (gpurun): bash tools/probes/bwd_ab.sh <tag> name1 name2 ...
R=$PWD; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
python tools/probes/scatter_variants.py run "$@" > $O/bwd_ab.log 2>&1
cat $O/bwd_ab.log | cut -c1-260
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  NFI_PROBE_LIBRARY=$R/build/variants/libnfi_bwd_$v.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ks_$v -o x -- python $R/tools/bench_train_backward.py 20 > /dev/null 2>&1
done
cd $R
python - "$O" "$@" <<'PY' | tee -a $O/bwd_ab.log
import csv, glob, collections, sys
O, names = sys.argv[1], sys.argv[2:]
for which in names:
    for f in glob.glob('%s/ks_%s/**/*kernel_trace.csv' % (O, which), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        print('==', which)
        for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:7]:
            v.sort()
            print('  %-46s n=%4d median %8.1f us  min %8.1f' % (name, len(v), v[len(v) // 2], v[0]))
PY
rm -rf $O/ks_*
