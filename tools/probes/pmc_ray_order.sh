R=$PWD; O=$R/gpurun_out/r3w; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in nohint hint; do
  if [ $cfg = nohint ]; then export NFI_NO_RAY_ORDER=1; else unset NFI_NO_RAY_ORDER; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${cfg}_f -o x -- python $R/tools/bench_train_backward.py 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${cfg}_t -o x -- python $R/tools/bench_train_backward.py 3 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for cfg in ('nohint','hint'):
    for kind in ('f','t'):
        for f in glob.glob('gpurun_out/r3w/%s_%s/**/*counter_collection.csv'%(cfg,kind), recursive=True):
            acc=collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'field_query_bwd' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
            print(cfg, {k:(sum(v)/len(v), len(v)) for k,v in acc.items()})
PY
