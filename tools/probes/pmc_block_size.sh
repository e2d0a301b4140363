# FETCH_SIZE / L2 hit rate / time of render_fwd_kernel for the three pixel-block sizes of the per-XCD work queues
# (tuning bits 5-6: 0 = default (32 px at 128^2), 96 = 16 px, 32 = 8 px).  bash tools/probes/pmc_block_size.sh  (GPU box)
# ROUND-3 PROBE: the tuning bits 5-8 were measurement knobs and left the product in round 4 (the default - the largest
# block that divides the image - had the least traffic: HISTORY.md section 7); to re-run this, put a library built from
# commit 4d6d769 in place (tools/probes/render_variants.py build base).
R=$PWD; O=$R/gpurun_out/r3blk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for t in 0 96 32; do
  NFI_TUNING=$t NFI_ITERS=10 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$t -o x -- python $R/tools/quick_bench.py > /dev/null 2>&1
  NFI_TUNING=$t NFI_ITERS=10 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/t_$t -o x -- python $R/tools/quick_bench.py > /dev/null 2>&1
  NFI_TUNING=$t python $R/tools/quick_bench.py 2>/dev/null | grep "B=8 radius=2.0" > $O/time_$t.txt
done
cd $R
python - <<'PY'
import csv, glob, collections
for t in (0, 96, 32):
    row = {}
    for kind in ('f', 't'):
        for f in glob.glob('gpurun_out/r3blk/%s_%d/**/*counter_collection.csv' % (kind, t), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                # the B=8 launches of the default render kernel are the largest-grid ones; take launches by name and keep the top values
                if 'render_fwd_kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
            for k, v in acc.items():
                v.sort(); top = v[-20:]
                row[k] = sum(top) / len(top)
    hit = row.get('TCC_HIT_sum', 0) / max(1.0, row.get('TCC_HIT_sum', 0) + row.get('TCC_MISS_sum', 0))
    print('tuning %3d: FETCH_SIZE %.0f KiB (x2 = %.2f GB fetched per launch)  L2 hit %.3f   %s' % (
        t, row.get('FETCH_SIZE', 0), row.get('FETCH_SIZE', 0) * 2 * 1024 / 1e9, hit, open('gpurun_out/r3blk/time_%d.txt' % t).read().strip()))
PY
